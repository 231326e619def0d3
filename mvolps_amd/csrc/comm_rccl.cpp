// comm_rccl.cpp -- libmvolps_rccl.so: the three operations of `mvx_comm` (include/mvx_dist.h) over RCCL.
// One communicator per process (one process per GPU), created on the device that is current in the calling thread
// (mvx_set_device / hipSetDevice before mvx_rccl_comm_create).  The node images are device buffers and go GPU to GPU
// (ncclSend / ncclRecv inside one group, over xGMI on an MI355X node); the two small host-side operations (MAX of a few
// hundred doubles, broadcast of the incumbent) are staged through a device scratch buffer.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/mvx_dist.h"

namespace {

struct RcclCtx {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int dev = 0;
  double *scratch = nullptr;
  size_t scratch_n = 0;
};

#define RC_HIP(x)                                                                                \
  do {                                                                                           \
    hipError_t e_ = (x);                                                                         \
    if (e_ != hipSuccess) {                                                                      \
      std::fprintf(stderr, "mvx rccl: %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 0x301;                                                                              \
    }                                                                                            \
  } while (0)
#define RC_NCCL(x)                                                                                \
  do {                                                                                            \
    ncclResult_t r_ = (x);                                                                        \
    if (r_ != ncclSuccess) {                                                                      \
      std::fprintf(stderr, "mvx rccl: %s at %s:%d\n", ncclGetErrorString(r_), __FILE__, __LINE__); \
      return 0x302;                                                                               \
    }                                                                                             \
  } while (0)

int ensure_scratch(RcclCtx *c, size_t n) {
  if (c->scratch_n >= n) return 0;
  if (c->scratch) RC_HIP(hipFree(c->scratch));
  c->scratch = nullptr;
  c->scratch_n = 0;
  size_t cap = 1024;
  while (cap < n) cap *= 2;
  RC_HIP(hipMalloc(&c->scratch, cap * sizeof(double)));
  c->scratch_n = cap;
  return 0;
}

int rccl_allreduce_max(void *ctx, double *v, size_t n) {
  RcclCtx *c = (RcclCtx *)ctx;
  if (n == 0) return 0;
  RC_HIP(hipSetDevice(c->dev));
  if (int rc = ensure_scratch(c, n)) return rc;
  RC_HIP(hipMemcpyAsync(c->scratch, v, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  RC_NCCL(ncclAllReduce(c->scratch, c->scratch, n, ncclDouble, ncclMax, c->comm, c->stream));
  RC_HIP(hipMemcpyAsync(v, c->scratch, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  RC_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int rccl_bcast(void *ctx, double *v, size_t n, int root) {
  RcclCtx *c = (RcclCtx *)ctx;
  if (n == 0) return 0;
  RC_HIP(hipSetDevice(c->dev));
  if (int rc = ensure_scratch(c, n)) return rc;
  RC_HIP(hipMemcpyAsync(c->scratch, v, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  RC_NCCL(ncclBroadcast(c->scratch, c->scratch, n, ncclDouble, root, c->comm, c->stream));
  RC_HIP(hipMemcpyAsync(v, c->scratch, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  RC_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

// The images were written by the engine on its own streams; mvx_pack_from returns after its copies have completed
// (it synchronises), so the buffers are ready when this is called, and the receiver reads them after the
// synchronisation below.
int rccl_exchange(void *ctx, const mvx_xfer *sends, int ns, const mvx_xfer *recvs, int nr) {
  RcclCtx *c = (RcclCtx *)ctx;
  if (ns == 0 && nr == 0) return 0;
  RC_HIP(hipSetDevice(c->dev));
  RC_NCCL(ncclGroupStart());
  // a failing send / receive must not leave the communicator inside an open group: note the first error, close the
  // group whatever happened, then report
  ncclResult_t first = ncclSuccess;
  for (int k = 0; k < ns && first == ncclSuccess; k++) first = ncclSend(sends[k].buf, sends[k].bytes, ncclUint8, sends[k].peer, c->comm, c->stream);
  for (int k = 0; k < nr && first == ncclSuccess; k++) first = ncclRecv(recvs[k].buf, recvs[k].bytes, ncclUint8, recvs[k].peer, c->comm, c->stream);
  const ncclResult_t closed = ncclGroupEnd();
  if (first != ncclSuccess || closed != ncclSuccess) {
    std::fprintf(stderr, "mvx rccl: exchange failed: %s\n", ncclGetErrorString(first != ncclSuccess ? first : closed));
    return 0x302;
  }
  RC_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

} // namespace

extern "C" int mvx_rccl_unique_id(void *id) {
  static_assert(sizeof(ncclUniqueId) <= MVX_RCCL_ID_BYTES, "id buffer too small");
  ncclUniqueId u;
  RC_NCCL(ncclGetUniqueId(&u));
  std::memset(id, 0, MVX_RCCL_ID_BYTES);
  std::memcpy(id, &u, sizeof(u));
  return 0;
}

extern "C" int mvx_rccl_comm_create(const void *id, int rank, int size, mvx_comm *out) {
  std::memset(out, 0, sizeof(*out));
  RcclCtx *c = new (std::nothrow) RcclCtx();
  if (!c) return 0x301;
  // nothing of a half-built context is left behind
  auto fail = [&](int code, const char *what) {
    std::fprintf(stderr, "mvx rccl: %s failed\n", what);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return code;
  };
  if (hipGetDevice(&c->dev) != hipSuccess) return fail(0x301, "hipGetDevice");
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    c->stream = nullptr;
    return fail(0x301, "hipStreamCreateWithFlags");
  }
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  if (ncclCommInitRank(&c->comm, size, u, rank) != ncclSuccess) return fail(0x302, "ncclCommInitRank");
  out->ctx = c;
  out->rank = rank;
  out->size = size;
  out->allreduce_max = rccl_allreduce_max;
  out->exchange = rccl_exchange;
  out->bcast = rccl_bcast;
  return 0;
}

extern "C" void mvx_rccl_comm_destroy(mvx_comm *m) {
  if (!m || !m->ctx) return;
  RcclCtx *c = (RcclCtx *)m->ctx;
  (void)hipSetDevice(c->dev);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  std::memset(m, 0, sizeof(*m));
}
