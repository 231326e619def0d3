/*
 * mvx_dist.h -- C ABI of the multi-GPU branch-and-bound entry: one process per GPU, node LPs farmed over the ranks.
 *
 * What is sharded is the unit the reference itself treats as independent: one B&B node = one glp_prob clone + its
 * glp_simplex calls (/root/reference/bs.cpp:114-117,269-288).  The loop of bs.cpp:96-327 runs on every rank in
 * lock-step over windows of the FIFO deque (util.cpp:165-166, bs.cpp:297-298); the incumbent (bs.cpp:90,172-174) and
 * the child bounds (bs.cpp:280,288) travel in small MAX all-reduces, and a child that has to change ranks travels as
 * the image mvx_pack_from writes (bounds + basis + tableau + appended cut rows), device to device.  Tree, oids, prune
 * labels, events and incumbent are those of mvx_branchAndBound on one GPU.
 *
 * The driver reaches its peers only through `mvx_comm` -- three operations -- so that the caller decides what carries
 * them: libmvolps_rccl.so (below) implements the table over RCCL for device buffers; the CPU tests plug
 * torch.distributed/gloo in through the same table (tests/test_dist_native.py).
 */
#ifndef MVX_DIST_H
#define MVX_DIST_H

#include <stddef.h>

#include "mvx_bnb.h"

#ifdef __cplusplus
extern "C" {
#endif

/* How a node travels.  Buffers come from buf_alloc: device memory for the gfx950 engine (mvx_pack_from writes
   there, RCCL sends from there), host memory for a CPU engine. */
typedef struct mvx_image_api {
  long long (*pack_size)(const void *P, const void *base);       /* mvx_pack_size_from */
  int (*pack)(const void *P, const void *base, void *buf);       /* mvx_pack_from */
  int (*unpack)(void *dst, const void *base, const void *buf);   /* mvx_unpack */
  void *(*buf_alloc)(size_t bytes);
  void (*buf_free)(void *buf);
} mvx_image_api;

const mvx_image_api *mvx_hip_image_api(void);
void *mvx_image_alloc(size_t bytes); /* device memory on the bound device; NULL when out of memory */
void mvx_image_free(void *buf);

typedef struct {
  void *buf;
  size_t bytes;
  int peer;
} mvx_xfer;

typedef struct mvx_comm {
  void *ctx;
  int rank, size;
  /* element-wise MAX over the ranks of n host doubles, in place; 0 on success */
  int (*allreduce_max)(void *ctx, double *v, size_t n);
  /* all sends and receives of one round, posted together (the k-th send from rank a to rank b matches the k-th
     receive that b posts from a); buffers are mvx_image_api buffers; returns when all of them have completed */
  int (*exchange)(void *ctx, const mvx_xfer *sends, int n_send, const mvx_xfer *recvs, int n_recv);
  /* n host doubles from rank `root` to every rank */
  int (*bcast)(void *ctx, double *v, size_t n, int root);
} mvx_comm;

typedef struct {
  int per_rank;   /* nodes a rank solves per window (default 64); the window is size * per_rank queue positions */
  int slack;      /* nodes beyond per_rank a rank may hold in one window before a child is sent away; < 0: per_rank / 4,
                     at least 1 */
  int roundrobin; /* 1: deal children round-robin (round 1's dealing, kept to measure the traffic against) */
} mvx_dist_params;

typedef struct {
  long long children, migrated, migrated_bytes, rounds;
  long long allreduces; /* MAX all-reduces made: one per round (children's bounds and their own window step together) */
} mvx_dist_stats;

void mvx_dist_default_params(mvx_dist_params *p);

/* The multi-rank counterpart of mvx_branchAndBound (bs.h:7).  Call it on every rank of `comm` with that rank's handle
   of the identical root problem; `res` comes out identical on every rank (mvx_bnb_free_result frees it).  FIFO node
   order only (params->node_strat 0: best-bound order picks by fresh child bounds and is not window-batchable).
   comm NULL = one rank.  Returns 0, MVX_EFAIL for an unsupported parameter, a transport error code of `comm`, or
   MVX_EDIST_NOCUT when a bug-compatible run meets a branched node that generates no cut (bs.cpp would re-add a cut
   pooled by an earlier node, cut.cpp:16-21, which is not carried between ranks). */
#define MVX_EDIST_NOCUT 0x201
int mvx_branchAndBound_dist(const mvx_lp_api *api, const mvx_image_api *img, void *root, const mvx_bnb_params *params,
                            const mvx_dist_params *dist, const mvx_comm *comm, mvx_bnb_result *res, mvx_dist_stats *stats);

/* ---- libmvolps_rccl.so: the table over RCCL (one communicator per process, device buffers, the engine's device) ----
   Rank 0 makes an id and hands it to the other ranks by whatever launched them (MPI, a file, torch.distributed's
   store); every rank then creates its communicator from it. */
#define MVX_RCCL_ID_BYTES 128
int mvx_rccl_unique_id(void *id);                                               /* ncclGetUniqueId */
int mvx_rccl_comm_create(const void *id, int rank, int size, mvx_comm *out);    /* ncclCommInitRank on the current device */
void mvx_rccl_comm_destroy(mvx_comm *c);

#ifdef __cplusplus
}
#endif
#endif
