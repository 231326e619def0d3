// cluster_probe.hip -- latency probe (not product code): what does one all-to-all exchange among NW workgroups cost
// inside ONE launch, by placement (all XCDs / one XCD), store flavour (write-through sc1 / L2-resident sc0), workgroup
// size and record size?  The exchange is the one the chained selection (k_chain) needs twice per simplex step: every
// workgroup reduces its candidates (one barrier), publishes a record of NF self-tagged 16-byte fields {lo, tag, hi, tag},
// and every wave of every workgroup sweeps all NW records (field-major: one dwordx4 load per field, lane = record) and
// reduces them with wave shuffles.
// usage: cluster_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
typedef unsigned u4 __attribute__((ext_vector_type(4)));

enum { AUX_SC0 = 1, AUX_SC1 = 16 };
constexpr int NWPAD = 64; // records per field row

// F = 0: sc0 stores (line stays in the XCD's L2), 1: sc1 (written through)
template <int F, int NF, int NT, int GATHER>
__global__ __launch_bounds__(NT) void k_xchg(unsigned *gran, int gran_bytes, int nw, int stride, int rounds, unsigned base, int *tmo, u64 *stamps, int *xcc,
                                             const double *__restrict__ T, int ld, int m) {
  constexpr int NWV = NT / 64;
  __shared__ double s_slot[NWV][NF];
  if ((int)blockIdx.x % stride) return;
  const int w = (int)blockIdx.x / stride, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (w >= nw) return;
  if (t == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[w] = (int)(id & 0xf);
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(gran, 0, gran_bytes, 0x00027000);
  unsigned acc = w + 1;
  bool fail = false;
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 1; r <= rounds && !fail; r++) {
    double a = 1.0 + (acc & 1023) + t;
    if (GATHER) { // a dependent strided gather, like the entering column's
      const int q = 1 + (acc % 8000);
      const int i = 1 + (w * NT + t) % m;
      a = T[(size_t)i * ld + q];
    }
    // workgroup reduce, one barrier: every wave leaves its best and a payload, every wave reduces the slots
    for (int o = 32; o; o >>= 1) a = fmax(a, __shfl_xor(a, o, 64));
    if (lane < NF) s_slot[wave][lane] = a + lane;
    __syncthreads();
    double bb = s_slot[0][0];
    int bw = 0;
#pragma unroll
    for (int k = 1; k < NWV; k++) if (s_slot[k][0] > bb) { bb = s_slot[k][0]; bw = k; }
    const unsigned tag = base + (unsigned)r;
    const unsigned reg_off = (unsigned)(r & 3) * (NF * NWPAD * 16);
    if (t < NF) {
      const double v = s_slot[bw][t];
      const u4 x = {(unsigned)__double2loint(v), tag, (unsigned)__double2hiint(v), tag};
      __builtin_amdgcn_raw_buffer_store_b128(x, rs, reg_off + (unsigned)(t * NWPAD + w) * 16, 0, F ? AUX_SC1 : AUX_SC0);
    }
    // every wave sweeps: lane k < nw takes record k, one 16-byte load per field
    double rec[NF];
    unsigned spins = 0;
    const int lk = lane < nw ? lane : 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int f = 0; f < NF; f++) {
        const u4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, reg_off + (unsigned)(f * NWPAD + lk) * 16, 0, AUX_SC1);
        ok &= (x.y == tag) & (x.w == tag);
        rec[f] = __hiloint2double((int)x.z, (int)x.x);
      }
      if (__all(ok)) break;
      if (++spins > 200000u) { fail = true; if (lane == 0) atomicExch(tmo, r); break; }
      __builtin_amdgcn_s_sleep(1);
    }
    // reduce the records: arg-max of field 0, payload from the winner's lane
    double key = lane < nw ? rec[0] : -1e300;
    int who = lane;
    for (int o = 32; o; o >>= 1) {
      const double k2 = __shfl_xor(key, o, 64);
      const int w2 = __shfl_xor(who, o, 64);
      if (k2 > key || (k2 == key && w2 < who)) { key = k2; who = w2; }
    }
    double pay = 0.0;
#pragma unroll
    for (int f = 1; f < NF; f++) pay += __shfl(rec[f], who, 64);
    acc = acc * 1664525u + (unsigned)(int)(key + pay);
    __syncthreads(); // s_slot reuse
  }
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  if (t == 0) { stamps[w * 2] = t0; stamps[w * 2 + 1] = t1; }
  if (t == 0 && acc == 0xdeadbeef) tmo[1] = 1;
}

struct Bufs { unsigned *gran; int gran_bytes; int *tmo; u64 *stamps; int *xcc; const double *T; int ld, m; unsigned base; };

template <int F, int NF, int NT, int GATHER>
static void run(const char *name, int nw, int stride, Bufs &B) {
  const int rounds = 400;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; rep++) {
    CK(hipMemset(B.tmo, 0, 64));
    B.base += 4096;
    CK(hipEventRecord(e0, 0));
    k_xchg<F, NF, NT, GATHER><<<nw * stride, NT>>>(B.gran, B.gran_bytes, nw, stride, rounds, B.base, B.tmo, B.stamps, B.xcc, B.T, B.ld, B.m);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    if (rep == 0) continue;
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int ht[2]; CK(hipMemcpy(ht, B.tmo, 8, hipMemcpyDeviceToHost));
    std::vector<u64> st(nw * 2); CK(hipMemcpy(st.data(), B.stamps, nw * 2 * 8, hipMemcpyDeviceToHost));
    std::vector<int> xc(nw); CK(hipMemcpy(xc.data(), B.xcc, nw * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int w = 0; w < nw; w++) worst = std::max(worst, (double)(st[w * 2 + 1] - st[w * 2]) / 100.0);
    int nx[16] = {0};
    for (int w = 0; w < nw; w++) nx[xc[w] & 15]++;
    int used = 0;
    for (int k = 0; k < 8; k++) used += nx[k] > 0;
    printf("{\"test\": \"%s\", \"store\": \"%s\", \"fields\": %d, \"threads\": %d, \"gather\": %d, \"nw\": %d, \"stride\": %d, \"us_per_round_event\": %.2f, "
           "\"us_per_round_inkernel\": %.2f, \"timeout_round\": %d, \"xcds_used\": %d}\n",
           name, F ? "sc1" : "sc0", NF, NT, GATHER, nw, stride, ms * 1e3 / rounds, worst / rounds, ht[0], used);
  }
}

int main() {
  const int m = 4096, n = 8192;
  const int ld = (n + 1 + 31) / 32 * 32;
  const size_t tb = (size_t)(m + 65) * ld * 8;
  double *T; CK(hipMalloc(&T, tb));
  {
    std::vector<double> h((size_t)(m + 65) * ld);
    for (size_t k = 0; k < h.size(); k++) h[k] = 1e-3 * (double)((k * 2654435761u) % 1000);
    CK(hipMemcpy(T, h.data(), tb, hipMemcpyHostToDevice));
  }
  Bufs B;
  B.gran_bytes = 4 * 8 * NWPAD * 16;
  CK(hipMalloc(&B.gran, B.gran_bytes)); CK(hipMemset(B.gran, 0, B.gran_bytes));
  CK(hipMalloc(&B.tmo, 64)); CK(hipMalloc(&B.stamps, 64 * 2 * 8)); CK(hipMalloc(&B.xcc, 64 * 4));
  B.T = T; B.ld = ld; B.m = m; B.base = 4096;
  // 8192 threads in all, three ways
  run<1, 7, 256, 0>("all_xcds", 32, 1, B);
  run<0, 7, 256, 0>("one_xcd", 32, 8, B);
  run<1, 7, 512, 0>("all_xcds", 16, 1, B);
  run<1, 7, 512, 0>("one_xcd", 16, 8, B);
  run<0, 7, 512, 0>("one_xcd", 16, 8, B);
  run<1, 7, 1024, 0>("all_xcds", 8, 1, B);
  run<0, 7, 1024, 0>("one_xcd", 8, 8, B);
  // with the dependent gather in front
  run<1, 7, 512, 1>("all_xcds", 16, 1, B);
  run<0, 7, 512, 1>("one_xcd", 16, 8, B);
  run<1, 7, 1024, 1>("all_xcds", 8, 1, B);
  run<0, 7, 1024, 1>("one_xcd", 8, 8, B);
  // fewer fields; smaller / larger clusters
  run<0, 4, 512, 0>("one_xcd", 16, 8, B);
  run<1, 4, 512, 0>("all_xcds", 16, 1, B);
  run<0, 7, 512, 0>("one_xcd", 4, 8, B);
  run<1, 7, 512, 0>("all_xcds", 4, 1, B);
  run<0, 7, 512, 0>("one_xcd", 32, 8, B);
  run<1, 7, 512, 0>("all_xcds", 32, 1, B);
  run<1, 7, 512, 0>("all_xcds", 64, 1, B);
  return 0;
}
