// overlap_probe.hip -- latency probe (not product code): what do small dependent "selection" launches cost while a
// streaming pass over a 268 MB tableau runs on another stream?  And what does an in-launch all-to-all among a few
// workgroups cost under the same load?  Arms: no reservation / high-priority stream / disjoint CU masks
// (hipExtStreamCreateWithCUMask).  The arithmetic is shaped like the real kernels but not checked.
// usage: overlap_probe [m n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef unsigned long long u64;

// ---- streaming pass: out of place, 16-row tiles, 16 B per lane, K rank-1 steps applied in registers
template <int TR>
__global__ __launch_bounds__(256) void k_stream(const double *__restrict__ Tin, double *__restrict__ Tout, const double *__restrict__ srow,
                                                const double *__restrict__ colq, int ld, int n, int K, int mcap) {
  const int j0 = 2 * ((int)blockIdx.x * 256 + (int)threadIdx.x);
  if (j0 > n) return;
  const int i0 = 1 + (int)blockIdx.y * TR;
  const double *base = Tin + (size_t)i0 * ld + j0;
  double2 v[TR];
#pragma unroll
  for (int r = 0; r < TR; r++) {
    v[r].x = __builtin_nontemporal_load(base + (size_t)r * ld);
    v[r].y = __builtin_nontemporal_load(base + (size_t)r * ld + 1);
  }
  for (int l = 0; l < K; l++) {
    const double2 s = *reinterpret_cast<const double2 *>(srow + (size_t)l * ld + j0);
    const double *cq = colq + (size_t)l * mcap + i0;
#pragma unroll
    for (int r = 0; r < TR; r++) {
      const double ci = cq[r];
      v[r].x = fma(-ci, s.x, v[r].x);
      v[r].y = fma(-ci, s.y, v[r].y);
    }
  }
  double *ob = Tout + (size_t)i0 * ld + j0;
#pragma unroll
  for (int r = 0; r < TR; r++) {
    __builtin_nontemporal_store(v[r].x, ob + (size_t)r * ld);
    __builtin_nontemporal_store(v[r].y, ob + (size_t)r * ld + 1);
  }
}

// ---- selection-like small kernel: level 1 control words, level 2 partials, level 3 strided column gather + carry
struct Ctl { int q, p, alive, pad; double part[64]; };
__global__ __launch_bounds__(256) void k_sel(Ctl *c, const double *__restrict__ T, const double *__restrict__ colq, int ld, int m, int mcap,
                                             int carry, int phase, int mrows) {
  __shared__ double lds[4];
  const int alive = c->alive; // level 1
  if (!alive) return;
  const int lane = threadIdx.x & 63;
  double best = c->part[lane]; // level 2
  for (int o = 32; o; o >>= 1) best = fmax(best, __shfl_down(best, o, 64));
  best = __shfl(best, 0, 64);
  const int q = 1 + ((int)(best * 1e6) + c->q) % 8000;
  const int i = 1 + (int)blockIdx.x * 256 + (int)threadIdx.x;
  double a = 0.0;
  if (i <= m) {
    a = phase ? T[(size_t)(1 + q % mrows) * ld + i] : T[(size_t)i * ld + q]; // level 3: row read / strided column gather
    for (int l = 0; l < carry; l++) a = fma(-colq[(size_t)l * mcap + (i % mrows)], 0.5, a);
  }
  for (int o = 32; o; o >>= 1) a = fmax(a, __shfl_down(a, o, 64));
  if (lane == 0) lds[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    c->part[blockIdx.x] = fmax(fmax(lds[0], lds[1]), fmax(lds[2], lds[3])) * 1e-3;
    if (blockIdx.x == 0) c->q = q;
  }
}

// ---- in-launch all-to-all among NW workgroups: each publishes G tagged granules per round, everyone sweeps all
__device__ __forceinline__ void gstore(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 gload(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int G>
__global__ __launch_bounds__(256) void k_cluster(u64 *gran, int rounds, int *tmo, u64 *stamps, const double *__restrict__ T, int ld, int m) {
  __shared__ unsigned s_val[64 * 8];
  __shared__ int s_fail;
  const int nw = gridDim.x, w = blockIdx.x, t = threadIdx.x;
  if (t == 0) s_fail = 0;
  __syncthreads();
  unsigned acc = w + 1;
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 1; r <= rounds; r++) {
    // some dependent memory work per round, like a gather that follows the exchange
    const int q = 1 + (acc % 8000);
    const int i = 1 + w * 256 + t;
    double a = (i <= m) ? T[(size_t)i * ld + q] : 0.0;
    acc += (unsigned)(a * 3.0);
    __syncthreads();
    // publish G granules {tag = r, value}
    if (t < G) gstore(gran + ((size_t)(r & 1) * nw + w) * G + t, ((u64)r << 32) | (acc + t));
    // sweep: wave 0 reads all nw*G granules of this round's parity
    if (t < 64) {
      const int total = nw * G;
      unsigned spins = 0;
      for (;;) {
        bool ok = true;
        for (int k = t; k < total; k += 64) {
          const u64 x = gload(gran + (size_t)(r & 1) * nw * G + k);
          ok &= (unsigned)(x >> 32) == (unsigned)r;
          s_val[k & 511] = (unsigned)x;
        }
        if (__all(ok)) break;
        if (++spins > 200000u) { if (t == 0) { s_fail = 1; atomicExch(tmo, r); } break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (s_fail) break;
    acc += s_val[(t * 7) % (nw * G > 512 ? 512 : nw * G)];
  }
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  if (t == 0) { stamps[w * 2] = t0; stamps[w * 2 + 1] = t1; }
  if (t == 0 && acc == 0xdeadbeef) tmo[1] = 1;
}

static double ms_between(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char **argv) {
  const int m = argc > 2 ? atoi(argv[1]) : 4096, n = argc > 2 ? atoi(argv[2]) : 8192;
  const int ld = (n + 1 + 31) / 32 * 32, mcap = m + 64;
  const size_t tb = (size_t)(mcap + 1) * ld * 8;
  const int K = 10, TR = 16;
  double *Ta, *Tb, *srow, *colq;
  CK(hipMalloc(&Ta, tb)); CK(hipMalloc(&Tb, tb));
  CK(hipMalloc(&srow, (size_t)32 * ld * 8)); CK(hipMalloc(&colq, (size_t)32 * (mcap + 1) * 8));
  {
    std::vector<double> h((size_t)(mcap + 1) * ld);
    for (size_t k = 0; k < h.size(); k++) h[k] = 1e-3 * (double)((k * 2654435761u) % 1000);
    CK(hipMemcpy(Ta, h.data(), tb, hipMemcpyHostToDevice));
    CK(hipMemcpy(Tb, h.data(), tb, hipMemcpyHostToDevice));
    CK(hipMemcpy(srow, h.data(), (size_t)32 * ld * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(colq, h.data(), (size_t)32 * (mcap + 1) * 8, hipMemcpyHostToDevice));
  }
  Ctl *ctl; CK(hipMalloc(&ctl, sizeof(Ctl)));
  Ctl hc{}; hc.alive = 1; for (int k = 0; k < 64; k++) hc.part[k] = 0.001 * k;
  CK(hipMemcpy(ctl, &hc, sizeof(Ctl), hipMemcpyHostToDevice));
  u64 *gran; int *tmo; u64 *stamps;
  CK(hipMalloc(&gran, 2 * 64 * 8 * 8)); CK(hipMalloc(&tmo, 64)); CK(hipMalloc(&stamps, 64 * 2 * 8));

  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d, \"m\": %d, \"n\": %d, \"tableau_MB\": %.1f}\n", prop.name, ncu, m, n, tb / 1e6);

  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t sB, sS, sShi, sBm, sSm;
  CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sS, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&sShi, hipStreamNonBlocking, hi));
  // CU masks: the selection stream gets `res` CUs per XCD... the mask is by CU index; take the top CUs of the index range and
  // also a strided variant, and see which (if any) behaves like a per-XCD reservation
  const int words = (ncu + 31) / 32;
  bool mask_ok = true;
  const int res = 32;
  std::vector<uint32_t> mb(words, 0), msel(words, 0);
  for (int cu = 0; cu < ncu; cu++) {
    const bool sel = (cu % 8) == 7; // every 8th CU: 32 of 256
    (sel ? msel : mb)[cu / 32] |= 1u << (cu % 32);
  }
  if (hipExtStreamCreateWithCUMask(&sBm, words, mb.data()) != hipSuccess) mask_ok = false;
  if (mask_ok && hipExtStreamCreateWithCUMask(&sSm, words, msel.data()) != hipSuccess) mask_ok = false;
  printf("{\"cu_mask_streams\": %s, \"priority_range\": [%d, %d], \"reserved_cus\": %d}\n", mask_ok ? "true" : "false", lo, hi, res);

  hipEvent_t e0, e1, e2, e3; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
  const dim3 gs((n + 2 + 511) / 512, (m + TR - 1) / TR);
  auto stream_pass = [&](hipStream_t s, int reps) {
    for (int r = 0; r < reps; r++) {
      if (r & 1) k_stream<TR><<<gs, 256, 0, s>>>(Tb, Ta, srow, colq, ld, n, K, mcap + 1);
      else k_stream<TR><<<gs, 256, 0, s>>>(Ta, Tb, srow, colq, ld, n, K, mcap + 1);
    }
  };
  const int nsel = 40;
  auto sel_chain = [&](hipStream_t s, int carry) {
    for (int k = 0; k < nsel; k++) {
      if (k & 1) k_sel<<<(n + 1 + 255) / 256, 256, 0, s>>>(ctl, Ta, colq, ld, n, mcap + 1, carry, 1, m);
      else k_sel<<<(m + 255) / 256, 256, 0, s>>>(ctl, Ta, colq, ld, m, mcap + 1, carry, 0, m);
    }
  };
  // warm-up
  stream_pass(sB, 4); sel_chain(sS, 10); CK(hipDeviceSynchronize());

  // 1. streaming alone (full chip / masked)
  for (int arm = 0; arm < (mask_ok ? 2 : 1); arm++) {
    hipStream_t s = arm ? sBm : sB;
    stream_pass(s, 2); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); stream_pass(s, 20); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    const double us = ms_between(e0, e1) * 1e3 / 20;
    printf("{\"test\": \"stream_alone\", \"masked\": %d, \"us_per_pass\": %.1f, \"TBps\": %.2f}\n", arm, us, 2.0 * (double)m * ld * 8 / us / 1e6);
  }
  // 2. selection chain alone
  for (int carry : {0, 10, 20}) {
    CK(hipEventRecord(e0, sS)); sel_chain(sS, carry); CK(hipEventRecord(e1, sS)); CK(hipStreamSynchronize(sS));
    printf("{\"test\": \"sel_alone\", \"carry\": %d, \"us_per_kernel\": %.2f}\n", carry, ms_between(e0, e1) * 1e3 / nsel);
  }
  if (mask_ok) {
    CK(hipEventRecord(e0, sSm)); sel_chain(sSm, 10); CK(hipEventRecord(e1, sSm)); CK(hipStreamSynchronize(sSm));
    printf("{\"test\": \"sel_alone_masked\", \"carry\": 10, \"us_per_kernel\": %.2f}\n", ms_between(e0, e1) * 1e3 / nsel);
  }
  // 3. both at once
  struct Arm { const char *name; hipStream_t b, s; };
  std::vector<Arm> arms = {{"plain", sB, sS}, {"sel_high_priority", sB, sShi}};
  if (mask_ok) arms.push_back({"cu_masks", sBm, sSm});
  for (auto &a : arms) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, a.b)); stream_pass(a.b, 12); CK(hipEventRecord(e1, a.b));
    CK(hipEventRecord(e2, a.s)); sel_chain(a.s, 10); CK(hipEventRecord(e3, a.s));
    CK(hipDeviceSynchronize());
    printf("{\"test\": \"both\", \"arm\": \"%s\", \"stream_us_per_pass\": %.1f, \"sel_us_per_kernel\": %.2f}\n", a.name, ms_between(e0, e1) * 1e3 / 12,
           ms_between(e2, e3) * 1e3 / nsel);
  }
  // 4. cluster exchange: alone and beside the stream
  for (int nw : {16, 32, 48}) {
    for (int load = 0; load < (mask_ok ? 3 : 2); load++) {
      hipStream_t sb = load == 2 ? sBm : sB, ss = load == 2 ? sSm : sShi;
      const int rounds = 200;
      CK(hipMemsetAsync(gran, 0, 2 * 64 * 8 * 8, ss)); CK(hipMemsetAsync(tmo, 0, 64, ss));
      CK(hipDeviceSynchronize());
      if (load) stream_pass(sb, 14);
      CK(hipEventRecord(e2, ss));
      k_cluster<6><<<nw, 256, 0, ss>>>(gran, rounds, tmo, stamps, Ta, ld, m);
      CK(hipEventRecord(e3, ss));
      CK(hipDeviceSynchronize());
      int ht[2]; CK(hipMemcpy(ht, tmo, 8, hipMemcpyDeviceToHost));
      std::vector<u64> st(64 * 2); CK(hipMemcpy(st.data(), stamps, 64 * 2 * 8, hipMemcpyDeviceToHost));
      double worst = 0;
      for (int w = 0; w < nw; w++) worst = std::max(worst, (double)(st[w * 2 + 1] - st[w * 2]) / 100.0);
      printf("{\"test\": \"cluster\", \"nw\": %d, \"load\": %d, \"granules\": 6, \"us_per_round_event\": %.2f, \"us_per_round_inkernel\": %.2f, \"timeout_round\": %d}\n", nw, load,
             ms_between(e2, e3) * 1e3 / rounds, worst / rounds, ht[0]);
    }
  }
  return 0;
}
