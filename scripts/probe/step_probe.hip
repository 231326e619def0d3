// step_probe.hip -- latency probe (not product code): ONE launch per simplex pivot on a cache-resident tableau.
// Every workgroup repeats the selection (pricing over row 0 -> entering column, strided column gather + ratio test ->
// leaving row) from the tableau side being read and writes its tile of the update to the other side (out of place, so
// no workgroup can see a half-updated pivot row or column).  Measures what such a launch costs back to back; the
// arithmetic is shaped like the real step but not checked.  usage: step_probe [m n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Hdr { double *T[2]; double *w[2]; int m, n, ld, side; long steps; };
struct Cand { double k; int idx; };
__device__ __forceinline__ Cand wbest(Cand c) {
  for (int o = 32; o; o >>= 1) {
    double k2 = __shfl_down(c.k, o, 64); int i2 = __shfl_down(c.idx, o, 64);
    if (k2 > c.k || (k2 == c.k && i2 < c.idx && i2 != 0)) { c.k = k2; c.idx = i2; }
  }
  return c;
}
__device__ Cand bbest(Cand c, Cand *lds) {
  c = wbest(c);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) lds[w] = c;
  __syncthreads();
  Cand r = lds[0];
  for (int k = 1; k < (int)blockDim.x / 64; k++) if (lds[k].k > r.k || (lds[k].k == r.k && lds[k].idx < r.idx && lds[k].idx != 0)) r = lds[k];
  __syncthreads();
  return r;
}

template <int TR>
__global__ __launch_bounds__(256) void k_step(Hdr *hdr2, int k) {
  __shared__ Cand lds[4];
  const Hdr h = hdr2[k & 1]; // level 1
  const double *Tc = h.T[h.side]; double *Tn = h.T[h.side ^ 1];
  const double *wc = h.w[h.side]; double *wn = h.w[h.side ^ 1];
  const int m = h.m, n = h.n; const size_t ld = h.ld;
  const int t = threadIdx.x;
  // level 2: price row 0
  Cand best{0.0, 0};
  for (int j = 1 + t; j <= n; j += 256) { const double d = Tc[j]; const double sc = d * d / wc[j]; if (sc > best.k) best = Cand{sc, j}; }
  best = bbest(best, lds);
  const int q = best.idx ? best.idx : 1;
  // level 3: column q, ratio test
  Cand rb{0.0, 0};
  for (int i = 1 + t; i <= m; i += 256) {
    const double a = Tc[i * ld + q], beta = Tc[i * ld];
    const double r = (fabs(a) > 1e-9) ? 1.0 / (1.0 + fabs(beta / a)) : 0.0; // larger key = smaller ratio
    if (r > rb.k) rb = Cand{r, i};
  }
  rb = bbest(rb, lds);
  const int p = rb.idx ? rb.idx : 1;
  // level 4: row p strip, own rows of column q, tile
  const int j0 = 2 * ((int)blockIdx.x * 256 + t);
  const int i0 = (int)blockIdx.y * TR; // row 0 (objective) is part of row block 0
  const double piv = Tc[p * ld + q];
  const double ip = (fabs(piv) > 1e-9) ? 1.0 / piv : 1.0;
  if (j0 <= n) {
    double2 s = *reinterpret_cast<const double2 *>(Tc + p * ld + j0);
    s.x *= ip; s.y *= ip;
    double2 v[TR]; double ci[TR];
#pragma unroll
    for (int r = 0; r < TR; r++) v[r] = *reinterpret_cast<const double2 *>(Tc + (size_t)(i0 + r) * ld + j0);
#pragma unroll
    for (int r = 0; r < TR; r++) ci[r] = Tc[(size_t)(i0 + r) * ld + q];
#pragma unroll
    for (int r = 0; r < TR; r++) {
      if (i0 + r == p) { v[r].x = -s.x; v[r].y = -s.y; }
      else { v[r].x = fma(-ci[r] * 1e-3, s.x, v[r].x); v[r].y = fma(-ci[r] * 1e-3, s.y, v[r].y); }
      *reinterpret_cast<double2 *>(Tn + (size_t)(i0 + r) * ld + j0) = v[r];
    }
    if (blockIdx.y == 0) { // devex weights forward
      const double w0 = wc[j0], w1 = wc[j0 + 1];
      wn[j0] = fmax(w0, s.x * s.x); wn[j0 + 1] = fmax(w1, s.y * s.y);
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && t == 0) { Hdr o = h; o.side ^= 1; o.steps++; hdr2[(k + 1) & 1] = o; }
}

template <int TR>
static void run(int m, int n, int reps) {
  const int ld = (n + 1 + 31) / 32 * 32;
  const int rows = (m + 1 + TR - 1) / TR * TR;
  const size_t bytes = (size_t)(rows + TR) * ld * 8;
  Hdr h{};
  for (int s = 0; s < 2; s++) { CK(hipMalloc(&h.T[s], bytes)); CK(hipMalloc(&h.w[s], (size_t)ld * 8 + 64)); }
  std::vector<double> T((size_t)(rows + TR) * ld), w(ld + 8, 1.0);
  unsigned long long z = 12345;
  for (auto &x : T) { z = z * 6364136223846793005ull + 1442695040888963407ull; x = (double)(z >> 40) / (1 << 24) + 0.1; }
  for (int s = 0; s < 2; s++) { CK(hipMemcpy(h.T[s], T.data(), bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(h.w[s], w.data(), (size_t)ld * 8, hipMemcpyHostToDevice)); }
  h.m = m; h.n = n; h.ld = ld; h.side = 0; h.steps = 0;
  Hdr *d; CK(hipMalloc(&d, 2 * sizeof(Hdr)));
  CK(hipMemcpy(d, &h, sizeof(Hdr), hipMemcpyHostToDevice));
  CK(hipMemcpy(d + 1, &h, sizeof(Hdr), hipMemcpyHostToDevice));
  dim3 grid(((n + 2) / 2 + 255) / 256, rows / TR);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  int k = 0;
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_step<TR>, grid, dim3(256), 0, 0, d, k++);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_step<TR>, grid, dim3(256), 0, 0, d, k++);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  Hdr out; CK(hipMemcpy(&out, d + (k & 1), sizeof(Hdr), hipMemcpyDeviceToHost));
  printf("{\"m\": %d, \"n\": %d, \"TR\": %d, \"workgroups\": %u, \"us_per_launch\": %.2f, \"steps_seen\": %ld}\n", m, n, TR, grid.x * grid.y, ms * 1e3 / reps, out.steps);
  for (int s = 0; s < 2; s++) { CK(hipFree(h.T[s])); CK(hipFree(h.w[s])); }
  CK(hipFree(d));
}

int main(int argc, char **argv) {
  const int sizes[][2] = {{256, 512}, {512, 1024}, {1024, 2048}, {1024, 4096}, {2048, 4096}};
  for (auto &s : sizes) { run<4>(s[0], s[1], 400); run<8>(s[0], s[1], 400); run<16>(s[0], s[1], 400); }
  return 0;
}
