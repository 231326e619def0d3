// stream_probe.hip -- design-space probe for the streamed rank-1 update (not product code):
// T[i][j] = fma(-c[i], s[j], T[i][j]) over a (rows x ld) fp64 tableau, variants of block size,
// rows per block (TR), doubles per lane (CPL) and launch bounds.  Interleaved rounds, HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int THREADS, int TR, int CPL, int MINW>
__global__ __launch_bounds__(THREADS, MINW) void k(double *T, const double *c, const double *s, int rows, int ld) {
  const int j0 = CPL * ((int)blockIdx.x * THREADS + (int)threadIdx.x);
  if (j0 >= ld) return;
  const int i0 = (int)blockIdx.y * TR;
  double2 sv[CPL / 2];
#pragma unroll
  for (int u = 0; u < CPL / 2; u++) sv[u] = *reinterpret_cast<const double2 *>(s + j0 + 2 * u);
  double *base = T + (size_t)i0 * ld + j0;
  double2 v[TR][CPL / 2];
  double ci[TR];
#pragma unroll
  for (int r = 0; r < TR; r++)
#pragma unroll
    for (int u = 0; u < CPL / 2; u++) v[r][u] = *reinterpret_cast<const double2 *>(base + (size_t)r * ld + 2 * u);
#pragma unroll
  for (int r = 0; r < TR; r++) ci[r] = c[i0 + r];
#pragma unroll
  for (int r = 0; r < TR; r++)
#pragma unroll
    for (int u = 0; u < CPL / 2; u++) {
      v[r][u].x = fma(-ci[r], sv[u].x, v[r][u].x);
      v[r][u].y = fma(-ci[r], sv[u].y, v[r][u].y);
    }
#pragma unroll
  for (int r = 0; r < TR; r++)
#pragma unroll
    for (int u = 0; u < CPL / 2; u++) *reinterpret_cast<double2 *>(base + (size_t)r * ld + 2 * u) = v[r][u];
}

// pivot-column tile (TR doubles) staged through LDS: one wave loads it, every lane reads it back
template <int THREADS, int TR>
__global__ __launch_bounds__(THREADS) void k_lds(double *T, const double *c, const double *s, int rows, int ld) {
  __shared__ double cs[TR];
  const int i0 = (int)blockIdx.y * TR;
  if (threadIdx.x < TR) cs[threadIdx.x] = c[i0 + threadIdx.x];
  const int j0 = 2 * ((int)blockIdx.x * THREADS + (int)threadIdx.x);
  const bool act = j0 < ld;
  double2 sv = act ? *reinterpret_cast<const double2 *>(s + j0) : double2{0, 0};
  double *base = T + (size_t)i0 * ld + j0;
  double2 v[TR];
  if (act) {
#pragma unroll
    for (int r = 0; r < TR; r++) v[r] = *reinterpret_cast<const double2 *>(base + (size_t)r * ld);
  }
  __syncthreads();
  if (!act) return;
#pragma unroll
  for (int r = 0; r < TR; r++) {
    const double ci = cs[r];
    v[r].x = fma(-ci, sv.x, v[r].x);
    v[r].y = fma(-ci, sv.y, v[r].y);
  }
#pragma unroll
  for (int r = 0; r < TR; r++) *reinterpret_cast<double2 *>(base + (size_t)r * ld) = v[r];
}
template <int THREADS, int TR>
void launch_lds(double *T, const double *c, const double *s, int rows, int ld, hipStream_t st) {
  dim3 grid((ld / 2 + THREADS - 1) / THREADS, rows / TR);
  hipLaunchKernelGGL((k_lds<THREADS, TR>), grid, dim3(THREADS), 0, st, T, c, s, rows, ld);
}

struct Var { const char *name; void (*launch)(double *, const double *, const double *, int, int, hipStream_t); };
template <int THREADS, int TR, int CPL, int MINW>
void launch(double *T, const double *c, const double *s, int rows, int ld, hipStream_t st) {
  dim3 grid((ld / CPL + THREADS - 1) / THREADS, rows / TR);
  hipLaunchKernelGGL((k<THREADS, TR, CPL, MINW>), grid, dim3(THREADS), 0, st, T, c, s, rows, ld);
}

int main() {
  const int rows = 4096 + 32, ld = 8224; // rows a multiple of every TR
  double *T, *c, *s;
  CK(hipMalloc(&T, (size_t)rows * ld * 8)); CK(hipMalloc(&c, rows * 8)); CK(hipMalloc(&s, ld * 8));
  std::vector<double> h((size_t)rows * ld);
  for (size_t i = 0; i < h.size(); i++) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0;
  CK(hipMemcpy(T, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  std::vector<double> hc(rows, 1e-9), hs(ld, 1e-9);
  CK(hipMemcpy(c, hc.data(), rows * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(s, hs.data(), ld * 8, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  std::vector<Var> vars = {
    {"t256 tr16 cpl2 w1", launch<256, 16, 2, 1>}, {"t256 tr16 cpl2 w4", launch<256, 16, 2, 4>}, {"t256 tr8  cpl2 w1", launch<256, 8, 2, 1>},
    {"t256 tr8  cpl4 w1", launch<256, 8, 4, 1>},  {"t256 tr16 cpl4 w1", launch<256, 16, 4, 1>}, {"t256 tr4  cpl4 w1", launch<256, 4, 4, 1>},
    {"t512 tr16 cpl2 w1", launch<512, 16, 2, 1>}, {"t512 tr8  cpl2 w1", launch<512, 8, 2, 1>},  {"t128 tr16 cpl2 w1", launch<128, 16, 2, 1>},
    {"t128 tr32 cpl2 w1", launch<128, 32, 2, 1>}, {"t256 tr32 cpl2 w1", launch<256, 32, 2, 1>}, {"t64  tr16 cpl4 w1", launch<64, 16, 4, 1>},
    {"t256 tr16 LDS colq", launch_lds<256, 16>}, {"t128 tr16 LDS colq", launch_lds<128, 16>}, {"t256 tr32 LDS colq", launch_lds<256, 32>},
    {"t256 tr12 cpl2 w1", launch<256, 12, 2, 1>}, {"t1024 tr8 cpl2 w1", launch<1024, 8, 2, 1>}, {"t256 tr8  cpl8 w1", launch<256, 8, 8, 1>},
  };
  const double bytes = 16.0 * rows * ld;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<std::vector<float>> times(vars.size());
  for (int round = 0; round < 6; round++)
    for (size_t v = 0; v < vars.size(); v++) {
      if (round == 0) { vars[v].launch(T, c, s, rows, ld, st); CK(hipStreamSynchronize(st)); }
      CK(hipEventRecord(a, st));
      for (int it = 0; it < 20; it++) vars[v].launch(T, c, s, rows, ld, st);
      CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); times[v].push_back(ms / 20);
    }
  for (size_t v = 0; v < vars.size(); v++) {
    std::sort(times[v].begin(), times[v].end());
    float med = times[v][times[v].size() / 2], mn = times[v][0];
    printf("%-20s median %7.2f us  %6.0f GB/s   best %7.2f us %6.0f GB/s\n", vars[v].name, med * 1e3, bytes / (med * 1e-3) / 1e9, mn * 1e3, bytes / (mn * 1e-3) / 1e9);
  }
  return 0;
}
