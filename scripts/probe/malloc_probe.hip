// How long does hipMalloc take by size, idle and beside a busy stream?  (slab arenas: engine.cpp slab_alloc)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(double *x, int n) {
  double v = x[threadIdx.x];
  for (int i = 0; i < n; i++) v = v * 1.0000001 + 1e-9;
  x[threadIdx.x] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  double *d;
  hipMalloc(&d, 4096);
  hipStream_t s;
  hipStreamCreate(&s);
  for (int busy = 0; busy < 2; busy++)
    for (size_t mb : {4, 32, 128, 512, 2048, 8192}) {
      std::vector<void *> ps;
      double t = 0;
      const int reps = mb >= 2048 ? 4 : 16;
      for (int r = 0; r < reps; r++) {
        if (busy) for (int k = 0; k < 8; k++) spin<<<256, 256, 0, s>>>(d, 20000);
        void *p;
        const double t0 = now();
        if (hipMalloc(&p, mb << 20) != hipSuccess) { printf("fail\n"); return 1; }
        t += now() - t0;
        ps.push_back(p);
        if (busy) hipStreamSynchronize(s);
      }
      const double t0 = now();
      for (void *p : ps) hipFree(p);
      printf("busy=%d %5zu MB: hipMalloc %.1f us  hipFree %.1f us\n", busy, mb, 1e6 * t / reps, 1e6 * (now() - t0) / reps);
    }
  return 0;
}
