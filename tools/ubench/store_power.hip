// Tuning aid: what a write-only stream costs in POWER (the GMM kernels run at the board's power limit,
// so joules per byte stored are clock).  Fills a 2 GB buffer over and over for a few seconds with one
// store flavour; run tools/store_power.sh, which samples rocm-smi from the side.
//   hipcc -O2 --offload-arch=gfx950 store_power.hip -o /tmp/store_power && /tmp/store_power <variant> <seconds>
//   variant 0: non-temporal dwordx4   1: plain dwordx4   2: sc1 (write-through) dwordx4   3: no stores, FP64 FMA only
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
typedef double v2d __attribute__((ext_vector_type(2)));
template <int V> __global__ __launch_bounds__(256) void k(v2d* d, long long n, double v) {
  const long long stride = (long long)gridDim.x * 256 * 4;
  v2d x = {v, v + 1.0};
  for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n; i += stride) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (V == 3) { for (int q = 0; q < 16; ++q) { x.x = __builtin_fma(x.x, 1.0000001, 0.5); x.y = __builtin_fma(x.y, 0.9999999, 0.25); } continue; }
      if (i + u * 256 >= n) continue;
      if (V == 0) __builtin_nontemporal_store(x, d + i + u * 256);
      if (V == 1) d[i + u * 256] = x;
      if (V == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(d + i + u * 256), "v"(x) : "memory");
    }
  }
  if (V == 3 && x.x + x.y == 12345.0) d[0] = x;
}
int main(int argc, char** argv) {
  const int variant = argc > 1 ? atoi(argv[1]) : 0;
  const double secs = argc > 2 ? atof(argv[2]) : 5.0;
  const long long bytes = 2LL << 30, n = bytes / 16;
  v2d* d; if (hipMalloc(&d, bytes) != hipSuccess) return 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const auto t0 = std::chrono::steady_clock::now();
  double ms_sum = 0; long long launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    hipEventRecord(e0, 0);
    for (int r = 0; r < 20; ++r) {
      if (variant == 0) hipLaunchKernelGGL(k<0>, dim3(8192), dim3(256), 0, 0, d, n, 1.5);
      if (variant == 1) hipLaunchKernelGGL(k<1>, dim3(8192), dim3(256), 0, 0, d, n, 1.5);
      if (variant == 2) hipLaunchKernelGGL(k<2>, dim3(8192), dim3(256), 0, 0, d, n, 1.5);
      if (variant == 3) hipLaunchKernelGGL(k<3>, dim3(8192), dim3(256), 0, 0, d, n, 1.5);
    }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms_sum += ms; launches += 20;
  }
  printf("variant %d: %.1f us per launch, %.0f GB/s\n", variant, 1e3 * ms_sum / launches, variant == 3 ? 0.0 : bytes / (ms_sum / launches * 1e-3) / 1e9);
  return 0;
}
