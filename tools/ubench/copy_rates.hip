// Tuning aid: what a plain streaming copy reaches on this GPU (read + written GB/s), by grid size,
// unroll and non-temporal hints -- the ceiling bench.py's copy_GBps should be close to.
//   hipcc -O3 --offload-arch=gfx950 copy_rates.hip -o copy_rates && ./copy_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v2d __attribute__((ext_vector_type(2)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void k(const v2d* __restrict__ src, v2d* __restrict__ dst, long long n) {
  const long long stride = (long long)gridDim.x * 256 * U;
  for (long long i = (long long)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
    v2d v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n) v[u] = NT ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * 256); else dst[i + u * 256] = v[u]; }
  }
}

template <int U, bool NT>
double run(const v2d* a, v2d* b, long long n, int grid) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double best = 0;
  for (int it = 0; it < 6; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<U, NT>), dim3(grid), dim3(256), 0, 0, a, b, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double g = 2.0 * 16.0 * n / (ms * 1e-3) / 1e9;
    if (it > 0 && g > best) best = g;
  }
  return best;
}

int main() {
  const long long bytes = 1ll << 30, n = bytes / 16;
  v2d *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes);
  printf("%-8s %10s %10s %10s %10s %10s %10s\n", "grid", "U1", "U2", "U4", "U1 nt", "U2 nt", "U4 nt");
  for (int grid : {512, 1024, 2048, 4096, 8192, 16384}) {
    printf("%-8d %10.0f %10.0f %10.0f %10.0f %10.0f %10.0f\n", grid, run<1, false>(a, b, n, grid), run<2, false>(a, b, n, grid),
           run<4, false>(a, b, n, grid), run<1, true>(a, b, n, grid), run<2, true>(a, b, n, grid), run<4, true>(a, b, n, grid));
  }
  return 0;
}
