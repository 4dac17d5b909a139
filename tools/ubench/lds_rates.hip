// Tuning aid: cost of the LDS instructions k_gmm_step uses (cycles per wave-instruction seen by the
// issuing wave), with 4, 8 and 12 waves per CU issuing together.
//   hipcc -O2 --offload-arch=gfx950 lds_rates.hip -o lds_rates && ./lds_rates
#include <hip/hip_runtime.h>
#include <stdio.h>

#define LDSP(p) ((__attribute__((address_space(3))) double*)(p))

template <int MODE>
__global__ void k(unsigned long long* out, int stride_mix) {
  extern __shared__ double s[];
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int j = 0; j < 16; ++j) s[j * nt + tid] = 1.0 + j;
  __syncthreads();
  // per-lane "random" table index (mode 3/4): a multiplicative hash of the lane id
  const int ridx = ((tid * 2654435761u) >> 20) & 127;
  double acc = 0.0;
  double v = 1.0 + tid;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 256; ++it) {
    if (MODE == 0) {          // ds_add_f64, thread-private slots
#pragma unroll
      for (int j = 0; j < 8; ++j) __builtin_amdgcn_ds_atomic_fadd_f64(LDSP(&s[j * nt + tid]), v);
    } else if (MODE == 1) {   // ds_read_b64 + v_add_f64 + ds_write_b64
#pragma unroll
      for (int j = 0; j < 8; ++j) { double* q = &s[j * nt + tid]; *q = *q + v; }
    } else if (MODE == 2) {   // ds_read_b64 conflict free
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += s[j * nt + ((tid + it) & (nt - 1))];
    } else if (MODE == 3) {   // ds_read_b128, per-lane table index (16-byte entries)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const double2 e = *reinterpret_cast<const double2*>(&s[2 * ((ridx + j * 17 + it) & 127)]); acc += e.x + e.y; }
    } else if (MODE == 4) {   // ds_read_b128, all lanes the same address (broadcast)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const double2 e = *reinterpret_cast<const double2*>(&s[2 * ((j * 17 + it) & 127)]); acc += e.x + e.y; }
    } else if (MODE == 5) {   // only the v_add_f64 of modes 2-4 (to subtract)
#pragma unroll
      for (int j = 0; j < 8; ++j) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc) : "v"(v)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc) : "v"(v)); }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  if (acc + s[tid] == 12345.0) out[0] = 1;
  if ((tid & 63) == 0) out[blockIdx.x * (nt / 64) + tid / 64] = t1 - t0;
}

typedef void (*kern_t)(unsigned long long*, int);
int main() {
  struct { const char* name; kern_t f; } tab[] = {
    {"ds_add_f64 (private slot)", k<0>}, {"ds_read_b64+add+ds_write_b64", k<1>}, {"ds_read_b64 linear (+v_add)", k<2>},
    {"ds_read_b128 per-lane idx (+2 v_add)", k<3>}, {"ds_read_b128 broadcast (+2 v_add)", k<4>}, {"2 v_add_f64 only", k<5>}};
  unsigned long long* d; hipMalloc(&d, 256 * 16 * 8);
  unsigned long long h[256 * 16];
  printf("%-40s %10s %10s %10s   shader cycles per loop element, per wave\n", "", "4 w/CU", "8 w/CU", "12 w/CU");
  for (auto& e : tab) {
    printf("%-40s", e.name);
    for (int wps = 1; wps <= 3; ++wps) {
      const int threads = 256 * wps, waves = 4 * wps;
      for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(e.f, dim3(256), dim3(threads), 16 * threads * sizeof(double), 0, d, 0);
      hipDeviceSynchronize();
      hipMemcpy(h, d, 256 * waves * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (int i = 0; i < 256 * waves; ++i) avg += (double)h[i];
      avg /= 256.0 * waves;
      printf(" %10.2f", avg / (256.0 * 8.0));
    }
    printf("\n");
  }
  return 0;
}
