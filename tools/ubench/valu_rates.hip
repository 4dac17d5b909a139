// Tuning aid: issue cost (shader cycles per wave-instruction) of the VALU instructions k_gmm_step
// is made of, at 1, 2 and 3 waves per SIMD.  Eight independent chains per instruction, 256 repeats.
//   hipcc -O2 --offload-arch=gfx950 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define DEFK(NAME, DECL, ASM8, SINK)                                                           \
  __global__ void NAME(unsigned long long* out, double seed) {                                 \
    DECL                                                                                       \
    unsigned long long t0 = __builtin_readcyclecounter();                                      \
    for (int it = 0; it < 256; ++it) { ASM8 }                                                  \
    unsigned long long t1 = __builtin_readcyclecounter();                                      \
    SINK                                                                                       \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
  }

#define D8 double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; double b = seed * 0.5 + 1.0, c = 0.25;
#define DSINK if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.0) out[0] = 1;
#define U8 unsigned a0 = (unsigned)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; unsigned b = a0 * 3 + 1;
#define USINK if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345u) out[0] = 1;
#define L8 unsigned long long a0 = (unsigned long long)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; unsigned b = (unsigned)a0 * 3 + 1, c = 0xD2511F53u;
#define LSINK if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345ull) out[0] = 1;

#define X_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
DEFK(k_fma_f64, D8, REP8(X_FMA64), DSINK)
#define X_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a##i) : "v"(b));
DEFK(k_mul_f64, D8, REP8(X_MUL64), DSINK)
#define X_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a##i) : "v"(b));
DEFK(k_add_f64, D8, REP8(X_ADD64), DSINK)
#define X_MAX64(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a##i) : "v"(b));
DEFK(k_max_f64, D8, REP8(X_MAX64), DSINK)
#define X_RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a##i));
DEFK(k_rcp_f64, D8, REP8(X_RCP64), DSINK)
#define X_RSQ64(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(a##i));
DEFK(k_rsq_f64, D8, REP8(X_RSQ64), DSINK)
#define X_SQRT64(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a##i));
DEFK(k_sqrt_f64, D8, REP8(X_SQRT64), DSINK)
#define X_RND64(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(a##i));
DEFK(k_rndne_f64, D8, REP8(X_RND64), DSINK)
#define X_LDEXP64(i) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a##i));
DEFK(k_ldexp_f64, D8, REP8(X_LDEXP64), DSINK)
#define X_CVT64(i) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(a##i) : "v"(u));
DEFK(k_cvt_f64_u32, D8 unsigned u = (unsigned)seed;, REP8(X_CVT64), DSINK)
#define X_CMP64(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, 0, 1, vcc" : : "v"(a##i), "v"(b), "v"(u) : "vcc");
#define X_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(a##i) : "v"(b), "v"(c) : "vcc");
DEFK(k_mad_u64_u32, L8, REP8(X_MAD64), LSINK)
#define X_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a##i) : "v"(b));
DEFK(k_xor_b32, U8, REP8(X_XOR), USINK)
#define X_XOR3(i) asm volatile("v_bfi_b32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b));
DEFK(k_bfi_b32, U8, REP8(X_XOR3), USINK)
#define X_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
DEFK(k_mul_lo_u32, U8, REP8(X_MULLO), USINK)
#define X_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
DEFK(k_mul_hi_u32, U8, REP8(X_MULHI), USINK)
#define X_ADD32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
DEFK(k_add_u32, U8, REP8(X_ADD32), USINK)
#define X_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b) : "vcc");
DEFK(k_cndmask_b32, U8, REP8(X_CND), USINK)
#define X_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(a##i));
DEFK(k_alignbit_b32, U8, REP8(X_ALIGN), USINK)
#define X_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a##i) : "v"(b));
DEFK(k_mov_b32, U8, REP8(X_MOV), USINK)
#define X_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a##i) : "v"(b));
DEFK(k_mul_u32_u24, U8, REP8(X_MUL24), USINK)
#define X_MADU24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a##i) : "v"(b));
DEFK(k_mad_u32_u24, U8, REP8(X_MADU24), USINK)
#define X_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b));
DEFK(k_fma_f32, U8, REP8(X_FMA32), USINK)
#define X_PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a##i));
DEFK(k_pk_fma_f32, L8, REP8(X_PKFMA32), LSINK)

typedef void (*kern_t)(unsigned long long*, double);
struct entry { const char* name; kern_t k; };

int main() {
  entry tab[] = {{"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_max_f64", k_max_f64},
                 {"v_rcp_f64", k_rcp_f64}, {"v_rsq_f64", k_rsq_f64}, {"v_sqrt_f64", k_sqrt_f64}, {"v_rndne_f64", k_rndne_f64},
                 {"v_ldexp_f64", k_ldexp_f64}, {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_mad_u64_u32", k_mad_u64_u32},
                 {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mul_u32_u24", k_mul_u32_u24},
                 {"v_mad_u32_u24", k_mad_u32_u24}, {"v_xor_b32", k_xor_b32}, {"v_bfi_b32", k_bfi_b32}, {"v_add_u32", k_add_u32},
                 {"v_cndmask_b32", k_cndmask_b32}, {"v_alignbit_b32", k_alignbit_b32}, {"v_mov_b32", k_mov_b32},
                 {"v_fma_f32", k_fma_f32}, {"v_pk_fma_f32", k_pk_fma_f32}};
  unsigned long long* d;
  hipMalloc(&d, 256 * 16 * sizeof(unsigned long long));
  unsigned long long h[256 * 16];
  printf("%-16s %10s %10s %10s   (shader cycles per wave-instruction, per SIMD = cycles / (waves/SIMD) below)\n", "instruction", "1 w/SIMD", "2 w/SIMD", "3 w/SIMD");
  for (auto& e : tab) {
    printf("%-16s", e.name);
    for (int wps = 1; wps <= 3; ++wps) {
      const int threads = 256 * wps, waves = 4 * wps;
      for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, d, 1.0);
      hipDeviceSynchronize();
      hipMemcpy(h, d, 256 * waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      double avg = 0;
      for (int i = 0; i < 256 * waves; ++i) avg += (double)h[i];
      avg /= 256.0 * waves;
      printf(" %10.2f", avg / (256.0 * 8.0) / wps);       // cycles the SIMD spends per wave-instruction
    }
    printf("\n");
  }
  hipFree(d);
  return 0;
}
