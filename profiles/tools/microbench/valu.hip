// cycles per VALU instruction, one wave per SIMD vs two waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int OP>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* cyc, int iters, int nwaves, int mm) {
  const int wave = threadIdx.x >> 6;
  unsigned long long msk = 0x5555555555555555ull + blockIdx.x; float2v v[8], c1 = {0.999f, 0.998f}, c2 = {0.001f, 0.002f};
  for (int i = 0; i < 8; ++i) { v[i][0] = threadIdx.x * 0.01f + i; v[i][1] = threadIdx.x * 0.02f + i; }
  half8 a8, b8; for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(i * 0.1f); b8[i] = (_Float16)(0.3f); }
  floatx4 acc[4] = {}; floatx16 acc2[2] = {};
  __syncthreads();
  const long long t0 = clock64();
  if (wave >= 4 && mm == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 200; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[u & 3], 0, 0, 0);
    }
  } else if (wave >= 4 && mm == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 100; ++u) acc2[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc2[u & 1], 0, 0, 0);
    }
  } else if (wave < nwaves) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i][0]) : "v"(c1[0]), "v"(c2[0]));
          if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
          if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i][0]));
          if (OP == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i][0]) : "v"(c1[0]));
          if (OP == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c2));
          if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
          if (OP == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i][0]));
          if (OP == 7) asm volatile("v_log_f32 %0, %0" : "+v"(v[i][0]));
          if (OP == 8) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i][0]) : "v"(c1[0]), "v"(c2[0]));
          if (OP == 9) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[i][0]));
          if (OP == 10) asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(v[i][0]), "v"(c1[0]) : "vcc");
          if (OP == 11) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i][0]) : "v"(c1[0]));
          if (OP == 12) asm volatile("v_pk_mov_b32 %0, %1, %2" : "=v"(v[i]) : "v"(c1), "v"(c2));
          if (OP == 13) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v[i][0]) : "v"(c1[0]), "v"(c2[0]));
          if (OP == 15) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(v[i][0]) : "v"(c1[0]), "v"(c2[0]));
          if (OP == 16) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(v[i][0]) : "v"(c1[0]), "v"(c2[0]), "s"(msk));
          if (OP == 17) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i][0]) : "v"(c2[0]), "s"(msk));
          if (OP == 18) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i][0]) : "v"(c1[0]), "v"(c2[0]));
          if (OP == 19) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i][0]) : "v"(c2[0]));
          if (OP == 20) { asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(v[i][0]), "v"(c1[0]) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i][0]) : "v"(c2[0]) : "vcc"); }
          if (OP == 21) { asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(v[i][0]), "v"(c1[0]) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %1, vcc\n v_cndmask_b32 %3, %3, %1, vcc" : "+v"(v[i][0]), "+v"(v[i][1]) : "v"(c2[0]), "v"(c1[1]) : "vcc"); }
          if (OP == 14) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i][0]) : "v"(c1[0]), "v"(c2[0]));
        }
      }
    }
  }
  const long long t1 = clock64();
  float r = 0;
  for (int i = 0; i < 8; ++i) r += v[i][0] + v[i][1];
  r += acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] + acc2[0][0] + acc2[1][0];
  out[blockIdx.x * 512 + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int OP>
void run(const char* name) {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 100;
  printf("%-18s", name);
  for (int cfg = 0; cfg < 4; ++cfg) {
    const int nw = cfg == 1 ? 8 : 4, mm = cfg == 2 ? 1 : cfg == 3 ? 2 : 0;
    for (int rep = 0; rep < 2; ++rep) k<OP><<<256, 512>>>(out, cyc, iters, nw, mm);
    (void)hipDeviceSynchronize();
    long long h[256 * 8];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0, mo = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < nw ? m : mo) += h[b * 8 + w];
    printf("  %s %6.2f", cfg == 0 ? "alone" : cfg == 1 ? "2w/SIMD" : cfg == 2 ? "|| mfma16x16x32" : "|| mfma32x32x16", m / 256 / nw / iters / 256);
    if (mm) printf(" (M %5.0f/3200)", mo / 256 / 4 / iters);
  }
  printf("\n");
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); run<4>("v_pk_add_f32"); run<5>("v_pk_mul_f32"); run<2>("v_exp_f32"); run<6>("v_rcp_f32");
  run<7>("v_log_f32"); run<3>("v_cndmask_b32"); run<8>("v_max3_f32"); run<9>("v_cvt_f16_f32"); run<10>("v_cmp_ge_f32"); run<11>("v_mov_b32");
  run<15>("cndmask indep vcc"); run<16>("cndmask indep sgpr"); run<17>("cndmask chain sgpr"); run<18>("v_add indep"); run<19>("v_add chain"); run<20>("cmp+cndmask"); run<12>("v_pk_mov_b32"); run<13>("v_cvt_pk_f16_f32"); run<14>("v_med3_f32");
  return 0;
}
