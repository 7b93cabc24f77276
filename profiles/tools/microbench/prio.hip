// VALU stream beside an MFMA stream on the same SIMD: role assignment (which wave is older) and s_setprio
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int SHAPE, int VPRIO, int MPRIO>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* cyc, int iters, int v_is_low) {
  const int wave = threadIdx.x >> 6;
  const bool isV = v_is_low ? wave < 4 : wave >= 4;
  float v[8], c1 = 0.999f, c2 = 0.001f;
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  half8 a8, b8; for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(i * 0.1f); b8[i] = (_Float16)(0.3f); }
  floatx4 acc[4] = {}; floatx16 acc2[2] = {};
  __syncthreads();
  const long long t0 = clock64();
  if (!isV) {
    __builtin_amdgcn_s_setprio(MPRIO);
    for (int it = 0; it < iters; ++it) {
      if (SHAPE == 0) {
#pragma unroll
        for (int u = 0; u < 200; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[u & 3], 0, 0, 0);
      } else {
#pragma unroll
        for (int u = 0; u < 100; ++u) acc2[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc2[u & 1], 0, 0, 0);
      }
    }
  } else {
    __builtin_amdgcn_s_setprio(VPRIO);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
      }
    }
  }
  const long long t1 = clock64();
  float r = 0;
  for (int i = 0; i < 8; ++i) r += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = r + acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] + acc2[0][0] + acc2[1][0];
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int SHAPE, int VPRIO, int MPRIO>
void run() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 100;
  for (int vlow = 0; vlow < 2; ++vlow) {
    for (int rep = 0; rep < 2; ++rep) k<SHAPE, VPRIO, MPRIO><<<256, 512>>>(out, cyc, iters, vlow);
    (void)hipDeviceSynchronize();
    long long h[256 * 8];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mv = 0, mm = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (((w < 4) == (vlow == 1)) ? mv : mm) += h[b * 8 + w];
    printf("%s  V prio %d M prio %d  V on waves %s: V %6.2f cyc/instr   M %5.0f cycles per 3200 of matrix work\n", SHAPE ? "32x32x16" : "16x16x32",
           VPRIO, MPRIO, vlow ? "0-3" : "4-7", mv / 1024 / iters / 256, mm / 1024 / iters);
  }
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0, 0, 0>(); run<1, 0, 0>();
  run<0, 3, 0>(); run<1, 3, 0>();
  run<0, 0, 3>(); run<1, 0, 3>();
  return 0;
}
