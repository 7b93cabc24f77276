// Micro-benchmark: how much VALU work of wave B co-issues beside back-to-back MFMAs of wave A on the same SIMD,
// for v_mfma_f32_16x16x32_f16 (16 cycles) vs v_mfma_f32_32x32x16_f16 (32 cycles) at equal flop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int MODE>   // SHAPE 0: 16x16x32, 1: 32x32x16 ; MODE 1: M only, 2: V only, 3: both, 4: V on all 8 waves, 5: M on all 8 waves
__global__ __launch_bounds__(512, 2) void k(float* out, long long* cyc, int iters, int nv) {
  const int wave = threadIdx.x >> 6;
  const bool isM = (MODE == 5) || (MODE != 4 && wave < 4);
  const bool act = (MODE == 3 || MODE == 4 || MODE == 5) || (MODE == 1 && isM) || (MODE == 2 && !isM);
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i * 0.01f); }
  float r = 0.f;
  __syncthreads();
  const long long t0 = clock64();
  if (act) {
    if (isM) {
      if (SHAPE == 0) {
        floatx4 c[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
          for (int u = 0; u < 144; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[u & 3], 0, 0, 0);
        }
        r = c[0][0] + c[1][1] + c[2][2] + c[3][3];
      } else {
        floatx16 c[2] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
          for (int u = 0; u < 72; ++u) c[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[u & 1], 0, 0, 0);
        }
        r = c[0][0] + c[1][5];
      }
    } else {
      float v[8]; float c1 = 0.999f, c2 = 0.001f;
      for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
      for (int it = 0; it < iters * (nv / 36); ++it) {
#pragma unroll
        for (int u = 0; u < 36; ++u) {
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
        }
      }
      for (int i = 0; i < 8; ++i) r += v[i];
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int SHAPE, int MODE>
void run(const char* name, int iters, int nv) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  for (int rep = 0; rep < 2; ++rep) k<SHAPE, MODE><<<256, 512>>>(out, cyc, iters, nv);
  hipDeviceSynchronize();
  long long h[256 * 8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0, v = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += h[b * 8 + w];
  printf("%-34s iters %d nv %d: waves0-3 %8.0f cyc/iter  waves4-7 %8.0f cyc/iter\n", name, iters, nv, m / 1024 / iters, v / 1024 / iters);
  hipFree(out); hipFree(cyc);
}

int main() {
  const int it = 200;
  // one "iteration" = 144 x 16x16x32 (2304 cycles) or 72 x 32x32x16 (2304 cycles); V: nv*8 fma
  run<0, 1>("16x16x32 M only", it, 0);
  run<1, 1>("32x32x16 M only", it, 0);
  run<0, 5>("16x16x32 M on all 8 waves", it, 0);
  run<1, 5>("32x32x16 M on all 8 waves", it, 0);
  for (int nv : {36, 72}) {          // 288 / 576 VALU per iteration
    run<0, 2>("V only (4 waves)", it, nv);
    run<0, 4>("V on all 8 waves", it, nv);
    run<0, 3>("16x16x32 M || V", it, nv);
    run<1, 3>("32x32x16 M || V", it, nv);
  }
  return 0;
}
