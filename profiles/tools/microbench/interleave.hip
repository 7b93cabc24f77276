// Micro-benchmark (round 3): MFMA and VALU interleaved INSIDE one wave - the structure asked for in VERDICT r2
// item 2 - instead of a matrix wave beside a vector wave.  Every wave of a 512-thread workgroup (two waves per
// SIMD) or of a 256-thread one (one wave per SIMD) runs
//     loop { v_mfma_f32_32x32x16_f16 ; NV x v_fma_f32 (8 independent chains) }
// with the order pinned by asm volatile.  Printed: shader cycles per loop iteration and wave.  The matrix pipe
// needs 32 cycles per instruction and wave, i.e. 64 per iteration with two waves per SIMD: if the time stays at
// that floor while NV grows, the vector instructions are free.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NV, int TRANS, int THREADS, int CH = 8>
__global__ __launch_bounds__(THREADS) void k(float* out, long long* cyc, int iters) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i * 0.01f); }
  floatx16 c0 = {}, c1 = {};
  float v[8]; float k1 = 0.999f, k2 = 0.001f;
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
      else       asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (TRANS && i == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(u + i) & 7]));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(u * NV + i) % CH]) : "v"(k1), "v"(k2));
      }
    }
  }
  const long long t1 = clock64();
  float r = c0[0] + c1[3];
  for (int i = 0; i < 8; ++i) r += v[i];
  out[blockIdx.x * THREADS + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NV, int TRANS, int THREADS, int CH = 8>
void run() {
  float* out; long long* cyc;
  const int iters = 4000, W = THREADS / 64;
  hipMalloc(&out, 256 * THREADS * 4); hipMalloc(&cyc, 256 * W * 8);
  k<NV, TRANS, THREADS, CH><<<256, THREADS>>>(out, cyc, iters);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<NV, TRANS, THREADS, CH><<<256, THREADS>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[256 * 16];
  hipMemcpy(h, cyc, 256 * W * 8, hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 256 * W; ++i) m += h[i];
  const double flop = 256.0 * W * iters * 8 * 32768.0;
  printf("chains %d  waves/SIMD %d  VALU per MFMA %2d (%s): %7.1f cycles per (MFMA + VALU group) and wave; wall %.3f ms = %.0f TFLOP/s of f16 MFMA, clock %.2f GHz\n", CH, W / 4, NV,
         TRANS ? "first one v_exp_f32" : "all v_fma_f32", m / (256.0 * W) / iters / 8, ms, flop / ms * 1e-9,
         m / (256.0 * W) / (ms * 1e6));
  hipFree(out); hipFree(cyc);
}

int main() {
  // dependent chains: the same 6 (or 8) vector instructions per MFMA as 1, 2, 4 or 8 independent chains
  run<6, 0, 256, 1>(); run<6, 0, 256, 2>(); run<6, 0, 256, 3>(); run<6, 0, 256, 4>(); run<6, 0, 256, 8>();
  run<8, 0, 256, 1>(); run<8, 0, 256, 2>(); run<8, 0, 256, 4>(); run<8, 0, 256, 8>();
  run<6, 0, 512, 1>(); run<6, 0, 512, 2>(); run<6, 0, 512, 4>(); run<6, 0, 512, 8>();
  run<8, 0, 512, 1>(); run<8, 0, 512, 2>(); run<8, 0, 512, 8>();
  run<0, 0, 512>(); run<4, 0, 512>(); run<8, 0, 512>(); run<12, 0, 512>(); run<16, 0, 512>();
  run<4, 1, 512>(); run<8, 1, 512>(); run<12, 1, 512>();
  run<0, 0, 256>(); run<4, 0, 256>(); run<6, 0, 256>(); run<8, 0, 256>();
  return 0;
}
