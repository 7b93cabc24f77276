// GEMM-level probe of the two matrix arithmetics of the fused RQS layer kernels: y = x W^T + b for ONE dense
// layer of the conditioner (reference: nn.Linear in fp32, normflow/nets/resnet.py:78-106) evaluated
//   mode VCNF_PROBE_F32      on v_mfma_f32_16x16x4_f32, accumulator started at the bias, k ascending - the chain of
//                            fused_layer.hip::dense_block (exact fp32 products, fp32 accumulation);
//   mode VCNF_PROBE_F16X3    on v_mfma_f32_32x32x16_f16 with both operands split by split_half.hpp::split8 -
//                            main += hi*hi, corr += hi*lo, corr += lo*hi per 16-deep k-step, result
//                            main + corr * 2^-11: the sequence of fused_layer_v6.hip's hidden and last layers;
//   mode VCNF_PROBE_F16X3_LL the same plus corr2 += lo*lo (result (corr2 * 2^-11 + corr) * 2^-11 + main): the
//                            sequence of fused_layer_v6.hip's first layer.
// Not on any product path: tests/test_gpu_gemm_error.py measures both arithmetics against an fp64 product with
// it.  One wave per 32 samples x 32 output rows; K % 16 == 0, N % 32 == 0.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"
#include "fused_common.hpp"
#include "split_half.hpp"

namespace vcnf {

__global__ __launch_bounds__(64) void gemm_probe_f16x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              long long B, int K, int N, int lolo, int relu,
                                                              int bias_after, int32_t* sat) {
  const int lane = threadIdx.x;
  const int c32 = lane & 31, kg = lane >> 5;
  const long long b0 = (long long)blockIdx.x * 32;
  const int n0 = blockIdx.y * 32;
  const long long brow = min(b0 + c32, B - 1);        // rows past the batch repeat the last one (never stored)
  floatx16 mainv, corr = {}, corr2 = {};
  // D layout: register r of lane l is row 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32
#pragma unroll
  for (int r = 0; r < 16; ++r) mainv[r] = (bias && !bias_after) ? bias[n0 + 8 * (r >> 2) + 4 * kg + (r & 3)] : 0.f;
  float satm = 0.f, wsat = 0.f;
  for (int t = 0; t < K / 16; ++t) {
    float av[8], bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      av[i] = w[(long long)(n0 + c32) * K + 16 * t + 8 * kg + i];     // A: lane l holds row l % 32, k-slots (l / 32, i)
      bv[i] = x[brow * K + 16 * t + 8 * kg + i];                      // B: lane l holds column l % 32, same k-slots
    }
    half8 ah, al, bh, bl;
    split8<false>(av, ah, al, wsat);
    if (relu) split8<true>(bv, bh, bl, satm); else split8<false>(bv, bh, bl, satm);
    mainv = mfma32h(ah, bh, mainv);
    corr = mfma32h(ah, bl, corr);
    corr = mfma32h(al, bh, corr);
    if (lolo) corr2 = mfma32h(al, bl, corr2);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = lolo ? fmaf(fmaf(corr2[r], kLoUnscale, corr[r]), kLoUnscale, mainv[r]) : fmaf(corr[r], kLoUnscale, mainv[r]);
    if (bias && bias_after) v += bias[n0 + 8 * (r >> 2) + 4 * kg + (r & 3)];
    if (b0 + c32 < B) y[(b0 + c32) * N + n0 + 8 * (r >> 2) + 4 * kg + (r & 3)] = v;
  }
  if (sat && satm > 65504.f) atomicAdd(sat, 1);
}

__global__ __launch_bounds__(64) void gemm_probe_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            long long B, int K, int N, int relu) {
  // v_mfma_f32_16x16x4_f32: A lane l = row l % 16, k = l / 16; B lane l = column l % 16, k = l / 16;
  // D register r of lane l = row 4 (l / 16) + r, column l % 16.  Four 16 x 16 blocks per wave.
  const int lane = threadIdx.x;
  const int m16 = lane & 15, q = lane >> 4;
  const long long b0 = (long long)blockIdx.x * 32;
  const int n0 = blockIdx.y * 32;
  floatx4 acc[2][2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[rb][cb][r] = bias ? bias[n0 + 16 * rb + 4 * q + r] : 0.f;
  for (int s = 0; s < K / 4; ++s) {
    float av[2], bv[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) av[rb] = w[(long long)(n0 + 16 * rb + m16) * K + 4 * s + q];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const float v = x[min(b0 + 16 * cb + m16, B - 1) * K + 4 * s + q];
      bv[cb] = relu ? fmaxf(v, 0.f) : v;
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma4(av[rb], bv[cb], acc[rb][cb]);
  }
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long b = b0 + 16 * cb + m16;
        if (b < B) y[b * N + n0 + 16 * rb + 4 * q + r] = acc[rb][cb][r];
      }
}

}  // namespace vcnf

extern "C" int vcnf_linear_probe_f32(const float* x, const float* weight, const float* bias, float* y, int64_t batch,
                                     int32_t in_features, int32_t out_features, int mode, int relu_input,
                                     int32_t* sat_count, void* stream) {
  if (!x || !weight || !y) return VCNF_ERR_NULL;
  if (batch <= 0 || in_features <= 0 || out_features <= 0 || in_features % 16 || out_features % 32 ||
      batch > ((int64_t)1 << 36))
    return VCNF_ERR_SHAPE;
  if (mode < VCNF_PROBE_F32 || mode > 3) return VCNF_ERR_UNSUPPORTED;   // 3: experiment - F16X3_LL with the bias added afterwards
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid((unsigned)((batch + 31) / 32), (unsigned)(out_features / 32));
  if (mode == VCNF_PROBE_F32)
    hipLaunchKernelGGL(vcnf::gemm_probe_f32_kernel, grid, dim3(64), 0, st, x, weight, bias, y, (long long)batch,
                       in_features, out_features, relu_input);
  else
    hipLaunchKernelGGL(vcnf::gemm_probe_f16x3_kernel, grid, dim3(64), 0, st, x, weight, bias, y, (long long)batch,
                       in_features, out_features, mode >= VCNF_PROBE_F16X3_LL ? 1 : 0, relu_input, mode == 3 ? 1 : 0, sat_count);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}
