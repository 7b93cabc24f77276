// The fp16 split-half ("fp16x3") operand arithmetic shared by the fused RQS layer kernel (fused_layer_v6.hip) and
// the GEMM-level probe (gemm_probe.hip): an fp32 value v travels as hi = fp16(v), lo = fp16((v - hi) * 2^11); a
// product keeps hi*hi + (hi*lo + lo*hi) * 2^-11 in two fp32 accumulators on v_mfma_f32_32x32x16_f16.
// Reference arithmetic this stands in for: fp32 nn.Linear, normflow/nets/resnet.py:92-106.
#pragma once
#include <hip/hip_runtime.h>

#include "fused_common.hpp"

namespace vcnf {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ floatx16 mfma32h(half8 a, half8 b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// hi / lo halves of eight values; the running maximum of what was clamped goes to ``satm``
template <bool RELU>
__device__ __forceinline__ void split8(const float (&v)[8], half8& hi, half8& lo, float& satm) {
#pragma unroll
  for (int i = 0; i < 8; i += 2)
    satm = RELU ? fmaxf(fmaxf(satm, v[i]), v[i + 1]) : fmaxf(fmaxf(satm, __builtin_fabsf(v[i])), __builtin_fabsf(v[i + 1]));
  // pin the running maximum here: left alone the compiler sinks these updates to the end of the tile and keeps
  // (spills) every value that ever went through a split until then
  asm volatile("" : "+v"(satm));
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const float x0 = __builtin_amdgcn_fmed3f(v[i], RELU ? 0.f : -65504.f, 65504.f);
    const float x1 = __builtin_amdgcn_fmed3f(v[i + 1], RELU ? 0.f : -65504.f, 65504.f);
    const half2v h2 = __builtin_convertvector(float2v{x0, x1}, half2v);
    hi[i] = h2[0];
    hi[i + 1] = h2[1];
    lo[i] = (_Float16)__builtin_fmaf((float)h2[0], -kLoScale, x0 * kLoScale);
    lo[i + 1] = (_Float16)__builtin_fmaf((float)h2[1], -kLoScale, x1 * kLoScale);
  }
}

}  // namespace vcnf
