// Fused elementwise pieces of a residual block of the conditioner in the TRAINING path (nets/resnet.py:38-57 and its
// autograd): PyTorch runs every ReLU, sigmoid, product and sum of the block and of its backward as a kernel of its own
// (about fourteen [B, 128] passes per block and step; 30 % of the C3 training step after the weight-gradient kernel).
// vcnf_amd/autograd.py::ResBlockFn keeps the block's GEMMs on the library / csrc/linear_wgrad.hip and does the rest with
// four maps:
//   op 0  gate forward          out0 = a + b * sigmoid(c)                        (a block input, b second Linear, c gate logits)
//   op 1  gate backward         out0 = a * s,  out1 = a * b * s * (1 - s),  s = sigmoid(c)     (a upstream gradient)
//   op 2  ReLU backward         out0 = a * (b > 0)                               (b the ReLU's output)
//   op 3  ReLU backward + skip  out0 = c + a * (b > 0)                           (c gradient of the skip connection)
// One thread per four elements (16-byte accesses) when the buffers allow it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"

namespace vcnf {

struct ResOpArgs {
  const float *a, *b, *c;
  float *o0, *o1;
  long long n;
};

__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + expf(-v)); }   // torch.sigmoid in fp32 (resnet.py:54-56); the fast __expf form was ~2 ulp off

template <int OP>
__device__ __forceinline__ void res_op(float a, float b, float c, float& o0, float& o1) {
  if (OP == 0) {
    o0 = a + b * sigm(c);
  } else if (OP == 1) {
    const float s = sigm(c);
    o0 = a * s;
    o1 = a * b * s * (1.f - s);
  } else if (OP == 2) {
    o0 = b > 0.f ? a : 0.f;
  } else {
    o0 = c + (b > 0.f ? a : 0.f);
  }
}

template <int OP, bool VEC>
__global__ __launch_bounds__(256) void resblock_op_kernel(const ResOpArgs r) {
  const long long stride = (long long)gridDim.x * 256;
  if (VEC) {
    const long long n4 = r.n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const float4 a = reinterpret_cast<const float4*>(r.a)[i];
      const float4 b = reinterpret_cast<const float4*>(r.b)[i];
      const float4 c = (OP == 2) ? a : reinterpret_cast<const float4*>(r.c)[i];
      float4 o0, o1 = make_float4(0.f, 0.f, 0.f, 0.f);
      res_op<OP>(a.x, b.x, c.x, o0.x, o1.x);
      res_op<OP>(a.y, b.y, c.y, o0.y, o1.y);
      res_op<OP>(a.z, b.z, c.z, o0.z, o1.z);
      res_op<OP>(a.w, b.w, c.w, o0.w, o1.w);
      reinterpret_cast<float4*>(r.o0)[i] = o0;
      if (OP == 1) reinterpret_cast<float4*>(r.o1)[i] = o1;
    }
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < r.n; i += stride) {
      float o0, o1 = 0.f;
      res_op<OP>(r.a[i], r.b[i], OP == 2 ? 0.f : r.c[i], o0, o1);
      r.o0[i] = o0;
      if (OP == 1) r.o1[i] = o1;
    }
  }
}

template <int OP>
static void launch_res(const ResOpArgs& r, bool vec, hipStream_t st) {
  long long work = vec ? (r.n >> 2) : r.n;
  long long blocks = (work + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  if (vec) hipLaunchKernelGGL((resblock_op_kernel<OP, true>), dim3((unsigned)blocks), dim3(256), 0, st, r);
  else hipLaunchKernelGGL((resblock_op_kernel<OP, false>), dim3((unsigned)blocks), dim3(256), 0, st, r);
}

}  // namespace vcnf

using namespace vcnf;

extern "C" int vcnf_resblock_elementwise_f32(int op, const float* a, const float* b, const float* c, float* out0,
                                             float* out1, int64_t n, void* stream) {
  if (op < 0 || op > 3) return VCNF_ERR_UNSUPPORTED;
  if (n < 0) return VCNF_ERR_SHAPE;
  if (n == 0) return VCNF_OK;
  if (!a || !b || !out0 || (op != 2 && !c) || (op == 1 && !out1)) return VCNF_ERR_NULL;
  ResOpArgs r{a, b, c, out0, out1, n};
  const uintptr_t bits = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out0) |
                         (c ? reinterpret_cast<uintptr_t>(c) : 0) | (out1 ? reinterpret_cast<uintptr_t>(out1) : 0);
  const bool vec = (n % 4 == 0) && (bits & 15) == 0;
  hipStream_t st = (hipStream_t)stream;
  switch (op) {
    case 0: launch_res<0>(r, vec, st); break;
    case 1: launch_res<1>(r, vec, st); break;
    case 2: launch_res<2>(r, vec, st); break;
    default: launch_res<3>(r, vec, st); break;
  }
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}
