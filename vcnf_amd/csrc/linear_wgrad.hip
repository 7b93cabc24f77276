// Weight and bias gradient of a dense layer of the conditioner for LARGE batches:
//     dW[o, i] = sum_b dy[b, o] x[b, i]        db[o] = sum_b dy[b, o]            x [B, IN], dy [B, OUT]
// (the reference gets both from autograd over nets/resnet.py:92-106, i.e. torch.nn.functional.linear's backward).
// These are GEMMs whose reduction runs over the batch and whose output is tiny (128 x 128 for a hidden layer of
// config C3): PyTorch-ROCm hands them to a library kernel that splits the OUTPUT into 32 x 32 tiles - 16 workgroups
// on a 256-CU device, 223 us per 131 072-sample layer, 36 % of the whole training step (profiles/
// r01_train_step_kernel_stats.csv) - and the bias gradient to a separate column-sum kernel.  Here the BATCH is split:
// workgroup = (block of 128 output rows, slice of the batch); its 4 waves own two 16-row blocks each and all column
// blocks, accumulate dy^T x of their slice on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation) with both
// operands read straight from global memory in matrix-operand layout (lane = 16 q + m: sample 4 j + q of k-step j,
// column m of the block - 64-byte row segments), sum the bias gradient from the same dy values, and write one partial
// [OUT, IN] (+ [OUT]) per slice; a second small kernel adds the slices in a fixed order (deterministic).
// IN: multiple of 16 up to 128; any OUT.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"
#include "fused_common.hpp"

namespace vcnf {

struct WgradArgs {
  const float* x;
  const float* dy;
  float* part_w;      // [S][OUT][IN]
  float* part_b;      // [S][OUT] or null
  long long B, chunk; // samples per slice (multiple of 16)
  int IN, OUT, S, row_tiles;
};

constexpr int kWgBlock = 256;

template <int NB>   // IN = 16 NB
__global__ __launch_bounds__(kWgBlock) void linear_wgrad_kernel(const WgradArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m16 = lane & 15;
  const int q = lane >> 4;
  const int rt = blockIdx.x % a.row_tiles;
  const int s = blockIdx.x / a.row_tiles;
  const int o0 = rt * 128 + wave * 32;                 // this wave's 32 output rows
  const long long b_lo = (long long)s * a.chunk;
  const long long b_hi = b_lo + a.chunk < a.B ? b_lo + a.chunk : a.B;
  floatx4 acc[2][NB];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = floatx4{0.f, 0.f, 0.f, 0.f};
  float bsum[2] = {0.f, 0.f};
  const bool rok0 = o0 + m16 < a.OUT, rok1 = o0 + 16 + m16 < a.OUT;
  constexpr int U = 4;                                 // k-steps (4 samples each) requested together
  for (long long b0 = b_lo; b0 < b_hi; b0 += 4 * U) {
    float av[U][2], xv[U][NB];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long b = b0 + 4 * u + q;
      const bool ok = b < b_hi;
      const float* dyr = a.dy + b * a.OUT + o0 + m16;
      av[u][0] = (ok && rok0) ? dyr[0] : 0.f;
      av[u][1] = (ok && rok1) ? dyr[16] : 0.f;
      const float* xr = a.x + b * a.IN + m16;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) xv[u][nb] = ok ? xr[16 * nb] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bsum[0] += av[u][0];
      bsum[1] += av[u][1];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        acc[0][nb] = mfma4(av[u][0], xv[u][nb], acc[0][nb]);
        acc[1][nb] = mfma4(av[u][1], xv[u][nb], acc[1][nb]);
      }
    }
  }
  // accumulator register r of lane (q, m16): row 4 q + r of the 16-row block, column m16 of the column block
  float* pw = a.part_w + (long long)s * a.OUT * a.IN;
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = o0 + 16 * mb + 4 * q + r;
      if (row < a.OUT) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) pw[(long long)row * a.IN + 16 * nb + m16] = acc[mb][nb][r];
      }
    }
  if (a.part_b) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      float t = bsum[mb];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      const int row = o0 + 16 * mb + m16;
      if (q == 0 && row < a.OUT) a.part_b[(long long)s * a.OUT + row] = t;
    }
  }
}

struct WreduceArgs {
  const float* part;
  float* out;
  long long n;       // elements of one slice
  int S, accumulate;
};

// column sums of the [S, n] partial results: a workgroup owns 64 columns, its four waves take every fourth slice
// (256-byte row segments, eight loads in flight), LDS adds the four in a fixed order
__global__ __launch_bounds__(kWgBlock) void wgrad_reduce_kernel(const WreduceArgs a) {
  __shared__ float sm[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long long col = (long long)blockIdx.x * 64 + c;
  float t = 0.f;
  if (col < a.n) {
    int s = g;
    for (; s + 28 < a.S; s += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = a.part[(long long)(s + 4 * u) * a.n + col];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; s < a.S; s += 4) t += a.part[(long long)s * a.n + col];
  }
  sm[g][c] = t;
  __syncthreads();
  if (g == 0 && col < a.n) {
    const float r = ((sm[0][c] + sm[1][c]) + sm[2][c]) + sm[3][c];
    a.out[col] = a.accumulate ? a.out[col] + r : r;
  }
}

template <int NB>
static void launch_wgrad(const WgradArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((linear_wgrad_kernel<NB>), dim3((unsigned)(a.row_tiles * a.S)), dim3(kWgBlock), 0, st, a);
}

}  // namespace vcnf

using namespace vcnf;

extern "C" int vcnf_linear_wgrad_supported(int32_t in_features, int32_t out_features) {
  return (in_features >= 16 && in_features <= 128 && in_features % 16 == 0 && out_features >= 1) ? 1 : 0;
}

/* batch slices the kernel uses: enough workgroups (row tiles x slices) for two waves on every SIMD, at least 256
 * samples per slice (the partial results are S x OUT x IN floats: 32 MB at 128 x 128) */
extern "C" int64_t vcnf_linear_wgrad_slices(int64_t batch, int32_t in_features, int32_t out_features) {
  if (!vcnf_linear_wgrad_supported(in_features, out_features) || batch < 1) return 0;
  const long long row_tiles = (out_features + 127) / 128;
  long long s = 512 / row_tiles;          // row tiles x slices <= 512 workgroups: two per CU, no third round for a few stragglers
  const long long most = (batch + 255) / 256;
  if (s > most) s = most;
  return s < 1 ? 1 : s;
}

extern "C" int vcnf_linear_wgrad_f32(const float* x, const float* dy, float* dw, float* db, float* workspace,
                                     int64_t workspace_floats, int64_t batch, int32_t in_features,
                                     int32_t out_features, int accumulate, void* stream) {
  if (!vcnf_linear_wgrad_supported(in_features, out_features)) return VCNF_ERR_UNSUPPORTED;
  if (batch < 1) return VCNF_ERR_SHAPE;
  if (!x || !dy || !dw || !workspace) return VCNF_ERR_NULL;
  const long long S = vcnf_linear_wgrad_slices(batch, in_features, out_features);
  const long long per = (long long)out_features * in_features + (db ? out_features : 0);
  if (workspace_floats < S * per) return VCNF_ERR_SHAPE;
  WgradArgs a;
  a.x = x; a.dy = dy; a.part_w = workspace;
  a.part_b = db ? workspace + S * (long long)out_features * in_features : nullptr;
  a.B = batch; a.IN = in_features; a.OUT = out_features; a.S = (int)S;
  a.row_tiles = (out_features + 127) / 128;
  a.chunk = ((batch + S - 1) / S + 15) / 16 * 16;
  hipStream_t st = (hipStream_t)stream;
  switch (in_features / 16) {
    case 1: launch_wgrad<1>(a, st); break;
    case 2: launch_wgrad<2>(a, st); break;
    case 3: launch_wgrad<3>(a, st); break;
    case 4: launch_wgrad<4>(a, st); break;
    case 5: launch_wgrad<5>(a, st); break;
    case 6: launch_wgrad<6>(a, st); break;
    case 7: launch_wgrad<7>(a, st); break;
    case 8: launch_wgrad<8>(a, st); break;
    default: return VCNF_ERR_UNSUPPORTED;
  }
  WreduceArgs r;
  r.part = a.part_w; r.out = dw; r.n = (long long)out_features * in_features; r.S = (int)S; r.accumulate = accumulate ? 1 : 0;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((r.n + 63) / 64)), dim3(kWgBlock), 0, st, r);
  if (db) {
    r.part = a.part_b; r.out = db; r.n = out_features;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((r.n + 63) / 64)), dim3(kWgBlock), 0, st, r);
  }
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}
