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
#include "split_half.hpp"

namespace vcnf {

struct WgradArgs {
  const float* x;
  const float* dy;
  float* part_w;      // [S][OUT][IN]
  float* part_b;      // [S][OUT] or null
  long long B, chunk; // samples per slice (multiple of 16)
  int IN, OUT, S, row_tiles;
  int col_tiles;      // split-half form: workgroups per row tile (64 input columns each)
  int relu_x;         // split-half form: the layer's input is relu(x) (x = the residual stream, nets/resnet.py:42)
  int32_t* sat;       // split-half form: tiles that clamped a value at +-65504 (or NULL)
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

// The same partial products on the fp16 split-half matrix path (round 3): dy^T x of the slice on
// v_mfma_f32_32x32x16_f16 with both operands split in registers (hi*hi + (hi*lo + lo*hi) * 2^-11, fp32 accumulation -
// the arithmetic of split_half.hpp; the reduction runs over the 16 samples of a k-step, so the accumulator rounds once
// per 16 samples instead of once per 4).  Wave = 32 output rows x all column blocks of 32 inputs; a lane's eight
// k-slots are eight consecutive samples of one column, i.e. eight 4-byte loads whose 32 lanes cover a 128-byte row
// segment - no LDS, no barrier.  The exact-fp32 kernel above is bound by its matrix instructions (157 TFLOP/s peak:
// 30 us for 131 072 x 128 x 128, 157 us for the 736-row last layer); this one by the vector work of the splits
// (~10 / 65 us).  Values beyond +-65504 are clamped and counted.
template <int KB>   // column blocks of 32 inputs per workgroup (1 or 2); wider layers are split over col_tiles workgroups
__global__ __launch_bounds__(kWgBlock, 2) void linear_wgrad_f16x3_kernel(const WgradArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c32 = lane & 31, kg = lane >> 5;
  const int ct = blockIdx.x % a.col_tiles;             // column tiles are neighbours in the grid: the second one
  const int rt = (blockIdx.x / a.col_tiles) % a.row_tiles;   // finds the slice's dy rows in L2
  const int s = blockIdx.x / (a.col_tiles * a.row_tiles);
  const int o0 = rt * 128 + wave * 32;                 // this wave's 32 output rows
  const int i0 = ct * 32 * KB;                         // first input column of this workgroup
  const long long b_lo = (long long)s * a.chunk;
  const long long b_hi = b_lo + a.chunk < a.B ? b_lo + a.chunk : a.B;
  floatx16 mainv[KB], corr[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) { mainv[kb] = floatx16{}; corr[kb] = floatx16{}; }
  float bsum = 0.f, satm = 0.f;
  const bool rok = o0 + c32 < a.OUT;
  bool cok[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) cok[kb] = i0 + 32 * kb + c32 < a.IN;
  // A: lane (c32, kg) holds row o0 + c32, k-slots = samples b0 + 8 kg + i;  B: column i0 + 32 kb + c32, same samples.
  // Two k-steps (32 samples) per round, the NEXT round's values requested before this round's are split and
  // multiplied: a round is ~0.3 us of vector / matrix work against ~2 us of memory latency, and two waves per SIMD do
  // not hide it (first version, one k-step at a time: 2.8 us per k-step).
  constexpr int U = 2;
  float av[2][U][8], xv[2][U][KB][8];
  // buffer loads through a descriptor rebuilt per round (base = the round's first row, size = what is left of the
  // slice: reads past the slice return 0 by the hardware range check, which covers the per-lane offset): the per-lane
  // part is (column, sample half) plus a small row offset - as 64-bit per-lane addresses the 48 requests of a round
  // held 96 registers of addresses (178 spilled).  Lanes of rows / columns outside the layer point far past the end.
  const int voa = rok ? (8 * kg * a.OUT + o0 + c32) * 4 : 0x7FF00000;
  int vox[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) vox[kb] = cok[kb] ? (8 * kg * a.IN + i0 + 32 * kb + c32) * 4 : 0x7FF00000;
#define VCNF_WG_LOAD(BUF, B0)                                                             \
  {                                                                                       \
    const long long rb_ = min((long long)(B0), b_hi);                                     \
    const __amdgpu_buffer_rsrc_t dyr_ = __builtin_amdgcn_make_buffer_rsrc(                \
        const_cast<float*>(a.dy) + rb_ * a.OUT, 0, (unsigned)((b_hi - rb_) * a.OUT * 4), 0x00020000); \
    const __amdgpu_buffer_rsrc_t xr_ = __builtin_amdgcn_make_buffer_rsrc(                 \
        const_cast<float*>(a.x) + rb_ * a.IN, 0, (unsigned)((b_hi - rb_) * a.IN * 4), 0x00020000); \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                       \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                     \
        av[BUF][u][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dyr_, voa + (16 * u + i) * a.OUT * 4, 0, 0)); \
        _Pragma("unroll") for (int kb = 0; kb < KB; ++kb)                                 \
          xv[BUF][u][kb][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr_, vox[kb] + (16 * u + i) * a.IN * 4, 0, 0)); \
      }                                                                                   \
    }                                                                                     \
  }
#define VCNF_WG_COMPUTE(BUF)                                                              \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                         \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                       \
      bsum += av[BUF][u][i];                                                              \
      satm = fmaxf(satm, av[BUF][u][i] == av[BUF][u][i] ? 0.f : __builtin_inff());        \
    }                                                                                     \
    half8 ah, al;                                                                         \
    split8<false>(av[BUF][u], ah, al, satm);                                              \
    _Pragma("unroll") for (int kb = 0; kb < KB; ++kb) {                                   \
      _Pragma("unroll") for (int i = 0; i < 8; ++i)                                       \
        satm = fmaxf(satm, xv[BUF][u][kb][i] == xv[BUF][u][kb][i] ? 0.f : __builtin_inff()); \
      half8 bh, bl;                                                                       \
      if (a.relu_x) split8<true>(xv[BUF][u][kb], bh, bl, satm);                           \
      else split8<false>(xv[BUF][u][kb], bh, bl, satm);                                   \
      mainv[kb] = mfma32h(ah, bh, mainv[kb]);                                             \
      corr[kb] = mfma32h(ah, bl, corr[kb]);                                               \
      corr[kb] = mfma32h(al, bh, corr[kb]);                                               \
    }                                                                                     \
  }
  if (o0 < a.OUT) {
    VCNF_WG_LOAD(0, b_lo)
    for (long long b0 = b_lo; b0 < b_hi; b0 += 32 * U) {
      VCNF_WG_LOAD(1, b0 + 16 * U)                    // (past the slice: every lane reads 0)
      __builtin_amdgcn_sched_barrier(0);
      VCNF_WG_COMPUTE(0)
      __builtin_amdgcn_sched_barrier(0);
      VCNF_WG_LOAD(0, b0 + 32 * U)
      __builtin_amdgcn_sched_barrier(0);
      VCNF_WG_COMPUTE(1)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef VCNF_WG_LOAD
#undef VCNF_WG_COMPUTE
  // accumulator register r of lane (c32, kg): row 8 (r / 4) + 4 kg + r % 4 of the 32-row block, column c32 of block kb
  float* pw = a.part_w + (long long)s * a.OUT * a.IN;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = o0 + 8 * (r >> 2) + 4 * kg + (r & 3);
    if (row < a.OUT) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
        if (cok[kb]) pw[(long long)row * a.IN + i0 + 32 * kb + c32] = fmaf(corr[kb][r], kLoUnscale, mainv[kb][r]);
    }
  }
  if (a.part_b && ct == 0) {
    float t = bsum;
    t += __shfl_xor(t, 32, 64);
    if (kg == 0 && rok) a.part_b[(long long)s * a.OUT + o0 + c32] = t;
  }
  if (a.sat && satm > 65504.f) atomicAdd(a.sat, 1);
}

struct WreduceArgs {
  const float* part;
  float* out;
  long long n;       // elements of one slice
  const float* part2;   // second set of partial results reduced by the same launch (the bias gradient), or NULL
  float* out2;
  long long n2;
  int blocks1;       // workgroups of the first set
  int S, accumulate;
};

// column sums of the [S, n] partial results: a workgroup owns 64 columns, its four waves take every fourth slice
// (256-byte row segments, eight loads in flight), LDS adds the four in a fixed order.  Weight and bias partials go
// through one launch (the workgroups past blocks1 own the second set): the same sums in the same order as two
// launches, one launch gap less per dense layer of the backward pass.
__global__ __launch_bounds__(kWgBlock) void wgrad_reduce_kernel(const WreduceArgs a) {
  __shared__ float sm[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const bool second = (int)blockIdx.x >= a.blocks1;
  const float* part = second ? a.part2 : a.part;
  float* out = second ? a.out2 : a.out;
  const long long n = second ? a.n2 : a.n;
  const long long col = (long long)(second ? blockIdx.x - a.blocks1 : blockIdx.x) * 64 + c;
  float t = 0.f;
  if (col < n) {
    int s = g;
    for (; s + 28 < a.S; s += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(long long)(s + 4 * u) * n + col];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; s < a.S; s += 4) t += part[(long long)s * n + col];
  }
  sm[g][c] = t;
  __syncthreads();
  if (g == 0 && col < n) {
    const float r = ((sm[0][c] + sm[1][c]) + sm[2][c]) + sm[3][c];
    out[col] = a.accumulate ? out[col] + r : r;
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

static int wgrad_run(const float* x, const float* dy, float* dw, float* db, float* workspace, int64_t workspace_floats,
                     int64_t batch, int32_t in_features, int32_t out_features, int accumulate, int f16x3,
                     int32_t* sat_count, void* stream);

extern "C" int vcnf_linear_wgrad_f32(const float* x, const float* dy, float* dw, float* db, float* workspace,
                                     int64_t workspace_floats, int64_t batch, int32_t in_features,
                                     int32_t out_features, int accumulate, void* stream) {
  return wgrad_run(x, dy, dw, db, workspace, workspace_floats, batch, in_features, out_features, accumulate, 0, nullptr, stream);
}

extern "C" int vcnf_linear_wgrad_f16x3_f32(const float* x, const float* dy, float* dw, float* db, float* workspace,
                                           int64_t workspace_floats, int64_t batch, int32_t in_features,
                                           int32_t out_features, int accumulate, int relu_input,
                                           int32_t* sat_count, void* stream) {
  return wgrad_run(x, dy, dw, db, workspace, workspace_floats, batch, in_features, out_features, accumulate,
                   relu_input ? 2 : 1, sat_count, stream);
}

static int wgrad_run(const float* x, const float* dy, float* dw, float* db, float* workspace, int64_t workspace_floats,
                     int64_t batch, int32_t in_features, int32_t out_features, int accumulate, int f16x3,
                     int32_t* sat_count, void* stream) {
  if (!vcnf_linear_wgrad_supported(in_features, out_features)) return VCNF_ERR_UNSUPPORTED;
  if (batch < 1) return VCNF_ERR_SHAPE;
  if (!x || !dy || !dw || !workspace) return VCNF_ERR_NULL;
  const long long S = vcnf_linear_wgrad_slices(batch, in_features, out_features);
  const long long per = (long long)out_features * in_features + (db ? out_features : 0);
  if (workspace_floats < S * per) return VCNF_ERR_SHAPE;
  WgradArgs a;
  a.sat = sat_count;
  a.col_tiles = 1;
  a.relu_x = f16x3 == 2 ? 1 : 0;              // f16x3: 0 exact fp32 kernel, 1 split-half, 2 split-half on relu(x)
  a.x = x; a.dy = dy; a.part_w = workspace;
  a.part_b = db ? workspace + S * (long long)out_features * in_features : nullptr;
  a.B = batch; a.IN = in_features; a.OUT = out_features; a.S = (int)S;
  a.row_tiles = (out_features + 127) / 128;
  a.chunk = ((batch + S - 1) / S + 15) / 16 * 16;
  hipStream_t st = (hipStream_t)stream;
  // (the split-half kernel addresses its operands through 32-bit buffer descriptors)
  if (f16x3 && ((long long)batch * out_features * 4 >= (1ll << 31) || (long long)batch * in_features * 4 >= (1ll << 31))) {
    if (f16x3 == 2) return VCNF_ERR_UNSUPPORTED;     // (the exact-fp32 kernel has no ReLU on load: the caller materialises relu(x))
    f16x3 = 0;
  }
  if (f16x3) {
    a.col_tiles = (in_features + 63) / 64;
    const dim3 grid((unsigned)(a.col_tiles * a.row_tiles * a.S));
    if (in_features <= 32) hipLaunchKernelGGL((linear_wgrad_f16x3_kernel<1>), grid, dim3(kWgBlock), 0, st, a);
    else hipLaunchKernelGGL((linear_wgrad_f16x3_kernel<2>), grid, dim3(kWgBlock), 0, st, a);
  } else
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
  r.blocks1 = (int)((r.n + 63) / 64);
  r.part2 = db ? a.part_b : nullptr; r.out2 = db; r.n2 = db ? out_features : 0;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(r.blocks1 + (r.n2 + 63) / 64)), dim3(kWgBlock), 0, st, r);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}
