// 1x1 convolution of an NCHW batch with its neighbouring bias adds and LeakyReLUs in ONE pass:
//     y[b, o, p] = act_out( sum_c W[o, c] act_in(x[b, c, p] + in_bias[c]) + out_bias[o] )
// This is the middle layer of Glow's coupling conditioner (nets/cnn.py:20-52: Conv3x3, LeakyReLU, Conv1x1, LeakyReLU,
// Conv3x3; flows/affine/glow.py:37-47) together with the bias + LeakyReLU that follow the first convolution and the
// bias + LeakyReLU that follow the 1x1 convolution: the caller runs the first 3x3 convolution WITHOUT its bias, this
// kernel applies that bias and activation on load and its own on store, and the hidden activations (256 channels:
// 64 x the coupling's own data) cross HBM twice instead of ten times (library convolution + 2 bias kernels + 2
// activation kernels, each a read and a write).
//
// A dense contraction over the channels, so it runs on the matrix cores: v_mfma_f32_32x32x16_f16 with both operands
// split into hi + lo * 2^-11 fp16 halves (22 significant bits, 3 instructions per product, fp32 accumulation - the
// scheme of fused_layer_v6.hip; values beyond +-65504 are clamped and counted in `sat`).
//   * workgroup = 8 waves; wave w owns output channels 32 w .. 32 w + 31 and keeps ITS weight fragments (hi and lo for
//     every k-step: <= 128 registers) for the whole launch - no weight traffic after the prologue;
//   * a pass handles 64 pixels (two 32-column blocks): every thread loads 8 consecutive input channels of one pixel
//     (lanes = consecutive pixels: coalesced rows), applies bias + activation, splits, and writes one 16-byte
//     B-fragment entry (hi) and one (lo) into LDS; after a barrier every wave runs, per 32-pixel column block, three
//     independent accumulator chains (hi x hi, hi x lo, lo x hi) over the k-steps on the LDS fragments and stores its
//     32 channels x 32 pixels (128-byte rows).
// C_in multiple of 16 up to 256, C_out up to 256.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"
#include "fused_common.hpp"

namespace vcnf {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct Conv1Args {
  const float* x;
  float* y;
  const uint4* wfrag;      // [8 row blocks][KS][hi | lo][64 lanes] 16-byte fragments
  const float* in_bias;    // [C_in] or null
  const float* out_bias;   // [C_out] or null
  long long npix, inner;
  int Cin, Cout, in_act, out_act;
  int inner_shift;         // log2(inner) when inner is a power of two, else -1
  float in_slope, out_slope;
  int32_t* sat;
};

constexpr int kC1Block = 512;
constexpr int kC1Pix = 64;

template <int KS>
__global__ __launch_bounds__(kC1Block, 2) void conv1x1_f16x3_kernel(const Conv1Args a) {
  extern __shared__ __align__(16) uint4 bfrag[];        // [KS][2 column blocks][hi | lo][64 lanes]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool active = wave * 32 < a.Cout;
  half8 wh[KS], wl[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    wh[ks] = __builtin_bit_cast(half8, a.wfrag[((wave * KS + ks) * 2 + 0) * 64 + lane]);
    wl[ks] = __builtin_bit_cast(half8, a.wfrag[((wave * KS + ks) * 2 + 1) * 64 + lane]);
  }
  const int px = tid & 63;            // loader role: pixel of the pass, channel groups kq, kq + 8, ...
  const int kq = tid >> 6;
  float satm = 0.f;
  const long long ntiles = (a.npix + kC1Pix - 1) / kC1Pix;
  // a thread's share of a pass: NIT groups of 8 input channels of its pixel, requested two groups at a time (16 rows
  // in flight per thread; all four at once, or the next pass during the matrix phase, spill 48 / 84 registers at
  // C_in = 256 on top of the 128 weight registers)
  constexpr int NIT = (2 * KS + 7) / 8;
  constexpr int NCH = NIT > 2 ? 2 : NIT;             // groups requested together
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long g = tile * kC1Pix + px;
    const bool ok = g < a.npix;
    const long long b = ok ? (a.inner_shift >= 0 ? (g >> a.inner_shift) : g / a.inner) : 0;
    const float* src = a.x + b * a.Cin * a.inner + (ok ? g - b * a.inner : 0);
    __syncthreads();                 // the previous pass's fragments are consumed
#pragma unroll 1
    for (int it0 = 0; it0 < NIT; it0 += NCH) {
      float v[NCH][8];
#pragma unroll
      for (int u = 0; u < NCH; ++u) {
        const int cg = kq + 8 * (it0 + u);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[u][i] = (ok && cg < 2 * KS) ? src[(long long)(8 * cg + i) * a.inner] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < NCH; ++u) {
        const int cg = kq + 8 * (it0 + u);
        if (it0 + u < NIT && cg < 2 * KS) {
          half8 hi, lo;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float t = v[u][i];
            if (a.in_bias) t += a.in_bias[8 * cg + i];
            if (a.in_act) t = t >= 0.f ? t : t * a.in_slope;
            satm = fmaxf(satm, __builtin_fabsf(t));
            t = __builtin_amdgcn_fmed3f(t, -65504.f, 65504.f);
            const _Float16 h = (_Float16)t;
            hi[i] = h;
            lo[i] = (_Float16)((t - (float)h) * kLoScale);
          }
          const int ks = cg >> 1, ln = 32 * (cg & 1) + (px & 31), ct = px >> 5;
          bfrag[((ks * 2 + ct) * 2 + 0) * 64 + ln] = __builtin_bit_cast(uint4, hi);
          bfrag[((ks * 2 + ct) * 2 + 1) * 64 + ln] = __builtin_bit_cast(uint4, lo);
        }
      }
    }
    __syncthreads();
    if (active) {
      // one 32-pixel column block at a time: three independent accumulator chains (main, hi x lo, lo x hi)
#pragma unroll 1
      for (int ct = 0; ct < 2; ++ct) {
        floatx16 mainv, ca, cb;
#pragma unroll
        for (int r = 0; r < 16; ++r) {       // the bias is the accumulator's start value (its loads hide behind the k-steps)
          const int row = 32 * wave + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
          mainv[r] = (a.out_bias && row < a.Cout) ? a.out_bias[row] : 0.f;
          ca[r] = 0.f;
          cb[r] = 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const half8 bh = __builtin_bit_cast(half8, bfrag[((ks * 2 + ct) * 2 + 0) * 64 + lane]);
          const half8 bl = __builtin_bit_cast(half8, bfrag[((ks * 2 + ct) * 2 + 1) * 64 + lane]);
          mainv = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], bh, mainv, 0, 0, 0);
          ca = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], bl, ca, 0, 0, 0);
          cb = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks], bh, cb, 0, 0, 0);
        }
        const long long g = tile * kC1Pix + 32 * ct + (lane & 31);
        if (g < a.npix) {
          const long long b = a.inner_shift >= 0 ? (g >> a.inner_shift) : g / a.inner;
          float* dst = a.y + b * a.Cout * a.inner + (g - b * a.inner);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = 32 * wave + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
            if (row < a.Cout) {
              float t = fmaf(ca[r] + cb[r], kLoUnscale, mainv[r]);
              if (a.out_act) t = t >= 0.f ? t : t * a.out_slope;
              dst[(long long)row * a.inner] = t;
            }
          }
        }
      }
    }
  }
  if (a.sat && satm > 65504.f) atomicAdd(a.sat, 1);
}

// Same computation with the raw rows of the NEXT pass travelling straight from global memory into LDS
// (buffer_load ... lds: no registers) while the current pass is in its matrix phase.  With the weights in registers
// only one 8-wave workgroup fits a CU, so in the kernel above a pass is: wait for its rows, convert, matrix
// instructions, stores - nothing overlaps (15 us per pass at 256 -> 256 channels).  Here a pass is: [barrier] convert
// raw -> fragments (LDS -> LDS), [barrier] request the next pass's rows, matrix instructions + stores.
// Needs inner % 4 == 0 and 16-byte aligned x (a lane moves 4 consecutive pixels of one channel).
template <int KS>
__global__ __launch_bounds__(kC1Block, 2) void conv1x1_f16x3_dma_kernel(const Conv1Args a) {
  extern __shared__ __align__(16) uint4 smem[];
  uint4* bfrag = smem;                                               // [KS][2 column blocks][hi | lo][64 lanes]
  float* raw = reinterpret_cast<float*>(smem + KS * 2 * 2 * 64);     // [16 KS channels][64 pixels]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool active = wave * 32 < a.Cout;
  half8 wh[KS], wl[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    wh[ks] = __builtin_bit_cast(half8, a.wfrag[((wave * KS + ks) * 2 + 0) * 64 + lane]);
    wl[ks] = __builtin_bit_cast(half8, a.wfrag[((wave * KS + ks) * 2 + 1) * 64 + lane]);
  }
  const int px = tid & 63;
  const int kq = tid >> 6;
  float satm = 0.f;
  const long long ntiles = (a.npix + kC1Pix - 1) / kC1Pix;
  const long long total = a.npix * a.Cin;                            // floats of x
  // request of a pass: wave w moves channels 4 w + 32 j .. + 3 (j < KS / 2), lane = (channel of the four, 4 pixels)
#define VCNF_C1_DMA(TILE)                                                                            \
  {                                                                                                  \
    const long long g0_ = (TILE) * kC1Pix;                                                           \
    const long long b0_ = a.inner_shift >= 0 ? (g0_ >> a.inner_shift) : g0_ / a.inner;                 \
    const long long base_ = b0_ * a.Cin * a.inner;              /* first image of the pass */        \
    const long long left_ = (total - base_) * 4;                                                     \
    const __amdgpu_buffer_rsrc_t xr_ = __builtin_amdgcn_make_buffer_rsrc(                            \
        const_cast<float*>(a.x) + base_, 0, (int)(left_ < 0x7fffffffLL ? left_ : 0x7fffffffLL), 0x00020000); \
    const long long g_ = g0_ + 4 * (lane & 15);                                                      \
    const long long bb_ = a.inner_shift >= 0 ? (g_ >> a.inner_shift) : g_ / a.inner;                   \
    const long long off_ = (bb_ - b0_) * a.Cin * a.inner + (g_ - bb_ * a.inner);                     \
    const bool in_ = g_ < a.npix;                                                                    \
    for (int j = 0; j < KS / 2 + (KS & 1); ++j) {                                                    \
      const int c0_ = 32 * j + 4 * wave;                                                             \
      if (c0_ < 16 * KS) {                                                                           \
        const long long e_ = off_ + (long long)(c0_ + (lane >> 4)) * a.inner;                        \
        dma16_to_lds(xr_, raw + c0_ * 64, in_ ? (int)(e_ * 4) : 0x7ffffff0, 0);                      \
      }                                                                                              \
    }                                                                                                \
  }
  if ((long long)blockIdx.x < ntiles) VCNF_C1_DMA((long long)blockIdx.x)
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    wait_vector_memory();            // this wave's rows of the pass have landed in LDS ...
    __syncthreads();                 // ... and everyone's; the previous pass's fragments are consumed
#pragma unroll 1
    for (int cg = kq; cg < 2 * KS; cg += 8) {
      half8 hi, lo;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float t = raw[(8 * cg + i) * 64 + px];
        if (a.in_bias) t += a.in_bias[8 * cg + i];
        if (a.in_act) t = t >= 0.f ? t : t * a.in_slope;
        satm = fmaxf(satm, __builtin_fabsf(t));
        t = __builtin_amdgcn_fmed3f(t, -65504.f, 65504.f);
        const _Float16 h = (_Float16)t;
        hi[i] = h;
        lo[i] = (_Float16)((t - (float)h) * kLoScale);
      }
      const int ks = cg >> 1, ln = 32 * (cg & 1) + (px & 31), ct = px >> 5;
      bfrag[((ks * 2 + ct) * 2 + 0) * 64 + ln] = __builtin_bit_cast(uint4, hi);
      bfrag[((ks * 2 + ct) * 2 + 1) * 64 + ln] = __builtin_bit_cast(uint4, lo);
    }
    __syncthreads();                 // fragments complete, raw rows consumed
    if (tile + gridDim.x < ntiles) VCNF_C1_DMA(tile + gridDim.x)
    if (active) {
#pragma unroll 1
      for (int ct = 0; ct < 2; ++ct) {
        floatx16 mainv, ca, cb;
#pragma unroll
        for (int r = 0; r < 16; ++r) {       // the bias is the accumulator's start value (its loads hide behind the k-steps)
          const int row = 32 * wave + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
          mainv[r] = (a.out_bias && row < a.Cout) ? a.out_bias[row] : 0.f;
          ca[r] = 0.f;
          cb[r] = 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const half8 bh = __builtin_bit_cast(half8, bfrag[((ks * 2 + ct) * 2 + 0) * 64 + lane]);
          const half8 bl = __builtin_bit_cast(half8, bfrag[((ks * 2 + ct) * 2 + 1) * 64 + lane]);
          mainv = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], bh, mainv, 0, 0, 0);
          ca = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], bl, ca, 0, 0, 0);
          cb = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks], bh, cb, 0, 0, 0);
        }
        const long long g = tile * kC1Pix + 32 * ct + (lane & 31);
        if (g < a.npix) {
          const long long b = a.inner_shift >= 0 ? (g >> a.inner_shift) : g / a.inner;
          float* dst = a.y + b * a.Cout * a.inner + (g - b * a.inner);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = 32 * wave + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
            if (row < a.Cout) {
              float t = fmaf(ca[r] + cb[r], kLoUnscale, mainv[r]);
              if (a.out_act) t = t >= 0.f ? t : t * a.out_slope;
              dst[(long long)row * a.inner] = t;
            }
          }
        }
      }
    }
  }
#undef VCNF_C1_DMA
  if (a.sat && satm > 65504.f) atomicAdd(a.sat, 1);
}

template <int KS>
static int launch_conv1(const Conv1Args& a, hipStream_t st) {
  const bool dma = (a.inner % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0);
  const size_t lds = (size_t)KS * 2 * 2 * 64 * 16 + (dma ? (size_t)KS * 16 * 64 * 4 : 0);
  static bool attr_set[2] = {false, false};
  if (!attr_set[dma]) {
    const void* fn = dma ? reinterpret_cast<const void*>(&conv1x1_f16x3_dma_kernel<KS>)
                         : reinterpret_cast<const void*>(&conv1x1_f16x3_kernel<KS>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VCNF_ERR_LAUNCH;
    attr_set[dma] = true;
  }
  const long long ntiles = (a.npix + kC1Pix - 1) / kC1Pix;
  const long long cap = 256;                 // one 8-wave workgroup per CU (128 weight registers per lane)
  dim3 grid((unsigned)(ntiles < cap ? ntiles : cap));
  if (dma)
    hipLaunchKernelGGL((conv1x1_f16x3_dma_kernel<KS>), grid, dim3(kC1Block), lds, st, a);
  else
    hipLaunchKernelGGL((conv1x1_f16x3_kernel<KS>), grid, dim3(kC1Block), lds, st, a);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

}  // namespace vcnf

using namespace vcnf;

extern "C" int vcnf_conv1x1_supported(int32_t c_in, int32_t c_out) {
  return (c_in >= 16 && c_in <= 256 && c_in % 16 == 0 && c_out >= 1 && c_out <= 256) ? 1 : 0;
}

extern "C" int64_t vcnf_conv1x1_pack_floats(int32_t c_in, int32_t c_out) {
  if (!vcnf_conv1x1_supported(c_in, c_out)) return 0;
  return (int64_t)8 * (c_in / 16) * 2 * 64 * 4;
}

extern "C" int vcnf_conv1x1_f16x3_f32(const float* x, float* y, const float* wpack, int64_t wpack_floats,
                                      const float* in_bias, const float* out_bias, int64_t batch, int32_t c_in,
                                      int32_t c_out, int64_t inner, int in_act, float in_slope, int out_act,
                                      float out_slope, int32_t* sat_count, void* stream) {
  if (!vcnf_conv1x1_supported(c_in, c_out)) return VCNF_ERR_UNSUPPORTED;
  if (batch < 0 || inner < 1) return VCNF_ERR_SHAPE;
  if (wpack_floats != vcnf_conv1x1_pack_floats(c_in, c_out)) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!x || !y || !wpack) return VCNF_ERR_NULL;
  if (reinterpret_cast<uintptr_t>(wpack) & 15) return VCNF_ERR_ALIGN;
  Conv1Args a;
  a.x = x; a.y = y; a.wfrag = reinterpret_cast<const uint4*>(wpack); a.in_bias = in_bias; a.out_bias = out_bias;
  a.npix = batch * inner; a.inner = inner; a.Cin = c_in; a.Cout = c_out;
  a.in_act = in_act ? 1 : 0; a.out_act = out_act ? 1 : 0; a.in_slope = in_slope; a.out_slope = out_slope;
  a.sat = sat_count;
  a.inner_shift = -1;
  for (int sh = 0; sh < 40; ++sh) if (((long long)1 << sh) == inner) a.inner_shift = sh;
  hipStream_t st = (hipStream_t)stream;
  switch (c_in / 16) {
    case 1: return launch_conv1<1>(a, st);
    case 2: return launch_conv1<2>(a, st);
    case 3: return launch_conv1<3>(a, st);
    case 4: return launch_conv1<4>(a, st);
    case 5: return launch_conv1<5>(a, st);
    case 6: return launch_conv1<6>(a, st);
    case 7: return launch_conv1<7>(a, st);
    case 8: return launch_conv1<8>(a, st);
    case 9: return launch_conv1<9>(a, st);
    case 10: return launch_conv1<10>(a, st);
    case 11: return launch_conv1<11>(a, st);
    case 12: return launch_conv1<12>(a, st);
    case 13: return launch_conv1<13>(a, st);
    case 14: return launch_conv1<14>(a, st);
    case 15: return launch_conv1<15>(a, st);
    case 16: return launch_conv1<16>(a, st);
    default: return VCNF_ERR_UNSUPPORTED;
  }
}
