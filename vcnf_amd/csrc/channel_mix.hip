// Per-pixel channel mixing of an image batch in one launch:
//     y[b, o, p] = sum_c M[o, c] x[b, c, p] + v[o]            x, y [B, C, inner] (NCHW with inner = H W)
// This is Glow's invertible 1x1 convolution (flows/mixing.py:57-128) composed with the ActNorm next to it
// (flows/normalization.py:8-38 over flows/affine/coupling.py:37-53) - the two mixers between the affine couplings
// of a GlowBlock (flows/affine/glow.py:12-74).  The host composes M and v from the layer parameters (vcnf_amd/flows/
// affine/glow.py: sampling direction M = diag(exp s) W^-1, v = t; density direction M = W diag(exp -s),
// v = -W (t exp -s)), so the pair costs one read and one write of the activations instead of two of each, and the
// 1x1 convolution no longer goes through a convolution library (which ran it as one GEMM per image).
//
// HBM-bound (8 C bytes per pixel against 2 C^2 flop): the products run on v_mfma_f32_16x16x4_f32 (exact fp32 products,
// fp32 accumulation) so that the vector unit only moves data.  A wave owns tiles of 16 pixels: M is its A operand
// (held in registers for the whole launch, rows padded to a multiple of 16), the pixels' channel values its B operand
// (lane = 16 q + m: channel 4 j + q of pixel m in k-step j - one 64-byte segment per channel row and tile), the
// accumulators come out as channel 16 ob + 4 q + r of pixel m.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"
#include "fused_common.hpp"

namespace vcnf {

struct MixArgs {
  const float* x;
  float* y;
  const float* M;      // [C, C] row-major
  const float* v;      // [C]
  long long npix;      // B * inner
  long long inner;
  int C;
};

constexpr int kMixBlock = 256;

// CB = row blocks of 16 output channels, KS = k-steps of 4 input channels (C = 4 KS <= 16 CB).  ROWS: inner = 1, the
// "pixels" are the rows of a [B, C] matrix (the LU-parameterised linear layer between spline couplings,
// flows/mixing.py:352-470): a lane's four output channels are adjacent in memory and leave as one 16-byte store.
template <int CB, int KS, bool ROWS>
__global__ __launch_bounds__(kMixBlock) void channel_mix_kernel(const MixArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int m16 = lane & 15;
  const int q = lane >> 4;
  const int C = a.C;
  float wa[CB][KS];
  floatx4 bias[CB];
#pragma unroll
  for (int ob = 0; ob < CB; ++ob) {
    const int row = 16 * ob + m16;
#pragma unroll
    for (int j = 0; j < KS; ++j) wa[ob][j] = row < C ? a.M[row * C + 4 * j + q] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[ob][r] = (16 * ob + 4 * q + r) < C ? a.v[16 * ob + 4 * q + r] : 0.f;
  }
  // two 16-pixel tiles per pass: both tiles' loads are in flight together and neighbouring 64-byte segments of a
  // channel row are requested by the same wave
  constexpr int NT = 2;
  const long long ntiles = (a.npix + 16 * NT - 1) / (16 * NT);
  const long long stride = (long long)gridDim.x * (kMixBlock / 64);
  const long long plane = (long long)C * a.inner;
  for (long long t = (long long)blockIdx.x * (kMixBlock / 64) + wave; t < ntiles; t += stride) {
    bool ok[NT];
    long long base[NT];
    float xv[NT][KS];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      const long long g = (t * NT + c) * 16 + m16;            // pixel index over (b, p)
      ok[c] = g < a.npix;
      const long long b = ROWS ? (ok[c] ? g : 0) : (ok[c] ? g / a.inner : 0);
      base[c] = ROWS ? b * C : b * plane + (ok[c] ? g - b * a.inner : 0);
#pragma unroll
      for (int j = 0; j < KS; ++j) xv[c][j] = ok[c] ? a.x[base[c] + (ROWS ? 4 * j + q : (4 * j + q) * a.inner)] : 0.f;
    }
    floatx4 acc[NT][CB];
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
      for (int ob = 0; ob < CB; ++ob) acc[c][ob] = bias[ob];
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
      for (int ob = 0; ob < CB; ++ob)
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c][ob] = mfma4(wa[ob][j], xv[c][j], acc[c][ob]);
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      if (ok[c]) {
#pragma unroll
        for (int ob = 0; ob < CB; ++ob) {
          if (ROWS) {
            if (16 * ob + 4 * q < C) *reinterpret_cast<floatx4*>(a.y + base[c] + 16 * ob + 4 * q) = acc[c][ob];   // C % 4 == 0
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 16 * ob + 4 * q + r;
              if (row < C) a.y[base[c] + row * a.inner] = acc[c][ob][r];
            }
          }
        }
      }
    }
  }
}

template <int CB, int KS>
static void launch_mix(const MixArgs& a, dim3 grid, hipStream_t st) {
  if (a.inner == 1 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0)
    hipLaunchKernelGGL((channel_mix_kernel<CB, KS, true>), grid, dim3(kMixBlock), 0, st, a);
  else
    hipLaunchKernelGGL((channel_mix_kernel<CB, KS, false>), grid, dim3(kMixBlock), 0, st, a);
}

}  // namespace vcnf

using namespace vcnf;

extern "C" int vcnf_channel_mix_supported(int32_t channels) {
  return (channels >= 4 && channels <= 64 && channels % 4 == 0) ? 1 : 0;
}

extern "C" int vcnf_channel_mix_f32(const float* x, float* y, const float* matrix, const float* shift,
                                    int64_t batch, int32_t channels, int64_t inner, void* stream) {
  if (!vcnf_channel_mix_supported(channels)) return VCNF_ERR_UNSUPPORTED;
  if (batch < 0 || inner < 1) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!x || !y || !matrix || !shift) return VCNF_ERR_NULL;
  MixArgs a;
  a.x = x; a.y = y; a.M = matrix; a.v = shift; a.npix = batch * inner; a.inner = inner; a.C = channels;
  const long long blocks = (a.npix + 127) / 128;
  const long long cap = 256 * 16;
  dim3 grid((unsigned)(blocks < cap ? blocks : cap));
  hipStream_t st = (hipStream_t)stream;
  const int cb = (channels + 15) / 16, ks = channels / 4;
  // one instantiation per (row blocks, k-steps): C = 4 ks, 16 (cb - 1) < C <= 16 cb
#define VCNF_MIX_CASE(CBV, KSV) if (cb == CBV && ks == KSV) { launch_mix<CBV, KSV>(a, grid, st); } else
  VCNF_MIX_CASE(1, 1) VCNF_MIX_CASE(1, 2) VCNF_MIX_CASE(1, 3) VCNF_MIX_CASE(1, 4)
  VCNF_MIX_CASE(2, 5) VCNF_MIX_CASE(2, 6) VCNF_MIX_CASE(2, 7) VCNF_MIX_CASE(2, 8)
  VCNF_MIX_CASE(3, 9) VCNF_MIX_CASE(3, 10) VCNF_MIX_CASE(3, 11) VCNF_MIX_CASE(3, 12)
  VCNF_MIX_CASE(4, 13) VCNF_MIX_CASE(4, 14) VCNF_MIX_CASE(4, 15) VCNF_MIX_CASE(4, 16)
  { return VCNF_ERR_UNSUPPORTED; }
#undef VCNF_MIX_CASE
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}
