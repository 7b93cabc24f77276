// First two layers of Glow's coupling conditioner in ONE launch (nets/cnn.py:20-52, flows/affine/glow.py:37-47):
//     h1 = leaky(conv3x3(x; W1, pad 1) + b1)            x [B, C_in, H, W], C_in small (half of the coupling's channels)
//     y  = leaky(W2 h1 + b2)                             256 hidden channels, W2 the 1x1 convolution
// The 256-channel activation h1 - 20 to 85 x the layer's input - never exists in memory: after csrc/conv1x1.hip it
// still made one round trip (written by the library's 3x3 convolution, read back by the 1x1 kernel), and that
// convolution was a library call of its own.  Both layers are dense contractions (K = 9 C_in for the 3x3 layer seen
// as a GEMM over its 3 x 3 x C_in patch), so both run on v_mfma_f32_32x32x16_f16 with fp16 split-half operands
// (hi + lo 2^-11, three instructions per product, fp32 accumulation; clamping at +-65504 counted in `sat`), as in
// conv1x1.hip, whose second half this kernel shares:
//   * wave w of the 8 owns hidden channels 32 w .. 32 w + 31 of BOTH layers; its W2 fragments stay in registers for
//     the whole launch, its W1 fragments (8-28 KB, L2-resident) stream in per pass;
//   * a pass is 64 consecutive pixels: (1) all threads gather the pass's 3x3 patches (zero padded at the image
//     borders) and write them as B fragments; (2) every wave multiplies its W1 rows with them, adds b1, applies the
//     activation, splits the result and writes it as the B fragments of the second layer - accumulator register r of
//     lane (column, half) is channel 32 w + 8 (r / 4) + 4 half + r % 4, i.e. four adjacent halves of a fragment entry;
//     (3) the 1x1 layer as in conv1x1.hip; (4) 128-byte row stores of y.
// C_in <= 24 (K <= 224), hidden = output = 256 channels.
//
// THIRD layer (template flag): the conditioner's last convolution, 3x3 from the 256 channels to C_out <= 56, is computed
// as nine 1x1 convolutions - z[t * C_out + o, p] = sum_c W3[o, c, tap t] y[c, p] for EVERY pixel p, no shift - whose
// results a small second kernel shifts and adds (col2im: out[o, (py, px)] = b3[o] + sum_t z[t C_out + o, (py + dy - 1,
// px + dx - 1)] over the taps that stay inside the image).  So the second 256-channel activation y never exists in
// memory either, a pass needs no halo, and the library's last convolution - half of the C4 step - becomes one more
// matrix phase per pass: after the 1x1 layer of a 32-pixel column block every wave splits its 32 rows of y into a
// third fragment buffer, and the waves share the 9 C_out rows of the tap matrix W3' (row blocks of 32, streamed from
// L2) for the column block.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"
#include "fused_common.hpp"

namespace vcnf {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));

struct Conv31Args {
  const float* x;
  float* y;
  const uint4* w1frag;     // [8 row blocks][KS1][hi | lo][64 lanes]
  const uint4* w2frag;     // [8 row blocks][16][hi | lo][64 lanes]
  const float* b1;         // [256] or null
  const float* b2;         // [256] or null
  long long npix;          // B * H * W
  int Cin, H, W, K1;       // K1 = 9 Cin
  float slope1, slope2;
  int32_t* sat;
  const uint4* w3frag;     // THIRD: [RB3 row blocks][16][hi | lo][64 lanes] of the tap matrix [9 C_out, 256]
  float* z;                // THIRD: [B, 9 C_out, H, W]
  int M3, RB3;             // THIRD: 9 C_out rows, row blocks of 32
};

constexpr int kC31Block = 512;
constexpr int kC31Pix = 64;
constexpr int kC31KS2 = 16;          // 256 hidden channels

template <int KS1, bool THIRD>
__global__ __launch_bounds__(kC31Block, 2) void conv3x3_1x1_f16x3_kernel(const Conv31Args a) {
  extern __shared__ __align__(16) uint4 smem[];
  uint4* bf1 = smem;                                  // [KS1][2 column blocks][hi | lo][64 lanes]
  uint4* bf2 = smem + KS1 * 2 * 2 * 64;               // [16][2][2][64]
  uint4* bf3 = bf2 + kC31KS2 * 2 * 2 * 64;            // THIRD: [16][hi | lo][64] fragments of y for ONE column block
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  half8 wh[kC31KS2], wl[kC31KS2];
#pragma unroll
  for (int ks = 0; ks < kC31KS2; ++ks) {
    wh[ks] = __builtin_bit_cast(half8, a.w2frag[((wave * kC31KS2 + ks) * 2 + 0) * 64 + lane]);
    wl[ks] = __builtin_bit_cast(half8, a.w2frag[((wave * kC31KS2 + ks) * 2 + 1) * 64 + lane]);
  }
  const uint4* w1 = a.w1frag + (long long)wave * KS1 * 2 * 64 + lane;
  const int hw = a.H * a.W;
  float satm = 0.f;
  const long long ntiles = (a.npix + kC31Pix - 1) / kC31Pix;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();                       // the previous pass's fragments (both layers) are consumed
    // ---- (1) patches of the pass's 64 pixels -> B fragments of the first layer: task = (pixel, 8 consecutive k)
#pragma unroll 1
    for (int id = tid; id < kC31Pix * 2 * KS1; id += kC31Block) {
      const int pxl = id & 63, kg8 = id >> 6;
      const long long g = tile * kC31Pix + pxl;
      const bool ok = g < a.npix;
      const long long b = ok ? g / hw : 0;
      const int p = ok ? (int)(g - b * hw) : 0;
      const int py = p / a.W, pxx = p - py * a.W;
      const float* img = a.x + b * a.Cin * hw;
      half8 hi, lo;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 8 * kg8 + i;
        float t = 0.f;
        if (ok && k < a.K1) {
          const int ci = k / 9, tp = k - 9 * ci;
          const int dy = tp / 3, dx = tp - 3 * dy;
          const int yy = py + dy - 1, xx = pxx + dx - 1;
          if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) t = img[(long long)ci * hw + yy * a.W + xx];
        }
        satm = fmaxf(satm, __builtin_fabsf(t));
        t = __builtin_amdgcn_fmed3f(t, -65504.f, 65504.f);
        const _Float16 h = (_Float16)t;
        hi[i] = h;
        lo[i] = (_Float16)((t - (float)h) * kLoScale);
      }
      const int ks = kg8 >> 1, ln = 32 * (kg8 & 1) + (pxl & 31), ct = pxl >> 5;
      bf1[((ks * 2 + ct) * 2 + 0) * 64 + ln] = __builtin_bit_cast(uint4, hi);
      bf1[((ks * 2 + ct) * 2 + 1) * 64 + ln] = __builtin_bit_cast(uint4, lo);
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    // ---- (2) first layer for this wave's 32 hidden channels, result -> B fragments of the second layer
#pragma unroll 1
    for (int ct = 0; ct < 2; ++ct) {
      floatx16 mainv, ca, cb;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = 32 * wave + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        mainv[r] = a.b1 ? a.b1[ch] : 0.f;
        ca[r] = 0.f;
        cb[r] = 0.f;
      }
      half8 ah = __builtin_bit_cast(half8, w1[0]), al = __builtin_bit_cast(half8, w1[64]);
#pragma unroll 1
      for (int ks = 0; ks < KS1; ++ks) {      // rolled: unrolled, the compiler requests every k-step's W1 fragments up front

        const half8 ahc = ah, alc = al;
        if (ks + 1 < KS1) {
          ah = __builtin_bit_cast(half8, w1[((ks + 1) * 2 + 0) * 64]);
          al = __builtin_bit_cast(half8, w1[((ks + 1) * 2 + 1) * 64]);
        }
        const half8 bh = __builtin_bit_cast(half8, bf1[((ks * 2 + ct) * 2 + 0) * 64 + lane]);
        const half8 bl = __builtin_bit_cast(half8, bf1[((ks * 2 + ct) * 2 + 1) * 64 + lane]);
        mainv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahc, bh, mainv, 0, 0, 0);
        ca = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahc, bl, ca, 0, 0, 0);
        cb = __builtin_amdgcn_mfma_f32_32x32x16_f16(alc, bh, cb, 0, 0, 0);
      }
      // register r = 4 j + c: channel 32 w + 8 j + 4 half + c -> k-step 2 w + j / 2, lane half j % 2, halves 4 half + c
      const int col = lane & 31, hh = lane >> 5;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        half4v h4, l4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float t = fmaf(ca[4 * j + c] + cb[4 * j + c], kLoUnscale, mainv[4 * j + c]);
          t = t >= 0.f ? t : t * a.slope1;
          satm = fmaxf(satm, __builtin_fabsf(t));
          t = __builtin_amdgcn_fmed3f(t, -65504.f, 65504.f);
          const _Float16 h = (_Float16)t;
          h4[c] = h;
          l4[c] = (_Float16)((t - (float)h) * kLoScale);
        }
        const int ks2 = 2 * wave + (j >> 1), ln = 32 * (j & 1) + col;
        uint2* eh = reinterpret_cast<uint2*>(bf2 + ((ks2 * 2 + ct) * 2 + 0) * 64 + ln) + hh;
        uint2* el = reinterpret_cast<uint2*>(bf2 + ((ks2 * 2 + ct) * 2 + 1) * 64 + ln) + hh;
        *eh = __builtin_bit_cast(uint2, h4);
        *el = __builtin_bit_cast(uint2, l4);
      }
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    // ---- (3) + (4) second layer (1x1) and stores
#pragma unroll 1
    for (int ct = 0; ct < 2; ++ct) {
      floatx16 mainv, ca, cb;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * wave + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        mainv[r] = a.b2 ? a.b2[row] : 0.f;
        ca[r] = 0.f;
        cb[r] = 0.f;
      }
#pragma unroll
      for (int ks = 0; ks < kC31KS2; ++ks) {
        const half8 bh = __builtin_bit_cast(half8, bf2[((ks * 2 + ct) * 2 + 0) * 64 + lane]);
        const half8 bl = __builtin_bit_cast(half8, bf2[((ks * 2 + ct) * 2 + 1) * 64 + lane]);
        mainv = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], bh, mainv, 0, 0, 0);
        ca = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], bl, ca, 0, 0, 0);
        cb = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks], bh, cb, 0, 0, 0);
      }
      const long long g = tile * kC31Pix + 32 * ct + (lane & 31);
      if constexpr (!THIRD) {
        if (g < a.npix) {
          const long long b = g / hw;
          float* dst = a.y + b * 256 * hw + (g - b * hw);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = 32 * wave + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
            float t = fmaf(ca[r] + cb[r], kLoUnscale, mainv[r]);
            t = t >= 0.f ? t : t * a.slope2;
            dst[(long long)row * hw] = t;
          }
        }
      } else {
        // ---- (5) y of this column block -> fragments (same register -> entry map as after the first layer)
        const int col = lane & 31, hh = lane >> 5;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          half4v h4, l4;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float t = fmaf(ca[4 * j + c] + cb[4 * j + c], kLoUnscale, mainv[4 * j + c]);
            t = t >= 0.f ? t : t * a.slope2;
            satm = fmaxf(satm, __builtin_fabsf(t));
            t = __builtin_amdgcn_fmed3f(t, -65504.f, 65504.f);
            const _Float16 h = (_Float16)t;
            h4[c] = h;
            l4[c] = (_Float16)((t - (float)h) * kLoScale);
          }
          const int ks3 = 2 * wave + (j >> 1), ln = 32 * (j & 1) + col;
          *(reinterpret_cast<uint2*>(bf3 + (ks3 * 2 + 0) * 64 + ln) + hh) = __builtin_bit_cast(uint2, h4);
          *(reinterpret_cast<uint2*>(bf3 + (ks3 * 2 + 1) * 64 + ln) + hh) = __builtin_bit_cast(uint2, l4);
        }
        __syncthreads();
        // ---- (6) nine 1x1 convolutions of the taps: row blocks of W3' shared out over the waves
        for (int rb = wave; rb < a.RB3; rb += 8) {
          floatx16 m3, c3a, c3b;
#pragma unroll
          for (int r = 0; r < 16; ++r) { m3[r] = 0.f; c3a[r] = 0.f; c3b[r] = 0.f; }
          const uint4* w3 = a.w3frag + (long long)rb * kC31KS2 * 2 * 64 + lane;
          half8 ah = __builtin_bit_cast(half8, w3[0]), al = __builtin_bit_cast(half8, w3[64]);
#pragma unroll 1
          for (int ks = 0; ks < kC31KS2; ++ks) {
            const half8 ahc = ah, alc = al;
            if (ks + 1 < kC31KS2) {
              ah = __builtin_bit_cast(half8, w3[((ks + 1) * 2 + 0) * 64]);
              al = __builtin_bit_cast(half8, w3[((ks + 1) * 2 + 1) * 64]);
            }
            const half8 bh = __builtin_bit_cast(half8, bf3[(ks * 2 + 0) * 64 + lane]);
            const half8 bl = __builtin_bit_cast(half8, bf3[(ks * 2 + 1) * 64 + lane]);
            m3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahc, bh, m3, 0, 0, 0);
            c3a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahc, bl, c3a, 0, 0, 0);
            c3b = __builtin_amdgcn_mfma_f32_32x32x16_f16(alc, bh, c3b, 0, 0, 0);
          }
          if (g < a.npix) {
            const long long b = g / hw;
            float* dst = a.z + b * a.M3 * hw + (g - b * hw);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = 32 * rb + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
              if (row < a.M3) dst[(long long)row * hw] = fmaf(c3a[r] + c3b[r], kLoUnscale, m3[r]);
            }
          }
        }
        __syncthreads();                   // the column block's fragments are consumed before the next one is written
      }
    }
  }
  if (a.sat && satm > 65504.f) atomicAdd(a.sat, 1);
}

template <int KS1, bool THIRD>
static int launch_c31t(const Conv31Args& a, hipStream_t st) {
  const size_t lds = ((size_t)KS1 + kC31KS2) * 2 * 2 * 64 * 16 + (THIRD ? (size_t)kC31KS2 * 2 * 64 * 16 : 0);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_1x1_f16x3_kernel<KS1, THIRD>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return VCNF_ERR_LAUNCH;
    attr_set = true;
  }
  const long long ntiles = (a.npix + kC31Pix - 1) / kC31Pix;
  const long long cap = 256;               // one 8-wave workgroup per CU (128 weight registers per lane)
  dim3 grid((unsigned)(ntiles < cap ? ntiles : cap));
  hipLaunchKernelGGL((conv3x3_1x1_f16x3_kernel<KS1, THIRD>), grid, dim3(kC31Block), lds, st, a);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

template <int KS1>
static int launch_c31(const Conv31Args& a, hipStream_t st) {
  return a.w3frag ? launch_c31t<KS1, true>(a, st) : launch_c31t<KS1, false>(a, st);
}

// col2im of the nine tap results: one thread per output element
struct Col2imArgs {
  const float* z;      // [B, 9 C, H, W], row t * C + o
  const float* bias;   // [C] or null
  float* out;          // [B, C, H, W]
  long long n;         // B * C * H * W
  int C, H, W;
};

__global__ __launch_bounds__(256) void col2im3x3_kernel(const Col2imArgs a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int hw = a.H * a.W;
  const long long bc = i / hw;
  const int p = (int)(i - bc * hw);
  const long long b = bc / a.C;
  const int o = (int)(bc - b * a.C);
  const int py = p / a.W, px = p - py * a.W;
  const float* zi = a.z + b * 9 * a.C * hw;
  float acc = a.bias ? a.bias[o] : 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
    if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) acc += zi[(long long)(t * a.C + o) * hw + yy * a.W + xx];
  }
  a.out[i] = acc;
}

}  // namespace vcnf

using namespace vcnf;

extern "C" int vcnf_conv3x3_1x1_supported(int32_t c_in, int32_t hidden, int32_t c_out) {
  return (c_in >= 1 && c_in <= 24 && hidden == 256 && c_out == 256) ? 1 : 0;
}

/* floats of the packed first-layer weights: A fragments of the [256, 9 c_in] patch matrix, k padded to 16 */
extern "C" int64_t vcnf_conv3x3_1x1_pack_floats(int32_t c_in) {
  if (c_in < 1 || c_in > 24) return 0;
  return (int64_t)8 * ((9 * c_in + 15) / 16) * 2 * 64 * 4;
}

static int dispatch_c31(const Conv31Args& a, int c_in, hipStream_t st) {
  switch ((9 * c_in + 15) / 16) {
    case 1: return launch_c31<1>(a, st);
    case 2: return launch_c31<2>(a, st);
    case 3: return launch_c31<3>(a, st);
    case 4: return launch_c31<4>(a, st);
    case 5: return launch_c31<5>(a, st);
    case 6: return launch_c31<6>(a, st);
    case 7: return launch_c31<7>(a, st);
    case 8: return launch_c31<8>(a, st);
    case 9: return launch_c31<9>(a, st);
    case 10: return launch_c31<10>(a, st);
    case 11: return launch_c31<11>(a, st);
    case 12: return launch_c31<12>(a, st);
    case 13: return launch_c31<13>(a, st);
    case 14: return launch_c31<14>(a, st);
    default: return VCNF_ERR_UNSUPPORTED;
  }
}

extern "C" int vcnf_conv3x3_1x1_f16x3_f32(const float* x, float* y, const float* w1pack, int64_t w1pack_floats,
                                          const float* w2pack, int64_t w2pack_floats, const float* b1, const float* b2,
                                          int64_t batch, int32_t c_in, int32_t height, int32_t width, float slope1,
                                          float slope2, int32_t* sat_count, void* stream) {
  if (!vcnf_conv3x3_1x1_supported(c_in, 256, 256)) return VCNF_ERR_UNSUPPORTED;
  if (batch < 0 || height < 1 || width < 1) return VCNF_ERR_SHAPE;
  if (w1pack_floats != vcnf_conv3x3_1x1_pack_floats(c_in) || w2pack_floats != (int64_t)8 * 16 * 2 * 64 * 4) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!x || !y || !w1pack || !w2pack) return VCNF_ERR_NULL;
  if ((reinterpret_cast<uintptr_t>(w1pack) | reinterpret_cast<uintptr_t>(w2pack)) & 15) return VCNF_ERR_ALIGN;
  Conv31Args a;
  a.x = x; a.y = y; a.w1frag = reinterpret_cast<const uint4*>(w1pack); a.w2frag = reinterpret_cast<const uint4*>(w2pack);
  a.b1 = b1; a.b2 = b2; a.npix = batch * (long long)height * width; a.Cin = c_in; a.H = height; a.W = width;
  a.K1 = 9 * c_in; a.slope1 = slope1; a.slope2 = slope2; a.sat = sat_count;
  a.w3frag = nullptr; a.z = nullptr; a.M3 = 0; a.RB3 = 0;
  return dispatch_c31(a, c_in, (hipStream_t)stream);
}

/* Whole Glow conditioner Conv3x3(c_in -> 256), LeakyReLU, Conv1x1(256 -> 256), LeakyReLU, Conv3x3(256 -> c_out) up to the
 * shift-and-add of the last layer's nine taps: z[b, t * c_out + o, p] = sum_c W3[o, c, tap t] y[b, c, p] */
extern "C" int vcnf_convnet3_supported(int32_t c_in, int32_t hidden, int32_t c_out) {
  return (c_in >= 1 && c_in <= 24 && hidden == 256 && c_out >= 1 && c_out <= 56) ? 1 : 0;
}

extern "C" int64_t vcnf_convnet3_w3_pack_floats(int32_t c_out) {
  if (c_out < 1 || c_out > 56) return 0;
  return (int64_t)((9 * c_out + 31) / 32) * 16 * 2 * 64 * 4;
}

extern "C" int vcnf_convnet3_taps_f16x3_f32(const float* x, float* z, const float* w1pack, int64_t w1pack_floats,
                                            const float* w2pack, int64_t w2pack_floats, const float* w3pack,
                                            int64_t w3pack_floats, const float* b1, const float* b2, int64_t batch,
                                            int32_t c_in, int32_t c_out, int32_t height, int32_t width, float slope1,
                                            float slope2, int32_t* sat_count, void* stream) {
  if (!vcnf_convnet3_supported(c_in, 256, c_out)) return VCNF_ERR_UNSUPPORTED;
  if (batch < 0 || height < 1 || width < 1) return VCNF_ERR_SHAPE;
  if (w1pack_floats != vcnf_conv3x3_1x1_pack_floats(c_in) || w2pack_floats != (int64_t)8 * 16 * 2 * 64 * 4 ||
      w3pack_floats != vcnf_convnet3_w3_pack_floats(c_out))
    return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!x || !z || !w1pack || !w2pack || !w3pack) return VCNF_ERR_NULL;
  if ((reinterpret_cast<uintptr_t>(w1pack) | reinterpret_cast<uintptr_t>(w2pack) | reinterpret_cast<uintptr_t>(w3pack)) & 15)
    return VCNF_ERR_ALIGN;
  Conv31Args a;
  a.x = x; a.y = nullptr; a.w1frag = reinterpret_cast<const uint4*>(w1pack); a.w2frag = reinterpret_cast<const uint4*>(w2pack);
  a.b1 = b1; a.b2 = b2; a.npix = batch * (long long)height * width; a.Cin = c_in; a.H = height; a.W = width;
  a.K1 = 9 * c_in; a.slope1 = slope1; a.slope2 = slope2; a.sat = sat_count;
  a.w3frag = reinterpret_cast<const uint4*>(w3pack); a.z = z; a.M3 = 9 * c_out; a.RB3 = (9 * c_out + 31) / 32;
  return dispatch_c31(a, c_in, (hipStream_t)stream);
}

/* out[b, o, py, px] = bias[o] + sum over the 3 x 3 taps t = (dy, dx) that stay inside the image of
 * z[b, t * channels + o, py + dy - 1, px + dx - 1]: the shift-and-add that turns the nine tap results of
 * vcnf_convnet3_taps_f16x3_f32 into the 3x3 convolution with padding 1 (nets/cnn.py:36-43). */
extern "C" int vcnf_col2im3x3_f32(const float* z, const float* bias, float* out, int64_t batch, int32_t channels,
                                  int32_t height, int32_t width, void* stream) {
  if (batch < 0 || channels < 1 || height < 1 || width < 1) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!z || !out) return VCNF_ERR_NULL;
  Col2imArgs a{z, bias, out, batch * (long long)channels * height * width, channels, height, width};
  const long long blocks = (a.n + 255) / 256;
  if (blocks > 0x7fffffffLL) return VCNF_ERR_SHAPE;
  hipLaunchKernelGGL(col2im3x3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}
