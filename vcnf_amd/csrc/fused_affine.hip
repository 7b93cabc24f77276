// Fused affine coupling layer (AffineCouplingBlock with an MLP conditioner of two hidden
// layers) for MI355X: conditioner + affine transform + log|det| in one launch.
//
// Reference path per layer (flows/affine/coupling.py:113-168, :247-258, nets/mlp.py:30-58):
// chunk, three Linear + two LeakyReLU kernels, slicing of the interleaved (shift, scale)
// parameters, the elementwise map, a row sum, a concatenation.  Config C2 (D = 32, MLP
// 16-64-64-32, 8 layers) spends its time in launch gaps and in round trips of [B, 64]
// activations through HBM.  Here a layer reads x once and writes y once.
//
// Work split: a wave owns 16 samples from input to output; nothing is exchanged between waves.
// The MLP runs on v_mfma_f32_16x16x4_f32 (exact fp32 products) with the weights as the A operand
// and the wave's 16 samples as the 16 columns of the B operand.  A layer's accumulators ARE the
// next layer's B operand: accumulator register r of row block pb holds unit 16 pb + 4 q + r of
// sample (lane & 15) in lane group q = lane >> 4, so the next layer consumes k-step j = 4 pb + r
// straight from registers and its weights are packed on the host in that k order
// (vcnf_amd/fused_affine.py).  Weights (<= 0.2 MB, hot in L1/L2) are fetched with one 16-byte
// buffer load per four matrix instructions.  The last layer's row order puts shift and scale of
// a feature side by side in one lane (rows 4 q + r: r = 0, 1 -> feature 8 ob + 2 q, r = 2, 3 ->
// the next one), which is exactly the reference's interleaved parameter layout.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"
#include "fused_common.hpp"
#include "split_half.hpp"

namespace vcnf {

struct FusedAffineArgs {
  const float* x;
  float* y;
  float* logdet;
  const float* wpack;
  unsigned wpack_bytes;
  long long B;
  int D, c_in, cond_off, d_t, t_off;
  int KI;            // first-layer k-steps = ceil(c_in / 4)
  int OB;            // output row blocks = ceil(n_out / 16)
  int scale_map, inverse, ld_mode;
  float ld_sign, slope;
  const int32_t* in_gather;    // optional: the layer sees x[:, in_gather]   (a Permute before the block)
  const int32_t* out_gather;   // optional: the result is y[:, out_gather]   (a Permute after the block)
  int off_b1, off_w2, off_b2, off_w3, off_b3;   // float offsets inside wpack (w1 at 0)
};

__device__ __forceinline__ float aff_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }

// One transformed element: same arithmetic as affine_coupling_kernel (coupling.py:124-136, :150-162).
__device__ __forceinline__ float affine_apply(float v, float shift, float sc, int scale_map, int inverse, float& acc) {
  if (scale_map == VCNF_SCALE_EXP) {
    if (inverse) { acc -= sc; return (v - shift) * expf(-sc); }
    acc += sc;
    return v * expf(sc) + shift;
  }
  const float sg = aff_sigmoid(sc + 2.f);
  const float lg = logf(sg);
  const bool divide = (scale_map == VCNF_SCALE_SIGMOID) != (inverse != 0);
  acc += divide ? -lg : lg;
  if (inverse) return divide ? (v - shift) / sg : (v - shift) * sg;
  return divide ? v / sg + shift : v * sg + shift;
}

// Geometry of one layer inside a row strip of D floats and the float offset of its packed weights.
struct FALayer {
  int w_off;                 // float offset of this layer's pack inside wpack
  int cond_off, c_in, t_off, d_t;
};

// One AffineCouplingBlock on the wave's 16-row LDS strip ``xs`` (row stride XS), in place; ``ld`` accumulates the
// lane's share of log|det|.  (Body shared by the one-layer kernel and the stack kernel.)
template <int KIG, int HB, int OBM>
__device__ __forceinline__ void affine_layer_on_strip(float* xs, int XS, const __amdgpu_buffer_rsrc_t wr, const FALayer lp,
                                                      const FusedAffineArgs& a, int voff, int qoff, int m16, int q,
                                                      bool scaled, float& ld) {
  const int wb = 4 * lp.w_off;               // byte offset of the layer's pack
  // ---- layer 1: natural k order, k-step s of lane group q reads input 4 s + q
  floatx4 h1[HB];
  {
    float xin[4 * KIG];
#pragma unroll
    for (int s = 0; s < 4 * KIG; ++s) {
      const int c = 4 * s + q;
      xin[s] = (s < a.KI && c < lp.c_in) ? xs[m16 * XS + lp.cond_off + c] : 0.f;
    }
#pragma unroll
    for (int nb = 0; nb < HB; ++nb) {
      floatx4 acc = wload(wr, qoff, wb + 4 * (a.off_b1 + 16 * nb));
#pragma unroll
      for (int g = 0; g < KIG; ++g) {
        const floatx4 w = wload(wr, voff, wb + 4 * ((nb * KIG + g) * 256));
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = mfma4(w[i], xin[4 * g + i], acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) h1[nb][r] = acc[r] > 0.f ? acc[r] : a.slope * acc[r];   // LeakyReLU, mlp.py:33
    }
  }
  // ---- layer 2: chained k order (k-step 4 pb + r <- accumulator r of row block pb)
  floatx4 h2[HB];
#pragma unroll
  for (int nb = 0; nb < HB; ++nb) {
    floatx4 acc = wload(wr, qoff, wb + 4 * (a.off_b2 + 16 * nb));
#pragma unroll
    for (int pb = 0; pb < HB; ++pb) {
      const floatx4 w = wload(wr, voff, wb + 4 * (a.off_w2 + (nb * HB + pb) * 256));
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = mfma4(w[r], h1[pb][r], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) h2[nb][r] = acc[r] > 0.f ? acc[r] : a.slope * acc[r];
  }
  // ---- layer 3 + affine map on the lane's features
#pragma unroll
  for (int ob = 0; ob < OBM; ++ob) {
    if (ob < a.OB) {
      floatx4 acc = wload(wr, qoff, wb + 4 * (a.off_b3 + 16 * ob));
#pragma unroll
      for (int pb = 0; pb < HB; ++pb) {
        const floatx4 w = wload(wr, voff, wb + 4 * (a.off_w3 + (ob * HB + pb) * 256));
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma4(w[r], h2[pb][r], acc);
      }
      if (scaled) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const int f = 8 * ob + 2 * q + pr;              // rows 4 q + 2 pr (shift), + 1 (scale)
          if (f < lp.d_t) {
            float* pz = xs + m16 * XS + lp.t_off + f;
            *pz = affine_apply(*pz, acc[2 * pr], acc[2 * pr + 1], a.scale_map, a.inverse, ld);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * ob + 4 * q + r;
          if (f < lp.d_t) {
            float* pz = xs + m16 * XS + lp.t_off + f;
            *pz = a.inverse ? *pz - acc[r] : *pz + acc[r];  // coupling.py:139-141 / :165-167
          }
        }
      }
    }
  }
}


// The same layer with the two 16 HB-deep dense layers (hidden -> hidden, hidden -> parameters) on the fp16
// split-half matrix path (round 3): v_mfma_f32_16x16x32_f16 with hi | lo halves of both operands, hi*hi +
// (hi*lo + lo*hi) 2^-11, fp32 accumulation - GEMM error at or below the fp32 chain's at these depths
// (tests/test_gpu_gemm_error.py) at 3 x 16 instead of 8 x 32 matrix-pipe cycles per 16 x 16 x 32 product.  The first
// layer (c_in <= 64 deep, raw inputs) stays on the exact fp32 instruction.  The accumulators of a layer are again the
// next layer's B operand: k-step t of the 16x16x32 instruction takes lane group q's rows 4 q .. 4 q + 3 of row blocks
// 2 t and 2 t + 1 (eight values = its eight k-slots), and the host packs the weights in that k order
// (vcnf_amd/fused_affine.py::pack_h3).  Range: the halves cannot carry |h| > 65504; the wave then returns false BEFORE
// anything is written and the caller evaluates the layer with the fp32 body above - no clamped value ever leaves.
template <int KIG, int HB, int OBM, int NCB>
__device__ __forceinline__ bool affine_layer_on_strip_h3(float* xs, int XS, const __amdgpu_buffer_rsrc_t wr,
                                                         const __amdgpu_buffer_rsrc_t wr3, const FALayer lp, int h3_off,
                                                         const FusedAffineArgs& a, int voff, int qoff, int m16, int q,
                                                         bool scaled, float (&ld)[NCB]) {
  static_assert(HB % 2 == 0, "two row blocks of 16 per k-step of 32");
  constexpr int NT = HB / 2;
  const int wb = 4 * lp.w_off;               // byte offset of the layer's fp32 pack (first layer, biases)
  const int hb3 = 4 * h3_off;                // byte offset of the layer's split-half fragments
  // NCB column blocks of 16 samples share every weight fragment (the kernel is bound by the fragment loads from
  // L1 otherwise: 28 KB per 16 samples and layer)
  // ---- layer 1 (exact fp32): natural k order, k-step s of lane group q reads input 4 s + q
  floatx4 h1[NCB][HB];
  {
    float xin[NCB][4 * KIG];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int s = 0; s < 4 * KIG; ++s) {
        const int c = 4 * s + q;
        xin[cb][s] = (s < a.KI && c < lp.c_in) ? xs[(16 * cb + m16) * XS + lp.cond_off + c] : 0.f;
      }
#pragma unroll
    for (int nb = 0; nb < HB; ++nb) {
      floatx4 acc[NCB];
      const floatx4 bias = wload(wr, qoff, wb + 4 * (a.off_b1 + 16 * nb));
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = bias;
#pragma unroll
      for (int g = 0; g < KIG; ++g) {
        const floatx4 w = wload(wr, voff, wb + 4 * ((nb * KIG + g) * 256));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) acc[cb] = mfma4(w[i], xin[cb][4 * g + i], acc[cb]);
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) h1[cb][nb][r] = acc[cb][r] > 0.f ? acc[cb][r] : a.slope * acc[cb][r];   // LeakyReLU, mlp.py:33
    }
  }
  float satm = 0.f;
  half8 bh[NCB][NT], bl[NCB][NT];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float v8[8] = {h1[cb][2 * t][0], h1[cb][2 * t][1], h1[cb][2 * t][2], h1[cb][2 * t][3],
                           h1[cb][2 * t + 1][0], h1[cb][2 * t + 1][1], h1[cb][2 * t + 1][2], h1[cb][2 * t + 1][3]};
      split8<false>(v8, bh[cb][t], bl[cb][t], satm);
    }
  if (__builtin_amdgcn_ballot_w64(!(satm <= 65504.f)) != 0) return false;
  // ---- layer 2 on split halves: fragments [nb][t][hi | lo][lane][8 halves]
  floatx4 h2[NCB][HB];
#pragma unroll
  for (int nb = 0; nb < HB; ++nb) {
    floatx4 acc[NCB], corr[NCB];
    const floatx4 bias = wload(wr, qoff, wb + 4 * (a.off_b2 + 16 * nb));
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) { acc[cb] = bias; corr[cb] = floatx4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const half8 ah = __builtin_bit_cast(half8, wload(wr3, voff, hb3 + 4 * (((nb * NT + t) * 2 + 0) * 256)));
      const half8 al = __builtin_bit_cast(half8, wload(wr3, voff, hb3 + 4 * (((nb * NT + t) * 2 + 1) * 256)));
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        acc[cb] = mfma16h(ah, bh[cb][t], acc[cb]);
        corr[cb] = mfma16h(ah, bl[cb][t], corr[cb]);
        corr[cb] = mfma16h(al, bh[cb][t], corr[cb]);
      }
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = fmaf(corr[cb][r], kLoUnscale, acc[cb][r]);
        h2[cb][nb][r] = v > 0.f ? v : a.slope * v;
      }
  }
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float v8[8] = {h2[cb][2 * t][0], h2[cb][2 * t][1], h2[cb][2 * t][2], h2[cb][2 * t][3],
                           h2[cb][2 * t + 1][0], h2[cb][2 * t + 1][1], h2[cb][2 * t + 1][2], h2[cb][2 * t + 1][3]};
      split8<false>(v8, bh[cb][t], bl[cb][t], satm);
    }
  if (__builtin_amdgcn_ballot_w64(!(satm <= 65504.f)) != 0) return false;
  // ---- layer 3 on split halves + affine map on the lane's features (as in the fp32 body)
  const int w3 = hb3 + 4 * (HB * NT * 2 * 256);
#pragma unroll
  for (int ob = 0; ob < OBM; ++ob) {
    if (ob < a.OB) {
      floatx4 acc[NCB], corr[NCB];
      const floatx4 bias = wload(wr, qoff, wb + 4 * (a.off_b3 + 16 * ob));
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) { acc[cb] = bias; corr[cb] = floatx4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const half8 ah = __builtin_bit_cast(half8, wload(wr3, voff, w3 + 4 * (((ob * NT + t) * 2 + 0) * 256)));
        const half8 al = __builtin_bit_cast(half8, wload(wr3, voff, w3 + 4 * (((ob * NT + t) * 2 + 1) * 256)));
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          acc[cb] = mfma16h(ah, bh[cb][t], acc[cb]);
          corr[cb] = mfma16h(ah, bl[cb][t], corr[cb]);
          corr[cb] = mfma16h(al, bh[cb][t], corr[cb]);
        }
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[cb][r] = fmaf(corr[cb][r], kLoUnscale, acc[cb][r]);
        if (scaled) {
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const int f = 8 * ob + 2 * q + pr;              // rows 4 q + 2 pr (shift), + 1 (scale)
            if (f < lp.d_t) {
              float* pz = xs + (16 * cb + m16) * XS + lp.t_off + f;
              *pz = affine_apply(*pz, acc[cb][2 * pr], acc[cb][2 * pr + 1], a.scale_map, a.inverse, ld[cb]);
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int f = 16 * ob + 4 * q + r;
            if (f < lp.d_t) {
              float* pz = xs + (16 * cb + m16) * XS + lp.t_off + f;
              *pz = a.inverse ? *pz - acc[cb][r] : *pz + acc[cb][r];  // coupling.py:139-141 / :165-167
            }
          }
        }
      }
    }
  }
  return true;
}

// column blocks per wave of the split-half stack kernel: two share each weight fragment (half the L1 traffic) but at
// 188 registers; measured at C2: 0.291 ms per pass against 0.255 with one (profiles/r03_c2_affine_stack.md)
#ifndef VCNF_FA_NCB
#define VCNF_FA_NCB 1
#endif
constexpr int kFABlock = 256;          // 4 waves x 16 samples

// KIG: first-layer k-step groups of four (c_in <= 16 KIG); HB: hidden row blocks (hidden = 16 HB,
// both hidden layers); OBM: most output row blocks (2 d_t or d_t <= 16 OBM).
template <int KIG, int HB, int OBM>
__global__ __launch_bounds__(kFABlock) void fused_affine_layer_kernel(const FusedAffineArgs a) {
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15;
  const int q = lane >> 4;
  const int D = a.D;
  const __amdgpu_buffer_rsrc_t wr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpack), 0, a.wpack_bytes, 0x00020000);
  const int voff = lane * 16;
  const int qoff = q * 16;
  const bool scaled = a.scale_map != VCNF_SCALE_NONE;

  // Waves are independent: each walks its own sequence of 16-sample tiles and stages them in its
  // private LDS strip, so there is no workgroup barrier anywhere in the kernel (LDS operations
  // of one wave execute in order; the staging writes are complete before the strip is read).
  // The strip is a plain copy of the 16 contiguous rows (stride D): no index arithmetic on the
  // way in or out, 16-byte transfers when D is a multiple of 4.
  const int XS = D;
  float* xs = smem + wave * 16 * XS;
  const long long nwt = (a.B + 15) / 16;                       // wave tiles
  const long long wstride = (long long)gridDim.x * (kFABlock / 64);
  const int n16 = 16 * D;
  // wave-tile t = blockIdx + gridDim * (wave + 4 k): with fewer tiles than waves on the chip every workgroup runs ONE
  // wave (2048 samples = 128 tiles on 128 compute units instead of 32; waves are independent, see above)
  for (long long wt = blockIdx.x + (long long)gridDim.x * wave; wt < nwt; wt += wstride) {
    const long long b0 = wt * 16;
    const int rows = (int)min(16LL, a.B - b0);
    const int nvalid = rows * D;
    const float* src = a.x + b0 * D;
    if (a.in_gather) {
      for (int e = lane; e < n16; e += 64) {
        const int r = e / D, c = e - r * D;
        xs[e] = e < nvalid ? src[r * D + a.in_gather[c]] : 0.f;
      }
    } else if ((D & 3) == 0) {
      for (int e = 4 * lane; e < n16; e += 256) {
        const float4 v = e < nvalid ? *reinterpret_cast<const float4*>(src + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(xs + e) = v;
      }
    } else {
      for (int e = lane; e < n16; e += 64) xs[e] = e < nvalid ? src[e] : 0.f;
    }

    float ld = 0.f;
    affine_layer_on_strip<KIG, HB, OBM>(xs, XS, wr, FALayer{0, a.cond_off, a.c_in, a.t_off, a.d_t}, a, voff, qoff, m16, q, scaled, ld);
    if (a.logdet) {
      ld += __shfl_xor(ld, 16, 64);
      ld += __shfl_xor(ld, 32, 64);
      if (q == 0 && m16 < rows) {
        const float o = a.ld_sign * ld;
        a.logdet[b0 + m16] = a.ld_mode ? a.logdet[b0 + m16] + o : o;
      }
    }
    float* dst = a.y + b0 * D;
    if (a.out_gather) {
      for (int e = lane; e < nvalid; e += 64) {
        const int r = e / D, c = e - r * D;
        dst[e] = xs[r * D + a.out_gather[c]];
      }
    } else if ((D & 3) == 0) {
      for (int e = 4 * lane; e < nvalid; e += 256) *reinterpret_cast<float4*>(dst + e) = *reinterpret_cast<const float4*>(xs + e);
    } else {
      for (int e = lane; e < nvalid; e += 64) dst[e] = xs[e];
    }
  }
}

// A run of AffineCouplingBlocks of ONE conditioner shape, with the column permutations between them, in one launch:
// the wave's 16 rows stay in its LDS strip from the first layer to the last (x is read once, y written once; a
// per-layer launch streams [B, D] through HBM and pays a launch gap per layer).  A permutation is a copy between the
// wave's two strips through its index row.
constexpr int kFAStackMax = 16;
struct FusedAffineStackArgs {
  FusedAffineArgs base;          // x, y, logdet, wpack, B, D, KI, OB, scale_map, inverse, ld_*, slope, weight offsets
  const int32_t* gathers;        // [rows][D] column index rows
  int n_layers, gather_after;    // index row applied to the result (-1: none)
  int layer_floats;              // floats of one layer's pack
  FALayer layer[kFAStackMax];
  int gather_before[kFAStackMax];   // index row applied to the columns before layer i (-1: none)
  const float* wpack_h3;         // split-half fragments of the second and third dense layer of every layer (or NULL)
  unsigned wpack_h3_bytes;
  int h3_layer_floats;
  int32_t* redo;                 // counts layer evaluations that fell back to the fp32 body (range)
};

// Column permutation of a wave's strip (flows/mixing.py:32-54): xo[r][c] = xs[r][idx[c]].  When D divides 64 a lane
// keeps ONE column for the whole copy (its index entry is loaded once, the row advances by 64 / D): no integer
// division and one index load per lane instead of one per element - the general loop cost as many vector instructions
// per layer as the whole conditioner (profiles/r03_c2_affine_stack.md).
__device__ __forceinline__ void strip_gather(const float* xs, float* xo, const int32_t* idx, int D, int n, int lane) {
  if ((64 % D) == 0) {
    const int c = lane & (D - 1);
    const int src = idx[c];
    const int step = 64 / D;
    int r = lane / D;
    for (int e = lane; e < n; e += 64, r += step) xo[e] = xs[r * D + src];
  } else {
    for (int e = lane; e < n; e += 64) {
      const int r = e / D, c = e - r * D;
      xo[e] = xs[r * D + idx[c]];
    }
  }
}

template <int KIG, int HB, int OBM, bool H3>
__global__ __launch_bounds__(kFABlock) void fused_affine_stack_kernel(const FusedAffineStackArgs sa) {
  extern __shared__ __align__(16) float smem[];
  constexpr int NCB = H3 ? VCNF_FA_NCB : 1;  // 16-sample column blocks per wave in the split-half form (sharing every weight fragment)
  constexpr int RW = 16 * NCB;               // rows of a wave's strip
  const FusedAffineArgs& a = sa.base;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15;
  const int q = lane >> 4;
  const int D = a.D;
  const __amdgpu_buffer_rsrc_t wr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpack), 0, a.wpack_bytes, 0x00020000);
  const int voff = lane * 16;
  const int qoff = q * 16;
  const bool scaled = a.scale_map != VCNF_SCALE_NONE;
  const int XS = D;
  const int nrw = RW * D;
  float* s0 = smem + wave * 2 * nrw;         // two strips per wave
  float* s1 = s0 + nrw;
  const long long nwt = (a.B + RW - 1) / RW;
  const long long wstride = (long long)gridDim.x * (kFABlock / 64);
  // wave-tile t = blockIdx + gridDim * (wave + 4 k): with fewer tiles than waves on the chip every workgroup runs ONE
  // wave (2048 samples = 128 tiles on 128 compute units instead of 32; waves are independent, see above)
  for (long long wt = blockIdx.x + (long long)gridDim.x * wave; wt < nwt; wt += wstride) {
    const long long b0 = wt * RW;
    const int rows = (int)min((long long)RW, a.B - b0);
    const int nvalid = rows * D;
    const float* src = a.x + b0 * D;
    float* xs = s0;
    float* xo = s1;
    if ((D & 3) == 0) {
      for (int e = 4 * lane; e < nrw; e += 256) {
        const float4 v = e < nvalid ? *reinterpret_cast<const float4*>(src + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(xs + e) = v;
      }
    } else {
      for (int e = lane; e < nrw; e += 64) xs[e] = e < nvalid ? src[e] : 0.f;
    }
    float ld[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) ld[cb] = 0.f;
    for (int li = 0; li < sa.n_layers; ++li) {
      const int gb = sa.gather_before[li];
      if (gb >= 0) {                                              // flows/mixing.py:32-54 as a strip-to-strip copy
        strip_gather(xs, xo, sa.gathers + gb * D, D, nrw, lane);
        float* t_ = xs; xs = xo; xo = t_;
      }
      if constexpr (H3) {
        const __amdgpu_buffer_rsrc_t wr3 =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sa.wpack_h3), 0, sa.wpack_h3_bytes, 0x00020000);
        float ld_try[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) ld_try[cb] = ld[cb];
        if (affine_layer_on_strip_h3<KIG, HB, OBM, NCB>(xs, XS, wr, wr3, sa.layer[li], li * sa.h3_layer_floats, a, voff,
                                                        qoff, m16, q, scaled, ld_try)) {
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) ld[cb] = ld_try[cb];
        } else {                           // a hidden activation beyond the fp16 range: the exact fp32 body, nothing was written
          if (lane == 0 && sa.redo) atomicAdd(sa.redo, 1);
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
            affine_layer_on_strip<KIG, HB, OBM>(xs + 16 * cb * XS, XS, wr, sa.layer[li], a, voff, qoff, m16, q, scaled, ld[cb]);
        }
      } else {
        affine_layer_on_strip<KIG, HB, OBM>(xs, XS, wr, sa.layer[li], a, voff, qoff, m16, q, scaled, ld[0]);
      }
    }
    if (sa.gather_after >= 0) {
      strip_gather(xs, xo, sa.gathers + sa.gather_after * D, D, nrw, lane);
      float* t_ = xs; xs = xo; xo = t_;
    }
    if (a.logdet) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        float v = ld[cb];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        const int row = 16 * cb + m16;
        if (q == 0 && row < rows) {
          const float o = a.ld_sign * v;
          a.logdet[b0 + row] = a.ld_mode ? a.logdet[b0 + row] + o : o;
        }
      }
    }
    float* dst = a.y + b0 * D;
    if ((D & 3) == 0) {
      for (int e = 4 * lane; e < nvalid; e += 256) *reinterpret_cast<float4*>(dst + e) = *reinterpret_cast<const float4*>(xs + e);
    } else {
      for (int e = lane; e < nvalid; e += 64) dst[e] = xs[e];
    }
  }
}

template <int KIG, int HB, int OBM>
static int launch_fa_stack(const FusedAffineStackArgs& sa, hipStream_t st) {
  const int rw = sa.wpack_h3 ? 16 * VCNF_FA_NCB : 16;     // rows of a wave's strip
  const size_t lds = (size_t)4 * 2 * rw * sa.base.D * sizeof(float);
  if (lds > 64 * 1024) return VCNF_ERR_SHAPE;
  const long long ntiles = (sa.base.B + rw - 1) / rw;          // one workgroup per wave-tile until the chip is full
  const long long cap = 256 * 8;
  dim3 grid((unsigned)(ntiles < cap ? ntiles : cap));
  if (sa.wpack_h3)
    hipLaunchKernelGGL((fused_affine_stack_kernel<KIG, HB, OBM, true>), grid, dim3(kFABlock), lds, st, sa);
  else
    hipLaunchKernelGGL((fused_affine_stack_kernel<KIG, HB, OBM, false>), grid, dim3(kFABlock), lds, st, sa);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

template <int KIG, int HB, int OBM>
static int launch_fa(const FusedAffineArgs& a, hipStream_t st) {
  const size_t lds = (size_t)4 * 16 * a.D * sizeof(float);
  if (lds > 64 * 1024) return VCNF_ERR_SHAPE;
  const long long ntiles = (a.B + 15) / 16;                    // one workgroup per wave-tile until the chip is full
  const long long cap = 256 * 8;
  dim3 grid((unsigned)(ntiles < cap ? ntiles : cap));
  hipLaunchKernelGGL((fused_affine_layer_kernel<KIG, HB, OBM>), grid, dim3(kFABlock), lds, st, a);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

static bool fa_shape_ok(int c_in, int hidden, int n_out, int features) {
  if (c_in < 1 || c_in > 64 || n_out < 1 || n_out > 128) return false;
  if (hidden != 32 && hidden != 64 && hidden != 128) return false;
  if (features < 2 || features > 1023) return false;
  return true;
}

}  // namespace vcnf

using namespace vcnf;

extern "C" int vcnf_affine_layer_fused_supported(int32_t c_in, int32_t hidden, int32_t n_out, int32_t features) {
  return fa_shape_ok(c_in, hidden, n_out, features) ? 1 : 0;
}

// the whole-stack kernel keeps two LDS strips of 16 samples x D per wave (512 D bytes per workgroup <= 64 KB)
extern "C" int vcnf_affine_stack_fused_supported(int32_t c_in, int32_t hidden, int32_t n_out, int32_t features) {
  return (fa_shape_ok(c_in, hidden, n_out, features) && (size_t)4 * 2 * 16 * features * sizeof(float) <= 64 * 1024) ? 1 : 0;
}

extern "C" int64_t vcnf_affine_layer_fused_pack_floats(int32_t c_in, int32_t hidden, int32_t n_out) {
  if (!fa_shape_ok(c_in, hidden, n_out, 2)) return 0;
  const int KIG = c_in <= 16 ? 1 : 4, HB = hidden / 16, OB = (n_out + 15) / 16;
  return (int64_t)HB * KIG * 256 + 16 * HB + (int64_t)HB * HB * 256 + 16 * HB + (int64_t)OB * HB * 256 + 16 * OB;
}

extern "C" int vcnf_affine_layer_fused_f32(const float* x, float* y, float* logdet, int64_t batch, int32_t features,
                                           int32_t cond_off, int32_t c_in, int32_t t_off, int32_t d_t,
                                           int32_t hidden, float leaky_slope, int scale_map,
                                           const float* wpack, int64_t wpack_floats,
                                           const int32_t* in_gather, const int32_t* out_gather,
                                           int inverse, int ld_mode, float ld_sign, void* stream) {
  const int n_out = scale_map == VCNF_SCALE_NONE ? d_t : 2 * d_t;
  if (batch < 0 || !fa_shape_ok(c_in, hidden, n_out, features)) return VCNF_ERR_SHAPE;
  if (cond_off < 0 || t_off < 0 || cond_off + c_in > features || t_off + d_t > features) return VCNF_ERR_SHAPE;
  if (scale_map < VCNF_SCALE_EXP || scale_map > VCNF_SCALE_NONE) return VCNF_ERR_UNSUPPORTED;
  if (ld_mode != VCNF_LD_STORE && ld_mode != VCNF_LD_ACCUM) return VCNF_ERR_UNSUPPORTED;
  if (wpack_floats != vcnf_affine_layer_fused_pack_floats(c_in, hidden, n_out)) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!x || !y || !wpack || (scale_map != VCNF_SCALE_NONE && !logdet)) return VCNF_ERR_NULL;
  FusedAffineArgs a;
  a.x = x; a.y = y; a.logdet = logdet; a.wpack = wpack; a.wpack_bytes = (unsigned)(wpack_floats * 4);
  a.B = batch; a.D = features; a.c_in = c_in; a.cond_off = cond_off; a.d_t = d_t; a.t_off = t_off;
  a.KI = (c_in + 3) / 4; a.OB = (n_out + 15) / 16;
  a.scale_map = scale_map; a.inverse = inverse ? 1 : 0; a.ld_mode = ld_mode; a.ld_sign = ld_sign; a.slope = leaky_slope;
  a.in_gather = in_gather; a.out_gather = out_gather;
  const int KIG = c_in <= 16 ? 1 : 4, HB = hidden / 16;
  a.off_b1 = HB * KIG * 256;
  a.off_w2 = a.off_b1 + 16 * HB;
  a.off_b2 = a.off_w2 + HB * HB * 256;
  a.off_w3 = a.off_b2 + 16 * HB;
  a.off_b3 = a.off_w3 + a.OB * HB * 256;
  hipStream_t st = (hipStream_t)stream;
  const bool small_out = a.OB <= 2;
#define VCNF_FA(KIG_, HB_)                                                              \
  return small_out ? launch_fa<KIG_, HB_, 2>(a, st) : launch_fa<KIG_, HB_, 8>(a, st);
  if (KIG == 1) {
    if (HB == 2) { VCNF_FA(1, 2) }
    if (HB == 4) { VCNF_FA(1, 4) }
    VCNF_FA(1, 8)
  }
  if (HB == 2) { VCNF_FA(4, 2) }
  if (HB == 4) { VCNF_FA(4, 4) }
  VCNF_FA(4, 8)
#undef VCNF_FA
}

extern "C" int64_t vcnf_affine_layer_fused_h3_pack_floats(int32_t c_in, int32_t hidden, int32_t n_out) {
  if (!fa_shape_ok(c_in, hidden, n_out, 2)) return 0;
  const int HB = hidden / 16, OB = (n_out + 15) / 16;
  return (int64_t)HB * HB * 256 + (int64_t)OB * HB * 256;      // W2 and W3 as hi | lo fp16 fragments: as many floats as in fp32
}

static int affine_stack(const float* x, float* y, float* logdet, int64_t batch, int32_t features,
                        int32_t n_layers, const vcnf_affine_stack_layer* layers, int32_t gather_after,
                        int32_t c_in, int32_t hidden, float leaky_slope, int scale_map,
                        const float* wpack, int64_t wpack_floats,
                        const int32_t* gathers, int32_t n_gather_rows,
                        int inverse, int ld_mode, float ld_sign,
                        const float* wpack_h3, int64_t wpack_h3_floats, int32_t* redo_count, void* stream) {
  if (!layers || n_layers < 1 || n_layers > kFAStackMax) return VCNF_ERR_SHAPE;
  const int d_t0 = layers[0].d_t;
  const int n_out = scale_map == VCNF_SCALE_NONE ? d_t0 : 2 * d_t0;
  if (batch < 0 || !fa_shape_ok(c_in, hidden, n_out, features)) return VCNF_ERR_SHAPE;
  if (scale_map < VCNF_SCALE_EXP || scale_map > VCNF_SCALE_NONE) return VCNF_ERR_UNSUPPORTED;
  if (ld_mode != VCNF_LD_STORE && ld_mode != VCNF_LD_ACCUM) return VCNF_ERR_UNSUPPORTED;
  const int64_t per_layer = vcnf_affine_layer_fused_pack_floats(c_in, hidden, n_out);
  if (wpack_floats != per_layer * n_layers) return VCNF_ERR_SHAPE;
  if (gather_after >= n_gather_rows || gather_after < -1) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!x || !y || !wpack || (scale_map != VCNF_SCALE_NONE && !logdet) || (n_gather_rows > 0 && !gathers)) return VCNF_ERR_NULL;
  FusedAffineStackArgs sa;
  FusedAffineArgs& a = sa.base;
  a.x = x; a.y = y; a.logdet = logdet; a.wpack = wpack; a.wpack_bytes = (unsigned)(wpack_floats * 4);
  a.B = batch; a.D = features; a.c_in = c_in; a.cond_off = 0; a.d_t = d_t0; a.t_off = 0;
  a.KI = (c_in + 3) / 4; a.OB = (n_out + 15) / 16;
  a.scale_map = scale_map; a.inverse = inverse ? 1 : 0; a.ld_mode = ld_mode; a.ld_sign = ld_sign; a.slope = leaky_slope;
  a.in_gather = nullptr; a.out_gather = nullptr;
  const int KIG = c_in <= 16 ? 1 : 4, HB = hidden / 16;
  a.off_b1 = HB * KIG * 256;
  a.off_w2 = a.off_b1 + 16 * HB;
  a.off_b2 = a.off_w2 + HB * HB * 256;
  a.off_w3 = a.off_b2 + 16 * HB;
  a.off_b3 = a.off_w3 + a.OB * HB * 256;
  sa.gathers = gathers; sa.n_layers = n_layers; sa.gather_after = gather_after; sa.layer_floats = (int)per_layer;
  sa.wpack_h3 = wpack_h3; sa.redo = redo_count;
  sa.h3_layer_floats = (int)vcnf_affine_layer_fused_h3_pack_floats(c_in, hidden, n_out);
  sa.wpack_h3_bytes = (unsigned)(wpack_h3_floats * 4);
  if (wpack_h3 && wpack_h3_floats != (int64_t)sa.h3_layer_floats * n_layers) return VCNF_ERR_SHAPE;
  for (int i = 0; i < n_layers; ++i) {
    const vcnf_affine_stack_layer& l = layers[i];
    if (l.d_t != d_t0 || l.cond_off < 0 || l.t_off < 0 || l.cond_off + c_in > features || l.t_off + l.d_t > features)
      return VCNF_ERR_SHAPE;
    if (l.gather_before >= n_gather_rows || l.gather_before < -1) return VCNF_ERR_SHAPE;
    sa.layer[i] = FALayer{(int)(per_layer * i), l.cond_off, c_in, l.t_off, l.d_t};
    sa.gather_before[i] = l.gather_before;
  }
  hipStream_t st = (hipStream_t)stream;
  const bool small_out = a.OB <= 2;
#define VCNF_FAS(KIG_, HB_)                                                             \
  return small_out ? launch_fa_stack<KIG_, HB_, 2>(sa, st) : launch_fa_stack<KIG_, HB_, 8>(sa, st);
  if (KIG == 1) {
    if (HB == 2) { VCNF_FAS(1, 2) }
    if (HB == 4) { VCNF_FAS(1, 4) }
    VCNF_FAS(1, 8)
  }
  if (HB == 2) { VCNF_FAS(4, 2) }
  if (HB == 4) { VCNF_FAS(4, 4) }
  VCNF_FAS(4, 8)
#undef VCNF_FAS
}

extern "C" int vcnf_affine_stack_fused_f32(const float* x, float* y, float* logdet, int64_t batch, int32_t features,
                                           int32_t n_layers, const vcnf_affine_stack_layer* layers, int32_t gather_after,
                                           int32_t c_in, int32_t hidden, float leaky_slope, int scale_map,
                                           const float* wpack, int64_t wpack_floats,
                                           const int32_t* gathers, int32_t n_gather_rows,
                                           int inverse, int ld_mode, float ld_sign, void* stream) {
  return affine_stack(x, y, logdet, batch, features, n_layers, layers, gather_after, c_in, hidden, leaky_slope, scale_map,
                      wpack, wpack_floats, gathers, n_gather_rows, inverse, ld_mode, ld_sign, nullptr, 0, nullptr, stream);
}

extern "C" int vcnf_affine_stack_fused_f16x3_f32(const float* x, float* y, float* logdet, int64_t batch, int32_t features,
                                                 int32_t n_layers, const vcnf_affine_stack_layer* layers,
                                                 int32_t gather_after, int32_t c_in, int32_t hidden, float leaky_slope,
                                                 int scale_map, const float* wpack, int64_t wpack_floats,
                                                 const float* wpack_h3, int64_t wpack_h3_floats,
                                                 const int32_t* gathers, int32_t n_gather_rows,
                                                 int inverse, int ld_mode, float ld_sign, int32_t* redo_count,
                                                 void* stream) {
  if (!wpack_h3) return VCNF_ERR_NULL;
  if (hidden % 32) return VCNF_ERR_UNSUPPORTED;
  return affine_stack(x, y, logdet, batch, features, n_layers, layers, gather_after, c_in, hidden, leaky_slope, scale_map,
                      wpack, wpack_floats, gathers, n_gather_rows, inverse, ld_mode, ld_sign, wpack_h3, wpack_h3_floats,
                      redo_count, stream);
}
