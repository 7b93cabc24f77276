// One whole RQS coupling layer in one kernel: conditioner input gather, the
// ResidualNet conditioner (dense layers on the fp32 matrix cores), the splines on
// both halves and the per-sample log|det| - for MI355X (gfx950, wave64).
//
// Why: with the layer split into GEMM kernels and a spline kernel, the conditioner
// output params[B, d_t*(3K-1)] (2944 B per sample at config C3) and ~12 hidden-state
// tensors cross HBM per layer; fused, a sample costs 4D (x) + 4C (context) + 4D (y)
// + 8 (log-det) bytes of HBM and everything else stays in registers / LDS.
//
// Matrix-core formulation (v_mfma_f32_16x16x4_f32, exact fp32 fma chains):
//   out^T[N x 16 samples] = W[N x Kin] * in^T[Kin x 16 samples]
// A operand = weight fragment, lane l holds W[16 nb + (l & 15)][k(l >> 4)];
// B operand = activations,     lane l holds in[sample l & 15][k(l >> 4)];
// result: lane l holds out[sample l & 15][16 nb + 4 (l >> 4) + r], r = 0..3.
// The result registers of one layer are used directly as the B operands of the next
// one: k-step (nb, r) of the next layer covers k = 16 nb + 4 q + r for lane group
// q = l >> 4, and the host packs the next layer's weight fragments in that k order.
// So hidden activations never leave registers.  The last layer's rows are permuted
// on the host so that lane group q receives all 3K-1 logits of feature 4 g + q
// (padded to a multiple of 4) in its own registers: the spline is then evaluated
// straight from the accumulators, with no cross-lane exchange at all.
//
// Reference arithmetic: flows/neural_spline/coupling.py:70-125 (layer),
// nets/resnet.py:38-57, :92-106 (conditioner), utils/splines.py:20-193 (splines).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/vcnf_hip.h"
#include "rqs_math.hpp"
#include "fused_common.hpp"

namespace vcnf {

// One 16-row output block of a dense layer for the wave's kCB column blocks:
//   acc[cb] += sum_s A(s) * B(cb, s),  s < 4 * NS4.
// The block's weight fragments are packed [s / 4][lane][4]: one 16-byte load per lane
// feeds 4 k-steps (8 matrix instructions).  B(cb, s) is a callable returning the
// activation operand (a register of the previous layer's result, optionally ReLU'd).
template <int NS4, int kCB, class BOp>
__device__ __forceinline__ void dense_block(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, BOp bop,
                                            floatx4 (&acc)[kCB]) {
  floatx4 a_cur = wload(rsrc, voff, soff);
#pragma unroll
  for (int s4 = 0; s4 < NS4; ++s4) {
    floatx4 a_nxt = a_cur;
    if (s4 + 1 < NS4) a_nxt = wload(rsrc, voff, soff + (s4 + 1) * 1024);
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
      for (int cb = 0; cb < kCB; ++cb) acc[cb] = mfma4(a_cur[cc], bop(cb, 4 * s4 + cc), acc[cb]);
    a_cur = a_nxt;
  }
}

// bias of row block nb in accumulator layout: lane group q holds rows 16 nb + 4 q + r
template <int kCB>
__device__ __forceinline__ void bias_block(__amdgpu_buffer_rsrc_t rsrc, int qoff, int soff, floatx4 (&acc)[kCB]) {
  const floatx4 v = wload(rsrc, qoff, soff);
#pragma unroll
  for (int cb = 0; cb < kCB; ++cb) acc[cb] = v;
}

template <int DI, int DT, int C, int H, int NBLK, int K, bool INV, int kCB, bool STACK>
__global__ __launch_bounds__(kFBlock, 2) void fused_rqs_layer_kernel(const FusedStackArgs sa) {
  // sa.n_layers coupling layers of one shape applied in order to each tile (1: the single-layer entry point); the
  // tile stays in LDS between layers, log|det| is summed over the layers in registers (core.py:144-183)
  const FusedArgs& a = sa.a;
  constexpr int kTile = 4 * kCB * 16;       // samples per workgroup tile
  static_assert(kTile == kFusedTile && kTile == 4 * kFusedFlagRows, "one tile = four 32-sample range flags");
  constexpr int D = DI + DT;
  constexpr int XS = D + 4;                 // padded LDS row strides (16-byte aligned rows)
  constexpr int CS = (C > 0 ? C : 4) + 4;
  constexpr int NB = H / 16;                // row blocks of a hidden layer
  constexpr int NS0 = (DI + C) / 4;         // k-steps of the first layer
  constexpr int NSH = H / 4;                // k-steps of a hidden->* layer
  constexpr int NSC = C / 4;
  constexpr int P = 3 * K - 1;
  constexpr int P4 = (P + 3) / 4;           // row-block quarters per feature in the last layer
  constexpr int NG = DT / 4;                // feature groups of the last layer
  constexpr int TABW = 3 * (K + 1);
  static_assert(DI % 4 == 0 && DT % 4 == 0 && C % 4 == 0 && H % 16 == 0, "shape family");

  extern __shared__ __align__(16) float smem[];
  float* xt = smem;                         // [kTile][XS]   x in, y out (in place)
  float* ct = xt + kTile * XS;              // [kTile][CS]
  float* tab = ct + kTile * CS;             // [DI][TABW]
  int* tfi = reinterpret_cast<int*>(tab + ((DI * TABW + 3) & ~3));
  int* idi = tfi + DT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int m16 = lane & 15;
  const int q = lane >> 4;
  const RqsConst& c = a.c;
  const bool shared = sa.lay[0].sh_w != nullptr;    // all layers of a stack or none
  // STACK = false: the single-layer instantiation (layer count known at compile time: the layer loop and its
  // bookkeeping fold away - with them in registers the kernel spilled and lost 8 % at 1M samples)
  const int nlay = STACK ? sa.n_layers : 1;

  if (a.redo) {
    // re-evaluation pass behind the split-half kernel: normally no tile is flagged - leave before any set-up
    // (index tables, knot tables) so that the launch costs a few microseconds
    const long long nt = (a.B + kTile - 1) / kTile, nf = (a.B + kFusedFlagRows - 1) / kFusedFlagRows;
    int any = 0;
    for (long long t = blockIdx.x; t < nt; t += gridDim.x)
      for (int k = 0; k < 4; ++k) any |= 4 * t + k < nf ? a.redo[4 * t + k] : 0;
    if (!any) return;
  }

  const long long ntiles = (a.B + kTile - 1) / kTile;
  // index vectors and knot tables of one layer -> LDS
  auto setup_layer = [&](const FusedLayerDesc& d) {
    for (int i = tid; i < DT; i += kFBlock) tfi[i] = d.tf_idx[i];
    for (int i = tid; i < DI; i += kFBlock) idi[i] = d.id_idx[i];
    if (shared) {
      // knot tables of the identity half: one thread per (feature, column: x knots | y knots | derivatives)
      for (int i = tid; i < 3 * DI; i += kFBlock) {
        const int f = i % DI;
        SplitLogits p{d.sh_w + f * K, d.sh_h + f * K, d.sh_d + f * (K - 1), K, 1.f, c.edge_logit, c.tails};
        rqs_build_table_part_k<K>(p, c, tab + f * TABW, 1, i / DI);
      }
    }
  };
  if (!STACK) setup_layer(sa.lay[0]);       // single layer: once per launch, outside the tile loop (the first
                                            // __syncthreads() of the loop publishes it)
  bool bad = false;
  int have = -1;                            // STACK: layer whose index vectors / knot tables are in LDS
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // re-evaluation pass behind the split-half kernel: only the tiles it flagged (and did not write)
    // (one flag per 32 rows: a flagged quarter is written, the others are left as the split-half kernel wrote them)
    const long long b0 = tile * kTile;
    const int rows = (int)min((long long)kTile, a.B - b0);
    unsigned wmask = 0xFu;
    if (a.redo) {
      wmask = 0u;
      for (int k = 0; k < 4; ++k)
        if (k * kFusedFlagRows < rows && a.redo[4 * tile + k] != 0) wmask |= 1u << k;
      if (wmask == 0u) continue;
    }
    __syncthreads();
    // ---- stage x rows and context rows (coalesced 16-byte loads, padded LDS rows)
    {
      constexpr int D4 = D / 4;
      const float4* sx = reinterpret_cast<const float4*>(a.x) + b0 * D4;
      for (int i = tid; i < kTile * D4; i += kFBlock) {
        const int r = i / D4, o = i - r * D4;
        const float4 v = r < rows ? sx[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(xt + r * XS + 4 * o) = v;
      }
      if (C > 0) {
        constexpr int C4 = (C > 0 ? C : 4) / 4;
        const float4* sc = reinterpret_cast<const float4*>(a.ctx) + b0 * C4;
        for (int i = tid; i < kTile * C4; i += kFBlock) {
          const int r = i / C4, o = i - r * C4;
          const float4 v = r < rows ? sc[i] : make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4*>(ct + r * CS + 4 * o) = v;
        }
      }
    }
    __syncthreads();

    float ld_tot[kCB];
#pragma unroll
    for (int cb = 0; cb < kCB; ++cb) ld_tot[cb] = 0.f;
   for (int l = 0; l < nlay; ++l) {
    const FusedLayerDesc& lay = sa.lay[l];
    if (STACK && have != l) {
      // this layer's index vectors and knot tables
      if (have >= 0) __syncthreads();
      setup_layer(lay);
      __syncthreads();
      have = l;
    }
    float ld_acc[kCB];
#pragma unroll
    for (int cb = 0; cb < kCB; ++cb) ld_acc[cb] = 0.f;

    // ---- identity half through the unconditional spline.  Sampling direction: before
    // the conditioner (it sees the transformed values, coupling.py:110-114); density
    // direction: after the operands were read (coupling.py:81-90).
#define VCNF_IDENTITY_PASS()                                                        \
  _Pragma("unroll") for (int cb = 0; cb < kCB; ++cb) {                              \
    const int mi = (wave * kCB + cb) * 16 + m16;                                    \
    for (int f = q; f < DI; f += 4) {                                               \
      float* px = xt + mi * XS + idi[f];                                            \
      const float xv = *px;                                                         \
      float yv = xv, lad = 0.f;                                                     \
      if (shared) rqs_point_table<INV>(xv, tab + f * TABW, c, yv, lad, bad);        \
      *px = yv;                                                                     \
      ld_acc[cb] += lad;                                                            \
    }                                                                               \
  }
    if (INV) {
      VCNF_IDENTITY_PASS()
      __syncthreads();
    }

    // ---- first-layer operand: identity features then context, natural k order
    float hin[kCB][NS0];
#pragma unroll
    for (int cb = 0; cb < kCB; ++cb) {
      const int m = (wave * kCB + cb) * 16 + m16;
#pragma unroll
      for (int s = 0; s < DI / 4; ++s) hin[cb][s] = xt[m * XS + idi[4 * s + q]];
#pragma unroll
      for (int s = 0; s < NSC; ++s) hin[cb][DI / 4 + s] = ct[m * CS + 4 * s + q];
    }
    if (!INV) {
      __syncthreads();
      VCNF_IDENTITY_PASS()
    }
#undef VCNF_IDENTITY_PASS

    // ---- conditioner trunk (nets/resnet.py:92-104); every layer one 16-row block at a time
    using L = PackLayout<DI, DT, C, H, NBLK, K>;
    constexpr int NS0_4 = NS0 / 4, NSH_4 = NSH / 4, NSC_4 = (NSC > 0 ? NSC : 4) / 4;
    const __amdgpu_buffer_rsrc_t wr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lay.wpack), 0, a.wpack_bytes, 0x00020000);
    const int voff = lane * 16;       // fragment loads: 16 bytes per lane
    const int qoff = q * 16;          // bias loads: 4 consecutive rows per lane group
    floatx4 h[kCB][NB];
    auto op_in = [&](int cb, int s) { return hin[cb][s]; };
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      floatx4 acc[kCB];
      bias_block(wr, qoff, 4 * (L::B0 + 16 * nb), acc);
      dense_block<NS0_4, kCB>(wr, voff, 4 * (L::W0 + nb * NS0 * 64), op_in, acc);
#pragma unroll
      for (int cb = 0; cb < kCB; ++cb) h[cb][nb] = acc[cb];
    }
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      const int base = L::BLK0 + blk * L::BLK;
      floatx4 t[kCB][NB];
      // context operand re-read from the LDS tile (keeps 2*C/4 registers free across the block)
      auto op_c = [&](int cb, int s) { return ct[((wave * kCB + cb) * 16 + m16) * CS + 4 * s + q]; };
      auto op_h = [&](int cb, int s) { return fmaxf(h[cb][s >> 2][s & 3], 0.f); };      // resnet.py:42
      auto op_t = [&](int cb, int s) { return fmaxf(t[cb][s >> 2][s & 3], 0.f); };      // :46
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {                                                // :43
        floatx4 acc[kCB];
        bias_block(wr, qoff, 4 * (base + L::BA + 16 * nb), acc);
        dense_block<NSH_4, kCB>(wr, voff, 4 * (base + L::WA + nb * NSH * 64), op_h, acc);
#pragma unroll
        for (int cb = 0; cb < kCB; ++cb) t[cb][nb] = acc[cb];
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {                                                // :48-57
        floatx4 acc[kCB];
        bias_block(wr, qoff, 4 * (base + L::BB + 16 * nb), acc);
        dense_block<NSH_4, kCB>(wr, voff, 4 * (base + L::WB + nb * NSH * 64), op_t, acc);
        if (C > 0) {                                                                   // GLU gate
          floatx4 gate[kCB];
          bias_block(wr, qoff, 4 * (base + L::BC + 16 * nb), gate);
          dense_block<NSC_4, kCB>(wr, voff, 4 * (base + L::WC + nb * (NSC > 0 ? NSC : 4) * 64), op_c, gate);
#pragma unroll
          for (int cb = 0; cb < kCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float sg = div_nr(1.f, 1.f + hw_exp2(-gate[cb][r] * kLog2e));
              h[cb][nb][r] = fmaf(acc[cb][r], sg, h[cb][nb][r]);
            }
        } else {
#pragma unroll
          for (int cb = 0; cb < kCB; ++cb) h[cb][nb] += acc[cb];
        }
      }
    }

    // ---- last layer + splines, 4 features (one per lane group) at a time
    auto op_f = [&](int cb, int s) { return h[cb][s >> 2][s & 3]; };
    for (int g = 0; g < NG; ++g) {
      floatx4 pa[kCB][P4];
#pragma unroll
      for (int b = 0; b < P4; ++b) {                                                   // resnet.py:105
        floatx4 acc[kCB];
        bias_block(wr, q * (16 * P4), 4 * (L::BF + g * 4 * (4 * P4) + 4 * b), acc);   // bf[g][q][4b..4b+3]
        dense_block<NSH_4, kCB>(wr, voff, 4 * (L::WF + (g * P4 + b) * NSH * 64), op_f, acc);
#pragma unroll
        for (int cb = 0; cb < kCB; ++cb) pa[cb][b] = acc[cb];
      }
      const int col = tfi[4 * g + q];
#pragma unroll
      for (int cb = 0; cb < kCB; ++cb) {
        const int m = (wave * kCB + cb) * 16 + m16;
        float* px = xt + m * XS + col;
        const float xv = *px;
        RegLogits<K, P4> p{pa[cb], c.wh_scale, c.edge_logit};
        float yv, lad;
        if (c.tails == 1 && !((xv >= c.lo_x) && (xv <= c.hi_x))) {
          yv = xv;
          lad = 0.f;
        } else {
          RqsBin sel;
          rqs_select<K, INV>(xv, p, c, c.wh_scale * kLog2e, sel);
          rqs_bin_eval<INV>(xv, sel, yv, lad, bad);
        }
        *px = yv;
        ld_acc[cb] += lad;
      }
    }

    // ---- per-sample log|det| of this layer: the 4 lane groups of a sample
#pragma unroll
    for (int cb = 0; cb < kCB; ++cb) {
      float v = ld_acc[cb];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      ld_tot[cb] += v;
    }
    __syncthreads();                        // the tile now holds this layer's output
   }   // layers
#pragma unroll
    for (int cb = 0; cb < kCB; ++cb) {
      const float v = ld_tot[cb];
      const int m = (wave * kCB + cb) * 16 + m16;
      if (q == 0 && m < rows && ((wmask >> (m / kFusedFlagRows)) & 1u)) {
        const float o = a.ld_sign * v;
        a.logdet[b0 + m] = a.ld_mode ? a.logdet[b0 + m] + o : o;
      }
    }
    {
      constexpr int D4 = D / 4;
      float4* dst = reinterpret_cast<float4*>(a.y) + b0 * D4;
      for (int i = tid; i < rows * D4; i += kFBlock) {
        const int r = i / D4, o = i - r * D4;
        if ((wmask >> (r / kFusedFlagRows)) & 1u) dst[i] = *reinterpret_cast<const float4*>(xt + r * XS + 4 * o);
      }
    }
  }
  if (INV && a.bad && bad) atomicAdd(a.bad, 1);
}

template <int DI, int DT, int C, int H, int NBLK, int K, int kCB>
static int launch_fused(const FusedStackArgs& sa, int inverse, hipStream_t st) {
  const FusedArgs& a = sa.a;
  constexpr int kTile = 4 * kCB * 16;
  constexpr int D = DI + DT;
  const size_t lds = ((size_t)kTile * (D + 4) + (size_t)kTile * ((C > 0 ? C : 4) + 4) +
                      ((DI * 3 * (K + 1) + 3) & ~3) + D) * 4 + 64;
  const long long ntiles = (a.B + kTile - 1) / kTile;
  const long long resident = 256 * 2;   // workgroups the chip holds at once
  dim3 grid((unsigned)(ntiles < resident ? ntiles : resident));
  if (sa.n_layers > 1) {
    if (inverse)
      hipLaunchKernelGGL((fused_rqs_layer_kernel<DI, DT, C, H, NBLK, K, true, kCB, true>), grid, dim3(kFBlock), lds, st, sa);
    else
      hipLaunchKernelGGL((fused_rqs_layer_kernel<DI, DT, C, H, NBLK, K, false, kCB, true>), grid, dim3(kFBlock), lds, st, sa);
  } else if (inverse)
    hipLaunchKernelGGL((fused_rqs_layer_kernel<DI, DT, C, H, NBLK, K, true, kCB, false>), grid, dim3(kFBlock), lds, st, sa);
  else
    hipLaunchKernelGGL((fused_rqs_layer_kernel<DI, DT, C, H, NBLK, K, false, kCB, false>), grid, dim3(kFBlock), lds, st, sa);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

// Shape family of the exact fp32 kernel: (d_id = d_t, ctx) with H = 128, 8 bins, NBLK residual blocks.
template <int NBLK>
static int launch_fused_f32_family(const FusedStackArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
  if (d_id == 32)
    return ctx_dim == 16 ? launch_fused<32, 32, 16, 128, NBLK, 8, 2>(a, inverse, st)
                         : launch_fused<32, 32, 0, 128, NBLK, 8, 2>(a, inverse, st);
  return ctx_dim == 16 ? launch_fused<16, 16, 16, 128, NBLK, 8, 2>(a, inverse, st)
                       : launch_fused<16, 16, 0, 128, NBLK, 8, 2>(a, inverse, st);
}

// One- and three-block layers are compiled as their own translation units (-DVCNF_F32_NBLK=1|3, build.py runs
// them in parallel); the unit without the macro holds the two-block kernels and the C entry points.
#if defined(VCNF_F32_NBLK) && VCNF_F32_NBLK == 1
int launch_fused_f32_b1(const FusedStackArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
  return launch_fused_f32_family<1>(a, d_id, ctx_dim, inverse, st);
}
#elif defined(VCNF_F32_NBLK) && VCNF_F32_NBLK == 3
int launch_fused_f32_b3(const FusedStackArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
  return launch_fused_f32_family<3>(a, d_id, ctx_dim, inverse, st);
}
#endif

}  // namespace vcnf

#ifndef VCNF_F32_NBLK
using namespace vcnf;

// shape family: d_id = d_t in {16, 32}, context 0 or 16, 1-3 residual blocks (hidden 128, 8 bins, linear
// tails), on both matrix paths.
template <int NBLK>
static int64_t pack_total(int d_id, int ctx_dim) {
  if (d_id == 32) return ctx_dim == 16 ? PackLayout<32, 32, 16, 128, NBLK, 8>::TOTAL : PackLayout<32, 32, 0, 128, NBLK, 8>::TOTAL;
  return ctx_dim == 16 ? PackLayout<16, 16, 16, 128, NBLK, 8>::TOTAL : PackLayout<16, 16, 0, 128, NBLK, 8>::TOTAL;
}

extern "C" int64_t vcnf_rqs_layer_fused_pack_floats(int32_t d_id, int32_t d_t, int32_t ctx_dim, int32_t num_blocks) {
  if (d_id != d_t || (d_id != 16 && d_id != 32) || (ctx_dim != 0 && ctx_dim != 16)) return 0;
  switch (num_blocks) {
    case 1: return pack_total<1>(d_id, ctx_dim);
    case 2: return pack_total<2>(d_id, ctx_dim);
    case 3: return pack_total<3>(d_id, ctx_dim);
    default: return 0;
  }
}

extern "C" int32_t vcnf_rqs_layer_fused_tile_rows(void) { return kFusedFlagRows; }

extern "C" int vcnf_rqs_layer_fused_supported(int32_t d_id, int32_t d_t, int32_t ctx_dim, int32_t hidden,
                                              int32_t num_blocks, int32_t num_bins, int32_t tails) {
  if (tails != VCNF_TAILS_LINEAR || num_bins != 8 || hidden != 128) return 0;
  return vcnf_rqs_layer_fused_pack_floats(d_id, d_t, ctx_dim, num_blocks) > 0 ? 1 : 0;
}

// fp16 split-half matrix path: fused_layer_v6.hip, one translation unit per number of residual blocks.
// (Earlier structures v2 - v6 are kept as text under profiles/tools/superseded/; they are no longer part
// of the library.)
// Batches up to this many samples take the 32-sample-tile kernel (fused_layer_v6s.hip): below it the 128-sample
// kernel leaves most of the chip idle (2048 samples = 16 workgroups), above it (>= 256 tiles of 128) its
// wave-group overlap wins.  Process-wide, set by vcnf_rqs_layer_fused_small_batch_rows().
static long long g_small_batch_rows = 16384;

extern "C" int64_t vcnf_rqs_layer_fused_small_batch_rows(int64_t rows) {
  const long long prev = g_small_batch_rows;
  if (rows >= 0) g_small_batch_rows = rows;
  return prev;
}

// sa.n_layers == 1: one layer; > 1: a run of layers in one launch (small-batch kernel on the split-half path: the
// 128-sample kernel has no layer loop)
static int launch_f16x3(const FusedStackArgs& sa, int d_id, int ctx_dim, int num_blocks, int inverse, hipStream_t st) {
  if (sa.n_layers > 1 || sa.a.B <= g_small_batch_rows) {
    if (num_blocks == 1) return launch_fused_v6s_b1(sa, d_id, ctx_dim, inverse, st);
    if (num_blocks == 2) return launch_fused_v6s_b2(sa, d_id, ctx_dim, inverse, st);
    return launch_fused_v6s_b3(sa, d_id, ctx_dim, inverse, st);
  }
  if (num_blocks == 1) return launch_fused_v6_b1(sa.a, d_id, ctx_dim, inverse, st);
  if (num_blocks == 2) return launch_fused_v6_b2(sa.a, d_id, ctx_dim, inverse, st);
  return launch_fused_v6_b3(sa.a, d_id, ctx_dim, inverse, st);
}

static int run_stack(const float* x, const float* context, float* y, float* logdet, int64_t batch,
                     const vcnf_rqs_stack_layer* layers, int32_t n_layers, int32_t d_t, int32_t d_id,
                     int32_t ctx_dim, int32_t hidden, int32_t num_blocks, int32_t precision, int64_t wpack_floats,
                     const vcnf_rqs_cfg* cfg, int inverse, int ld_mode, float ld_sign, int32_t* bad_disc,
                     int32_t* sat_count, int32_t* redo_tiles, void* stream) {
  if (!cfg || !layers) return VCNF_ERR_NULL;
  if (n_layers < 1 || n_layers > kMaxStackLayers) return VCNF_ERR_SHAPE;
  if (!vcnf_rqs_layer_fused_supported(d_id, d_t, ctx_dim, hidden, num_blocks, cfg->num_bins, cfg->tails))
    return VCNF_ERR_UNSUPPORTED;
  if (precision != VCNF_PREC_F32 && precision != VCNF_PREC_F16X3) return VCNF_ERR_UNSUPPORTED;
  if (batch < 0) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!x || !y || !logdet || (ctx_dim > 0 && !context)) return VCNF_ERR_NULL;
  const bool any_sh = layers[0].shared_w != nullptr;
  for (int l = 0; l < n_layers; ++l) {
    const vcnf_rqs_stack_layer& q = layers[l];
    if (!q.wpack || !q.transform_idx || !q.identity_idx) return VCNF_ERR_NULL;
    if ((q.shared_w != nullptr) != any_sh || (q.shared_h != nullptr) != any_sh || (q.shared_d != nullptr) != any_sh)
      return VCNF_ERR_NULL;                 // every layer of a run with the unconditional spline, or none
  }
  if (ld_mode != VCNF_LD_STORE && ld_mode != VCNF_LD_ACCUM) return VCNF_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(context)) & 15)
    return VCNF_ERR_ALIGN;
  if ((double)cfg->min_bin_width * cfg->num_bins > 1.0 || (double)cfg->min_bin_height * cfg->num_bins > 1.0)
    return VCNF_ERR_VALUE;
  if (wpack_floats != vcnf_rqs_layer_fused_pack_floats(d_id, d_t, ctx_dim, num_blocks)) return VCNF_ERR_SHAPE;

  FusedStackArgs sa;
  FusedArgs& a = sa.a;
  a.x = x; a.ctx = context; a.y = y; a.logdet = logdet;
  a.tf_idx = layers[0].transform_idx; a.id_idx = layers[0].identity_idx;
  a.sh_w = layers[0].shared_w; a.sh_h = layers[0].shared_h; a.sh_d = layers[0].shared_d;
  a.wpack = layers[0].wpack; a.wpack_bytes = (unsigned)(wpack_floats * 4);
  a.bad = bad_disc; a.sat = sat_count; a.redo = redo_tiles; a.B = batch; a.ld_mode = ld_mode; a.ld_sign = ld_sign;
  const int K = cfg->num_bins;
  a.c.K = K; a.c.tails = cfg->tails;
  a.c.lo_x = cfg->left; a.c.hi_x = cfg->right; a.c.span_x = (float)((double)cfg->right - (double)cfg->left);
  a.c.lo_y = cfg->bottom; a.c.hi_y = cfg->top; a.c.span_y = (float)((double)cfg->top - (double)cfg->bottom);
  a.c.min_w = cfg->min_bin_width; a.c.min_h = cfg->min_bin_height; a.c.min_d = cfg->min_derivative;
  a.c.free_w = (float)(1.0 - (double)cfg->min_bin_width * K);
  a.c.free_h = (float)(1.0 - (double)cfg->min_bin_height * K);
  a.c.wh_scale = cfg->wh_scale;
  a.c.edge_logit = (float)log(exp(1.0 - (double)cfg->min_derivative) - 1.0);
  sa.n_layers = n_layers;
  for (int l = 0; l < kMaxStackLayers; ++l) {
    const vcnf_rqs_stack_layer& q = layers[l < n_layers ? l : 0];
    sa.lay[l] = FusedLayerDesc{q.transform_idx, q.identity_idx, q.shared_w, q.shared_h, q.shared_d, q.wpack};
  }
  hipStream_t st = (hipStream_t)stream;
  if (precision == VCNF_PREC_F16X3) return launch_f16x3(sa, d_id, ctx_dim, num_blocks, inverse, st);
  if (num_blocks == 1) return launch_fused_f32_b1(sa, d_id, ctx_dim, inverse, st);
  if (num_blocks == 3) return launch_fused_f32_b3(sa, d_id, ctx_dim, inverse, st);
  return launch_fused_f32_family<2>(sa, d_id, ctx_dim, inverse, st);
}

extern "C" int32_t vcnf_rqs_stack_fused_max_layers(void) { return kMaxStackLayers; }

extern "C" int vcnf_rqs_stack_fused_f32(const float* x, const float* context, float* y, float* logdet,
                                        int64_t batch, const vcnf_rqs_stack_layer* layers, int32_t n_layers,
                                        int32_t d_t, int32_t d_id, int32_t ctx_dim, int32_t hidden,
                                        int32_t num_blocks, int32_t precision, int64_t wpack_floats,
                                        const vcnf_rqs_cfg* cfg, int inverse, int ld_mode, float ld_sign,
                                        int32_t* bad_disc, int32_t* sat_count, int32_t* redo_tiles, void* stream) {
  return run_stack(x, context, y, logdet, batch, layers, n_layers, d_t, d_id, ctx_dim, hidden, num_blocks, precision,
                   wpack_floats, cfg, inverse, ld_mode, ld_sign, bad_disc, sat_count, redo_tiles, stream);
}

extern "C" int vcnf_rqs_layer_fused_f32(const float* x, const float* context, float* y, float* logdet,
                                        int64_t batch, const int32_t* transform_idx, int32_t d_t,
                                        const int32_t* identity_idx, int32_t d_id, int32_t ctx_dim,
                                        int32_t hidden, int32_t num_blocks, int32_t precision,
                                        const float* wpack, int64_t wpack_floats,
                                        const float* shared_w, const float* shared_h, const float* shared_d,
                                        const vcnf_rqs_cfg* cfg, int inverse,
                                        int ld_mode, float ld_sign, int32_t* bad_disc, int32_t* sat_count,
                                        int32_t* redo_tiles, void* stream) {
  if (!wpack) return VCNF_ERR_NULL;
  if (batch > 0 && (!transform_idx || !identity_idx)) return VCNF_ERR_NULL;
  const bool any_sh = shared_w || shared_h || shared_d;
  if (any_sh && !(shared_w && shared_h && shared_d)) return VCNF_ERR_NULL;
  const vcnf_rqs_stack_layer one = {transform_idx, identity_idx, wpack, shared_w, shared_h, shared_d};
  return run_stack(x, context, y, logdet, batch, &one, 1, d_t, d_id, ctx_dim, hidden, num_blocks, precision,
                   wpack_floats, cfg, inverse, ld_mode, ld_sign, bad_disc, sat_count, redo_tiles, stream);
}
#endif  // VCNF_F32_NBLK
