// Fused RQS coupling layer, second structure: fp16 "split-half" matrix path.
//
// Same contract as fused_layer.hip (whole layer per launch: conditioner + splines +
// log|det|), built for the 16x faster v_mfma_f32_16x16x32_f16.  An fp32 value v is split
// into hi = fp16(v) and lo = fp16((v - hi) * 2^11) - 22 significant bits - and a product
// keeps hi*hi + (hi*lo + lo*hi) * 2^-11 in two fp32 accumulators (3 matrix instructions
// per 32-deep k-step; the dropped lo*lo term is 2^-22 relative).  Measured on the C3
// stack the result is indistinguishable from the fp32 matrix path: the spline's own
// conditioning, not the conditioner GEMMs, sets the error (scratch numbers in DESIGN.md).
//
// With matrix time cut ~5x the layer is bound by operand delivery, so the work split is
// chosen to fetch every weight byte ONCE per 128-sample tile and workgroup:
//   * 512 threads = 8 waves per workgroup, one workgroup per CU, tile = 128 samples
//     (8 column blocks of 16);
//   * trunk layers: wave w owns hidden row block w (16 units) for ALL 8 column blocks and
//     keeps its weight slice in registers for the whole layer ("weight stationary");
//     activations travel between layers through LDS, already split into fp16 hi/lo and
//     stored in B-fragment order, so a consumer reads its operand with one ds_read_b128;
//     producers hold their outputs in registers until every wave has finished reading
//     (one barrier), then overwrite the buffer in place;
//   * last layer + splines: wave w owns column block w (16 samples) with its operands in
//     registers; the 393 KB of last-layer weights stream through a 48 KB LDS window, one
//     feature group (6 row blocks) at a time, shared by the 8 waves.
// The first layer and the GLU gates see raw inputs (unbounded range) and stay on the
// fp32 matrix instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"
#include "rqs_math.hpp"
#include "fused_common.hpp"

namespace vcnf {

// NW waves per workgroup (4 or 8).  Trunk: wave w owns hidden row blocks w*RBW .. +RBW-1 for
// all NW column blocks of the tile; last layer: wave w owns column block w.  Tile = 16*NW
// samples.  NW = 8: one 512-thread workgroup per CU, every weight byte fetched once per 128
// samples; NW = 4: two independent 256-thread workgroups per CU (their phases interleave and
// hide each other's barriers and load latencies), weights fetched once per 64 samples.
template <int DI, int DT, int C, int H, int NBLK, int K, bool INV, int NW>
__global__ __launch_bounds__(64 * NW, 2) void fused_rqs_layer_v2_kernel(const FusedArgs a) {
  static_assert(H == 128 && (NW == 4 || NW == 8), "8 hidden row blocks over NW waves");
  constexpr int kV2Block = 64 * NW;
  constexpr int kV2Tile = 16 * NW;
  constexpr int kV2CB = NW;
  constexpr int RBW = 8 / NW;
  constexpr int D = DI + DT;
  constexpr int XS = D + 4;
  constexpr int CS = (C > 0 ? C : 4) + 4;
  constexpr int NS0 = (DI + C) / 4;         // fp32 k-steps of the first layer
  constexpr int NS0_4 = NS0 / 4;
  constexpr int NSC = C / 4;
  constexpr int NS32 = H / 32;              // fp16 k-steps of a hidden->* layer (4)
  constexpr int P = 3 * K - 1;
  constexpr int P4 = (P + 3) / 4;           // 6
  constexpr int NG = DT / 4;                // 8
  constexpr int TABW = 3 * (K + 1);
  using L = PackLayout<DI, DT, C, H, NBLK, K>;
  constexpr int HALF_W = (H / 16) * (H / 4) * 64 / 2;     // floats of the hi half of a hidden layer
  constexpr int HALF_F = NG * P4 * (H / 4) * 64 / 2;

  extern __shared__ __align__(16) float smem[];
  float* xt = smem;                                        // [128][XS]  x in, y out (in place)
  float* ct = xt + kV2Tile * XS;                           // [128][CS]
  float* tab = ct + kV2Tile * CS;                          // [DI][TABW]
  float* ldt = tab + ((DI * TABW + 3) & ~3);               // [128] identity-half log|det|
  int* tfi = reinterpret_cast<int*>(ldt + kV2Tile);
  int* idi = tfi + DT;
  // activation fragments [s][cb][lane][8 halves]: hi then lo (32 KB each); the same 64 KB
  // serve as the last layer's weight window [b][s][hi|lo][lane][8 halves] (48 KB)
  uint4* act = reinterpret_cast<uint4*>(idi + DI + ((4 - ((DT + DI) & 3)) & 3));
  uint4* act_hi = act;
  uint4* act_lo = act + NS32 * kV2CB * 64;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: feeds scalar offsets
  const int m16 = lane & 15;
  const int q = lane >> 4;
  const RqsConst& c = a.c;
  const bool shared = a.sh_w != nullptr;

  for (int i = tid; i < DT; i += kV2Block) tfi[i] = a.tf_idx[i];
  for (int i = tid; i < DI; i += kV2Block) idi[i] = a.id_idx[i];
  if (shared) {
    for (int f = tid; f < DI; f += kV2Block) {
      SplitLogits p{a.sh_w + f * K, a.sh_h + f * K, a.sh_d + f * (K - 1), K, 1.f, c.edge_logit, c.tails};
      rqs_build_table(p, c, tab + f * TABW);
    }
  }

  const __amdgpu_buffer_rsrc_t wr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpack), 0, a.wpack_bytes, 0x00020000);
  const int voff = lane * 16;
  const int qoff = q * 16;
  // where this wave's outputs go in the next layer's operand: row block nb -> k-step nb/2,
  // 8-byte half nb&1 of the lane's 16-byte fragment
  uint2* frag_hi = reinterpret_cast<uint2*>(act_hi);
  uint2* frag_lo = reinterpret_cast<uint2*>(act_lo);

  const long long ntiles = (a.B + kV2Tile - 1) / kV2Tile;
  bool bad = false;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long b0 = tile * kV2Tile;
    const int rows = (int)min((long long)kV2Tile, a.B - b0);
    __syncthreads();
    {   // ---- stage x and context rows
      constexpr int D4 = D / 4;
      const float4* sx = reinterpret_cast<const float4*>(a.x) + b0 * D4;
      for (int i = tid; i < kV2Tile * D4; i += kV2Block) {
        const int r = i / D4, o = i - r * D4;
        const float4 v = r < rows ? sx[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(xt + r * XS + 4 * o) = v;
      }
      if (C > 0) {
        constexpr int C4 = (C > 0 ? C : 4) / 4;
        const float4* sc = reinterpret_cast<const float4*>(a.ctx) + b0 * C4;
        for (int i = tid; i < kV2Tile * C4; i += kV2Block) {
          const int r = i / C4, o = i - r * C4;
          const float4 v = r < rows ? sc[i] : make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4*>(ct + r * CS + 4 * o) = v;
        }
      }
    }
    __syncthreads();

    // ---- identity half through the unconditional spline: 4 lanes per sample, each lane a run
    // of DI/4 features; per-sample log|det| of this half parked in LDS.
#define VCNF_IDENTITY_PASS()                                                              \
  for (int mi = tid >> 2; mi < kV2Tile; mi += kV2Block / 4) {                             \
    float lsum = 0.f;                                                                     \
    _Pragma("unroll") for (int k = 0; k < DI / 4; ++k) {                                  \
      const int f = (tid & 3) * (DI / 4) + k;                                             \
      float* px = xt + mi * XS + idi[f];                                                  \
      const float xv = *px;                                                               \
      float yv = xv, lad = 0.f;                                                           \
      if (shared) rqs_point_table<INV, K>(xv, tab + f * TABW, c, yv, lad, bad);           \
      *px = yv;                                                                           \
      lsum += lad;                                                                        \
    }                                                                                     \
    lsum += __shfl_xor(lsum, 1, 64);                                                      \
    lsum += __shfl_xor(lsum, 2, 64);                                                      \
    if ((tid & 3) == 0) ldt[mi] = lsum;                                                   \
  }
    if (INV) {
      VCNF_IDENTITY_PASS()
      __syncthreads();
    }

    // ---- first layer on the fp32 instruction: this wave's row blocks, all column blocks
    floatx4 h[RBW][kV2CB];
    {
      floatx4 w0[RBW][NS0_4], bias[RBW];
#pragma unroll
      for (int rb = 0; rb < RBW; ++rb) {
        const int nb = wave * RBW + rb;
#pragma unroll
        for (int s4 = 0; s4 < NS0_4; ++s4) w0[rb][s4] = wload(wr, voff, 4 * (L::W0 + (nb * NS0_4 + s4) * 256));
        bias[rb] = wload(wr, qoff, 4 * (L::B0 + 16 * nb));
      }
      // operand columns of this lane: identity feature 4 s + q (read once, not per use)
      int xcol[DI / 4];
#pragma unroll
      for (int s = 0; s < DI / 4; ++s) xcol[s] = idi[4 * s + q];
      float bv[kV2CB][NS0];
#pragma unroll
      for (int cb = 0; cb < kV2CB; ++cb) {
        const float* xr = xt + (cb * 16 + m16) * XS;
        const float* cr = ct + (cb * 16 + m16) * CS;
#pragma unroll
        for (int s = 0; s < NS0; ++s) bv[cb][s] = s < DI / 4 ? xr[xcol[s]] : cr[4 * (s - DI / 4) + q];
      }
#pragma unroll
      for (int cb = 0; cb < kV2CB; ++cb) {
        floatx4 acc[RBW];
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) acc[rb] = bias[rb];
#pragma unroll
        for (int s = 0; s < NS0; ++s) {
#pragma unroll
          for (int rb = 0; rb < RBW; ++rb) acc[rb] = mfma4(w0[rb][s >> 2][s & 3], bv[cb][s], acc[rb]);
        }
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) h[rb][cb] = acc[rb];
      }
      // schedule: a column block's operand reads run one block ahead of its matrix work
      // (left alone the scheduler puts every LDS read right before the MFMA that uses it
      // and each MFMA then waits out a full LDS round trip)
      __builtin_amdgcn_sched_group_barrier(0x100, NS0, 0);
#pragma unroll
      for (int cb = 0; cb + 1 < kV2CB; ++cb) {
        __builtin_amdgcn_sched_group_barrier(0x100, NS0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NS0 * RBW, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NS0 * RBW, 0);
    }
    if (!INV) {
      __syncthreads();
      VCNF_IDENTITY_PASS()
    }
#undef VCNF_IDENTITY_PASS

    // publish relu(h) as the operand of the first hidden layer
#define VCNF_PUBLISH(SRC, RELU)                                                           \
  _Pragma("unroll") for (int rb = 0; rb < RBW; ++rb) {                                    \
    const int nb = wave * RBW + rb;                                                       \
    _Pragma("unroll") for (int cb = 0; cb < kV2CB; ++cb) {                                \
      half4 phi, plo;                                                                     \
      split4<RELU>(SRC[rb][cb], phi, plo);                                                \
      const int at = (((nb >> 1) * kV2CB + cb) * 64 + lane) * 2 + (nb & 1);               \
      frag_hi[at] = __builtin_bit_cast(uint2, phi);                                       \
      frag_lo[at] = __builtin_bit_cast(uint2, plo);                                       \
    }                                                                                     \
  }
    VCNF_PUBLISH(h, true)
    __syncthreads();

    // one hidden->hidden layer for this wave's row block: weights stationary in registers,
    // operands streamed from LDS.  OUT[cb] = bias + W_slice * operand(cb).
#define VCNF_HIDDEN_LAYER(WOFF, BOFF, OUT)                                                \
  {                                                                                       \
    half8 ahi[RBW][NS32], alo[RBW][NS32];                                                 \
    floatx4 bias[RBW];                                                                    \
    _Pragma("unroll") for (int rb = 0; rb < RBW; ++rb) {                                  \
      const int nb = wave * RBW + rb;                                                     \
      _Pragma("unroll") for (int s = 0; s < NS32; ++s) {                                  \
        ahi[rb][s] = __builtin_bit_cast(half8, wload(wr, voff, 4 * ((WOFF) + (nb * NS32 + s) * 256)));          \
        alo[rb][s] = __builtin_bit_cast(half8, wload(wr, voff, 4 * ((WOFF) + HALF_W + (nb * NS32 + s) * 256))); \
      }                                                                                   \
      bias[rb] = wload(wr, qoff, 4 * ((BOFF) + 16 * nb));                                 \
    }                                                                                     \
    _Pragma("unroll") for (int cb = 0; cb < kV2CB; ++cb) {                                \
      floatx4 mainv[RBW], corr[RBW];                                                      \
      _Pragma("unroll") for (int rb = 0; rb < RBW; ++rb) {                                \
        mainv[rb] = bias[rb];                                                             \
        corr[rb] = floatx4{0.f, 0.f, 0.f, 0.f};                                           \
      }                                                                                   \
      _Pragma("unroll") for (int s = 0; s < NS32; ++s) {                                  \
        const half8 bhi = __builtin_bit_cast(half8, act_hi[(s * kV2CB + cb) * 64 + lane]);                      \
        const half8 blo = __builtin_bit_cast(half8, act_lo[(s * kV2CB + cb) * 64 + lane]);                      \
        _Pragma("unroll") for (int rb = 0; rb < RBW; ++rb) mainv[rb] = mfma16h(ahi[rb][s], bhi, mainv[rb]);     \
        _Pragma("unroll") for (int rb = 0; rb < RBW; ++rb) corr[rb] = mfma16h(ahi[rb][s], blo, corr[rb]);       \
        _Pragma("unroll") for (int rb = 0; rb < RBW; ++rb) corr[rb] = mfma16h(alo[rb][s], bhi, corr[rb]);       \
      }                                                                                   \
      _Pragma("unroll") for (int rb = 0; rb < RBW; ++rb)                                  \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                     \
          OUT[rb][cb][r] = fmaf(corr[rb][r], kLoUnscale, mainv[rb][r]);                   \
    }                                                                                     \
  }

#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      const int base = L::BLK0 + blk * L::BLK;
      floatx4 t[RBW][kV2CB];
      VCNF_HIDDEN_LAYER(base + L::WA, base + L::BA, t)                                  // resnet.py:42-43
      __syncthreads();                       // every wave is done reading relu(h)
      VCNF_PUBLISH(t, true)                                                             // :46
      __syncthreads();
      VCNF_HIDDEN_LAYER(base + L::WB, base + L::BB, t)                                  // :48
      if (C > 0) {                                                                      // :49-56 GLU gate, fp32
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) {
          const int nb = wave * RBW + rb;
          const floatx4 wc = wload(wr, voff, 4 * (base + L::WC + nb * (NSC > 0 ? NSC : 4) * 64));
          const floatx4 bc = wload(wr, qoff, 4 * (base + L::BC + 16 * nb));
#pragma unroll
          for (int cb = 0; cb < kV2CB; ++cb) {
            const float* cr = ct + (cb * 16 + m16) * CS;
            floatx4 gate = bc;
#pragma unroll
            for (int s = 0; s < NSC; ++s) gate = mfma4(wc[s], cr[4 * s + q], gate);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float sg = div_nr(1.f, 1.f + hw_exp2(-gate[r] * kLog2e));
              h[rb][cb][r] = fmaf(t[rb][cb][r], sg, h[rb][cb][r]);                      // :57
            }
          }
        }
      } else {
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
          for (int cb = 0; cb < kV2CB; ++cb) h[rb][cb] += t[rb][cb];
      }
      __syncthreads();                       // every wave is done reading relu(t)
      if (blk + 1 < NBLK) {
        VCNF_PUBLISH(h, true)
      } else {
        VCNF_PUBLISH(h, false)               // the last layer takes h itself (resnet.py:105)
      }
      __syncthreads();
    }
#undef VCNF_HIDDEN_LAYER
#undef VCNF_PUBLISH

    // ---- last layer + splines: wave w now owns column block w (samples 16w .. 16w+15)
    half8 fhi[NS32], flo[NS32];
#pragma unroll
    for (int s = 0; s < NS32; ++s) {
      fhi[s] = __builtin_bit_cast(half8, act_hi[(s * kV2CB + wave) * 64 + lane]);
      flo[s] = __builtin_bit_cast(half8, act_lo[(s * kV2CB + wave) * 64 + lane]);
    }
    float ld_acc = 0.f;
    const int mrow = wave * 16 + m16;
    for (int g = 0; g < NG; ++g) {
      __syncthreads();                       // window free (operands read / previous group consumed)
      {   // stage the group's weights: [b][s][hi|lo][lane] 16-byte fragments, 48 KB
        constexpr int NFRAG = P4 * NS32 * 2 * 64;
        for (int i = tid; i < NFRAG; i += kV2Block) {
          const int ln = i & 63, part = (i >> 6) & 1, bs = i >> 7;     // bs = b * NS32 + s
          act[i] = __builtin_bit_cast(uint4, wload(wr, 4 * (L::WF + part * HALF_F + (g * P4 * NS32 + bs) * 256) + ln * 16, 0));
        }
      }
      __syncthreads();
      floatx4 pa[P4];
#pragma unroll
      for (int b = 0; b < P4; ++b) {
        floatx4 mainv = wload(wr, q * (16 * P4), 4 * (L::BF + g * 4 * (4 * P4) + 4 * b));   // bf[g][q][4b..]
        floatx4 corr = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS32; ++s) {
          const half8 ahi = __builtin_bit_cast(half8, act[((b * NS32 + s) * 2 + 0) * 64 + lane]);
          const half8 alo = __builtin_bit_cast(half8, act[((b * NS32 + s) * 2 + 1) * 64 + lane]);
          mainv = mfma16h(ahi, fhi[s], mainv);
          corr = mfma16h(ahi, flo[s], corr);
          corr = mfma16h(alo, fhi[s], corr);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) pa[b][r] = fmaf(corr[r], kLoUnscale, mainv[r]);
      }
      float* px = xt + mrow * XS + tfi[4 * g + q];
      const float xv = *px;
      RegLogits<K, P4> p{pa, c.wh_scale, c.edge_logit};
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
      ld_acc += lad;
    }

    // ---- per-sample log|det|: 4 lane groups of the sample + the identity half
    ld_acc += __shfl_xor(ld_acc, 16, 64);
    ld_acc += __shfl_xor(ld_acc, 32, 64);
    if (q == 0 && mrow < rows) {
      const float o = a.ld_sign * (ld_acc + ldt[mrow]);
      a.logdet[b0 + mrow] = a.ld_mode ? a.logdet[b0 + mrow] + o : o;
    }
    __syncthreads();
    {
      constexpr int D4 = D / 4;
      float4* dst = reinterpret_cast<float4*>(a.y) + b0 * D4;
      for (int i = tid; i < rows * D4; i += kV2Block) {
        const int r = i / D4, o = i - r * D4;
        dst[i] = *reinterpret_cast<const float4*>(xt + r * XS + 4 * o);
      }
    }
  }
  if (INV && a.bad && bad) atomicAdd(a.bad, 1);
}

template <int DI, int DT, int C, int H, int NBLK, int K, int NW>
static int launch_v2(const FusedArgs& a, int inverse, hipStream_t st) {
  constexpr int D = DI + DT;
  constexpr int TILE = 16 * NW;
  constexpr size_t ACT = (size_t)2 * (H / 32) * NW * 64 * 16;      // activation fragments (hi + lo)
  constexpr size_t WIN = (size_t)((3 * K + 2) / 4) * (H / 32) * 2 * 64 * 16;   // last-layer weight window
  const size_t lds = ((size_t)TILE * (D + 4) + (size_t)TILE * ((C > 0 ? C : 4) + 4) +
                      ((DI * 3 * (K + 1) + 3) & ~3) + TILE + D + 4) * 4 + (ACT > WIN ? ACT : WIN) + 64;
  static bool attr_set[2] = {false, false};
  if (!attr_set[inverse ? 1 : 0]) {
    hipError_t e;
    if (inverse)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_rqs_layer_v2_kernel<DI, DT, C, H, NBLK, K, true, NW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_rqs_layer_v2_kernel<DI, DT, C, H, NBLK, K, false, NW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return VCNF_ERR_LAUNCH;
    attr_set[inverse ? 1 : 0] = true;
  }
  const long long ntiles = (a.B + TILE - 1) / TILE;
  const long long resident = 256 * (NW == 8 ? 1 : 2);
  dim3 grid((unsigned)(ntiles < resident ? ntiles : resident));
  if (inverse)
    hipLaunchKernelGGL((fused_rqs_layer_v2_kernel<DI, DT, C, H, NBLK, K, true, NW>), grid, dim3(64 * NW), lds, st, a);
  else
    hipLaunchKernelGGL((fused_rqs_layer_v2_kernel<DI, DT, C, H, NBLK, K, false, NW>), grid, dim3(64 * NW), lds, st, a);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

int launch_fused_v2_c16(const FusedArgs& a, int inverse, hipStream_t st) {
  return launch_v2<32, 32, 16, 128, 2, 8, 8>(a, inverse, st);
}

int launch_fused_v2_c0(const FusedArgs& a, int inverse, hipStream_t st) {
  return launch_v2<32, 32, 0, 128, 2, 8, 8>(a, inverse, st);
}

}  // namespace vcnf
