// Fused RQS coupling layer, fp16 split-half matrix path, third structure.
//
// Same contract, packed-weight layout and arithmetic as fused_layer_v2.hip; what changes is
// which wave computes what.  v2 gives every wave one 16x16 output tile per operand fetched from
// LDS (3 matrix instructions per 2 KB read in the trunk, per 2 KB in the last layer): at
// 128 B/clk of LDS bandwidth against one 16x16x32 instruction per SIMD every 16 clocks a CU
// can feed at most 0.5 KB per instruction, so both phases of v2 sit on the LDS pipe.  Here
// every operand read from LDS feeds TWO output tiles (0.33 KB per instruction):
//   * trunk: wave w owns hidden row blocks {2 (w&3), 2 (w&3) + 1} for the column half
//     (w>>2): its weight slice (2 row blocks, hi + lo) is stationary in registers, each
//     activation fragment it reads is used by both row blocks; its outputs are exactly one
//     16-byte operand fragment of the next layer (one ds_write_b128 per column block);
//   * last layer + splines: wave w owns the column-block PAIR (w&3) for feature groups of
//     parity (w>>2); the LDS window holds two feature groups (96 KB), every weight fragment
//     read from it multiplies both column blocks.  4 staging rounds per tile instead of 8.
// The next layer's stationary weights are requested before the barriers that separate two
// layers, so their L2 latency overlaps the publish / barrier sequence.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"
#include "rqs_math.hpp"
#include "fused_common.hpp"

#ifndef VCNF_ABL
#define VCNF_ABL 0
#endif

namespace vcnf {

template <int DI, int DT, int C, int H, int NBLK, int K, bool INV>
__global__ __launch_bounds__(512, 2) void fused_rqs_layer_v3_kernel(const FusedArgs a) {
  static_assert(H == 128, "8 hidden row blocks = 4 row pairs x 2 column halves over 8 waves");
  constexpr int kBlock = 512;
  constexpr int kTile = 128;
  constexpr int kCB = 8;                    // 16-sample column blocks per tile
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
  static_assert(NG % 2 == 0, "feature groups are processed two per round");
  constexpr int TABW = 3 * (K + 1);
  using L = PackLayout<DI, DT, C, H, NBLK, K>;
  constexpr int HALF_W = (H / 16) * (H / 4) * 64 / 2;     // floats of the hi half of a hidden layer
  constexpr int HALF_F = NG * P4 * (H / 4) * 64 / 2;
  constexpr int GFRAG = P4 * NS32 * 2 * 64;               // 16-byte fragments of one feature group

  extern __shared__ __align__(16) float smem[];
  float* xt = smem;                                        // [128][XS]  x in, y out (in place)
  float* ct = xt + kTile * XS;                             // [128][CS]
  float* tab = ct + kTile * CS;                            // [DI][TABW]
  float* ldt = tab + ((DI * TABW + 3) & ~3);               // [128] identity-half log|det|
  int* tfi = reinterpret_cast<int*>(ldt + kTile);
  int* idi = tfi + DT;
  // activation fragments [s][cb][lane][8 halves]: hi then lo (32 KB each); the same region
  // (96 KB) is the last layer's weight window [group parity][b][s][hi|lo][lane][8 halves]
  uint4* act = reinterpret_cast<uint4*>(idi + DI + ((4 - ((DT + DI) & 3)) & 3));
  uint4* act_hi = act;
  uint4* act_lo = act + NS32 * kCB * 64;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: feeds scalar offsets
  const int rp = wave & 3;                  // trunk: row-block pair; last layer: column-block pair
  const int ch = wave >> 2;                 // trunk: column half;    last layer: group parity
  const int m16 = lane & 15;
  const int q = lane >> 4;
  const RqsConst& c = a.c;
  const bool shared = a.sh_w != nullptr;

  for (int i = tid; i < DT; i += kBlock) tfi[i] = a.tf_idx[i];
  for (int i = tid; i < DI; i += kBlock) idi[i] = a.id_idx[i];
  if (shared) {
    for (int f = tid; f < DI; f += kBlock) {
      SplitLogits p{a.sh_w + f * K, a.sh_h + f * K, a.sh_d + f * (K - 1), K, 1.f, c.edge_logit, c.tails};
      rqs_build_table(p, c, tab + f * TABW);
    }
  }

  const __amdgpu_buffer_rsrc_t wr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpack), 0, a.wpack_bytes, 0x00020000);
  const int voff = lane * 16;
  const int qoff = q * 16;

  const long long ntiles = (a.B + kTile - 1) / kTile;
  bool bad = false;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long b0 = tile * kTile;
    const int rows = (int)min((long long)kTile, a.B - b0);
    __syncthreads();
    {   // ---- stage x and context rows
      constexpr int D4 = D / 4;
      const float4* sx = reinterpret_cast<const float4*>(a.x) + b0 * D4;
      for (int i = tid; i < kTile * D4; i += kBlock) {
        const int r = i / D4, o = i - r * D4;
        const float4 v = r < rows ? sx[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(xt + r * XS + 4 * o) = v;
      }
      if (C > 0) {
        constexpr int C4 = (C > 0 ? C : 4) / 4;
        const float4* sc = reinterpret_cast<const float4*>(a.ctx) + b0 * C4;
        for (int i = tid; i < kTile * C4; i += kBlock) {
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
  for (int mi = tid >> 2; mi < kTile; mi += kBlock / 4) {                                 \
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

    // stationary weights of a hidden->hidden layer for this wave's two row blocks
    half8 ahi[2][NS32], alo[2][NS32];
    floatx4 abias[2];
#define VCNF_LOAD_HIDDEN(WOFF, BOFF)                                                      \
  _Pragma("unroll") for (int rb = 0; rb < 2; ++rb) {                                      \
    const int nb = 2 * rp + rb;                                                           \
    _Pragma("unroll") for (int s = 0; s < NS32; ++s) {                                    \
      ahi[rb][s] = __builtin_bit_cast(half8, wload(wr, voff, 4 * ((WOFF) + (nb * NS32 + s) * 256)));          \
      alo[rb][s] = __builtin_bit_cast(half8, wload(wr, voff, 4 * ((WOFF) + HALF_W + (nb * NS32 + s) * 256))); \
    }                                                                                     \
    abias[rb] = wload(wr, qoff, 4 * ((BOFF) + 16 * nb));                                  \
  }

    // ---- first layer on the fp32 instruction: 2 row blocks x 4 column blocks per wave
    floatx4 h[2][4];
    {
      floatx4 w0[2][NS0_4], bias[2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const int nb = 2 * rp + rb;
#pragma unroll
        for (int s4 = 0; s4 < NS0_4; ++s4) w0[rb][s4] = wload(wr, voff, 4 * (L::W0 + (nb * NS0_4 + s4) * 256));
        bias[rb] = wload(wr, qoff, 4 * (L::B0 + 16 * nb));
      }
      int xcol[DI / 4];
#pragma unroll
      for (int s = 0; s < DI / 4; ++s) xcol[s] = idi[4 * s + q];
      float bv[4][NS0];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* xr = xt + ((4 * ch + j) * 16 + m16) * XS;
        const float* cr = ct + ((4 * ch + j) * 16 + m16) * CS;
#pragma unroll
        for (int s = 0; s < NS0; ++s) bv[j][s] = s < DI / 4 ? xr[xcol[s]] : cr[4 * (s - DI / 4) + q];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        floatx4 acc[2] = {bias[0], bias[1]};
#pragma unroll
        for (int s = 0; s < NS0; ++s) {
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) acc[rb] = mfma4(w0[rb][s >> 2][s & 3], bv[j][s], acc[rb]);
        }
        h[0][j] = acc[0];
        h[1][j] = acc[1];
      }
    }
    __builtin_amdgcn_sched_barrier(0);       // keep the 72 registers of the next loads out of the block above
    VCNF_LOAD_HIDDEN(L::BLK0 + L::WA, L::BLK0 + L::BA)
    if (!INV) {
      __syncthreads();
      VCNF_IDENTITY_PASS()
    }
#undef VCNF_IDENTITY_PASS

    // publish: the wave's two row blocks are the two 8-byte halves of one operand fragment
#define VCNF_PUBLISH(SRC, RELU)                                                           \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                         \
    half4 h0, l0, h1, l1;                                                                 \
    if (VCNF_ABL == 7) {                                                                  \
      h0 = __builtin_bit_cast(half4, make_float2(SRC[0][j][0], SRC[0][j][1]));            \
      l0 = __builtin_bit_cast(half4, make_float2(SRC[0][j][2], SRC[0][j][3]));            \
      h1 = __builtin_bit_cast(half4, make_float2(SRC[1][j][0], SRC[1][j][1]));            \
      l1 = __builtin_bit_cast(half4, make_float2(SRC[1][j][2], SRC[1][j][3]));            \
    } else {                                                                              \
    split4<RELU>(SRC[0][j], h0, l0);                                                      \
    split4<RELU>(SRC[1][j], h1, l1);                                                      \
    }                                                                                     \
    const int at = (rp * kCB + 4 * ch + j) * 64 + lane;                                   \
    act_hi[at] = __builtin_bit_cast(uint4, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7)); \
    act_lo[at] = __builtin_bit_cast(uint4, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7)); \
  }
    VCNF_PUBLISH(h, true)
    __syncthreads();

    // OUT[rb][j] = bias + W_slice(rb) * operand(column block 4 ch + j).  The operand fragments of
    // column block j + 1 are requested before the matrix work of block j starts (left alone the
    // compiler reads each fragment pair right before its six instructions and every k-step
    // waits out an LDS round trip: measured 0.47 ms of a 2.0 ms launch).
#define VCNF_READ_B(T)                                                                    \
  {                                                                                       \
    rh[(T) % 3] = __builtin_bit_cast(half8, act_hi[(((VCNF_ABL == 8 ? 0 : (T)) & 3) * kCB + 4 * ch + ((VCNF_ABL == 8 ? 0 : (T)) >> 2)) * 64 + lane]); \
    rl[(T) % 3] = __builtin_bit_cast(half8, act_lo[(((VCNF_ABL == 8 ? 0 : (T)) & 3) * kCB + 4 * ch + ((VCNF_ABL == 8 ? 0 : (T)) >> 2)) * 64 + lane]); \
  }
#define VCNF_HIDDEN_COMPUTE(OUT)                                                          \
  {                                                                                       \
    half8 rh[3], rl[3];                      /* ring: step t = 4 j + s uses slot t % 3 */  \
    floatx4 mainv[2], corr[2];                                                            \
    VCNF_READ_B(0)                                                                        \
    VCNF_READ_B(1)                                                                        \
    _Pragma("unroll") for (int st_ = 0; st_ < 4 * NS32; ++st_) {                                \
      const int j = st_ >> 2, s = st_ & 3;                                                    \
      if (st_ + 2 < 4 * NS32) {                                                             \
        VCNF_READ_B(st_ + 2)                                                                \
      }                                                                                   \
      if (s == 0) {                                                                       \
        _Pragma("unroll") for (int rb = 0; rb < 2; ++rb) {                                \
          mainv[rb] = abias[rb];                                                          \
          corr[rb] = floatx4{0.f, 0.f, 0.f, 0.f};                                         \
        }                                                                                 \
      }                                                                                   \
      _Pragma("unroll") for (int rb = 0; rb < 2; ++rb) mainv[rb] = mfma16h(ahi[rb][s], rh[st_ % 3], mainv[rb]); \
      _Pragma("unroll") for (int rb = 0; rb < 2; ++rb) corr[rb] = mfma16h(ahi[rb][s], rl[st_ % 3], corr[rb]);   \
      _Pragma("unroll") for (int rb = 0; rb < 2; ++rb) corr[rb] = mfma16h(alo[rb][s], rh[st_ % 3], corr[rb]);   \
      if (s == NS32 - 1) {                                                                \
        _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                  \
          _Pragma("unroll") for (int r = 0; r < 4; ++r)                                   \
            OUT[rb][j][r] = fmaf(corr[rb][r], kLoUnscale, mainv[rb][r]);                  \
      }                                                                                   \
    }                                                                                     \
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                    \
    _Pragma("unroll") for (int st_ = 0; st_ + 2 < 4 * NS32; ++st_) {                            \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                  \
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                  \
    }                                                                                     \
    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);                                   \
  }

#pragma unroll
    for (int blk = 0; blk < (VCNF_ABL == 5 ? 0 : NBLK); ++blk) {
      const int base = L::BLK0 + blk * L::BLK;
      floatx4 t[2][4];
      VCNF_HIDDEN_COMPUTE(t)                                                            // resnet.py:42-43
      __builtin_amdgcn_sched_barrier(0);
      VCNF_LOAD_HIDDEN(base + L::WB, base + L::BB)
      __syncthreads();                       // every wave is done reading relu(h)
      VCNF_PUBLISH(t, true)                                                             // :46
      floatx4 wc[2], bc[2];                  // gate weights: requested now, used after the second layer
      if (C > 0) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          wc[rb] = wload(wr, voff, 4 * (base + L::WC + (2 * rp + rb) * (NSC > 0 ? NSC : 4) * 64));
          bc[rb] = wload(wr, qoff, 4 * (base + L::BC + 16 * (2 * rp + rb)));
        }
      }
      __syncthreads();
      VCNF_HIDDEN_COMPUTE(t)                                                            // :48
      __builtin_amdgcn_sched_barrier(0);
      if (blk + 1 < NBLK) {
        VCNF_LOAD_HIDDEN(base + L::BLK + L::WA, base + L::BLK + L::BA)
      }
      if (C > 0 && VCNF_ABL != 6) {                                                     // :49-56 GLU gate, fp32
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float* cr = ct + ((4 * ch + j) * 16 + m16) * CS;
            floatx4 gate = bc[rb];
#pragma unroll
            for (int s = 0; s < NSC; ++s) gate = mfma4(wc[rb][s], cr[4 * s + q], gate);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float sg = div_nr(1.f, 1.f + hw_exp2(-gate[r] * kLog2e));
              h[rb][j][r] = fmaf(t[rb][j][r], sg, h[rb][j][r]);                         // :57
            }
          }
        }
      } else {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int j = 0; j < 4; ++j) h[rb][j] += t[rb][j];
      }
      __syncthreads();                       // every wave is done reading relu(t)
      if (blk + 1 < NBLK) {
        VCNF_PUBLISH(h, true)
      } else {
        VCNF_PUBLISH(h, false)               // the last layer takes h itself (resnet.py:105)
      }
      __syncthreads();
    }
#undef VCNF_HIDDEN_COMPUTE
#undef VCNF_READ_B
#undef VCNF_LOAD_HIDDEN
#undef VCNF_PUBLISH

    // ---- last layer + splines: wave owns column blocks 2 rp, 2 rp + 1 for groups of parity ch
    half8 fhi[2][NS32], flo[2][NS32];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int s = 0; s < NS32; ++s) {
        fhi[j][s] = __builtin_bit_cast(half8, act_hi[(s * kCB + 2 * rp + j) * 64 + lane]);
        flo[j][s] = __builtin_bit_cast(half8, act_lo[(s * kCB + 2 * rp + j) * 64 + lane]);
      }
    }
    float ld_acc[2] = {0.f, 0.f};
    const uint4* win = act + ch * GFRAG;
    // staging of two groups (96 KB): thread i moves fragments i, i + 512, ... (12 per thread)
    constexpr int NSTG = 2 * GFRAG / kBlock;
    static_assert(2 * GFRAG % kBlock == 0 && NSTG % 2 == 0, "staging split in two halves");
#define VCNF_STAGE_LOAD(DST, K0, RND)                                                     \
  _Pragma("unroll") for (int k = 0; k < NSTG / 2; ++k) {                                  \
    const int i = tid + ((K0) + k) * kBlock;                                              \
    const int par = i >= GFRAG ? 1 : 0;                                                   \
    const int rem = i - par * GFRAG;                                                      \
    const int ln = rem & 63, part = (rem >> 6) & 1, bs = rem >> 7;                        \
    DST[k] = wload(wr, 4 * (L::WF + part * HALF_F + ((2 * (RND) + par) * P4 * NS32 + bs) * 256) + ln * 16, 0); \
  }
#define VCNF_STAGE_STORE(SRC, K0)                                                         \
  _Pragma("unroll") for (int k = 0; k < NSTG / 2; ++k) act[tid + ((K0) + k) * kBlock] = __builtin_bit_cast(uint4, SRC[k]);
    __syncthreads();                         // every wave has its operand fragments
    {
      floatx4 stg[NSTG / 2];
      VCNF_STAGE_LOAD(stg, 0, 0)
      VCNF_STAGE_STORE(stg, 0)
      VCNF_STAGE_LOAD(stg, NSTG / 2, 0)
      VCNF_STAGE_STORE(stg, NSTG / 2)
    }
    for (int rnd = 0; rnd < (VCNF_ABL == 3 ? 0 : NG / 2); ++rnd) {
      const int g = 2 * rnd + ch;
      floatx4 pa[2][P4];
#pragma unroll
      for (int b = 0; b < P4; ++b)           // bias: lands in the accumulator registers, requested before the barrier
        pa[0][b] = wload(wr, q * (16 * P4), 4 * (L::BF + g * 4 * (4 * P4) + 4 * b));   // bf[g][q][4b..]
      __syncthreads();                       // window staged
      {
        half8 wh[3], wl[3];                  // ring: step t = 4 b + s uses slot t % 3, read two steps ahead
        floatx4 mainv[2], corr[2];
#define VCNF_READ_W(T)                                                                    \
  {                                                                                       \
    wh[(T) % 3] = __builtin_bit_cast(half8, win[((T) * 2 + 0) * 64 + lane]);              \
    wl[(T) % 3] = __builtin_bit_cast(half8, win[((T) * 2 + 1) * 64 + lane]);              \
  }
        VCNF_READ_W(0)
        VCNF_READ_W(1)
#pragma unroll
        for (int t = 0; t < P4 * NS32; ++t) {
          const int b = t >> 2, s = t & 3;
          if (t + 2 < P4 * NS32) {
            VCNF_READ_W(t + 2)
          }
          if (s == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              mainv[j] = pa[0][b];
              corr[j] = floatx4{0.f, 0.f, 0.f, 0.f};
            }
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) mainv[j] = mfma16h(wh[t % 3], fhi[j][s], mainv[j]);
#pragma unroll
          for (int j = 0; j < 2; ++j) corr[j] = mfma16h(wh[t % 3], flo[j][s], corr[j]);
#pragma unroll
          for (int j = 0; j < 2; ++j) corr[j] = mfma16h(wl[t % 3], fhi[j][s], corr[j]);
          if (s == NS32 - 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) pa[j][b][r] = fmaf(corr[j][r], kLoUnscale, mainv[j][r]);
          }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int t = 0; t + 2 < P4 * NS32; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
#undef VCNF_READ_W
      }
      __syncthreads();                       // every wave is done with the window
      // the next round's weights travel while the splines are evaluated (VALU only)
      const bool more = rnd + 1 < NG / 2;
      floatx4 stg[NSTG / 2];
      if (more && VCNF_ABL != 2) {
        VCNF_STAGE_LOAD(stg, 0, rnd + 1)
      }
      const int col = tfi[4 * g + q];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float* px = xt + ((2 * rp + j) * 16 + m16) * XS + col;
        const float xv = *px;
        RegLogits<K, P4> p{pa[j], c.wh_scale, c.edge_logit};
        float yv, lad;
#if VCNF_ABL == 1
        yv = xv; lad = 0.f;
        _Pragma("unroll") for (int b = 0; b < P4; ++b) { yv += pa[j][b][0] + pa[j][b][1]; lad += pa[j][b][2] + pa[j][b][3]; }
#else
        if (c.tails == 1 && !((xv >= c.lo_x) && (xv <= c.hi_x))) {
          yv = xv;
          lad = 0.f;
        } else {
          RqsBin sel;
          rqs_select<K, INV>(xv, p, c, c.wh_scale * kLog2e, sel);
          rqs_bin_eval<INV>(xv, sel, yv, lad, bad);
        }
#endif
        *px = yv;
        ld_acc[j] += lad;
        if (more && VCNF_ABL != 2) {
          if (j == 0) {
            VCNF_STAGE_STORE(stg, 0)
            VCNF_STAGE_LOAD(stg, NSTG / 2, rnd + 1)
          } else {
            VCNF_STAGE_STORE(stg, NSTG / 2)
          }
        }
      }
    }
#undef VCNF_STAGE_LOAD
#undef VCNF_STAGE_STORE

    // ---- per-sample log|det|: this wave covered one group parity of its samples; the partner
    // wave (other parity) adds its share through LDS (ldt already holds the identity half)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      ld_acc[j] += __shfl_xor(ld_acc[j], 16, 64);
      ld_acc[j] += __shfl_xor(ld_acc[j], 32, 64);
    }
    if (ch == 1 && q == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) ldt[(2 * rp + j) * 16 + m16] += ld_acc[j];
    }
    __syncthreads();
    if (ch == 0 && q == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int mrow = (2 * rp + j) * 16 + m16;
        if (mrow < rows) {
          const float o = a.ld_sign * (ld_acc[j] + ldt[mrow]);
          a.logdet[b0 + mrow] = a.ld_mode ? a.logdet[b0 + mrow] + o : o;
        }
      }
    }
    {
      constexpr int D4 = D / 4;
      float4* dst = reinterpret_cast<float4*>(a.y) + b0 * D4;
      for (int i = tid; i < rows * D4; i += kBlock) {
        const int r = i / D4, o = i - r * D4;
        dst[i] = *reinterpret_cast<const float4*>(xt + r * XS + 4 * o);
      }
    }
  }
  if (INV && a.bad && bad) atomicAdd(a.bad, 1);
}

template <int DI, int DT, int C, int H, int NBLK, int K>
static int launch_v3(const FusedArgs& a, int inverse, hipStream_t st) {
  constexpr int D = DI + DT;
  constexpr int TILE = 128;
  constexpr size_t WIN = (size_t)2 * ((3 * K + 2) / 4) * (H / 32) * 2 * 64 * 16;   // two feature groups
  const size_t lds = ((size_t)TILE * (D + 4) + (size_t)TILE * ((C > 0 ? C : 4) + 4) +
                      ((DI * 3 * (K + 1) + 3) & ~3) + TILE + D + 4) * 4 + WIN + 64;
  static bool attr_set[2] = {false, false};
  if (!attr_set[inverse ? 1 : 0]) {
    hipError_t e;
    if (inverse)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_rqs_layer_v3_kernel<DI, DT, C, H, NBLK, K, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_rqs_layer_v3_kernel<DI, DT, C, H, NBLK, K, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return VCNF_ERR_LAUNCH;
    attr_set[inverse ? 1 : 0] = true;
  }
  const long long ntiles = (a.B + TILE - 1) / TILE;
  dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256));
  if (inverse)
    hipLaunchKernelGGL((fused_rqs_layer_v3_kernel<DI, DT, C, H, NBLK, K, true>), grid, dim3(512), lds, st, a);
  else
    hipLaunchKernelGGL((fused_rqs_layer_v3_kernel<DI, DT, C, H, NBLK, K, false>), grid, dim3(512), lds, st, a);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

int launch_fused_v3_c16(const FusedArgs& a, int inverse, hipStream_t st) {
  return launch_v3<32, 32, 16, 128, 2, 8>(a, inverse, st);
}

int launch_fused_v3_c0(const FusedArgs& a, int inverse, hipStream_t st) {
  return launch_v3<32, 32, 0, 128, 2, 8>(a, inverse, st);
}

}  // namespace vcnf
