// Fused RQS coupling layer, fp16 split-half matrix path, fourth structure: the two wave groups
// of a workgroup run the SAME straight-line step sequence one step apart ("ping-pong").
//
// Same contract, packed-weight layout, arithmetic and work split as fused_layer_v3.hip.  PMC
// counters on v3 (profiles/r01_fused_v3_pmc.txt): matrix pipe busy 32 %, VALU 26 %, LDS 18 % of
// the cycles, waves waiting 47 %: every step of v3 is bracketed by workgroup barriers, so all
// eight waves do matrix work together and then vector work together, and neither pipe ever
// covers for the other.  Here:
//   * group A = waves 0-3, group B = waves 4-7 (wave w and w+4 share a SIMD);
//   * the tile's work is a sequence of steps alternating M (matrix: first layer, hidden layer,
//     last-layer block) and V (vector: gate, hi/lo split + publish, identity-half splines,
//     transformed-half splines, staging, next weights requested), one workgroup barrier after
//     each step;
//   * group B executes one extra barrier before the sequence and group A one after it.  A
//     hardware barrier only counts arrivals, so from then on A is in step k while B is in step
//     k-1: on every SIMD one wave is in an M step and the other in a V step;
//   * that is legal because within a sequence the groups touch disjoint LDS: in the trunk a
//     group owns one half of the tile's samples end to end (its layers read only what its own
//     waves published), in the last layer a group owns one parity of the feature groups and
//     its own 48 KB half of the weight window.  The sequences are re-aligned (the extra
//     barriers) where the groups exchange data: before the last layer reads all activations
//     and before the outputs are written.
// (A first attempt drove the steps from a loop with a switch on (slot - group): the register
// allocator then spills whole accumulator arrays around every branch, 2.3x slower; kept as
// profiles/tools/fused_layer_v4_loop_switch.hip.txt.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"
#include "rqs_math.hpp"
#include "fused_common.hpp"

#ifndef VCNF_ABL
#define VCNF_ABL 0
#endif
// -DVCNF_TIME=1: s_memtime stamps at every barrier; wave 0 of workgroup 0 leaves the per-phase sums in
// the first output row (timing builds only, read by profiles/tools/v4_phase_timing.py)
#ifndef VCNF_TIME
#define VCNF_TIME 0
#endif
#if VCNF_TIME
#define VCNF_T(I) { const long long t_ = clock64(); tacc[I] += t_ - tlast; tlast = t_; }
#else
#define VCNF_T(I)
#endif

namespace vcnf {

template <int DI, int DT, int C, int H, int NBLK, int K, bool INV>
__global__ __launch_bounds__(512, 2) void fused_rqs_layer_v4_kernel(const FusedArgs a) {
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
  // The fragment region comes first: at LDS offset 0 every fragment address is one per-lane
  // base register plus a 16-bit immediate.
  // activation fragments [s][cb][lane][8 halves]: hi then lo (32 KB each); the same region
  // (96 KB) is the last layer's weight window [group parity][b][s][hi|lo][lane][8 halves]
  uint4* act = reinterpret_cast<uint4*>(smem);
  uint4* act_hi = act;
  uint4* act_lo = act + NS32 * kCB * 64;
  float* xt = smem + 2 * GFRAG * 4;                        // [128][XS]  x in, y out (in place)
  float* ct = xt + kTile * XS;                             // [128][CS]
  float* tab = ct + kTile * CS;                            // [DI][TABW]
  float* ldt = tab + ((DI * TABW + 3) & ~3);               // [128] identity-half log|det|
  int* tfi = reinterpret_cast<int*>(ldt + kTile);
  int* idi = tfi + DT;

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
#if VCNF_TIME
  long long tacc[16], tlast = clock64();
  for (int i = 0; i < 16; ++i) tacc[i] = 0;
#endif
  // rows of the next tile travel in registers: bounds-checked buffer loads (rows past the batch read 0)
  float4 xpre[kTile * (D / 4) / kBlock], cpre[C > 0 ? (kTile * (C / 4) + kBlock - 1) / kBlock : 1];
  static_assert(kTile * (D / 4) % kBlock == 0 && (C == 0 || kTile * (C / 4) == kBlock), "rows per thread");
#define VCNF_PREFETCH_ROWS(TILE)                                                          \
  {                                                                                       \
    const long long pb0 = min((TILE) * kTile, a.B);                                       \
    const long long left = (a.B - pb0) * (D * 4);                                         \
    const __amdgpu_buffer_rsrc_t xr_ = __builtin_amdgcn_make_buffer_rsrc(                 \
        const_cast<float*>(a.x) + pb0 * D, 0, (int)min(left, (long long)(kTile * D * 4)), 0x00020000); \
    _Pragma("unroll") for (int k = 0; k < kTile * (D / 4) / kBlock; ++k)                  \
      xpre[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr_, (tid + kBlock * k) * 16, 0, 0)); \
    if (C > 0) {                                                                          \
      const long long leftc = (a.B - pb0) * (C * 4);                                      \
      const __amdgpu_buffer_rsrc_t cr_ = __builtin_amdgcn_make_buffer_rsrc(               \
          const_cast<float*>(a.ctx) + pb0 * C, 0, (int)min(leftc, (long long)(kTile * C * 4)), 0x00020000); \
      cpre[0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(cr_, tid * 16, 0, 0)); \
    }                                                                                     \
  }
  VCNF_PREFETCH_ROWS((long long)blockIdx.x)
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long b0 = tile * kTile;
    const int rows = (int)min((long long)kTile, a.B - b0);
    { VCNF_T(0) __syncthreads(); VCNF_T(15) }
    {   // ---- x and context rows: requested during the previous tile's last vector step (or before the loop)
      constexpr int D4 = D / 4;
#pragma unroll
      for (int k = 0; k < kTile * D4 / kBlock; ++k) {
        const int i = tid + kBlock * k;
        const int r = i / D4, o = i - r * D4;
        *reinterpret_cast<float4*>(xt + r * XS + 4 * o) = xpre[k];
      }
      if (C > 0) {
        constexpr int C4 = (C > 0 ? C : 4) / 4;
#pragma unroll
        for (int k = 0; k < kTile * C4 / kBlock; ++k) {
          const int i = tid + kBlock * k;
          const int r = i / C4, o = i - r * C4;
          *reinterpret_cast<float4*>(ct + r * CS + 4 * o) = cpre[k];
        }
      }
    }
    { VCNF_T(1) __syncthreads(); VCNF_T(15) }

    // ---- identity half through the unconditional spline: 4 lanes per sample, each lane a run
    // of DI/4 features; per-sample log|det| of this half parked in LDS.
    // (branch-free per feature: points outside the interval are evaluated at the left end and selected
    // to the identity afterwards, so the table reads of UNR features are in flight together)
#define VCNF_IDENTITY_ROW(ROW, UNR)                                                       \
  {                                                                                       \
    const int mi = (ROW);                                                                 \
    float lsum = 0.f;                                                                     \
    if (shared) {                                                                         \
      _Pragma("unroll") for (int k0 = 0; k0 < DI / 4; k0 += (UNR)) {                      \
        float yv[UNR], lad[UNR], xv[UNR];                                                 \
        bool in_[UNR];                                                                    \
        float* px[UNR];                                                                   \
        _Pragma("unroll") for (int k = 0; k < (UNR); ++k) {                               \
          const int f = (tid & 3) * (DI / 4) + k0 + k;                                    \
          px[k] = xt + mi * XS + idi[f];                                                  \
          xv[k] = *px[k];                                                                 \
          in_[k] = (xv[k] >= c.lo_x) && (xv[k] <= c.hi_x);                                \
        }                                                                                 \
        _Pragma("unroll") for (int k = 0; k < (UNR); ++k) {                               \
          const int f = (tid & 3) * (DI / 4) + k0 + k;                                    \
          bool bad1 = false;                                                              \
          rqs_point_table_inside<INV, K>(in_[k] ? xv[k] : c.lo_x, tab + f * TABW, yv[k], lad[k], bad1); \
          bad = bad || bad1;                                                              \
        }                                                                                 \
        _Pragma("unroll") for (int k = 0; k < (UNR); ++k) {                               \
          *px[k] = in_[k] ? yv[k] : xv[k];                                                \
          lsum += in_[k] ? lad[k] : 0.f;                                                  \
        }                                                                                 \
      }                                                                                   \
    }                                                                                     \
    lsum += __shfl_xor(lsum, 1, 64);                                                      \
    lsum += __shfl_xor(lsum, 2, 64);                                                      \
    if ((tid & 3) == 0) ldt[mi] = lsum;                                                   \
  }
    // first-layer operands of this wave's four column blocks.  Direction whose conditioner sees the
    // transformed identity half: read after that half went through its splines; other direction: read
    // first (raw values), then the identity half is transformed in place - with all eight waves at once
    // in both directions (inside the ping-pong sequence each group's share was on the critical path).
    float bv[4][NS0];
#define VCNF_READ_FIRST_OPERANDS()                                                        \
  {                                                                                       \
    int xcol[DI / 4];                                                                     \
    _Pragma("unroll") for (int s = 0; s < DI / 4; ++s) xcol[s] = idi[4 * s + q];          \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                       \
      const float* xr = xt + ((4 * ch + j) * 16 + m16) * XS;                              \
      const float* cr = ct + ((4 * ch + j) * 16 + m16) * CS;                              \
      _Pragma("unroll") for (int s = 0; s < NS0; ++s)                                     \
        bv[j][s] = s < DI / 4 ? xr[xcol[s < DI / 4 ? s : 0]] : cr[4 * (s - DI / 4) + q];  \
    }                                                                                     \
  }
    if (!INV) {
      VCNF_READ_FIRST_OPERANDS()
      { VCNF_T(2) __syncthreads(); VCNF_T(15) }            // every wave has its raw operands
    }
    VCNF_IDENTITY_ROW(tid >> 2, DI / 4)
    { VCNF_T(2) __syncthreads(); VCNF_T(15) }
    if (INV) {
      VCNF_READ_FIRST_OPERANDS()
    }
#undef VCNF_READ_FIRST_OPERANDS
    if (ch == 1) { VCNF_T(14) __syncthreads(); VCNF_T(15) }            // ---- group B now runs one step behind group A

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
    { VCNF_T(3) __syncthreads(); VCNF_T(15) }                         // ---- end of step M0
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
    // ---- step V1: first hidden layer's weights requested, relu(h) published
    VCNF_LOAD_HIDDEN(L::BLK0 + L::WA, L::BLK0 + L::BA)
    VCNF_PUBLISH(h, true)
    { VCNF_T(4) __syncthreads(); VCNF_T(15) }
#undef VCNF_IDENTITY_ROW

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
      // ---- step M: first layer of the block                                          resnet.py:42-43
      VCNF_HIDDEN_COMPUTE(t)
      { VCNF_T(5) __syncthreads(); VCNF_T(15) }
      // ---- step V: publish relu(t) (:46); second layer's and gate weights requested
      VCNF_LOAD_HIDDEN(base + L::WB, base + L::BB)
      floatx4 wc[2], bc[2];
      if (C > 0) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          wc[rb] = wload(wr, voff, 4 * (base + L::WC + (2 * rp + rb) * (NSC > 0 ? NSC : 4) * 64));
          bc[rb] = wload(wr, qoff, 4 * (base + L::BC + 16 * (2 * rp + rb)));
        }
      }
      VCNF_PUBLISH(t, true)
      { VCNF_T(6) __syncthreads(); VCNF_T(15) }
      // ---- step M: second layer of the block                                         :48
      // and the gate pre-activations (fp32 instruction): matrix work belongs in the matrix step, the
      // vector step that follows is the longer one of its pair (profiles/r01_fused_v4_phase_cycles.md)
      VCNF_HIDDEN_COMPUTE(t)
      floatx4 gate[2][4];
      if (C > 0 && VCNF_ABL != 6) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float* cr = ct + ((4 * ch + j) * 16 + m16) * CS;
          float cv[NSC > 0 ? NSC : 1];
#pragma unroll
          for (int s = 0; s < NSC; ++s) cv[s] = cr[4 * s + q];
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) {
            gate[rb][j] = bc[rb];
#pragma unroll
            for (int s = 0; s < NSC; ++s) gate[rb][j] = mfma4(wc[rb][s], cv[s], gate[rb][j]);
          }
        }
      }
      { VCNF_T(5) __syncthreads(); VCNF_T(15) }
      // ---- step V: GLU gate (sigmoid of the pre-activations), residual update, publish :49-57
      if (blk + 1 < NBLK) {
        VCNF_LOAD_HIDDEN(base + L::BLK + L::WA, base + L::BLK + L::BA)
      }
      if (C > 0 && VCNF_ABL != 6) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float sg = div_nr(1.f, 1.f + hw_exp2(-gate[rb][j][r] * kLog2e));
              h[rb][j][r] = fmaf(t[rb][j][r], sg, h[rb][j][r]);
            }
          }
        }
      } else {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int j = 0; j < 4; ++j) h[rb][j] += t[rb][j];
      }
      if (blk + 1 < NBLK) {
        VCNF_PUBLISH(h, true)
      } else {
        VCNF_PUBLISH(h, false)               // the last layer takes h itself (resnet.py:105)
      }
      { VCNF_T(7) __syncthreads(); VCNF_T(15) }
    }
    if (ch == 0) { VCNF_T(14) __syncthreads(); VCNF_T(15) }            // ---- groups re-aligned: all activations are published
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
    uint4* win = act + ch * GFRAG;           // this group's half of the window: one feature group, 48 KB
    const int gtid = tid & 255;              // thread index inside the group
    constexpr int NSTG = GFRAG / 256;        // 16-byte fragments a thread moves per staged group (12)
    static_assert(GFRAG % 256 == 0 && NSTG % 2 == 0, "staging split in two halves");
    constexpr int NR = NG / 2;               // rounds = feature groups per wave group
    // fragment i of feature group G: i = (b * NS32 + s) * 128 + part * 64 + lane; thread gtid moves
    // i = gtid + 256 k, i.e. (b * NS32 + s) = (gtid >> 7) + 2 k with part and lane fixed: one
    // per-thread byte offset + a wave-uniform offset per k
    const int stg_voff = 4 * (((gtid >> 6) & 1) * HALF_F + (gtid >> 7) * 256) + (gtid & 63) * 16;
#define VCNF_STAGE_LOAD(DST, K0, G)                                                       \
  _Pragma("unroll") for (int k = 0; k < NSTG / 2; ++k)                                    \
    DST[k] = wload(wr, stg_voff, 4 * (L::WF + ((G) * P4 * NS32 + 2 * ((K0) + k)) * 256));
#define VCNF_STAGE_STORE(SRC, K0)                                                         \
  _Pragma("unroll") for (int k = 0; k < NSTG / 2; ++k) win[gtid + ((K0) + k) * 256] = __builtin_bit_cast(uint4, SRC[k]);
    // the same fragments straight from global memory into the window (buffer_load ... lds: no registers,
    // no LDS write instructions; the wave's 64 lanes land at consecutive 16-byte slots of a wave-uniform base)
#define VCNF_STAGE_DMA(G)                                                                 \
  _Pragma("unroll") for (int k = 0; k < NSTG; ++k)                                        \
    dma16_to_lds(wr, win + (gtid & ~63) + k * 256, stg_voff, 4 * (L::WF + ((G) * P4 * NS32 + 2 * k) * 256));
    floatx4 pa[2][P4];
#define VCNF_LOAD_BIAS(G)                                                                 \
  _Pragma("unroll") for (int b = 0; b < P4; ++b)                                          \
    pa[0][b] = wload(wr, q * (16 * P4), 4 * (L::BF + (G) * 4 * (4 * P4) + 4 * b));      /* bf[g][q][4b..] */
    VCNF_LOAD_BIAS(ch)
    { VCNF_T(8) __syncthreads(); VCNF_T(15) }                         // every wave has its operand fragments: the window may be written
    {   // first feature group of each wave group
      floatx4 stg[NSTG / 2];
      VCNF_STAGE_LOAD(stg, 0, ch)
      VCNF_STAGE_STORE(stg, 0)
      VCNF_STAGE_LOAD(stg, NSTG / 2, ch)
      VCNF_STAGE_STORE(stg, NSTG / 2)
    }
    { VCNF_T(9) __syncthreads(); VCNF_T(15) }
    if (ch == 1) { VCNF_T(14) __syncthreads(); VCNF_T(15) }            // ---- group B one step behind again
    for (int rnd = 0; rnd < (VCNF_ABL == 3 ? 0 : NR); ++rnd) {
      const int g = 2 * rnd + ch;
      // the two elements this lane transforms in the vector step: read before the matrix step, so the
      // vector step does not start with two dependent LDS round trips
      const int col = tfi[4 * g + q];
      float xin[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) xin[j] = xt[((2 * rp + j) * 16 + m16) * XS + col];
      {
        // ---- step M: 144 matrix instructions on the group's window, fragments read two steps ahead
        half8 wh[3], wl[3];
        floatx4 mainv[2], corr[2];
#define VCNF_READ_W(T)                                                                    \
  {                                                                                       \
    wh[(T) % 3] = __builtin_bit_cast(half8, win[((T) * 2 + 0) * 64 + lane]);              \
    wl[(T) % 3] = __builtin_bit_cast(half8, win[((T) * 2 + 1) * 64 + lane]);              \
  }
        VCNF_READ_W(0)
        VCNF_READ_W(1)
#pragma unroll
        for (int u = 0; u < P4 * NS32; ++u) {
          const int b = u >> 2, s = u & 3;
          if (u + 2 < P4 * NS32) {
            VCNF_READ_W(u + 2)
          }
          if (s == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              mainv[j] = pa[0][b];
              corr[j] = floatx4{0.f, 0.f, 0.f, 0.f};
            }
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) mainv[j] = mfma16h(wh[u % 3], fhi[j][s], mainv[j]);
#pragma unroll
          for (int j = 0; j < 2; ++j) corr[j] = mfma16h(wh[u % 3], flo[j][s], corr[j]);
#pragma unroll
          for (int j = 0; j < 2; ++j) corr[j] = mfma16h(wl[u % 3], fhi[j][s], corr[j]);
          if (s == NS32 - 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) pa[j][b][r] = fmaf(corr[j][r], kLoUnscale, mainv[j][r]);
          }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int u = 0; u + 2 < P4 * NS32; ++u) {
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
#undef VCNF_READ_W
      }
      { VCNF_T(10) __syncthreads(); VCNF_T(15) }
      {
        // ---- step V: two spline evaluations per lane; the group's next window and bias travel meanwhile
        const bool more = rnd + 1 < NR;
        if (!more) {                           // last vector step of the tile: the next tile's rows are requested
          VCNF_PREFETCH_ROWS(tile + gridDim.x)
        }
        if (more && VCNF_ABL != 2) {
          VCNF_STAGE_DMA(g + 2)
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float* px = xt + ((2 * rp + j) * 16 + m16) * XS + col;
          const float xv = xin[j];
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
          if (more && j == 0) {
            VCNF_LOAD_BIAS(g + 2)
          }
        }
      }
      // this wave's part of the window must have landed before the other waves of the group read it
      // (vmcnt(0); the loads were requested a whole vector step ago)
      if (rnd + 1 < NR && VCNF_ABL != 2) wait_vector_memory();
      { VCNF_T(11) __syncthreads(); VCNF_T(15) }
    }
    if (ch == 0) { VCNF_T(14) __syncthreads(); VCNF_T(15) }            // ---- groups re-aligned: every spline of the tile is done
#undef VCNF_PREFETCH_ROWS
#undef VCNF_STAGE_LOAD
#undef VCNF_STAGE_STORE
#undef VCNF_STAGE_DMA
#undef VCNF_LOAD_BIAS

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
    { VCNF_T(12) __syncthreads(); VCNF_T(15) }
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
#if VCNF_TIME
  VCNF_T(13)
  if (blockIdx.x == 0 && tid == 0) {
    for (int i = 0; i < 16; ++i) a.y[i] = (float)tacc[i];
  }
#endif
  if (INV && a.bad && bad) atomicAdd(a.bad, 1);
}

template <int DI, int DT, int C, int H, int NBLK, int K>
static int launch_v4(const FusedArgs& a, int inverse, hipStream_t st) {
  constexpr int D = DI + DT;
  constexpr int TILE = 128;
  constexpr size_t WIN = (size_t)2 * ((3 * K + 2) / 4) * (H / 32) * 2 * 64 * 16;   // two feature groups
  const size_t lds = ((size_t)TILE * (D + 4) + (size_t)TILE * ((C > 0 ? C : 4) + 4) +
                      ((DI * 3 * (K + 1) + 3) & ~3) + TILE + D + 4) * 4 + WIN + 64;
  static bool attr_set[2] = {false, false};
  if (!attr_set[inverse ? 1 : 0]) {
    hipError_t e;
    if (inverse)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_rqs_layer_v4_kernel<DI, DT, C, H, NBLK, K, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_rqs_layer_v4_kernel<DI, DT, C, H, NBLK, K, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return VCNF_ERR_LAUNCH;
    attr_set[inverse ? 1 : 0] = true;
  }
  const long long ntiles = (a.B + TILE - 1) / TILE;
  dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256));
  if (inverse)
    hipLaunchKernelGGL((fused_rqs_layer_v4_kernel<DI, DT, C, H, NBLK, K, true>), grid, dim3(512), lds, st, a);
  else
    hipLaunchKernelGGL((fused_rqs_layer_v4_kernel<DI, DT, C, H, NBLK, K, false>), grid, dim3(512), lds, st, a);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

// Shape family of the fused fp16 split-half kernel: (d_id = d_t, ctx, residual blocks) with H = 128, 8 bins.
template <int NBLK>
static int launch_v4_family(const FusedArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
  if (d_id == 32) {
    return ctx_dim == 16 ? launch_v4<32, 32, 16, 128, NBLK, 8>(a, inverse, st)
                         : launch_v4<32, 32, 0, 128, NBLK, 8>(a, inverse, st);
  }
  return ctx_dim == 16 ? launch_v4<16, 16, 16, 128, NBLK, 8>(a, inverse, st)
                       : launch_v4<16, 16, 0, 128, NBLK, 8>(a, inverse, st);
}

// One translation unit per number of residual blocks (-DVCNF_V4_NBLK=1|2|3; each takes ~45 s to
// compile, build.py runs them in parallel); fused_layer.hip dispatches to launch_fused_v4_b<N>.
#ifndef VCNF_V4_NBLK
#define VCNF_V4_NBLK 2
#endif
#if VCNF_V4_NBLK == 1
int launch_fused_v4_b1(const FusedArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
  return launch_v4_family<1>(a, d_id, ctx_dim, inverse, st);
}
#elif VCNF_V4_NBLK == 2
int launch_fused_v4_b2(const FusedArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
  return launch_v4_family<2>(a, d_id, ctx_dim, inverse, st);
}
#else
int launch_fused_v4_b3(const FusedArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
  return launch_v4_family<3>(a, d_id, ctx_dim, inverse, st);
}
#endif

}  // namespace vcnf
