// Fused RQS coupling layer, fp16 split-half matrix path, fifth structure: activation-stationary waves.
//
// Same contract, packed-weight layout and arithmetic as fused_layer_v4.hip.  What v4's measurements
// said: its 128-sample tile takes ~110 k cycles against 41 k of matrix work and
// 43 k of vector work; the rest is synchronisation - every layer's activations are published to LDS
// for the other waves, 25 workgroup barriers per tile, and the two pipes only overlap where two waves
// of different groups happen to hold different kinds of step.  Here:
//   * a wave owns 32 samples (two 16-sample column blocks) from the first conditioner layer to the
//     spline outputs.  The result registers of one layer become the operand registers of the next
//     (rows 4q..4q+3 of two row blocks are the 8 k-values lane group q supplies to a
//     v_mfma_f32_16x16x32_f16; the host packs the weights in that k order - the same order v4's
//     publish step produces), so activations never touch LDS and no wave waits for another wave's data;
//   * 4 waves per workgroup, one workgroup per CU: one wave per SIMD with the whole 512-register
//     budget (residual stream 64 + operand halves 128 + accumulators, no spills).  Overlap of the
//     matrix and vector pipes comes from inside the wave: the vector work of a finished row-block
//     pair / feature group (split into halves, GLU gate, spline evaluation) is independent of the
//     matrix instructions of the next one and is scheduled between them;
//   * LDS holds weights only: a ring of two 48 KB chunks (half a hidden layer, or one feature group
//     of the last layer), filled by all 256 threads one chunk ahead; one workgroup barrier per chunk
//     (17 per 128 samples) and wave-private strips for the coalesced x -> y row traffic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"
#include "rqs_math.hpp"
#include "fused_common.hpp"

#ifndef VCNF_ABL
#define VCNF_ABL 0
#endif
#ifndef VCNF_TIME
#define VCNF_TIME 0
#endif

namespace vcnf {

template <int DI, int DT, int C, int H, int NBLK, int K, bool INV>
__global__ __launch_bounds__(256) void fused_rqs_layer_v5_kernel(const FusedArgs a) {
  static_assert(H == 128, "8 hidden row blocks");
  static_assert(C == 0 || C == 16, "gate weights: one fragment per row block");
  constexpr int kBlock = 256;
  constexpr int kTile = 128;
  constexpr int kWS = 32;                   // samples per wave
  constexpr int D = DI + DT;
  constexpr int D4 = D / 4;
  constexpr int XS = D + 4;
  constexpr int CC = C > 0 ? C : 4;
  constexpr int C4 = CC / 4;
  constexpr int CS = CC + 4;
  constexpr int NS0 = (DI + C) / 4;         // fp32 k-steps of the first layer
  constexpr int NS0_4 = NS0 / 4;
  constexpr int NSC = C / 4;
  constexpr int NS32 = H / 32;              // fp16 k-steps of a hidden->* layer (4)
  constexpr int NB = H / 16;
  constexpr int P = 3 * K - 1;
  constexpr int P4 = (P + 3) / 4;           // 6
  constexpr int NG = DT / 4;
  static_assert(NG % 2 == 0 && NG >= 2, "feature groups are processed two per loop iteration");
  static_assert(NS0 % 4 == 0 && NB * NS0_4 <= 24 && P4 * NS32 == 24, "chunk shapes");
  constexpr int TABW = 3 * (K + 1);
  using L = PackLayout<DI, DT, C, H, NBLK, K>;
  constexpr int HALF_W = NB * (H / 4) * 64 / 2;           // floats of the hi half of a hidden layer
  constexpr int HALF_F = NG * P4 * (H / 4) * 64 / 2;
  constexpr int RING = 48 * 64;                           // 16-byte fragments per chunk buffer

  extern __shared__ __align__(16) float smem[];
  uint4* ring = reinterpret_cast<uint4*>(smem);           // [2][48][64]
  float* xs = smem + 2 * RING * 4;                        // [4 waves][32][XS]
  float* cs = xs + 4 * kWS * XS;                          // [4 waves][32][CS]
  float* tab = cs + 4 * kWS * CS;                         // [DI][TABW]
  int* tfi = reinterpret_cast<int*>(tab + ((DI * TABW + 3) & ~3));
  int* idi = tfi + DT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15;
  const int q = lane >> 4;
  const RqsConst& c = a.c;
  const bool shared = a.sh_w != nullptr;
  float* xw = xs + wave * kWS * XS;
  float* cw = cs + wave * kWS * CS;

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
  const int toff = tid * 16;                // staging: thread t moves fragments (t >> 6) + 4 k

  // staging registers of the chunk in flight: up to 12 fragments per thread
  floatx4 st[12];
  // hidden half-layer chunk: 16 hi fragments (row block 4 HF + r, k-step s) then 16 lo fragments
#define V5_LOAD_HID(WOFF, HF)                                                             \
  _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                         \
    st[k] = wload(wr, toff, 4 * ((WOFF) + (HF) * 16 * 256 + k * 1024));                   \
    st[4 + k] = wload(wr, toff, 4 * ((WOFF) + HALF_W + (HF) * 16 * 256 + k * 1024));      \
  }
#define V5_STORE_HID()                                                                    \
  _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                         \
    nxt[(4 * k) * 64 + tid] = __builtin_bit_cast(uint4, st[k]);                           \
    nxt[(16 + 4 * k) * 64 + tid] = __builtin_bit_cast(uint4, st[4 + k]);                  \
  }
  // 24 + 24 fragments starting at float offset OFF (hi) and OFF + HALF_F (lo): one feature group of
  // the last layer, or (hi part only meaningful) the first layer's 8 NS0_4 fragments
#define V5_LOAD_WIDE(OFF)                                                                 \
  _Pragma("unroll") for (int k = 0; k < 6; ++k) {                                         \
    st[k] = wload(wr, toff, 4 * ((OFF) + k * 1024));                                      \
    st[6 + k] = wload(wr, toff, 4 * ((OFF) + HALF_F + k * 1024));                         \
  }
#define V5_STORE_WIDE()                                                                   \
  _Pragma("unroll") for (int k = 0; k < 6; ++k) {                                         \
    nxt[(4 * k) * 64 + tid] = __builtin_bit_cast(uint4, st[k]);                           \
    nxt[(24 + 4 * k) * 64 + tid] = __builtin_bit_cast(uint4, st[6 + k]);                  \
  }
#if VCNF_TIME
#define V5_T(I) { const long long t_ = clock64(); tacc[I] += t_ - tlast; tlast = t_; }
#else
#define V5_T(I)
#endif
#if VCNF_TIME == 2
#define V5_TU(I) { const long long t_ = clock64(); tu[I] += t_ - tlast2; tlast2 = t_; }
#define V5_TU0() tlast2 = clock64();
#else
#define V5_TU(I)
#define V5_TU0()
#endif
#define V5_FLIP()                                                                         \
  V5_T(V5_BUCKET)                                                                         \
  __syncthreads();                                                                        \
  V5_T(1)                                                                                 \
  buf ^= 1;                                                                               \
  cur = ring + buf * RING;                                                                \
  nxt = ring + (buf ^ 1) * RING;

  int buf = 0;
  const uint4* cur = ring;
  uint4* nxt = ring + RING;
  {   // the first layer's weights of the first tile
    nxt = ring;
    V5_LOAD_WIDE(L::W0)
    V5_STORE_WIDE()
    nxt = ring + RING;
  }
  __syncthreads();

  const long long ntiles = (a.B + kTile - 1) / kTile;
  bool bad = false;
#if VCNF_TIME
  long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#endif
#if VCNF_TIME == 2
  long long tu[26], tlast2 = 0;
  for (int i = 0; i < 26; ++i) tu[i] = 0;
#endif
#define V5_FENCE() __builtin_amdgcn_sched_barrier(0);
  // tile-invariant column indices of this lane
  int xcol[DI / 4];
#pragma unroll
  for (int s = 0; s < DI / 4; ++s) xcol[s] = idi[4 * s + q];

  // rows of x / context of the wave's next tile travel in registers while the current tile computes
  float4 xpre[kWS * D4 / 64], cpre[C > 0 ? kWS * C4 / 64 : 1];
#define V5_PREFETCH_ROWS(TILE)                                                            \
  {                                                                                       \
    const long long pb0 = min((TILE) * kTile + wave * kWS, a.B);                          \
    const long long left = (a.B - pb0) * (D * 4);         /* bytes of x from this wave's first row */ \
    const __amdgpu_buffer_rsrc_t xr_ = __builtin_amdgcn_make_buffer_rsrc(                 \
        const_cast<float*>(a.x) + pb0 * D, 0, (int)min(left, (long long)(kWS * D * 4)), 0x00020000); \
    _Pragma("unroll") for (int k = 0; k < kWS * D4 / 64; ++k)                             \
      xpre[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr_, (lane + 64 * k) * 16, 0, 0)); \
    if (C > 0) {                                                                          \
      const long long leftc = (a.B - pb0) * (CC * 4);                                     \
      const __amdgpu_buffer_rsrc_t cr_ = __builtin_amdgcn_make_buffer_rsrc(               \
          const_cast<float*>(a.ctx) + pb0 * CC, 0, (int)min(leftc, (long long)(kWS * CC * 4)), 0x00020000); \
      _Pragma("unroll") for (int k = 0; k < kWS * C4 / 64; ++k)                           \
        cpre[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(cr_, (lane + 64 * k) * 16, 0, 0)); \
    }                                                                                     \
  }
  V5_PREFETCH_ROWS((long long)blockIdx.x)

  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long b0 = tile * kTile + wave * kWS;        // first sample of this wave
    const int rows = (int)max(0LL, min((long long)kWS, a.B - b0));
    V5_T(7)
#define V5_BUCKET 2
    // ================= step 0: first layer (its chunk is in `cur`), next = first half of block 0's WA
    V5_LOAD_HID(L::BLK0 + L::WA, 0)
    {   // the wave's 32 rows of x and context into its strip
#pragma unroll
      for (int k = 0; k < kWS * D4 / 64; ++k) {
        const int i = lane + 64 * k;
        const int r = i / D4, o = i - r * D4;
        *reinterpret_cast<float4*>(xw + r * XS + 4 * o) = xpre[k];
      }
      if (C > 0) {
#pragma unroll
        for (int k = 0; k < kWS * C4 / 64; ++k) {
          const int i = lane + 64 * k;
          const int r = i / C4, o = i - r * C4;
          *reinterpret_cast<float4*>(cw + r * CS + 4 * o) = cpre[k];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    V5_T(8)
    // identity half through the unconditional spline: 4 lanes per sample, each a run of DI/4
    // features; the per-sample log|det| of this half is parked in the strip's padding column
#define V5_IDENTITY()                                                                     \
  if (shared) {                                                                           \
    float lsum[2] = {0.f, 0.f};                                                           \
    _Pragma("unroll") for (int k = 0; k < DI / 4; ++k) {                                  \
      const int f = (lane & 3) * (DI / 4) + k;                                            \
      const int col = idi[f];                                                             \
      _Pragma("unroll") for (int ps = 0; ps < 2; ++ps) {                                  \
        float* px = xw + (16 * ps + (lane >> 2)) * XS + col;                              \
        const float xv = *px;                                                             \
        const bool in_ = (xv >= c.lo_x) && (xv <= c.hi_x);                                \
        float yv, lad;                                                                    \
        bool bad1 = false;                                                                \
        rqs_point_table_inside<INV, K>(in_ ? xv : c.lo_x, tab + f * TABW, yv, lad, bad1); \
        *px = in_ ? yv : xv;                                                              \
        lsum[ps] += in_ ? lad : 0.f;                                                      \
        bad = bad || bad1;                                                                \
      }                                                                                   \
    }                                                                                     \
    _Pragma("unroll") for (int ps = 0; ps < 2; ++ps) {                                    \
      lsum[ps] += __shfl_xor(lsum[ps], 1, 64);                                            \
      lsum[ps] += __shfl_xor(lsum[ps], 2, 64);                                            \
      if ((lane & 3) == 0) xw[(16 * ps + (lane >> 2)) * XS + D] = lsum[ps];               \
    }                                                                                     \
  } else if (lane < kWS) {                                                                \
    xw[lane * XS + D] = 0.f;                                                              \
  }                                                                                       \
  __builtin_amdgcn_wave_barrier();
    if (INV) {                               // the conditioner sees the transformed identity half
      V5_IDENTITY()
    }

    floatx4 h[NB][2];                        // residual stream: rows 16 nb + 4 q + r of column block j
    float cv[2][NSC > 0 ? NSC : 1];          // context operands of the gates (fp32 instruction)
    float bv[2][NS0];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float* xr = xw + (16 * j + m16) * XS;
      const float* cr = cw + (16 * j + m16) * CS;
#pragma unroll
      for (int s = 0; s < NS0; ++s) bv[j][s] = s < DI / 4 ? xr[xcol[s < DI / 4 ? s : 0]] : cr[4 * (s - DI / 4) + q];
#pragma unroll
      for (int s = 0; s < NSC; ++s) cv[j][s] = cr[4 * s + q];
    }
    __builtin_amdgcn_wave_barrier();
    if (!INV) {
      V5_IDENTITY()
    }
#undef V5_IDENTITY
    V5_T(9)
    // operand halves: bh/bl[s][j] = k-chunk s (row blocks 2s, 2s+1) of column block j
    half8 bh[NS32][2], bl[NS32][2], b2h[NS32][2], b2l[NS32][2];
    half4 qh[2], ql[2];                      // halves of the operand fragment being assembled
    // one value -> its hi / lo halves (ReLU when LOWER = 0, saturation only when LOWER = -65504)
#define V5_SPLIT1(X, LOWER, R, V)                                                         \
  {                                                                                       \
    const float x_ = __builtin_amdgcn_fmed3f(X, LOWER, 65504.f);                          \
    const _Float16 hv_ = (_Float16)x_;                                                    \
    qh[R][V] = hv_;                                                                       \
    ql[R][V] = (_Float16)((x_ - (float)hv_) * kLoScale);                                  \
  }
#define V5_PACK(DH, DL, S, J)                                                             \
  {                                                                                       \
    DH[S][J] = __builtin_shufflevector(qh[0], qh[1], 0, 1, 2, 3, 4, 5, 6, 7);             \
    DL[S][J] = __builtin_shufflevector(ql[0], ql[1], 0, 1, 2, 3, 4, 5, 6, 7);             \
  }
    // The vector work of a pending unit (2 row blocks x 2 column blocks x 4 rows) is cut into 16
    // values n = 4 * (2 j + rb) + r; value n rides with 3 matrix instructions of the next unit in one
    // fenced mini-region (a transition unit, whose result is k-chunk 3 of the very next layer, is
    // done two values at a time in the first half of the slot).
    // first layer: bias, relu, operand chunk PI of block 0
#define V5_VAL_0(PI, N)                                                                   \
  {                                                                                       \
    const int rb_ = ((N) >> 2) & 1, j_ = (N) >> 3, r_ = (N) & 3;                          \
    h[2 * (PI) + rb_][j_][r_] += b0v[2 * (PI) + rb_][r_];                                 \
    V5_SPLIT1(h[2 * (PI) + rb_][j_][r_], 0.f, rb_, r_)                                    \
    if (rb_ == 1 && r_ == 3) V5_PACK(bh, bl, PI, j_)                                      \
  }
    // first layer of a block: pending unit -> relu -> operand chunk PI of the second layer
#define V5_VAL_A(PI, N)                                                                   \
  {                                                                                       \
    const int rb_ = ((N) >> 2) & 1, j_ = (N) >> 3, r_ = (N) & 3;                          \
    const float t_ = fmaf(pc[rb_][j_][r_], kLoUnscale, pm[rb_][j_][r_]) + pbias[rb_][r_]; \
    V5_SPLIT1(t_, 0.f, rb_, r_)                                                           \
    if (rb_ == 1 && r_ == 3) V5_PACK(b2h, b2l, PI, j_)                                    \
  }
    // second layer: gate, residual update of row blocks 2 PI, 2 PI + 1, operand chunk PI of what
    // follows (relu for another block, h itself for the last layer, resnet.py:105)
#define V5_VAL_B(PI, RELU, N)                                                             \
  {                                                                                       \
    const int rb_ = ((N) >> 2) & 1, j_ = (N) >> 3, r_ = (N) & 3;                          \
    const float t2 = fmaf(pc[rb_][j_][r_], kLoUnscale, pm[rb_][j_][r_]) + pbias[rb_][r_]; \
    if (C > 0) {                                                                          \
      const float sg = div_nr(1.f, 1.f + hw_exp2(-pg[rb_][j_][r_] * kLog2e));             \
      h[2 * (PI) + rb_][j_][r_] = fmaf(t2, sg, h[2 * (PI) + rb_][j_][r_]);                \
    } else {                                                                              \
      h[2 * (PI) + rb_][j_][r_] += t2;                                                    \
    }                                                                                     \
    V5_SPLIT1(h[2 * (PI) + rb_][j_][r_], (RELU) ? 0.f : -65504.f, rb_, r_)                \
    if (rb_ == 1 && r_ == 3) V5_PACK(bh, bl, PI, j_)                                      \
  }
#define V5_VAL_NONE(N)
    // pending unit: accumulators (main, correction), gate pre-activations, bias rows
    floatx4 pm[2][2], pc[2][2], pg[2][2], pbias[2];
    floatx4 b0v[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) b0v[nb] = wload(wr, qoff, 4 * (L::B0 + 16 * nb));
#pragma unroll
    for (int p = 0; p < NB / 2; ++p) {
      // matrix: row blocks 2p, 2p+1 (fp32 instruction), 3 instructions + one value of pair p - 1 per mini-region
      floatx4 w0[2][NS0_4];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int s4 = 0; s4 < NS0_4; ++s4)
          w0[rb][s4] = __builtin_bit_cast(floatx4, cur[((2 * p + rb) * NS0_4 + s4) * 64 + lane]);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int j = 0; j < 2; ++j) h[2 * p + rb][j] = floatx4{0.f, 0.f, 0.f, 0.f};
      V5_FENCE()
#pragma unroll
      for (int n = 0; n < 16; ++n) {
#pragma unroll
        for (int i = 0; i < (4 * NS0) / 16; ++i) {
          const int m = n * ((4 * NS0) / 16) + i;          // instruction m: k-step m / 4, (rb, j) = m % 4
          const int s = m >> 2, rb = (m >> 1) & 1, j = m & 1;
          h[2 * p + rb][j] = mfma4(w0[rb][s >> 2][s & 3], bv[j][s], h[2 * p + rb][j]);
        }
        if (p > 0) {
          V5_VAL_0(p - 1, n)
        }
        V5_FENCE()
      }
    }
    V5_STORE_HID()
    V5_FLIP()

    // fragments of k-step S of the chunk-local row blocks RBL0, RBL0 + 1
#define V5_PAIR_FRAGS(RBL0, S, SLOT)                                                      \
  _Pragma("unroll") for (int rb = 0; rb < 2; ++rb) {                                      \
    fa[SLOT][rb] = __builtin_bit_cast(half8, cur[(((RBL0) + rb) * NS32 + (S)) * 64 + lane]);       \
    fl[SLOT][rb] = __builtin_bit_cast(half8, cur[(16 + ((RBL0) + rb) * NS32 + (S)) * 64 + lane]);  \
  }
    // One hidden slot: 16 mini-regions (k-step s, quarter v), each 3 of the 12 matrix instructions of
    // the k-step (+ 1 of the gate's when GATED) and value(s) of the pending unit; the 4 fragment reads of
    // k-step s + 1 sit in mini-region (s, 0).  DOUBLE: a transition unit, values 2 mu, 2 mu + 1 in the
    // first 8 mini-regions.
#define V5_PAIR_SLOT(RBL0, BH, BL, GATED, VAL, DOUBLE)                                    \
  {                                                                                       \
    half8 fa[2][2], fl[2][2];                                                             \
    _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                      \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                     \
        nm[rb][j] = floatx4{0.f, 0.f, 0.f, 0.f};                                          \
        nc[rb][j] = floatx4{0.f, 0.f, 0.f, 0.f};                                          \
        ng[rb][j] = floatx4{0.f, 0.f, 0.f, 0.f};                                          \
      }                                                                                   \
    V5_PAIR_FRAGS(RBL0, 0, 0)                                                             \
    V5_FENCE()                                                                            \
    _Pragma("unroll") for (int mu = 0; mu < 16; ++mu) {                                   \
      const int s = mu >> 2, v = mu & 3;                                                  \
      if (v == 0 && s + 1 < NS32) {                                                       \
        V5_PAIR_FRAGS(RBL0, s + 1, (s + 1) & 1)                                           \
      }                                                                                   \
      _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                     \
        const int m = 3 * v + i, j = m / 6, t = m % 6;                                    \
        if (t < 2) nm[t][j] = mfma16h(fa[s & 1][t], BH[s][j], nm[t][j]);                  \
        else if (t < 4) nc[t - 2][j] = mfma16h(fa[s & 1][t - 2], BL[s][j], nc[t - 2][j]); \
        else nc[t - 4][j] = mfma16h(fl[s & 1][t - 4], BH[s][j], nc[t - 4][j]);            \
      }                                                                                   \
      if ((GATED) && C > 0 && s >= 2) {      /* gate of (row block rb, column block s - 2), k-step v */ \
        _Pragma("unroll") for (int rb = 0; rb < 2; ++rb) {                                \
          if (v == 0) ng[rb][s - 2] = sbc[(RBL0) + rb];                                   \
          ng[rb][s - 2] = mfma4(swc[(RBL0) + rb][v], cv[s - 2][v], ng[rb][s - 2]);        \
        }                                                                                 \
      }                                                                                   \
      if (DOUBLE) {                                                                       \
        if (mu < 8) {                                                                     \
          VAL(2 * mu)                                                                     \
          VAL(2 * mu + 1)                                                                 \
        }                                                                                 \
      } else {                                                                            \
        VAL(mu)                                                                           \
      }                                                                                   \
      V5_FENCE()                                                                          \
    }                                                                                     \
  }
#define V5_ROTATE()                                                                       \
  _Pragma("unroll") for (int rb = 0; rb < 2; ++rb) {                                      \
    pbias[rb] = nbias[rb];                                                                \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                       \
      pm[rb][j] = nm[rb][j];                                                              \
      pc[rb][j] = nc[rb][j];                                                              \
      pg[rb][j] = ng[rb][j];                                                              \
    }                                                                                     \
  }

#undef V5_BUCKET
#define V5_BUCKET 3
#pragma unroll
    for (int blk = 0; blk < (VCNF_ABL == 5 ? 0 : NBLK); ++blk) {
      const int base = L::BLK0 + blk * L::BLK;
      // ================= first layer of the block, two half-layer chunks          resnet.py:42-46
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        floatx4 sb[4], swc[4], sbc[4];       // small loads first: vector memory returns in order
#pragma unroll
        for (int rbl = 0; rbl < 4; ++rbl) sb[rbl] = wload(wr, qoff, 4 * (base + L::BA + 16 * (4 * hf + rbl)));
        V5_FENCE()
        if (hf == 0) {
          V5_LOAD_HID(base + L::WA, 1)
        } else {
          V5_LOAD_HID(base + L::WB, 0)
        }
        V5_FENCE()
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const int pi = 2 * hf + pl;        // pair index inside the layer
          floatx4 nm[2][2], nc[2][2], ng[2][2], nbias[2] = {sb[2 * pl], sb[2 * pl + 1]};
          if (pi > 0) {
#define V5_VAL(N) V5_VAL_A(pi - 1, N)
            V5_PAIR_SLOT(2 * pl, bh, bl, false, V5_VAL, false)
#undef V5_VAL
          } else if (blk > 0) {
#define V5_VAL(N) V5_VAL_B(NS32 - 1, true, N)
            V5_PAIR_SLOT(2 * pl, bh, bl, false, V5_VAL, true)      // last unit of the previous block
#undef V5_VAL
          } else {
#define V5_VAL(N) V5_VAL_0(NS32 - 1, N)
            V5_PAIR_SLOT(2 * pl, bh, bl, false, V5_VAL, true)      // last unit of the first layer
#undef V5_VAL
          }
          V5_ROTATE()
        }
        V5_STORE_HID()
        V5_FLIP()
      }
      // ================= second layer, GLU gate on the context, residual update      :48-57
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        floatx4 sb[4], swc[4], sbc[4];
#pragma unroll
        for (int rbl = 0; rbl < 4; ++rbl) {
          const int rb = 4 * hf + rbl;
          sb[rbl] = wload(wr, qoff, 4 * (base + L::BB + 16 * rb));
          if (C > 0) {
            swc[rbl] = wload(wr, voff, 4 * (base + L::WC + rb * (NSC > 0 ? NSC : 4) * 64));
            sbc[rbl] = wload(wr, qoff, 4 * (base + L::BC + 16 * rb));
          }
        }
        V5_FENCE()
        if (hf == 0) {
          V5_LOAD_HID(base + L::WB, 1)
        } else if (blk + 1 < NBLK) {
          V5_LOAD_HID(base + L::BLK + L::WA, 0)
        } else {
          V5_LOAD_WIDE(L::WF)
        }
        V5_FENCE()
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const int pi = 2 * hf + pl;
          floatx4 nm[2][2], nc[2][2], ng[2][2], nbias[2] = {sb[2 * pl], sb[2 * pl + 1]};
          if (pi > 0) {
#define V5_VAL(N) V5_VAL_B(pi - 1, (blk + 1 < NBLK), N)
            V5_PAIR_SLOT(2 * pl, b2h, b2l, true, V5_VAL, false)
#undef V5_VAL
          } else {
#define V5_VAL(N) V5_VAL_A(NS32 - 1, N)
            V5_PAIR_SLOT(2 * pl, b2h, b2l, true, V5_VAL, true)
#undef V5_VAL
          }
          V5_ROTATE()
        }
        if (hf == 1 && blk + 1 == NBLK) {
          V5_STORE_WIDE()
        } else {
          V5_STORE_HID()
        }
        V5_FLIP()
      }
    }

    // ================= last layer + splines: one feature group per chunk; the splines of group g - 1
    // are evaluated, piece by piece, in the slot that runs the matrix work of group g
    float ld_acc[2] = {0.f, 0.f};
    floatx4 pa[2][P4], pb[2][P4];
    // fragments of step u = b * NS32 + s are requested two steps ahead (ring of 3)
#define V5_GROUP_FRAGS(U)                                                                 \
  {                                                                                       \
    wh[(U) % 3] = __builtin_bit_cast(half8, cur[(U) * 64 + lane]);                        \
    wl[(U) % 3] = __builtin_bit_cast(half8, cur[(24 + (U)) * 64 + lane]);                 \
  }
    // One group slot: 24 mini-regions (parameter block b, k-step s): 6 matrix instructions, the fragment
    // reads two steps ahead, PIECE(u) of the vector work; the accumulators of block b are folded
    // into PA one mini-region after their last matrix instruction.
#define V5_GROUP_SLOT(PA, G, PIECE, STAGE_OFF)                                            \
  {                                                                                       \
    floatx4 gb[P4];                                                                       \
    _Pragma("unroll") for (int b = 0; b < P4; ++b)                                        \
      gb[b] = wload(wr, q * (16 * P4), 4 * (L::BF + (G) * 4 * (4 * P4) + 4 * b));         \
    V5_FENCE()                                                                            \
    V5_LOAD_WIDE(STAGE_OFF)                                                               \
    half8 wh[3], wl[3];                                                                   \
    floatx4 mainv[2][2], corr[2][2];                                                      \
    V5_GROUP_FRAGS(0)                                                                     \
    V5_GROUP_FRAGS(1)                                                                     \
    V5_FENCE()                                                                            \
    V5_TU(24)                                                                             \
    _Pragma("unroll") for (int u = 0; u < P4 * NS32; ++u) {                               \
      const int b = u / NS32, s = u % NS32;                                               \
      if (u + 2 < P4 * NS32) {                                                            \
        V5_GROUP_FRAGS(u + 2)                                                             \
      }                                                                                   \
      if (s == 0) {                                                                       \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                   \
          mainv[b & 1][j] = floatx4{0.f, 0.f, 0.f, 0.f};                                  \
          corr[b & 1][j] = floatx4{0.f, 0.f, 0.f, 0.f};                                   \
        }                                                                                 \
      }                                                                                   \
      if (VCNF_ABL != 13) {                                                               \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) mainv[b & 1][j] = mfma16h(wh[u % 3], bh[s][j], mainv[b & 1][j]); \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) corr[b & 1][j] = mfma16h(wh[u % 3], bl[s][j], corr[b & 1][j]);   \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) corr[b & 1][j] = mfma16h(wl[u % 3], bh[s][j], corr[b & 1][j]);   \
      }                                                                                   \
      if (s == 1 && b > 0) {                                                              \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                     \
          _Pragma("unroll") for (int r = 0; r < 4; ++r)                                   \
            PA[j][b - 1][r] = fmaf(corr[(b - 1) & 1][j][r], kLoUnscale, mainv[(b - 1) & 1][j][r]) + gb[b - 1][r]; \
      }                                                                                   \
      PIECE(u)                                                                            \
      V5_FENCE()                                                                          \
      V5_TU(u)                                                                            \
    }                                                                                     \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                         \
      _Pragma("unroll") for (int r = 0; r < 4; ++r)                                       \
        PA[j][P4 - 1][r] = fmaf(corr[(P4 - 1) & 1][j][r], kLoUnscale, mainv[(P4 - 1) & 1][j][r]) + gb[P4 - 1][r]; \
  }
    // Spline evaluation of one feature group in pieces (branch-free: points outside the interval
    // are evaluated at the left end and selected to the identity afterwards), two elements per lane
#define V5_SPLINE_BEGIN(G)                                                                \
  float* px[2];                                                                           \
  float xv[2], xe[2];                                                                     \
  bool inside[2];                                                                         \
  RqsStaged<K, INV> sp[2];                                                                \
  {                                                                                       \
    const int col = tfi[4 * (G) + q];                                                     \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                       \
      px[j] = xw + (16 * j + m16) * XS + col;                                             \
      xv[j] = *px[j];                                                                     \
      inside[j] = (xv[j] >= c.lo_x) && (xv[j] <= c.hi_x);                                 \
      xe[j] = inside[j] ? xv[j] : c.lo_x;                                                 \
    }                                                                                     \
  }                                                                                       \
  V5_FENCE()
    // piece U of 2 K + 3: exponentials of bin U (maxima first), normalisers, knot U - K - 1,
    // derivatives, the selected bin
#define V5_SPLINE_PIECE(PA, U)                                                            \
  if ((U) < 2 * K + 3) {                                                                  \
    _Pragma("unroll") for (int j = 0; j < (VCNF_ABL == 12 ? 0 : 2); ++j) {                \
      RegLogits<K, P4> p{PA[j], c.wh_scale, c.edge_logit};                                \
      if ((U) < K) {                                                                      \
        if ((U) == 0) sp[j].maxima(p);                                                    \
        sp[j].exps(p, c.wh_scale * kLog2e, (U), (U) + 1);                                   \
      } else if ((U) == K) {                                                              \
        sp[j].normalise(p, c);                                                            \
      } else if ((U) <= 2 * K) {                                                          \
        sp[j].knots(xe[j], p, c, (U) - K - 1, (U) - K);                                     \
      } else if ((U) == 2 * K + 1) {                                                      \
        sp[j].derivatives(c);                                                             \
      } else {                                                                            \
        float yv, lad;                                                                    \
        bool bad1 = false;                                                                \
        rqs_bin_eval<INV>(xe[j], sp[j].sel, yv, lad, bad1);                               \
        *px[j] = inside[j] ? yv : xv[j];                                                  \
        ld_acc[j] += inside[j] ? lad : 0.f;                                               \
        bad = bad || bad1;                                                                \
      }                                                                                   \
    }                                                                                     \
  }
#define V5_PIECE_NONE(U)
#undef V5_BUCKET
#define V5_BUCKET 4
    {   // slot of group 0: its matrix work; the trunk's last pending unit feeds k-chunk 3 of this very
        // slot and is done first (not hidden); the next tile's rows are requested here
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        V5_VAL_B(NS32 - 1, false, n)
      }
      V5_FENCE()
      V5_GROUP_SLOT(pa, 0, V5_PIECE_NONE, L::WF + 1 * P4 * NS32 * 256)
      V5_STORE_WIDE()
      V5_FLIP()
    }
#undef V5_BUCKET
#define V5_BUCKET 5
    for (int gi = 1; gi < (VCNF_ABL == 3 ? 0 : NG); gi += 2) {
      {   // group gi -> pb, splines of group gi - 1 from pa
        const int next_off = gi + 1 < NG ? L::WF + (gi + 1) * P4 * NS32 * 256 : L::W0;
        V5_TU0()
        V5_SPLINE_BEGIN(gi - 1)
#define V5_PIECE(U) V5_SPLINE_PIECE(pa, U)
        V5_GROUP_SLOT(pb, gi, V5_PIECE, next_off)
#undef V5_PIECE
        if (gi + 1 == NG) {                  // last slot of the tile: the next tile's rows are requested
          V5_PREFETCH_ROWS(tile + gridDim.x)
        }
        V5_STORE_WIDE()
        V5_FLIP()
        V5_TU(25)
      }
      if (gi + 1 < NG) {   // group gi + 1 -> pa, splines of group gi from pb
        V5_TU0()
        V5_SPLINE_BEGIN(gi)
#define V5_PIECE(U) V5_SPLINE_PIECE(pb, U)
        V5_GROUP_SLOT(pa, gi + 1, V5_PIECE, L::WF + (gi + 2) * P4 * NS32 * 256)
#undef V5_PIECE
        V5_STORE_WIDE()
        V5_FLIP()
      }
    }
    if (VCNF_ABL != 3) {
      V5_SPLINE_BEGIN(NG - 1)
#pragma unroll
      for (int u = 0; u < 2 * K + 3; ++u) {
        V5_SPLINE_PIECE(pb, u)
      }
    }
    __builtin_amdgcn_wave_barrier();

    V5_T(6)
    // ---- per-sample log|det| and the wave's rows of y
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float v = ld_acc[j];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int mrow = 16 * j + m16;
      if (q == 0 && mrow < rows) {
        const float o = a.ld_sign * (v + xw[mrow * XS + D]);
        a.logdet[b0 + mrow] = a.ld_mode ? a.logdet[b0 + mrow] + o : o;
      }
    }
    {
      float4* dst = reinterpret_cast<float4*>(a.y) + b0 * D4;
#pragma unroll
      for (int k = 0; k < kWS * D4 / 64; ++k) {
        const int i = lane + 64 * k;
        const int r = i / D4, o = i - r * D4;
        if (r < rows) dst[i] = *reinterpret_cast<const float4*>(xw + r * XS + 4 * o);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
#undef V5_BUCKET
#if VCNF_TIME
  if (blockIdx.x == 0 && tid == 0) {
    for (int i = 0; i < 12; ++i) a.y[i] = (float)tacc[i];
#if VCNF_TIME == 2
    for (int i = 0; i < 26; ++i) a.y[16 + i] = (float)tu[i];
#endif
  }
#endif
  if (INV && a.bad && bad) atomicAdd(a.bad, 1);
#undef V5_LOAD_HID
#undef V5_STORE_HID
#undef V5_LOAD_WIDE
#undef V5_STORE_WIDE
#undef V5_FLIP
#undef V5_T
#undef V5_FENCE
#undef V5_PREFETCH_ROWS
#undef V5_SPLIT1
#undef V5_PACK
#undef V5_VAL_0
#undef V5_VAL_A
#undef V5_VAL_B
#undef V5_VAL_NONE
#undef V5_PAIR_FRAGS
#undef V5_PAIR_SLOT
#undef V5_ROTATE
#undef V5_GROUP_FRAGS
#undef V5_GROUP_SLOT
#undef V5_SPLINE_BEGIN
#undef V5_SPLINE_PIECE
#undef V5_PIECE_NONE
}

template <int DI, int DT, int C, int H, int NBLK, int K>
static int launch_v5(const FusedArgs& a, int inverse, hipStream_t st) {
  constexpr int D = DI + DT;
  const size_t lds = (size_t)2 * 48 * 64 * 16 +
                     ((size_t)4 * 32 * (D + 4) + (size_t)4 * 32 * ((C > 0 ? C : 4) + 4) +
                      ((DI * 3 * (K + 1) + 3) & ~3) + D + 4) * 4 + 64;
  static bool attr_set[2] = {false, false};
  if (!attr_set[inverse ? 1 : 0]) {
    hipError_t e;
    if (inverse)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_rqs_layer_v5_kernel<DI, DT, C, H, NBLK, K, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_rqs_layer_v5_kernel<DI, DT, C, H, NBLK, K, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return VCNF_ERR_LAUNCH;
    attr_set[inverse ? 1 : 0] = true;
  }
  const long long ntiles = (a.B + 127) / 128;
  dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256));
  if (inverse)
    hipLaunchKernelGGL((fused_rqs_layer_v5_kernel<DI, DT, C, H, NBLK, K, true>), grid, dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL((fused_rqs_layer_v5_kernel<DI, DT, C, H, NBLK, K, false>), grid, dim3(256), lds, st, a);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

int launch_fused_v5_c16(const FusedArgs& a, int inverse, hipStream_t st) {
  return launch_v5<32, 32, 16, 128, 2, 8>(a, inverse, st);
}

}  // namespace vcnf
