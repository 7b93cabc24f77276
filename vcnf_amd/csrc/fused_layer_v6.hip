// Fused RQS coupling layer, fp16 split-half matrix path ("fp16x3"), sixth structure.
//
// Same contract as fused_layer.hip (exact fp32 path): one launch evaluates a whole
// PiecewiseRationalQuadraticCoupling layer - identity-half spline, ResidualNet conditioner, transformed-half
// splines, per-sample log|det| (reference: flows/neural_spline/coupling.py:70-125, 309-343; nets/resnet.py:60-106;
// utils/splines.py:20-193).  What changed against the fourth structure (profiles/tools/superseded/), each change
// following a measurement kept under profiles/r02_*:
//   * v_mfma_f32_32x32x16_f16 instead of 16x16x32: same matrix-pipe time per flop, but an instruction holds
//     the SIMD's vector issue for 8 of 32 cycles instead of 8 of 16, so the partner wave's vector step keeps
//     ~80 % of its issue rate beside a matrix step (v_fma beside back-to-back MFMAs: 6.1 against 8.1 cycles);
//   * the transformed-half spline is evaluated in "exp-sum space" (rqs_lean.hpp): ~190 vector instructions
//     instead of ~290, no packed-f32 instructions (a vector step built from v_pk_* made NO progress beside
//     the partner's matrix instructions: the translation unit is compiled with -fno-slp-vectorize);
//   * the 1/sqrt(hidden) scale of the width / height logits, log2(e) of every exponential and of the gate
//     sigmoid are folded into the packed weights on the host (vcnf_amd/fused.py);
//   * first layer and context gates run on the split-half instruction as well (K = 48 and 16 are whole k-steps
//     of 16), their inputs split once per tile (the first layer keeps the lo*lo product: its inputs are raw
//     fp32 data); every split saturates at +-65504 and the kernel counts workgroups in which a value was clamped
//     (FusedArgs::sat, surfaced by nf.check_saturation());
//   * the last layer's bias sits in an LDS table (its loads were issued at the end of a vector step and waited
//     for at the head of the next matrix step), a split costs 3.5 instead of 6.5 vector instructions per value
//     (v_cvt_pk_f16_f32 + v_fma_mix*_f16), barriers wait for LDS only.
// Two other structures were built and measured this round and are kept as text under profiles/tools/superseded/:
// v7 (two independent 256-thread workgroups per CU, no x / y tile in LDS: same speed, 1.6x the HBM traffic) and v8
// (this kernel with software barriers per wave group: slower); record in profiles/r02_fused_layer_structures.md.
//
// Work split (unchanged in spirit): 512 threads = wave groups A (waves 0-3) and B (4-7); wave w and w+4 share
// a SIMD and run the same step sequence one step apart, so that one is in a matrix step (M) while the other is
// in a vector step (V).  Tile = 128 samples = 4 column blocks of 32.  Trunk: wave = (32-row block rp, sample
// half ch) with the layer's weights for its rows stationary in registers and activations travelling through
// LDS as ready-made B fragments (hi | lo halves).  Last layer: wave = (column block rp, feature-group parity
// ch); its 32 samples' activations are stationary in registers and the 48 KB of weights of one group of four
// features stream through the group's half of an LDS window by buffer_load ... lds.
//
// MFMA 32x32x16 layouts (measured, scratch probe recorded in profiles/r02_mfma_32x32x16_layout.txt):
//   A: lane l holds row l % 32, k-slots (l / 32, 0..7);  B: lane l holds column l % 32, same k-slots;
//   D: register r of lane l is row 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32.
// A layer's accumulators become the next layer's B operand without any shuffle: k-step t, slot (kg, i) is hidden
// unit 16 t + 8 (i / 4) + 4 kg + i % 4; the host packs the weights in that k order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"
#include "rqs_math.hpp"
#include "fused_common.hpp"
#include "rqs_lean.hpp"
#include "split_half.hpp"

#ifndef VCNF_ABL
#define VCNF_ABL 0
#endif
// -DVCNF_TIME=1: s_memtime stamps at every barrier; wave 0 of workgroup 0 leaves the per-phase sums in
// the first output row (timing builds only, read by profiles/tools/v6_phase_timing.py)
#ifndef VCNF_TIME
#define VCNF_TIME 0
#endif
// operand fragments are requested VCNF_AHEAD steps (of three matrix instructions) before their use
#ifndef VCNF_AHEAD
#define VCNF_AHEAD 1
#endif
#if VCNF_TIME
#define VCNF_T(I) { const long long t_ = clock64(); tacc[I] += t_ - tlast; tlast = t_; }
#else
#define VCNF_T(I)
#endif

// Workgroup barrier.  Not __syncthreads(): its release fence drains EVERY outstanding vector-memory operation
// (s_waitcnt vmcnt(0)) in front of every barrier - the next layer's weights requested in this step, the next
// tile's rows - and puts their latency into every step.  What the steps exchange through the barrier is LDS data
// only: the wave's LDS operations are complete (lgkmcnt(0)), window pieces written by buffer_load ... lds are
// waited for explicitly by the wave that requested them (wait_vector_memory).  The inline assembly is opaque to
// the compiler (memory clobber: no access moves across it) and fenced for the instruction scheduler (matrix
// instructions are no memory operations and were otherwise sunk below the barrier that ends their step, into the
// vector step, where the next layer's weights are already being loaded: both weight sets live, spills).
#define VCNF_SYNC() { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

namespace vcnf {

template <int DI, int DT, int C, int H, int NBLK, int K, bool INV>
__global__ __launch_bounds__(512, 2) void fused_rqs_layer_v6_kernel(const FusedArgs a) {
  static_assert(H == 128 && K == 8, "4 row blocks of 32 over 4 waves; 3 K - 1 = 23 logits: two features per 48 rows");
  static_assert((DI == 16 || DI == 32) && DT == DI && (C == 0 || C == 16), "shape family");
  constexpr int kBlock = 512;
  constexpr int kTile = 128;
  constexpr int NCB = 4;                    // 32-sample column blocks per tile
  constexpr int D = DI + DT;
  constexpr int XS = D + 4;
  constexpr int NTX = DI / 16;              // k-steps of the identity features in the first layer
  constexpr int NTC = C / 16;               // k-steps of the context (0 or 1)
  constexpr int NT0 = NTX + NTC;
  constexpr int NTH = H / 16;               // k-steps of a hidden->* layer (8)
  constexpr int P = 3 * K - 1;
  constexpr int NG = DT / 4;                // feature groups (4 features = 96 rows = 3 row blocks)
  constexpr int NR = NG / 2;                // rounds = feature groups per wave group
  constexpr int TABW = 3 * (K + 1);
  using L = PackLayout6<DI, DT, C, H, NBLK, K>;
  constexpr int GFRAG = 3 * NTH * 2 * 64;   // 16-byte fragments of one feature group (48 KB)
  constexpr int RING = VCNF_AHEAD + 1;

  extern __shared__ __align__(16) float smem[];
  // fragment region first (LDS offset 0: every fragment address is a per-lane base + 16-bit immediate).
  // Trunk: activation fragments [t][cb][lane] of 16 bytes, hi (32 KB) then lo (32 KB); last layer: the weight
  // window [group parity][b][t][hi|lo][lane] (96 KB).
  uint4* act = reinterpret_cast<uint4*>(smem);
  uint4* act_hi = act;
  uint4* act_lo = act + NTH * NCB * 64;
  float* xt = smem + 2 * GFRAG * 4;                        // [128][XS]  x in, y out (in place)
  uint4* ctxf = reinterpret_cast<uint4*>(xt + kTile * XS); // [cb][hi|lo][lane] context fragments (8 KB)
  float* tab = reinterpret_cast<float*>(ctxf + (C > 0 ? NCB * 2 * 64 : 0));   // [DI][TABW]
  float* ldt = tab + ((DI * TABW + 3) & ~3);               // [128] identity-half log|det|
  int* tfi = reinterpret_cast<int*>(ldt + kTile);
  int* idi = tfi + DT;
  float* biasf = reinterpret_cast<float*>(idi + DI + 4);   // [NG][lane half][48] last-layer bias (3 KB)
  int* tflag = reinterpret_cast<int*>(biasf + NG * 96);    // this tile held a value the fp16 halves cannot carry

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: feeds scalar offsets
  const int rp = wave & 3;                  // trunk: 32-row block; last layer: column block
  const int ch = wave >> 2;                 // trunk: sample half;  last layer: feature-group parity
  const int c32 = lane & 31;
  const int kg = lane >> 5;
  const RqsConst& c = a.c;
  const bool shared = a.sh_w != nullptr;
  const LeanConst lc = make_lean_const(c);

  // the second-dispatched wave group loses the issue arbitration on every step (MI355X_MICROARCH.md, two waves per
  // SIMD, item 4): one static priority for it, no per-step flips (+0.5 %)
  if (ch == 1) __builtin_amdgcn_s_setprio(1);
  for (int i = tid; i < NG * 96; i += kBlock) biasf[i] = a.wpack[L::BF + i];
  for (int i = tid; i < DT; i += kBlock) tfi[i] = a.tf_idx[i];
  for (int i = tid; i < DI; i += kBlock) idi[i] = a.id_idx[i];
  if (shared) {
    // knot tables of the identity half: one thread per (feature, column: x knots | y knots | derivatives)
    for (int i = tid; i < 3 * DI; i += kBlock) {
      const int f = i % DI;
      SplitLogits p{a.sh_w + f * K, a.sh_h + f * K, a.sh_d + f * (K - 1), K, 1.f, c.edge_logit, c.tails};
      rqs_build_table_part_k<K>(p, c, tab + f * TABW, 1, i / DI);
    }
  }

  const __amdgpu_buffer_rsrc_t wr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpack), 0, a.wpack_bytes, 0x00020000);
  const int voff = lane * 16;
  const int boff = kg * 64;                 // bias rows of this lane half: [nb][kg][16] floats

  const long long ntiles = (a.B + kTile - 1) / kTile;
  bool bad = false;
  float satm = 0.f;
#if VCNF_TIME
  long long tacc[16], tlast = clock64();
  const long long tstart = tlast, rstart = wall_clock64();
  for (int i = 0; i < 16; ++i) tacc[i] = 0;
#endif
  // rows of the next tile travel in registers: bounds-checked buffer loads (rows past the batch read 0)
  float4 xpre[kTile * (D / 4) / kBlock], cpre[1];
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
    satm = 0.f;
    { VCNF_T(0) VCNF_SYNC(); VCNF_T(15) }
    {   // ---- x rows -> LDS tile; context row quarter -> half of a B fragment (hi | lo)
      constexpr int D4 = D / 4;
      if (tid == 0) *tflag = 0;
#pragma unroll
      for (int k = 0; k < kTile * D4 / kBlock; ++k) {
        const int i = tid + kBlock * k;
        const int r = i / D4, o = i - r * D4;
        *reinterpret_cast<float4*>(xt + r * XS + 4 * o) = xpre[k];
      }
      if (C > 0) {
        // thread holds context columns 4 (tid & 3) .. + 3 of row tid >> 2: k-slots (kg', 4 half .. + 3)
        const int r = tid >> 2, part = tid & 3;
        half4 h4, l4;
        const float cv[4] = {cpre[0].x, cpre[0].y, cpre[0].z, cpre[0].w};
        // raw inputs: a NaN counts as out of range too (fmaxf drops NaNs; the reference propagates them)
#pragma unroll
        for (int i = 0; i < 4; ++i) satm = fmaxf(satm, cv[i] == cv[i] ? __builtin_fabsf(cv[i]) : __builtin_inff());
        asm volatile("" : "+v"(satm));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x = __builtin_amdgcn_fmed3f(cv[i], -65504.f, 65504.f);
          const _Float16 hv = (_Float16)x;
          h4[i] = hv;
          l4[i] = (_Float16)((x - (float)hv) * kLoScale);
        }
        const int at = (((r >> 5) * 2) * 64 + (r & 31) + 32 * (part >> 1));          // uint4 index of the hi fragment entry
        uint2* dh = reinterpret_cast<uint2*>(ctxf + at) + (part & 1);
        uint2* dl = reinterpret_cast<uint2*>(ctxf + at + 64) + (part & 1);
        *dh = __builtin_bit_cast(uint2, h4);
        *dl = __builtin_bit_cast(uint2, l4);
      }
    }
    { VCNF_T(1) VCNF_SYNC(); VCNF_T(15) }

    // ---- identity half through the unconditional spline: 4 lanes per sample, each lane a run of DI/4
    // features (branch-free: points outside the interval are evaluated at the left end and selected to the
    // identity afterwards).  The same thread holds exactly one (half) B fragment of the first layer's input:
    // raw values in the density direction (coupling.py:78-81), transformed ones in the sampling direction
    // (:110-114) - split to hi | lo and stored in the activation region.
    {
      constexpr int UNR = DI / 4;
      const int mi = tid >> 2, part = tid & 3;
      float lsum = 0.f;
      float xv[UNR], fv[UNR];
      float* px[UNR];
#pragma unroll
      for (int k = 0; k < UNR; ++k) {
        px[k] = xt + mi * XS + idi[part * UNR + k];
        xv[k] = *px[k];
        fv[k] = xv[k];
      }
      if (shared) {
        float yv[UNR], lad[UNR];
        bool in_[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
          in_[k] = (xv[k] >= c.lo_x) && (xv[k] <= c.hi_x);
          bool bad1 = false;
          rqs_point_table_inside<INV, K>(in_[k] ? xv[k] : c.lo_x, tab + (part * UNR + k) * TABW, yv[k], lad[k], bad1);
          bad = bad || (bad1 && in_[k]);
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
          const float o = in_[k] ? yv[k] : xv[k];
          *px[k] = o;
          if (INV) fv[k] = o;
          lsum += in_[k] ? lad[k] : 0.f;
        }
      }
      lsum += __shfl_xor(lsum, 1, 64);
      lsum += __shfl_xor(lsum, 2, 64);
      if (part == 0) ldt[mi] = lsum;
      // raw inputs of the conditioner: NaN / Inf count as out of range (fmaxf in the splits drops NaNs)
#pragma unroll
      for (int k = 0; k < UNR; ++k) satm = fmaxf(satm, fv[k] == fv[k] ? 0.f : __builtin_inff());
      // fragment: first-layer k = 16 t + 8 kg' + i  <->  identity feature part * UNR + k
      const int k0 = part * UNR;                               // first feature of this thread
      const int t0 = k0 >> 4, kg0 = (k0 >> 3) & 1;
      const int at = (t0 * NCB + (mi >> 5)) * 64 + (mi & 31) + 32 * kg0;
      if (UNR == 8) {
        half8 h8, l8;
        float f8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) f8[k] = fv[k < UNR ? k : 0];
        split8<false>(f8, h8, l8, satm);
        act_hi[at] = __builtin_bit_cast(uint4, h8);
        act_lo[at] = __builtin_bit_cast(uint4, l8);
      } else {
        half4 h4, l4;
        satm = fmaxf(fmaxf(satm, __builtin_fabsf(fv[0])), __builtin_fabsf(fv[1]));
        satm = fmaxf(fmaxf(satm, __builtin_fabsf(fv[2 < UNR ? 2 : 0])), __builtin_fabsf(fv[3 < UNR ? 3 : 0]));
        asm volatile("" : "+v"(satm));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x = __builtin_amdgcn_fmed3f(fv[i < UNR ? i : 0], -65504.f, 65504.f);
          const _Float16 hv = (_Float16)x;
          h4[i] = hv;
          l4[i] = (_Float16)((x - (float)hv) * kLoScale);
        }
        reinterpret_cast<uint2*>(act_hi + at)[(k0 >> 2) & 1] = __builtin_bit_cast(uint2, h4);
        reinterpret_cast<uint2*>(act_lo + at)[(k0 >> 2) & 1] = __builtin_bit_cast(uint2, l4);
      }
    }
    { VCNF_T(2) VCNF_SYNC(); VCNF_T(15) }
    if (ch == 1) { VCNF_T(14) VCNF_SYNC(); VCNF_T(15) }            // ---- group B now runs one step behind group A

    // stationary weights of a hidden->hidden layer for this wave's 32 rows, bias in accumulator order
    half8 ahi[NTH], alo[NTH];
    floatx16 abias;
#define VCNF_LOAD_BIAS16(DST, FOFF)                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                      \
    const floatx4 b4_ = wload(wr, boff, 4 * ((FOFF) + 32 * rp) + 16 * i_);                \
    DST[4 * i_ + 0] = b4_[0]; DST[4 * i_ + 1] = b4_[1]; DST[4 * i_ + 2] = b4_[2]; DST[4 * i_ + 3] = b4_[3]; \
  }
#define VCNF_LOAD_HIDDEN(WOFF, BOFF)                                                      \
  {                                                                                       \
    _Pragma("unroll") for (int t = 0; t < NTH; ++t) {                                     \
      ahi[t] = __builtin_bit_cast(half8, wload(wr, voff, 4 * ((WOFF) + ((rp * NTH + t) * 2 + 0) * 256))); \
      alo[t] = __builtin_bit_cast(half8, wload(wr, voff, 4 * ((WOFF) + ((rp * NTH + t) * 2 + 1) * 256))); \
    }                                                                                     \
    VCNF_LOAD_BIAS16(abias, BOFF)                                                         \
  }

    // ---- step M0: first layer, 32 rows x 2 column blocks per wave                      resnet.py:92-99
    floatx16 h[2];
    {
      half8 w0h[NT0], w0l[NT0];
      floatx16 bias0;
#pragma unroll
      for (int t = 0; t < NT0; ++t) {
        w0h[t] = __builtin_bit_cast(half8, wload(wr, voff, 4 * (L::W0 + ((rp * NT0 + t) * 2 + 0) * 256)));
        w0l[t] = __builtin_bit_cast(half8, wload(wr, voff, 4 * (L::W0 + ((rp * NT0 + t) * 2 + 1) * 256)));
      }
      VCNF_LOAD_BIAS16(bias0, L::B0)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cb = 2 * ch + j;
        floatx16 mainv = bias0, corr = {}, corr2 = {};
#pragma unroll
        for (int t = 0; t < NT0; ++t) {
          const half8 bh = __builtin_bit_cast(half8, t < NTX ? act_hi[(t * NCB + cb) * 64 + lane] : ctxf[(cb * 2 + 0) * 64 + lane]);
          const half8 bl = __builtin_bit_cast(half8, t < NTX ? act_lo[(t * NCB + cb) * 64 + lane] : ctxf[(cb * 2 + 1) * 64 + lane]);
          mainv = mfma32h(w0h[t], bh, mainv);
          corr = mfma32h(w0h[t], bl, corr);
          corr = mfma32h(w0l[t], bh, corr);
          corr2 = mfma32h(w0l[t], bl, corr2);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) h[j][r] = fmaf(fmaf(corr2[r], kLoUnscale, corr[r]), kLoUnscale, mainv[r]);
      }
    }
    { VCNF_T(3) VCNF_SYNC(); VCNF_T(15) }                         // ---- end of step M0
    // publish: registers 8 hh .. 8 hh + 7 of a column block are the eight k-slots of k-step 2 rp + hh
#define VCNF_PUBLISH(SRC, RELU)                                                           \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                         \
    _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                    \
      float v8_[8];                                                                       \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) v8_[i] = SRC[j][8 * hh + i];          \
      half8 h8_, l8_;                                                                     \
      split8<RELU>(v8_, h8_, l8_, satm);                                                  \
      const int at = ((2 * rp + hh) * NCB + 2 * ch + j) * 64 + lane;                      \
      act_hi[at] = __builtin_bit_cast(uint4, h8_);                                        \
      act_lo[at] = __builtin_bit_cast(uint4, l8_);                                        \
    }                                                                                     \
  }
    // ---- step V1: first hidden layer's weights requested, relu(h) published
    VCNF_LOAD_HIDDEN(L::BLK0 + L::WA, L::BLK0 + L::BA)
    VCNF_PUBLISH(h, true)
    { VCNF_T(4) VCNF_SYNC(); VCNF_T(15) }

    // OUT[j] = bias + W_slice * operand(column block 2 ch + j).  Operand fragments are requested two steps
    // (six matrix instructions) before their use: ring of three, pinned with sched_group_barrier.
#define VCNF_READ_B(T)                                                                    \
  {                                                                                       \
    rh[(T) % RING] = __builtin_bit_cast(half8, act_hi[((((T) & 7)) * NCB + 2 * ch + ((T) >> 3)) * 64 + lane]); \
    rl[(T) % RING] = __builtin_bit_cast(half8, act_lo[((((T) & 7)) * NCB + 2 * ch + ((T) >> 3)) * 64 + lane]); \
  }
#define VCNF_HIDDEN_COMPUTE(OUT)                                                          \
  {                                                                                       \
    half8 rh[RING], rl[RING];                /* ring: step st = 8 j + t uses slot st % RING */ \
    floatx16 corr;                                                                        \
    _Pragma("unroll") for (int st_ = 0; st_ < VCNF_AHEAD; ++st_) {                        \
      VCNF_READ_B(st_)                                                                    \
    }                                                                                     \
    _Pragma("unroll") for (int st_ = 0; st_ < 2 * NTH; ++st_) {                           \
      const int j = st_ >> 3, tk_ = st_ & 7;                                              \
      if (st_ + VCNF_AHEAD < 2 * NTH) {                                                   \
        VCNF_READ_B(st_ + VCNF_AHEAD)                                                     \
      }                                                                                   \
      if (tk_ == 0) {                                                                       \
        OUT[j] = abias;                                                                   \
        corr = floatx16{};                                                                \
      }                                                                                   \
      OUT[j] = mfma32h(ahi[tk_], rh[st_ % RING], OUT[j]);                                      \
      corr = mfma32h(ahi[tk_], rl[st_ % RING], corr);                                          \
      corr = mfma32h(alo[tk_], rh[st_ % RING], corr);                                          \
      if (tk_ == NTH - 1) {                                                                 \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) OUT[j][r] = fmaf(corr[r], kLoUnscale, OUT[j][r]); \
      }                                                                                   \
    }                                                                                     \
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * VCNF_AHEAD, 0);                       \
    _Pragma("unroll") for (int st_ = 0; st_ + VCNF_AHEAD < 2 * NTH; ++st_) {              \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                  \
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                  \
    }                                                                                     \
    __builtin_amdgcn_sched_group_barrier(0x008, 3 * VCNF_AHEAD, 0);                       \
  }

#pragma unroll
    for (int blk = 0; blk < (VCNF_ABL == 5 ? 0 : NBLK); ++blk) {
      const int base = L::BLK0 + blk * L::BLK;
      floatx16 t[2];
      // ---- step M: first layer of the block                                          resnet.py:42-43
      VCNF_HIDDEN_COMPUTE(t)
      { VCNF_T(5) VCNF_SYNC(); VCNF_T(15) }
      // ---- step V: publish relu(t) (:46); second layer's and gate weights requested
      VCNF_LOAD_HIDDEN(base + L::WB, base + L::BB)
      half8 wch, wcl;
      floatx16 gate[2], gbias;
      if (C > 0) {
        wch = __builtin_bit_cast(half8, wload(wr, voff, 4 * (base + L::WC + (rp * 2 + 0) * 256)));
        wcl = __builtin_bit_cast(half8, wload(wr, voff, 4 * (base + L::WC + (rp * 2 + 1) * 256)));
        VCNF_LOAD_BIAS16(gbias, base + L::BC)
      }
      VCNF_PUBLISH(t, true)
      { VCNF_T(6) VCNF_SYNC(); VCNF_T(15) }
      // ---- step M: second layer of the block (:48) and the gate pre-activations (:53)
      VCNF_HIDDEN_COMPUTE(t)
      if (C > 0 && VCNF_ABL != 6) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const half8 bh = __builtin_bit_cast(half8, ctxf[((2 * ch + j) * 2 + 0) * 64 + lane]);
          const half8 bl = __builtin_bit_cast(half8, ctxf[((2 * ch + j) * 2 + 1) * 64 + lane]);
          // 16-deep layer: the accumulator rounds once, so the operands' 22 bits are what is left of the error -
          // the lo*lo term is kept like in the first layer (tests/test_gpu_gemm_error.py)
          floatx16 corr = {}, corr2 = {};
          gate[j] = mfma32h(wch, bh, gbias);
          corr = mfma32h(wch, bl, corr);
          corr = mfma32h(wcl, bh, corr);
          corr2 = mfma32h(wcl, bl, corr2);
#pragma unroll
          for (int r = 0; r < 16; ++r) gate[j][r] = fmaf(fmaf(corr2[r], kLoUnscale, corr[r]), kLoUnscale, gate[j][r]);
        }
      }
      { VCNF_T(5) VCNF_SYNC(); VCNF_T(15) }
      // ---- step V: GLU gate (the packed gate weights carry log2 e: sigmoid(g) = 1 / (1 + 2^-g')),
      // residual update, publish                                                       :49-57
      if (blk + 1 < NBLK) {
        VCNF_LOAD_HIDDEN(base + L::BLK + L::WA, base + L::BLK + L::BA)
      }
      if (C > 0 && VCNF_ABL != 6) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float sg = hw_rcp(1.f + hw_exp2(-gate[j][r]));
            h[j][r] = fmaf(t[j][r], sg, h[j][r]);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) h[j] += t[j];
      }
      if (blk + 1 < NBLK) {
        VCNF_PUBLISH(h, true)
      } else {
        VCNF_PUBLISH(h, false)               // the last layer takes h itself (resnet.py:105)
      }
      { VCNF_T(7) VCNF_SYNC(); VCNF_T(15) }
    }
    if (ch == 0) { VCNF_T(14) VCNF_SYNC(); VCNF_T(15) }            // ---- groups re-aligned: all activations are published
#undef VCNF_HIDDEN_COMPUTE
#undef VCNF_READ_B
#undef VCNF_LOAD_HIDDEN
#undef VCNF_PUBLISH

    // ---- last layer + splines: wave owns column block rp for the feature groups of parity ch
    half8 fhi[NTH], flo[NTH];
#pragma unroll
    for (int t = 0; t < NTH; ++t) {
      fhi[t] = __builtin_bit_cast(half8, act_hi[(t * NCB + rp) * 64 + lane]);
      flo[t] = __builtin_bit_cast(half8, act_lo[(t * NCB + rp) * 64 + lane]);
    }
    float ld_acc = 0.f;
    uint4* win = act + ch * GFRAG;           // this group's half of the window: one feature group, 48 KB
    const int gtid = tid & 255;              // thread index inside the group
    constexpr int NSTG = GFRAG / 256;        // 16-byte fragments a thread moves per staged group (12)
    static_assert(GFRAG % 256 == 0, "whole fragments per thread");
    // fragment i = gtid + 256 k of feature group G sits at float offset WF + G * 4 GFRAG + 4 i: straight from
    // global memory into the window (buffer_load ... lds: the wave's 64 lanes land at consecutive 16-byte slots)
#define VCNF_STAGE_DMA(G)                                                                 \
  _Pragma("unroll") for (int k = 0; k < NSTG; ++k)                                        \
    dma16_to_lds(wr, win + (gtid & ~63) + k * 256, gtid * 16, 4 * (L::WF + (G) * (4 * GFRAG) + k * 1024));
    floatx16 pa[3];
#define VCNF_LOAD_BIASF(G)                                                                \
  _Pragma("unroll") for (int b = 0; b < 3; ++b) {                                         \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                    \
      const floatx4 b4_ = *reinterpret_cast<const floatx4*>(biasf + (G) * 96 + kg * 48 + 16 * b + 4 * i_); \
      pa[b][4 * i_ + 0] = b4_[0]; pa[b][4 * i_ + 1] = b4_[1]; pa[b][4 * i_ + 2] = b4_[2]; pa[b][4 * i_ + 3] = b4_[3]; \
    }                                                                                     \
  }
    VCNF_LOAD_BIASF(ch)
    { VCNF_T(8) VCNF_SYNC(); VCNF_T(15) }                         // every wave has its operand fragments: the window may be written
    VCNF_STAGE_DMA(ch)                                                 // first feature group of each wave group
    wait_vector_memory();
    { VCNF_T(9) VCNF_SYNC(); VCNF_T(15) }
    if (ch == 1) { VCNF_T(14) VCNF_SYNC(); VCNF_T(15) }            // ---- group B one step behind again
    for (int rnd = 0; rnd < (VCNF_ABL == 3 ? 0 : NR); ++rnd) {
      const int g = 2 * rnd + ch;
      // the two elements this lane transforms (features 4 g + 2 kg + {0, 1} of sample c32)
      float* px[2];
      float xin[2];
#pragma unroll
      for (int f2 = 0; f2 < 2; ++f2) {
        px[f2] = xt + (rp * 32 + c32) * XS + tfi[4 * g + 2 * kg + f2];
        xin[f2] = *px[f2];
      }
      {
        // ---- step M: 72 matrix instructions on the group's window, fragments read two steps ahead
        half8 wh[RING], wl[RING];
        floatx16 corr;
#define VCNF_READ_W(T)                                                                    \
  {                                                                                       \
    wh[(T) % RING] = __builtin_bit_cast(half8, win[((T) * 2 + 0) * 64 + lane]);           \
    wl[(T) % RING] = __builtin_bit_cast(half8, win[((T) * 2 + 1) * 64 + lane]);           \
  }
#pragma unroll
        for (int u = 0; u < VCNF_AHEAD; ++u) {
          VCNF_READ_W(u)
        }
#pragma unroll
        for (int u = 0; u < 3 * NTH; ++u) {
          const int b = u >> 3, t = u & 7;
          if (u + VCNF_AHEAD < 3 * NTH) {
            VCNF_READ_W(u + VCNF_AHEAD)
          }
          if (t == 0) corr = floatx16{};
          pa[b] = mfma32h(wh[u % RING], fhi[t], pa[b]);
          corr = mfma32h(wh[u % RING], flo[t], corr);
          corr = mfma32h(wl[u % RING], fhi[t], corr);
          if (t == NTH - 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pa[b][r] = fmaf(corr[r], kLoUnscale, pa[b][r]);
          }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * VCNF_AHEAD, 0);
#pragma unroll
        for (int u = 0; u + VCNF_AHEAD < 3 * NTH; ++u) {
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * VCNF_AHEAD, 0);
#undef VCNF_READ_W
      }
      { VCNF_T(10) VCNF_SYNC(); VCNF_T(15) }
      {
        // ---- step V: two spline evaluations per lane; the group's next window and bias travel meanwhile
        const bool more = rnd + 1 < NR;
        if (!more) {                           // last vector step of the tile: the next tile's rows are requested
          VCNF_PREFETCH_ROWS(tile + gridDim.x)
        }
        if (more && VCNF_ABL != 2) {
          VCNF_STAGE_DMA(g + 2)
        }
        float yv[2], lad[2];
#pragma unroll
        for (int f2 = 0; f2 < 2; ++f2) {
          // logits of feature f2: accumulator entries 24 f2 + 0 .. 22 (entry v = register v % 16 of block v / 16)
          float lg[P];
#pragma unroll
          for (int tl = 0; tl < P; ++tl) lg[tl] = pa[(24 * f2 + tl) >> 4][(24 * f2 + tl) & 15];
          rqs_lean_eval<8, INV>(xin[f2], lg, lc, yv[f2], lad[f2], bad);
        }
        *px[0] = yv[0];
        *px[1] = yv[1];
        ld_acc += lad[0] + lad[1];
        if (more) {
          VCNF_LOAD_BIASF(g + 2)
        }
      }
      // this wave's part of the window must have landed before the other waves of the group read it
      // (vmcnt(0); the loads were requested a whole vector step ago)
      if (rnd + 1 < NR && VCNF_ABL != 2) wait_vector_memory();
      { VCNF_T(11) VCNF_SYNC(); VCNF_T(15) }
    }
    if (ch == 0) { VCNF_T(14) VCNF_SYNC(); VCNF_T(15) }            // ---- groups re-aligned: every spline of the tile is done
#undef VCNF_PREFETCH_ROWS
#undef VCNF_STAGE_DMA
#undef VCNF_LOAD_BIASF

    // ---- per-sample log|det|: this wave covered one group parity of its samples; the partner
    // wave (other parity) adds its share through LDS (ldt already holds the identity half)
    ld_acc += __shfl_xor(ld_acc, 32, 64);
    if (ch == 1 && kg == 0) ldt[rp * 32 + c32] += ld_acc;
    if (satm > 65504.f) *tflag = 1;
    { VCNF_T(12) VCNF_SYNC(); VCNF_T(15) }
    // A tile that held a non-finite input or a value beyond the fp16 range is not written at all when the caller
    // gave a flag array: the exact fp32 kernel evaluates it from the untouched inputs (vcnf_rqs_layer_fused_f32,
    // redo_tiles).  Without the array the clamped results are stored and only counted (sat).
    const bool over = *tflag != 0;
    if (tid == 0 && over && a.sat) atomicAdd(a.sat, 1);
    if (a.redo && tid < kTile / kFusedFlagRows && b0 + tid * kFusedFlagRows < a.B)   // one flag per 32 rows
      a.redo[tile * (kTile / kFusedFlagRows) + tid] = over ? 1 : 0;
    if (over && a.redo) continue;
    if (ch == 0 && kg == 0) {
      const int mrow = rp * 32 + c32;
      if (mrow < rows) {
        const float o = a.ld_sign * (ld_acc + ldt[mrow]);
        a.logdet[b0 + mrow] = a.ld_mode ? a.logdet[b0 + mrow] + o : o;
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
    a.y[16] = (float)(clock64() - tstart);             // shader cycles of the whole kernel ...
    a.y[17] = (float)(wall_clock64() - rstart);        // ... and 100 MHz ticks: in-kernel clock = 100 MHz * y[16] / y[17]
  }
#endif
  if (INV && a.bad && bad) atomicAdd(a.bad, 1);
}

template <int DI, int DT, int C, int H, int NBLK, int K>
static int launch_v6(const FusedArgs& a, int inverse, hipStream_t st) {
  constexpr int D = DI + DT;
  constexpr int TILE = 128;
  constexpr size_t WIN = (size_t)2 * 3 * (H / 16) * 2 * 64 * 16;   // two feature groups
  const size_t lds = ((size_t)TILE * (D + 4) + ((DI * 3 * (K + 1) + 3) & ~3) + TILE + D + 8 + (DT / 4) * 96) * 4 +
                     (C > 0 ? 4 * 2 * 64 * 16 : 0) + WIN + 64;
  static bool attr_set[2] = {false, false};
  if (!attr_set[inverse ? 1 : 0]) {
    hipError_t e;
    if (inverse)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_rqs_layer_v6_kernel<DI, DT, C, H, NBLK, K, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_rqs_layer_v6_kernel<DI, DT, C, H, NBLK, K, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return VCNF_ERR_LAUNCH;
    attr_set[inverse ? 1 : 0] = true;
  }
  const long long ntiles = (a.B + TILE - 1) / TILE;
  dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256));
  if (inverse)
    hipLaunchKernelGGL((fused_rqs_layer_v6_kernel<DI, DT, C, H, NBLK, K, true>), grid, dim3(512), lds, st, a);
  else
    hipLaunchKernelGGL((fused_rqs_layer_v6_kernel<DI, DT, C, H, NBLK, K, false>), grid, dim3(512), lds, st, a);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

// Shape family of the fused fp16 split-half kernel: (d_id = d_t, ctx, residual blocks) with H = 128, 8 bins.
template <int NBLK>
static int launch_v6_family(const FusedArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
#ifdef VCNF_DEV_ONLY
  return launch_v6<32, 32, 16, 128, NBLK, 8>(a, inverse, st);
#else
  if (d_id == 32) {
    return ctx_dim == 16 ? launch_v6<32, 32, 16, 128, NBLK, 8>(a, inverse, st)
                         : launch_v6<32, 32, 0, 128, NBLK, 8>(a, inverse, st);
  }
  return ctx_dim == 16 ? launch_v6<16, 16, 16, 128, NBLK, 8>(a, inverse, st)
                       : launch_v6<16, 16, 0, 128, NBLK, 8>(a, inverse, st);
#endif
}

// One translation unit per number of residual blocks (-DVCNF_V6_NBLK=1|2|3; build.py runs them in
// parallel); fused_layer.hip dispatches to launch_fused_v6_b<N>.
#ifndef VCNF_V6_NBLK
#define VCNF_V6_NBLK 2
#endif
#if VCNF_V6_NBLK == 1
int launch_fused_v6_b1(const FusedArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
  return launch_v6_family<1>(a, d_id, ctx_dim, inverse, st);
}
#elif VCNF_V6_NBLK == 2
int launch_fused_v6_b2(const FusedArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
  return launch_v6_family<2>(a, d_id, ctx_dim, inverse, st);
}
#else
int launch_fused_v6_b3(const FusedArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
  return launch_v6_family<3>(a, d_id, ctx_dim, inverse, st);
}
#endif

}  // namespace vcnf
