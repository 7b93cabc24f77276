// Fused RQS coupling layer, fp16 split-half matrix path, SMALL-BATCH form: 32-sample tiles.
//
// Same contract, same packed weights (PackLayout6, vcnf_amd/fused.py::pack_layer_h3) and the same arithmetic as
// fused_layer_v6.hip - every dense layer accumulates the same matrix instructions in the same order, so y is
// bitwise the 128-sample kernel's and log|det| differs only by the order of its per-sample sum - for the batch
// sizes the reference's own drivers use (/root/reference/run.py:45-47: 1024 - 2048 samples; reference path:
// flows/neural_spline/coupling.py:70-125, 309-343; nets/resnet.py:60-106; utils/splines.py:20-193).  At 2048
// samples the 128-sample kernel occupies 16 of the 256 compute units for the time of a whole tile (~40 us per
// layer); here a tile is one 32-sample column block of v_mfma_f32_32x32x16_f16, so the same batch spreads over
// 64 workgroups that each take a quarter of the work and none of the wave-group choreography:
//   * 512 threads; the trunk runs on waves 0-3 (wave = 32-row block, weights for its rows stationary in
//     registers, requested one layer ahead), activations ping-pong between two LDS fragment buffers so that a
//     layer costs ONE barrier;
//   * last layer + splines: wave = feature group (4 features = 96 rows), the tile's activations stationary in
//     registers, the group's 48 KB of weights straight from L2 into a register ring eight fragments deep (no
//     LDS window: every fragment is used once per workgroup);
//   * identity half: 16 lanes per sample.
// Range safety as in the 128-sample kernel, per 32 samples: a tile that held a non-finite input or a value the
// fp16 halves cannot carry writes nothing and sets its flag (FusedArgs::redo, one entry per 32 samples).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"
#include "rqs_math.hpp"
#include "fused_common.hpp"
#include "rqs_lean.hpp"
#include "split_half.hpp"

// workgroup barrier that waits for LDS traffic only (see fused_layer_v6.hip)
#define VCNF_SYNC() { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

// -DVCNF_TIME=1 (timing builds only, profiles/tools/v6s_phase_timing.py): waves 0 and 4 of workgroup 0 stamp the
// 100 MHz wall clock and the shader clock at every phase boundary and leave them in the first output rows
#ifndef VCNF_TIME
#define VCNF_TIME 0
#endif
#if VCNF_TIME
#define VCNF_TS(I) { tsw[I] = wall_clock64(); tsc[I] = clock64(); }
#else
#define VCNF_TS(I) {}
#endif

namespace vcnf {

// hi / lo halves of one value, the arithmetic of split8 (split_half.hpp)
template <bool RELU>
__device__ __forceinline__ void split1(float v, _Float16& hi, _Float16& lo) {
  const float x = __builtin_amdgcn_fmed3f(v, RELU ? 0.f : -65504.f, 65504.f);
  hi = (_Float16)x;
  lo = (_Float16)__builtin_fmaf((float)hi, -kLoScale, x * kLoScale);
}

template <int DI, int DT, int C, int H, int NBLK, int K, bool INV>
__global__ __launch_bounds__(512, 2) void fused_rqs_layer_v6s_kernel(const FusedStackArgs sa) {
  const FusedArgs& a = sa.a;
  static_assert(H == 128 && K == 8, "4 row blocks of 32 over 4 waves; 3 K - 1 = 23 logits: two features per 48 rows");
  static_assert((DI == 16 || DI == 32) && DT == DI && (C == 0 || C == 16), "shape family");
  constexpr int kBlock = 512;
  constexpr int kTile = kFusedFlagRows;     // 32 samples = one column block
  constexpr int D = DI + DT;
  constexpr int XS = D + 4;
  constexpr int NTX = DI / 16;              // k-steps of the identity features in the first layer
  constexpr int NTC = C / 16;               // k-steps of the context (0 or 1)
  constexpr int NT0 = NTX + NTC;
  constexpr int NTH = H / 16;               // k-steps of a hidden->* layer (8)
  constexpr int P = 3 * K - 1;
  constexpr int NG = DT / 4;                // feature groups (4 features = 96 rows = 3 row blocks)
  constexpr int TABW = 3 * (K + 1);
  constexpr int UNR = DI / 16;              // identity features per lane (16 lanes per sample)
  using L = PackLayout6<DI, DT, C, H, NBLK, K>;
  constexpr int GFRAG = 3 * NTH * 2 * 64;   // 16-byte fragments of one feature group (48 KB)
  constexpr int RING = 8;                   // hi | lo fragment pairs in flight per wave in the last layer
  constexpr int ABUF = NTH * 64;            // uint4 entries of one half (hi or lo) of one activation buffer

  extern __shared__ __align__(16) float smem[];
  // activation fragments [buffer][hi | lo][t][lane] of 16 bytes
  uint4* act = reinterpret_cast<uint4*>(smem);
  float* xt = smem + 4 * ABUF * 4;                         // [32][XS]  x in, y out (in place)
  uint4* ctxf = reinterpret_cast<uint4*>(xt + kTile * XS); // [hi | lo][lane] context fragment (2 KB)
  float* ldt = reinterpret_cast<float*>(ctxf + (C > 0 ? 2 * 64 : 0));   // [32] identity-half log|det|
  float* ldx = ldt + kTile;                                // [8][32] per-wave shares of the transformed half
  int* tflag = reinterpret_cast<int*>(ldx + 8 * kTile);
  // per-layer tables, two sets (the set of the next layer is written while this layer runs):
  //   tab [DI][TABW] knot tables of the identity half | tfi [DT] | idi [DI + 4] | biasf [NG][lane half][48]
  constexpr int TABF = (DI * TABW + 3) & ~3;
  constexpr int TSET = TABF + DT + DI + 4 + NG * 96;
  float* tsets = reinterpret_cast<float*>(tflag + 4);
#define VCNF_ACT_HI(BUF) (act + (BUF) * 2 * ABUF)
#define VCNF_ACT_LO(BUF) (act + (BUF) * 2 * ABUF + ABUF)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c32 = lane & 31;
  const int kg = lane >> 5;
  const RqsConst& c = a.c;
  const bool shared = sa.lay[0].sh_w != nullptr;     // all layers of a stack or none
  LeanConst lc = make_lean_const(c);
  {
    // wave-uniform: keep them in scalar registers (as vector registers they were spilled around the last layer and
    // reloaded from scratch inside the latency-bound phases)
    float* f = &lc.lo_x;
#pragma unroll
    for (int i = 0; i < 12; ++i)
      f[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, f[i])));
  }
  const bool trunk = wave < 4;
  const int rp = wave & 3;                  // trunk: 32-row block
#if VCNF_TIME
  long long tsw[16], tsc[16];
  for (int i = 0; i < 16; ++i) { tsw[i] = 0; tsc[i] = 0; }
  VCNF_TS(0)
#endif

  const int voff = lane * 16;
  const int boff = kg * 64;                 // bias rows of this lane half: [nb][kg][16] floats
  const long long ntiles = (a.B + kTile - 1) / kTile;

  const int nlay = sa.n_layers;
  // ---- the first tile's rows are requested before anything else
  constexpr int D4 = D / 4;
  float4 xpre = make_float4(0.f, 0.f, 0.f, 0.f), cpre = make_float4(0.f, 0.f, 0.f, 0.f);
#define VCNF_PREFETCH_ROWS(TILE)                                                          \
  {                                                                                       \
    const long long pb0 = (TILE) * kTile;                                                 \
    const int prow = (int)min((long long)kTile, a.B - pb0);                               \
    if (tid < kTile * D4) {                                                               \
      const __amdgpu_buffer_rsrc_t xr_ = __builtin_amdgcn_make_buffer_rsrc(               \
          const_cast<float*>(a.x) + pb0 * D, 0, prow * D * 4, 0x00020000);                \
      xpre = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr_, tid * 16, 0, 0)); \
    }                                                                                     \
    if (C > 0 && tid < kTile * (C / 4)) {                                                 \
      const __amdgpu_buffer_rsrc_t cr_ = __builtin_amdgcn_make_buffer_rsrc(               \
          const_cast<float*>(a.ctx) + pb0 * C, 0, prow * C * 4, 0x00020000);              \
      cpre = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(cr_, tid * 16, 0, 0)); \
    }                                                                                     \
  }
  static_assert(kTile * D4 <= kBlock, "one 16-byte piece of the tile per thread");
  VCNF_PREFETCH_ROWS((long long)blockIdx.x)
  __builtin_amdgcn_sched_barrier(0);

  // ---- per-layer tables (index vectors, last-layer bias, knot tables of the identity half), built by the waves that
  // sit out the trunk, in two steps so that no barrier waits for memory: ISSUE requests a layer's values into
  // registers, COMMIT (a few barriers later, when they have long arrived) writes table set SET.  The set of layer
  // l + 1 is built while layer l runs.
  constexpr int NSET = (NG * 96 + 255) / 256;
  static_assert(3 * DI <= 256 && DT + DI <= 256, "one table column / index entry per thread of waves 4-7");
#define VCNF_SETUP_ISSUE(LAY)                                                             \
  {                                                                                       \
    const int u_ = tid - 256;                                                             \
    _Pragma("unroll") for (int k = 0; k < NSET; ++k)                                      \
      bset[k] = u_ + 256 * k < NG * 96 ? (LAY).wpack[L::BF + u_ + 256 * k] : 0.f;         \
    iset = u_ < DT ? (LAY).tf_idx[u_] : (u_ < DT + DI ? (LAY).id_idx[u_ - DT] : 0);       \
    if (shared && u_ < 3 * DI) {                                                          \
      const int f_ = u_ % DI;                                                             \
      SplitLogits p_{(LAY).sh_w + f_ * K, (LAY).sh_h + f_ * K, (LAY).sh_d + f_ * (K - 1), K, 1.f, c.edge_logit, c.tails}; \
      rqs_table_column_load<K>(p_, u_ / DI, tcol);                                        \
    }                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  }
#define VCNF_SETUP_COMMIT(SET)                                                            \
  {                                                                                       \
    float* ts_ = tsets + (SET) * TSET;                                                    \
    const int u_ = tid - 256;                                                             \
    _Pragma("unroll") for (int k = 0; k < NSET; ++k)                                      \
      if (u_ + 256 * k < NG * 96) (ts_ + TABF + DT + DI + 4)[u_ + 256 * k] = bset[k];     \
    if (u_ < DT + DI) reinterpret_cast<int*>(ts_ + TABF)[u_] = iset;                      \
    if (shared && u_ < 3 * DI) rqs_table_column_build<K>(tcol, 1.f, c, ts_ + (u_ % DI) * TABW, 1, u_ / DI); \
  }
  if (!trunk) {
    float bset[NSET], tcol[K + 1];
    int iset;
    VCNF_SETUP_ISSUE(sa.lay[0])
    VCNF_SETUP_COMMIT(0)
  }

  bool bad = false;
  int step = 0;                             // layers done so far: table set of the current layer = step & 1
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long b0 = tile * kTile;
    const int rows = (int)min((long long)kTile, a.B - b0);
    float satm = 0.f;
    float ldsum = 0.f;                      // threads 0-31: the sample's log|det| over the layers done so far
   for (int l = 0; l < nlay; ++l, ++step) {
    const FusedLayerDesc& lay = sa.lay[l];
    const FusedLayerDesc& nxt = sa.lay[l + 1 < nlay ? l + 1 : 0];
    const bool first = l == 0;                                    // the tile's rows come from memory
    const bool more = l + 1 < nlay || tile + gridDim.x < ntiles;   // another layer follows in this workgroup
    const __amdgpu_buffer_rsrc_t wr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lay.wpack), 0, a.wpack_bytes, 0x00020000);
    float* tab = tsets + (step & 1) * TSET;
    int* tfi = reinterpret_cast<int*>(tab + TABF);
    int* idi = tfi + DT;
    float* biasf = reinterpret_cast<float*>(idi + DI + 4);
    float ld_acc = 0.f;
    // The phases before the first layer (three barriers), in two pieces: _a stages the tile's rows (its wait for the
    // prefetched rows would also wait for anything requested after them), then each wave kind requests what it needs
    // first - the trunk waves their first two layers' weights, the others their feature group's first fragments -
    // and _b (identity half) runs while those travel.  (Values defined in one `if (trunk)` block and used in a later
    // one were spilled at every join; inside ONE region per wave kind they stay in registers.)
    auto prologue_a = [&]() {
    VCNF_TS(1)
    VCNF_SYNC();
    VCNF_TS(2)
    {   // ---- x rows -> LDS tile (rows past the batch were read as 0); context row quarter -> half of a B fragment
      if (tid == 0) *tflag = 0;
      if (tid < kTile * D4) {
        const int r = tid / D4, o = tid - r * D4;
        *reinterpret_cast<float4*>(xt + r * XS + 4 * o) = xpre;
      }
      if (C > 0 && tid < kTile * (C / 4)) {
        // thread holds context columns 4 (tid & 3) .. + 3 of row tid >> 2: k-slots (kg', 4 half .. + 3)
        const int r = tid >> 2, part = tid & 3;
        half4 h4, l4;
        const float cv[4] = {cpre.x, cpre.y, cpre.z, cpre.w};
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
        const int at = r + 32 * (part >> 1);                  // uint4 index of the hi fragment entry
        uint2* dh = reinterpret_cast<uint2*>(ctxf + at) + (part & 1);
        uint2* dl = reinterpret_cast<uint2*>(ctxf + at + 64) + (part & 1);
        *dh = __builtin_bit_cast(uint2, h4);
        *dl = __builtin_bit_cast(uint2, l4);
      }
    }
    VCNF_TS(3)
    };
    auto prologue_b = [&]() {
    VCNF_SYNC();
    VCNF_TS(4)

    // ---- identity half through the unconditional spline: 16 lanes per sample, each lane a run of DI/16
    // features (branch-free).  The lane writes its part of the first layer's B fragment: raw values in the
    // density direction (coupling.py:78-81), transformed ones in the sampling direction (:110-114).
    {
      const int mi = tid >> 4, part = tid & 15;
      float lsum = 0.f;
      _Float16* fh = reinterpret_cast<_Float16*>(VCNF_ACT_HI(0));
      _Float16* fl = reinterpret_cast<_Float16*>(VCNF_ACT_LO(0));
#pragma unroll
      for (int k = 0; k < UNR; ++k) {
        const int f = part * UNR + k;
        float* px = xt + mi * XS + idi[f];
        const float xv = *px;
        float fv = xv;
        if (shared) {
          const bool in_ = (xv >= c.lo_x) && (xv <= c.hi_x);
          float yv, lad;
          bool bad1 = false;
          rqs_point_table_inside<INV, K>(in_ ? xv : c.lo_x, tab + f * TABW, yv, lad, bad1);
          bad = bad || (bad1 && in_);
          const float o = in_ ? yv : xv;
          *px = o;
          if (INV) fv = o;
          lsum += in_ ? lad : 0.f;
        }
        // raw inputs of the conditioner: NaN / Inf count as out of range
        satm = fmaxf(fmaxf(satm, __builtin_fabsf(fv)), fv == fv ? 0.f : __builtin_inff());
        asm volatile("" : "+v"(satm));
        _Float16 hv, lv;
        split1<false>(fv, hv, lv);
        // first-layer k = 16 t + 8 kg' + i  <->  identity feature f
        const int at = ((f >> 4) * 64 + mi + 32 * ((f >> 3) & 1)) * 8 + (f & 7);
        fh[at] = hv;
        fl[at] = lv;
      }
      lsum += __shfl_xor(lsum, 1, 64);
      lsum += __shfl_xor(lsum, 2, 64);
      lsum += __shfl_xor(lsum, 4, 64);
      lsum += __shfl_xor(lsum, 8, 64);
      if (part == 0) ldt[mi] = lsum;
    }
    VCNF_TS(5)
    VCNF_SYNC();
    VCNF_TS(6)
    };

    // ---- last layer + splines: wave = feature group g, the tile's activations stationary in registers; wh / wl hold
    // the group's first fragments (requested by the caller before its last barrier), the rest follows RING steps ahead
#define VCNF_READ_W(WH, WL, RNG, G, U)                                                    \
  {                                                                                       \
    WH[(U) % (RNG)] = __builtin_bit_cast(half8, wload(wr, voff, 4 * (L::WF + (G) * (4 * GFRAG)) + 4 * (((U) * 2 + 0) * 256))); \
    WL[(U) % (RNG)] = __builtin_bit_cast(half8, wload(wr, voff, 4 * (L::WF + (G) * (4 * GFRAG)) + 4 * (((U) * 2 + 1) * 256))); \
  }
#define VCNF_RING_FILL(WH, WL, RNG, G)                                                    \
  {                                                                                       \
    _Pragma("unroll") for (int u_ = 0; u_ < (RNG); ++u_) VCNF_READ_W(WH, WL, RNG, G, u_)  \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  }
    auto last_layer = [&](auto& wh, auto& wl, const int g) {
      constexpr int RING = (int)(sizeof(wh) / sizeof(wh[0]));
      VCNF_TS(10)
      half8 fhi[NTH], flo[NTH];
#pragma unroll
      for (int t = 0; t < NTH; ++t) {
        fhi[t] = __builtin_bit_cast(half8, VCNF_ACT_HI(1)[t * 64 + lane]);
        flo[t] = __builtin_bit_cast(half8, VCNF_ACT_LO(1)[t * 64 + lane]);
      }
      floatx16 pa[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
#pragma unroll
        for (int i_ = 0; i_ < 4; ++i_) {
          const floatx4 b4_ = *reinterpret_cast<const floatx4*>(biasf + g * 96 + kg * 48 + 16 * b + 4 * i_);
          pa[b][4 * i_ + 0] = b4_[0]; pa[b][4 * i_ + 1] = b4_[1]; pa[b][4 * i_ + 2] = b4_[2]; pa[b][4 * i_ + 3] = b4_[3];
        }
      }
      // the two elements this lane transforms (features 4 g + 2 kg + {0, 1} of sample c32)
      float* px[2];
      float xin[2];
#pragma unroll
      for (int f2 = 0; f2 < 2; ++f2) {
        px[f2] = xt + c32 * XS + tfi[4 * g + 2 * kg + f2];
        xin[f2] = *px[f2];
      }
      {
        floatx16 corr;
#pragma unroll
        for (int u = 0; u < 3 * NTH; ++u) {
          const int b = u >> 3, t = u & 7;
          if (t == 0) corr = floatx16{};
          pa[b] = mfma32h(wh[u % RING], fhi[t], pa[b]);
          corr = mfma32h(wh[u % RING], flo[t], corr);
          corr = mfma32h(wl[u % RING], fhi[t], corr);
          if (t == NTH - 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pa[b][r] = fmaf(corr[r], kLoUnscale, pa[b][r]);
          }
          // keep the written order: a fragment pair is requested RING steps (of 3 matrix instructions) before its
          // use - left alone the scheduler sinks every request to just above its use (two in flight, L2 latency
          // per step)
          __builtin_amdgcn_sched_barrier(0);
          if (u + RING < 3 * NTH) {
            VCNF_READ_W(wh, wl, RING, g, u + RING)
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      VCNF_TS(11)
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
    };
    static_assert(NG <= 8, "one feature group per wave");
    const int gw = (NG == 8 || wave < NG) ? wave : 0;     // waves without a group (d_t = 16: waves 4-7) request group 0's fragments and drop them

    // stationary weights of a hidden->hidden layer for this wave's 32 rows, bias in accumulator order
    // two register sets: a layer's weights are requested BEFORE the previous layer's matrix instructions are issued
    // (a step of the trunk moves 26 KB per wave through the compute unit's 64 bytes / clock: 0.7 us, as long as the
    // step's own matrix and publish work; requested behind it, the two add up - 1.45 us per step measured)
    half8 ahi[2][NTH], alo[2][NTH];
    floatx16 abias;
#define VCNF_LOAD_BIAS16(DST, FOFF)                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                      \
    const floatx4 b4_ = wload(wr, boff, 4 * ((FOFF) + 32 * rp) + 16 * i_);                \
    DST[4 * i_ + 0] = b4_[0]; DST[4 * i_ + 1] = b4_[1]; DST[4 * i_ + 2] = b4_[2]; DST[4 * i_ + 3] = b4_[3]; \
  }
#define VCNF_LOAD_HIDDEN(SET, WOFF)                                                       \
  {                                                                                       \
    _Pragma("unroll") for (int t = 0; t < NTH; ++t) {                                     \
      ahi[SET][t] = __builtin_bit_cast(half8, wload(wr, voff, 4 * ((WOFF) + ((rp * NTH + t) * 2 + 0) * 256))); \
      alo[SET][t] = __builtin_bit_cast(half8, wload(wr, voff, 4 * ((WOFF) + ((rp * NTH + t) * 2 + 1) * 256))); \
    }                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  }
    // publish into buffer BUF: registers 8 hh .. 8 hh + 7 are the eight k-slots of k-step 2 rp + hh
#define VCNF_PUBLISH(SRC, RELU, BUF)                                                      \
  _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                      \
    float v8_[8];                                                                         \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) v8_[i] = SRC[8 * hh + i];               \
    half8 h8_, l8_;                                                                       \
    split8<RELU>(v8_, h8_, l8_, satm);                                                    \
    VCNF_ACT_HI(BUF)[(2 * rp + hh) * 64 + lane] = __builtin_bit_cast(uint4, h8_);         \
    VCNF_ACT_LO(BUF)[(2 * rp + hh) * 64 + lane] = __builtin_bit_cast(uint4, l8_);         \
  }
    // OUT = bias + W_slice(register set SET) * operand(buffer BUF)
#define VCNF_HIDDEN_COMPUTE(OUT, BUF, SET)                                                \
  {                                                                                       \
    /* operand fragments are read AHEAD k-steps before their use (ring, pinned with sched_group_barrier: left */ \
    /* alone every k-step waits for its own two LDS reads - 1 us per layer measured) */     \
    constexpr int AH_ = 2, RB_ = AH_ + 1;                                                 \
    half8 rh[RB_], rl[RB_];                                                               \
    floatx16 corr = {};                                                                   \
    OUT = abias;                                                                          \
    _Pragma("unroll") for (int tk_ = 0; tk_ < AH_; ++tk_) {                               \
      rh[tk_ % RB_] = __builtin_bit_cast(half8, VCNF_ACT_HI(BUF)[tk_ * 64 + lane]);       \
      rl[tk_ % RB_] = __builtin_bit_cast(half8, VCNF_ACT_LO(BUF)[tk_ * 64 + lane]);       \
    }                                                                                     \
    _Pragma("unroll") for (int tk_ = 0; tk_ < NTH; ++tk_) {                               \
      if (tk_ + AH_ < NTH) {                                                              \
        rh[(tk_ + AH_) % RB_] = __builtin_bit_cast(half8, VCNF_ACT_HI(BUF)[(tk_ + AH_) * 64 + lane]); \
        rl[(tk_ + AH_) % RB_] = __builtin_bit_cast(half8, VCNF_ACT_LO(BUF)[(tk_ + AH_) * 64 + lane]); \
      }                                                                                   \
      OUT = mfma32h(ahi[SET][tk_], rh[tk_ % RB_], OUT);                                   \
      corr = mfma32h(ahi[SET][tk_], rl[tk_ % RB_], corr);                                 \
      corr = mfma32h(alo[SET][tk_], rh[tk_ % RB_], corr);                                 \
    }                                                                                     \
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * AH_, 0);                              \
    _Pragma("unroll") for (int tk_ = 0; tk_ + AH_ < NTH; ++tk_) {                         \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                  \
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                  \
    }                                                                                     \
    __builtin_amdgcn_sched_group_barrier(0x008, 3 * AH_, 0);                              \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) OUT[r] = fmaf(corr[r], kLoUnscale, OUT[r]); \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  }

    // ---- conditioner trunk on waves 0-3; the other waves pass the same number of barriers (a barrier counts
    // arrivals, not program addresses).  One region without joins: the weights requested a layer ahead stay in
    // registers across the barriers (as separate `if (trunk)` blocks per step they were spilled at every join).
    if (trunk) {
      // ---- first layer (operand in buffer 0), relu(h) -> buffer 1                      resnet.py:92-99
      floatx16 h;
      if (first) prologue_a();
      {
        half8 w0h[NT0], w0l[NT0];
        floatx16 bias0;
#pragma unroll
        for (int t = 0; t < NT0; ++t) {
          w0h[t] = __builtin_bit_cast(half8, wload(wr, voff, 4 * (L::W0 + ((rp * NT0 + t) * 2 + 0) * 256)));
          w0l[t] = __builtin_bit_cast(half8, wload(wr, voff, 4 * (L::W0 + ((rp * NT0 + t) * 2 + 1) * 256)));
        }
        VCNF_LOAD_BIAS16(bias0, L::B0)
        VCNF_LOAD_BIAS16(abias, L::BLK0 + L::BA)
        VCNF_LOAD_HIDDEN(0, L::BLK0 + L::WA)
        prologue_b();
        floatx16 mainv = bias0, corr = {}, corr2 = {};
#pragma unroll
        for (int t = 0; t < NT0; ++t) {
          const half8 bh = __builtin_bit_cast(half8, t < NTX ? VCNF_ACT_HI(0)[t * 64 + lane] : ctxf[lane]);
          const half8 bl = __builtin_bit_cast(half8, t < NTX ? VCNF_ACT_LO(0)[t * 64 + lane] : ctxf[64 + lane]);
          mainv = mfma32h(w0h[t], bh, mainv);
          corr = mfma32h(w0h[t], bl, corr);
          corr = mfma32h(w0l[t], bh, corr);
          corr2 = mfma32h(w0l[t], bl, corr2);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) h[r] = fmaf(fmaf(corr2[r], kLoUnscale, corr[r]), kLoUnscale, mainv[r]);
        VCNF_PUBLISH(h, true, 1)
      }
      VCNF_TS(7)
      VCNF_SYNC();
#pragma unroll
      for (int blk = 0; blk < NBLK; ++blk) {
        const int base = L::BLK0 + blk * L::BLK;
        floatx16 t;
        half8 wch, wcl;
        floatx16 gbias;
        // ---- first layer of the block (operand: relu(h) in buffer 1), relu(t) -> buffer 0   resnet.py:42-46
        VCNF_LOAD_HIDDEN(1, base + L::WB)
        VCNF_HIDDEN_COMPUTE(t, 1, 0)
        VCNF_LOAD_BIAS16(abias, base + L::BB)
        if (C > 0) {
          wch = __builtin_bit_cast(half8, wload(wr, voff, 4 * (base + L::WC + (rp * 2 + 0) * 256)));
          wcl = __builtin_bit_cast(half8, wload(wr, voff, 4 * (base + L::WC + (rp * 2 + 1) * 256)));
          VCNF_LOAD_BIAS16(gbias, base + L::BC)
        }
        VCNF_PUBLISH(t, true, 0)
        if (blk == 0) { VCNF_TS(8) }
        VCNF_SYNC();
        // ---- second layer (:48), gate pre-activations (:53), GLU gate and residual update (:49-57); the next
        // operand (relu(h), or h itself for the last layer, resnet.py:105) -> buffer 1
        if (blk + 1 < NBLK) {
          VCNF_LOAD_HIDDEN(0, base + L::BLK + L::WA)
        }
        VCNF_HIDDEN_COMPUTE(t, 0, 1)
        if (blk + 1 < NBLK) {
          VCNF_LOAD_BIAS16(abias, base + L::BLK + L::BA)
        }
        if (C > 0) {
          const half8 bh = __builtin_bit_cast(half8, ctxf[lane]);
          const half8 bl = __builtin_bit_cast(half8, ctxf[64 + lane]);
          floatx16 gate, corr = {}, corr2 = {};
          gate = mfma32h(wch, bh, gbias);
          corr = mfma32h(wch, bl, corr);
          corr = mfma32h(wcl, bh, corr);
          corr2 = mfma32h(wcl, bl, corr2);
#pragma unroll
          for (int r = 0; r < 16; ++r) gate[r] = fmaf(fmaf(corr2[r], kLoUnscale, corr[r]), kLoUnscale, gate[r]);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float sg = hw_rcp(1.f + hw_exp2(-gate[r]));
            h[r] = fmaf(t[r], sg, h[r]);
          }
        } else {
          h += t;
        }
        if (blk + 1 < NBLK) {
          VCNF_PUBLISH(h, true, 1)
          if (blk == 0) { VCNF_TS(9) }
          VCNF_SYNC();
        }
      }
      {
        constexpr int RT = 8;
        half8 wh[RT], wl[RT];
        VCNF_RING_FILL(wh, wl, RT, gw)       // unconditional: a conditional fill is a join (spill / reload of the ring)
        VCNF_PUBLISH(h, false, 1)
        if (NBLK == 1) { VCNF_TS(9) }
        VCNF_SYNC();
        if (NG == 8 || wave < NG) last_layer(wh, wl, gw);
      }
    } else {
      // these waves wait through the trunk with a free register file: a deeper ring, filled at once
      constexpr int RW = 8;
      half8 wh[RW], wl[RW];
      float bset[NSET], tcol[K + 1];
      int iset;
      if (first) prologue_a();
      VCNF_RING_FILL(wh, wl, RW, gw)
      if (more) VCNF_SETUP_ISSUE(nxt)
      prologue_b();
      VCNF_SYNC();
#pragma unroll
      for (int blk = 0; blk < NBLK; ++blk) {
        VCNF_SYNC();
        if (blk == 0 && more) VCNF_SETUP_COMMIT((step + 1) & 1)
        VCNF_SYNC();
      }
      if (NG == 8 || wave < NG) last_layer(wh, wl, gw);
    }
#undef VCNF_READ_W
#undef VCNF_RING_FILL
#undef VCNF_HIDDEN_COMPUTE
#undef VCNF_LOAD_HIDDEN
#undef VCNF_LOAD_BIAS16
#undef VCNF_PUBLISH

    if (l + 1 == nlay && tile + gridDim.x < ntiles) {
      VCNF_PREFETCH_ROWS(tile + gridDim.x)
    }
    // ---- per-sample log|det| of this layer: identity half (ldt) + the eight waves' shares, added in a fixed order
    ld_acc += __shfl_xor(ld_acc, 32, 64);
    if (kg == 0) ldx[wave * kTile + c32] = ld_acc;
    if (satm > 65504.f) *tflag = 1;
    VCNF_TS(12)
    VCNF_SYNC();
    VCNF_TS(13)
    if (tid < kTile) {
      float v = ldt[tid];
#pragma unroll
      for (int w = 0; w < 8; ++w) v += ldx[w * kTile + tid];
      ldsum += v;
    }
   }   // layers
    // A tile that held a non-finite input or a value beyond the fp16 range is not written at all when the caller
    // gave a flag array: the exact fp32 kernel evaluates it from the untouched inputs (vcnf_rqs_layer_fused_f32,
    // redo_tiles).  Without the array the clamped results are stored and only counted (sat).
    const bool over = *tflag != 0;
    if (tid == 0) {
      if (a.redo) a.redo[tile] = over ? 1 : 0;
      if (over && a.sat) atomicAdd(a.sat, 1);
    }
    if (over && a.redo) continue;
    if (tid < rows) {
      const float o = a.ld_sign * ldsum;
      a.logdet[b0 + tid] = a.ld_mode ? a.logdet[b0 + tid] + o : o;
    }
    {
      float4* dst = reinterpret_cast<float4*>(a.y) + b0 * D4;
      for (int i = tid; i < rows * D4; i += kBlock) {
        const int r = i / D4, o = i - r * D4;
        dst[i] = *reinterpret_cast<const float4*>(xt + r * XS + 4 * o);
      }
    }
  }
#if VCNF_TIME
  VCNF_TS(14)
  if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VCNF_TS(15)
    float* o = a.y + (wave == 0 ? 0 : 32);
    for (int i = 0; i < 16; ++i) { o[i] = (float)(tsw[i] - tsw[0]); o[16 + i] = (float)(tsc[i] - tsc[0]); }
  }
#endif
  if (INV && a.bad && bad) atomicAdd(a.bad, 1);
}

template <int DI, int DT, int C, int H, int NBLK, int K>
static int launch_v6s(const FusedStackArgs& sa, int inverse, hipStream_t st) {
  const FusedArgs& a = sa.a;
  constexpr int D = DI + DT;
  constexpr int TILE = kFusedFlagRows;
  constexpr size_t TSET = ((DI * 3 * (K + 1) + 3) & ~3) + DT + DI + 4 + (DT / 4) * 96;       // one set of per-layer tables
  const size_t lds = (size_t)4 * (H / 16) * 64 * 16 + (C > 0 ? 2 * 64 * 16 : 0) +
                     ((size_t)TILE * (D + 4) + TILE + 8 * TILE + 4 + 2 * TSET) * 4 + 64;
  const long long ntiles = (a.B + TILE - 1) / TILE;
  dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256));
  if (inverse)
    hipLaunchKernelGGL((fused_rqs_layer_v6s_kernel<DI, DT, C, H, NBLK, K, true>), grid, dim3(512), lds, st, sa);
  else
    hipLaunchKernelGGL((fused_rqs_layer_v6s_kernel<DI, DT, C, H, NBLK, K, false>), grid, dim3(512), lds, st, sa);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

template <int NBLK>
static int launch_v6s_family(const FusedStackArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
  if (d_id == 32) {
    return ctx_dim == 16 ? launch_v6s<32, 32, 16, 128, NBLK, 8>(a, inverse, st)
                         : launch_v6s<32, 32, 0, 128, NBLK, 8>(a, inverse, st);
  }
  return ctx_dim == 16 ? launch_v6s<16, 16, 16, 128, NBLK, 8>(a, inverse, st)
                       : launch_v6s<16, 16, 0, 128, NBLK, 8>(a, inverse, st);
}

// One translation unit per number of residual blocks (-DVCNF_V6_NBLK=1|2|3), like fused_layer_v6.hip.
#ifndef VCNF_V6_NBLK
#define VCNF_V6_NBLK 2
#endif
#if VCNF_V6_NBLK == 1
int launch_fused_v6s_b1(const FusedStackArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
  return launch_v6s_family<1>(a, d_id, ctx_dim, inverse, st);
}
#elif VCNF_V6_NBLK == 2
int launch_fused_v6s_b2(const FusedStackArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
  return launch_v6s_family<2>(a, d_id, ctx_dim, inverse, st);
}
#else
int launch_fused_v6s_b3(const FusedStackArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
  return launch_v6s_family<3>(a, d_id, ctx_dim, inverse, st);
}
#endif

}  // namespace vcnf
