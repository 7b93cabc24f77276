// Fused RQS coupling layer, fp16 split-half matrix path ("fp16x3"), seventh structure.
//
// Same contract as fused_layer.hip (exact fp32 path): one launch evaluates a whole
// PiecewiseRationalQuadraticCoupling layer - identity-half spline, ResidualNet conditioner, transformed-half
// splines, per-sample log|det| (reference: flows/neural_spline/coupling.py:70-125, 309-343; nets/resnet.py:60-106;
// utils/splines.py:20-193).  What changed against the fourth structure (profiles/tools/superseded/), each change
// following a measurement kept under profiles/r02_* (the sixth structure had all of them but the work split):
//   * v_mfma_f32_32x32x16_f16 instead of 16x16x32: same matrix-pipe time per flop, but an instruction holds
//     the SIMD's vector issue for 8 of 32 cycles instead of 8 of 16, so the partner wave's vector step keeps
//     ~80 % of its issue rate beside a matrix step (v_fma beside back-to-back MFMAs: 6.1 against 8.1 cycles);
//   * the transformed-half spline is evaluated in "exp-sum space" (rqs_lean.hpp): ~190 vector instructions
//     instead of ~290, no packed-f32 instructions (a vector step built from v_pk_* made NO progress beside
//     the partner's matrix instructions: the translation unit is compiled with -fno-slp-vectorize);
//   * the 1/sqrt(hidden) scale of the width / height logits, log2(e) of every exponential and of the gate
//     sigmoid are folded into the packed weights on the host (vcnf_amd/fused.py);
//   * first layer and context gates run on the split-half instruction as well (K = 48 and 16 are whole k-steps
//     of 16), their inputs split once per tile; every split saturates at +-65504 and the kernel counts
//     workgroups in which a value was clamped (FusedArgs::sat, surfaced by nf.check_saturation()).
//
// Work split: a workgroup is FOUR waves (one per SIMD) and a CU holds TWO such workgroups (<= 80 KB of LDS and
// 256 registers each).  The fourth structure ran two wave groups of one 512-thread workgroup one step apart
// ("ping-pong": one in a matrix step M while the other is in a vector step V); its steps were coupled through the
// workgroup barrier, every M / V pair lasted as long as its longer half and a third of all wave cycles were spent
// waiting at barriers (profiles/r02_fused_v6_phase_cycles.md).  Here the two groups are separate workgroups:
// barriers are group-local, the two workgroups of a CU drift freely against each other and the SIMD's matrix and
// vector pipes are shared by whatever the two resident waves happen to be doing.
// Tile = 128 samples = 4 column blocks of 32.  Trunk (two passes of 64 samples): wave = 32-row block rp with the
// layer's weights for its rows stationary in registers, activations travel through LDS as ready-made B
// fragments (hi | lo halves).  Last layer: wave = column block rp; its 32 samples' activations are stationary in
// registers and the 48 KB of weights of one group of four features stream through an LDS window (which overlays
// the activation fragments) by buffer_load ... lds.  A workgroup has no room for a resident x / y tile (80 KB)
// and 4-byte scattered stores in bursts made the kernel twice as slow (one L2 request per dword, 8192 per tile at
// once).  So rows are only touched in 16-byte chunks: at the start of a tile the threads read the rows coalesced
// and scatter the identity half into first-layer input fragments (2-byte LDS writes); in round g a lane loads the
// chunk that holds its two transformed features AND two identity features (for the alternating masks of
// wrapper.py:47 a chunk is exactly that: layout flag in FusedArgs::c via tf_idx / id_idx, checked per launch),
// transforms all four - the identity pair through the unconditional spline's LDS tables - and stores the
// chunk.  Other masks take the same path with four 4-byte accesses per lane and round (slower, same results).
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

#ifndef VCNF_ABL
#define VCNF_ABL 0
#endif
// -DVCNF_TIME=1: s_memtime stamps at every barrier; wave 0 of workgroup 0 leaves the per-phase sums in
// the first output row (timing builds only, read by profiles/tools/v7_phase_timing.py)
#ifndef VCNF_TIME
#define VCNF_TIME 0
#endif
// operand fragments are requested VCNF_AHEAD steps (of three matrix instructions) before their use
#ifndef VCNF_AHEAD
#define VCNF_AHEAD 2
#endif
// cache policy bits (aux) of the row loads at the start of a tile, the chunk loads of the rounds, the chunk stores
#ifndef VCNF_AUX_ROWS
#define VCNF_AUX_ROWS 0
#endif
#ifndef VCNF_AUX_XIN
#define VCNF_AUX_XIN 0
#endif
#ifndef VCNF_AUX_Y
#define VCNF_AUX_Y 0
#endif
// the chunks a lane produces are stored every VCNF_YFLUSH rounds, back to back
#ifndef VCNF_YFLUSH
#define VCNF_YFLUSH 1
#endif
#if VCNF_TIME
#define VCNF_T(I) { const long long t_ = clock64(); tacc[I] += t_ - tlast; tlast = t_; }
#else
#define VCNF_T(I)
#endif

// Workgroup barrier.  Not __syncthreads(): its release fence drains EVERY outstanding vector-memory operation
// (s_waitcnt vmcnt(0)) - the y stores of this round, the x chunk requested for the next one, the next layer's
// weights - in front of every barrier, i.e. it puts their whole latency into every step (the kernel ran at
// 1.4 ms instead of 0.75 with it).  What the steps exchange through the barrier is LDS data only: the wave's
// LDS operations are complete (lgkmcnt(0)), window pieces written by buffer_load ... lds are waited for
// explicitly (VCNF_WAIT_DMA) by the wave that requested them.  The inline assembly is opaque to the compiler
// (memory clobber: no access moves across it) and fenced for the instruction scheduler (matrix instructions
// are no memory operations and were otherwise sunk below the barrier that ends their step).
#define VCNF_SYNC() { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
// all but the N youngest vector-memory operations of this wave are complete (they complete in issue order)
#define VCNF_WAIT_VM(N) __builtin_amdgcn_s_waitcnt(0x0F70 | (N))

namespace vcnf {

typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ floatx16 mfma32h(half8 a, half8 b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// hi / lo halves of eight values; the running maximum of what was clamped goes to ``satm``
template <bool RELU>
__device__ __forceinline__ void split8(const float (&v)[8], half8& hi, half8& lo, float& satm) {
#pragma unroll
  for (int i = 0; i < 8; i += 2)
    satm = RELU ? fmaxf(fmaxf(satm, v[i]), v[i + 1]) : fmaxf(fmaxf(satm, __builtin_fabsf(v[i])), __builtin_fabsf(v[i + 1]));
  // pin the running maximum here: left alone the compiler sinks these updates to the end of the tile and keeps
  // (spills) every value that ever went through a split until then
  asm volatile("" : "+v"(satm));
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x = __builtin_amdgcn_fmed3f(v[i], RELU ? 0.f : -65504.f, 65504.f);
    const _Float16 hv = (_Float16)x;
    hi[i] = hv;
    lo[i] = (_Float16)((x - (float)hv) * kLoScale);
  }
}

template <int DI, int DT, int C, int H, int NBLK, int K, bool INV>
__global__ __launch_bounds__(256, 2) void fused_rqs_layer_v7_kernel(const FusedArgs a) {
  static_assert(H == 128 && K == 8, "4 row blocks of 32 over 4 waves; 3 K - 1 = 23 logits: two features per 48 rows");
  static_assert((DI == 16 || DI == 32) && DT == DI && (C == 0 || C == 16), "shape family");
  constexpr int kBlock = 256;
  constexpr int kTile = 128;
  constexpr int NCB = 4;                    // 32-sample column blocks per tile
  constexpr int D = DI + DT;
  constexpr int NTX = DI / 16;              // k-steps of the identity features in the first layer
  constexpr int NTC = C / 16;               // k-steps of the context (0 or 1)
  constexpr int NT0 = NTX + NTC;
  constexpr int NTH = H / 16;               // k-steps of a hidden->* layer (8)
  constexpr int P = 3 * K - 1;
  constexpr int NG = DT / 4;                // feature groups (4 features = 96 rows = 3 row blocks) = rounds
  constexpr int TABW = 3 * (K + 1);
  constexpr int UNR = DI / 4;               // identity features per thread and row
  using L = PackLayout6<DI, DT, C, H, NBLK, K>;
  constexpr int GFRAG = 3 * NTH * 2 * 64;   // 16-byte fragments of one feature group (48 KB)
  constexpr int RING = VCNF_AHEAD + 1;

  extern __shared__ __align__(16) float smem[];
  // fragment region first (LDS offset 0: every fragment address is a per-lane base + 16-bit immediate).
  // Trunk: activation fragments [t][cb][lane] of 16 bytes, hi (32 KB) then lo (32 KB); last layer: the weight
  // window [b][t][hi|lo][lane] of one feature group (48 KB) over the same bytes.
  uint4* act = reinterpret_cast<uint4*>(smem);
  uint4* act_hi = act;
  uint4* act_lo = act + NTH * NCB * 64;
  uint4* ctxf = act + 2 * NTH * NCB * 64;                  // [cb][hi|lo][lane] context fragments (8 KB)
  float* tab = reinterpret_cast<float*>(ctxf + (C > 0 ? NCB * 2 * 64 : 0));   // [DI][TABW]
  int* tfi = reinterpret_cast<int*>(tab + ((DI * TABW + 3) & ~3));
  int* idi = tfi + DT;
  float* biasf = reinterpret_cast<float*>(idi + DI + 4);   // [NG][lane half][48] last-layer bias (3 KB)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int rp = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave: trunk 32-row block, last-layer column block
  const int c32 = lane & 31;
  const int kg = lane >> 5;
  const RqsConst& c = a.c;
  const bool shared = a.sh_w != nullptr;
  const LeanConst lc = make_lean_const(c);

  for (int i = tid; i < NG * 96; i += kBlock) biasf[i] = a.wpack[L::BF + i];
  for (int i = tid; i < DT; i += kBlock) tfi[i] = a.tf_idx[i];
  for (int i = tid; i < DI; i += kBlock) idi[i] = a.id_idx[i];
  if (shared) {
    for (int f = tid; f < DI; f += kBlock) {
      SplitLogits p{a.sh_w + f * K, a.sh_h + f * K, a.sh_d + f * (K - 1), K, 1.f, c.edge_logit, c.tails};
      rqs_build_table(p, c, tab + f * TABW);
    }
  }
  // row chunks: 16-byte chunk i = tid + 256 k of the tile is columns 4 (tid % (D/4)) .. + 3 of row i / (D/4);
  // position of each of these columns in the identity half (or -1)
  constexpr int D4 = D / 4;
  static_assert(kBlock % D4 == 0, "a thread keeps its columns over the chunks it moves");
  int idpos[4] = {-1, -1, -1, -1};
  for (int p = 0; p < DI; ++p) {
    const int col = a.id_idx[p];
#pragma unroll
    for (int q = 0; q < 4; ++q) if (col == 4 * (tid % D4) + q) idpos[q] = p;
  }
  // chunk layout of the rounds: 1 = identity pair at dwords 0, 2 and transformed pair at 1, 3 of the lane's chunk,
  // 2 = the other way round, 0 = no such chunk (four separate 4-byte accesses)
  int chunked = 3;
  for (int j = 0; j < DT; j += 2) {
    const int i0 = a.id_idx[j], i1 = a.id_idx[j + 1], t0 = a.tf_idx[j], t1 = a.tf_idx[j + 1];
    const bool lay1 = (i0 & 3) == 0 && i1 == i0 + 2 && t0 == i0 + 1 && t1 == i0 + 3;
    const bool lay2 = (t0 & 3) == 0 && t1 == t0 + 2 && i0 == t0 + 1 && i1 == t0 + 3;
    chunked &= (lay1 ? 1 : 0) | (lay2 ? 2 : 0);
  }
  chunked = __builtin_amdgcn_readfirstlane(chunked);

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
  // x / y / context of a tile through bounds-checked buffer descriptors: rows past the batch read 0 and
  // their stores are dropped
#define VCNF_TILE_RSRC(PTR, TILE, WIDTH)                                                  \
  __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PTR) + min((TILE) * kTile, a.B) * (WIDTH), 0,   \
      (int)min((a.B - min((TILE) * kTile, a.B)) * ((WIDTH) * 4), (long long)(kTile * (WIDTH) * 4)), 0x00020000)
  // 16-byte row chunks of a tile: loads into registers (coalesced: D/4 lanes per row)
  constexpr int NCH = kTile * D4 / kBlock;
// (the scalar offset is an opaque zero: the three loads of a tile's rows must stay three loads - merged into one
// they would keep 32 registers alive through the whole tile)
#define VCNF_LOAD_ROWS(DST, RSRC)                                                         \
  {                                                                                       \
    int zero_ = 0;                                                                        \
    asm volatile("" : "+s"(zero_));                                                       \
    _Pragma("unroll") for (int k = 0; k < NCH; ++k)                                       \
      DST[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(RSRC, (tid + kBlock * k) * 16, zero_, VCNF_AUX_ROWS)); \
  }
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long b0 = tile * kTile;
    const int rows = (int)min((long long)kTile, a.B - b0);
    const __amdgpu_buffer_rsrc_t xr = VCNF_TILE_RSRC(a.x, tile, D);
    const __amdgpu_buffer_rsrc_t yr = VCNF_TILE_RSRC(a.y, tile, D);
    { VCNF_T(0) VCNF_SYNC(); VCNF_T(15) }                             // the previous tile's window is no longer read
    {
      // ---- rows (coalesced 16-byte chunks) and context rows; the identity half goes into the first layer's
      // input fragments: raw in the density direction (coupling.py:78-81), transformed by the unconditional
      // spline in the sampling direction (:110-114; the rounds transform it again for y and log|det|)
      float4 xrow[NCH], cpre[2];
      if (VCNF_ABL == 33) { for (int k = 0; k < NCH; ++k) xrow[k] = float4{0.01f * tid, 0.1f, -0.2f, 0.3f * k}; } else
      VCNF_LOAD_ROWS(xrow, xr)
      if (C > 0) {
        const __amdgpu_buffer_rsrc_t cr_ = VCNF_TILE_RSRC(a.ctx, tile, (C > 0 ? C : 4));
#pragma unroll
        for (int m = 0; m < 2; ++m)
          cpre[m] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(cr_, (tid + kBlock * m) * 16, 0, 0));
      }
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int r = (tid + kBlock * k) / D4;
        const float v4[4] = {xrow[k].x, xrow[k].y, xrow[k].z, xrow[k].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (idpos[q] < 0) continue;
          const int p = idpos[q];
          float v = v4[q];
          if (INV && shared) {
            const bool in_ = (v >= c.lo_x) && (v <= c.hi_x);
            float yv_, lad_;
            bool bad1 = false;
            rqs_point_table_inside<INV, K>(in_ ? v : c.lo_x, tab + p * TABW, yv_, lad_, bad1);
            v = in_ ? yv_ : v;
          }
          satm = fmaxf(satm, __builtin_fabsf(v));
          const float x = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
          const _Float16 hv = (_Float16)x;
          const _Float16 lv = (_Float16)((x - (float)hv) * kLoScale);
          // first-layer k = 16 t + 8 kg' + i  <->  identity feature p
          const int at = ((p >> 4) * NCB + (r >> 5)) * 64 + (r & 31) + 32 * ((p >> 3) & 1);
          reinterpret_cast<_Float16*>(act_hi + at)[p & 7] = hv;
          reinterpret_cast<_Float16*>(act_lo + at)[p & 7] = lv;
        }
        asm volatile("" : "+v"(satm));
      }
      if (C > 0) {
        // float4 i = tid + 256 m holds columns 4 (i & 3) .. + 3 of row i >> 2: k-slots (kg', 4 half .. + 3)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int i = tid + kBlock * m;
          const int r = i >> 2, part = i & 3;
          half4 h4, l4;
          const float cv[4] = {cpre[m].x, cpre[m].y, cpre[m].z, cpre[m].w};
          satm = fmaxf(fmaxf(satm, __builtin_fabsf(cv[0])), __builtin_fabsf(cv[1]));
          satm = fmaxf(fmaxf(satm, __builtin_fabsf(cv[2])), __builtin_fabsf(cv[3]));
          asm volatile("" : "+v"(satm));
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float x = __builtin_amdgcn_fmed3f(cv[q], -65504.f, 65504.f);
            const _Float16 hv = (_Float16)x;
            h4[q] = hv;
            l4[q] = (_Float16)((x - (float)hv) * kLoScale);
          }
          const int at = (((r >> 5) * 2) * 64 + (r & 31) + 32 * (part >> 1));          // uint4 index of the hi fragment entry
          reinterpret_cast<uint2*>(ctxf + at)[part & 1] = __builtin_bit_cast(uint2, h4);
          reinterpret_cast<uint2*>(ctxf + at + 64)[part & 1] = __builtin_bit_cast(uint2, l4);
        }
      }
    }
    { VCNF_T(2) VCNF_SYNC(); VCNF_T(15) }

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
    // publish: registers 8 hh .. 8 hh + 7 of a column block are the eight k-slots of k-step 2 rp + hh
#define VCNF_PUBLISH(SRC, RELU)                                                           \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                         \
    _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                    \
      float v8_[8];                                                                       \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) v8_[i] = SRC[j][8 * hh + i];          \
      half8 h8_, l8_;                                                                     \
      split8<RELU>(v8_, h8_, l8_, satm);                                                  \
      ph[((2 * rp + hh) * NCB + j) * 64] = __builtin_bit_cast(uint4, h8_);                \
      pl[((2 * rp + hh) * NCB + j) * 64] = __builtin_bit_cast(uint4, l8_);                \
    }                                                                                     \
  }
    // OUT[j] = bias + W_slice * operand(column block 2 pass + j).  Operand fragments are requested VCNF_AHEAD
    // steps (of three matrix instructions) before their use, pinned with sched_group_barrier.
#define VCNF_READ_B(T)                                                                    \
  {                                                                                       \
    rh[(T) % RING] = __builtin_bit_cast(half8, ph[(((T) & 7) * NCB + ((T) >> 3)) * 64]);  \
    rl[(T) % RING] = __builtin_bit_cast(half8, pl[(((T) & 7) * NCB + ((T) >> 3)) * 64]);  \
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
      if (tk_ == 0) {                                                                     \
        OUT[j] = abias;                                                                   \
        corr = floatx16{};                                                                \
      }                                                                                   \
      if (VCNF_ABL != 15) {                                                               \
      OUT[j] = mfma32h(ahi[tk_], rh[st_ % RING], OUT[j]);                                 \
      corr = mfma32h(ahi[tk_], rl[st_ % RING], corr);                                     \
      corr = mfma32h(alo[tk_], rh[st_ % RING], corr);                                     \
      } else { OUT[j][tk_] += (float)rh[st_ % RING][0] + (float)rl[st_ % RING][1] + (float)ahi[tk_][0] + (float)alo[tk_][1]; } \
      if (tk_ == NTH - 1) {                                                               \
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

    // ---- trunk, two passes of two column blocks (register budget: 32 rows x 64 samples of h, t and gate)
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      // this pass's fragments: column block 2 pass + j sits 64 j + 128 pass fragments behind the lane's base
      uint4* ph = act_hi + lane + 128 * pass;
      uint4* pl = act_lo + lane + 128 * pass;
      const uint4* pc = ctxf + lane + 256 * pass;
      // ---- step M0: first layer, 32 rows x 2 column blocks                               resnet.py:92-99
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
          floatx16 mainv = bias0, corr = {}, corr2 = {};
#pragma unroll
          for (int t = 0; t < NT0; ++t) {
            const half8 bh = __builtin_bit_cast(half8, t < NTX ? ph[(t * NCB + j) * 64] : pc[(j * 2 + 0) * 64]);
            const half8 bl = __builtin_bit_cast(half8, t < NTX ? pl[(t * NCB + j) * 64] : pc[(j * 2 + 1) * 64]);
            mainv = mfma32h(w0h[t], bh, mainv);
            corr = mfma32h(w0h[t], bl, corr);
            corr = mfma32h(w0l[t], bh, corr);
            corr2 = mfma32h(w0l[t], bl, corr2);          // lo * lo: three k-steps only, kept for the raw inputs
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) h[j][r] = fmaf(fmaf(corr2[r], kLoUnscale, corr[r]), kLoUnscale, mainv[r]);
        }
      }
      { VCNF_T(3) VCNF_SYNC(); VCNF_T(15) }                           // every wave has read the input fragments
      // ---- step V1: first hidden layer's weights requested, relu(h) published
      VCNF_LOAD_HIDDEN(L::BLK0 + L::WA, L::BLK0 + L::BA)
      VCNF_PUBLISH(h, true)
      { VCNF_T(4) VCNF_SYNC(); VCNF_T(15) }
#pragma unroll
      for (int blk = 0; blk < (VCNF_ABL == 5 ? 0 : NBLK); ++blk) {
        const int base = L::BLK0 + blk * L::BLK;
        floatx16 t[2];
        // ---- step M: first layer of the block                                        resnet.py:42-43
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
            const half8 bh = __builtin_bit_cast(half8, pc[(j * 2 + 0) * 64]);
            const half8 bl = __builtin_bit_cast(half8, pc[(j * 2 + 1) * 64]);
            floatx16 corr = {};
            gate[j] = mfma32h(wch, bh, gbias);
            corr = mfma32h(wch, bl, corr);
            corr = mfma32h(wcl, bh, corr);
#pragma unroll
            for (int r = 0; r < 16; ++r) gate[j][r] = fmaf(corr[r], kLoUnscale, gate[j][r]);
          }
        }
        { VCNF_T(5) VCNF_SYNC(); VCNF_T(15) }
        // ---- step V: GLU gate (the packed gate weights carry log2 e: sigmoid(g) = 1 / (1 + 2^-g')),
        // residual update, publish                                                     :49-57
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
          VCNF_PUBLISH(h, false)             // the last layer takes h itself (resnet.py:105)
        }
        { VCNF_T(7) VCNF_SYNC(); VCNF_T(15) }
      }
    }
#undef VCNF_HIDDEN_COMPUTE
#undef VCNF_READ_B
#undef VCNF_LOAD_HIDDEN
#undef VCNF_PUBLISH

    // ---- last layer + splines: wave owns column block rp for every feature group
    half8 fhi[NTH], flo[NTH];
#pragma unroll
    for (int t = 0; t < NTH; ++t) {
      fhi[t] = __builtin_bit_cast(half8, act_hi[(t * NCB + rp) * 64 + lane]);
      flo[t] = __builtin_bit_cast(half8, act_lo[(t * NCB + rp) * 64 + lane]);
    }
    float ld_acc = 0.f;
    uint4* win = act;                        // window of one feature group, 48 KB
    constexpr int NSTG = GFRAG / kBlock;     // 16-byte fragments a thread moves per staged group (12)
    static_assert(GFRAG % kBlock == 0, "whole fragments per thread");
    // fragment i = tid + 256 k of feature group G sits at float offset WF + G * 4 GFRAG + 4 i: straight from
    // global memory into the window (buffer_load ... lds: the wave's 64 lanes land at consecutive 16-byte slots)
#define VCNF_STAGE_DMA(G)                                                                 \
  _Pragma("unroll") for (int k = 0; k < NSTG; ++k)                                        \
    dma16_to_lds(wr, win + (tid & ~63) + k * 256, tid * 16, 4 * (L::WF + (G) * (4 * GFRAG) + k * 1024));
    floatx16 pa[3];
    const int fboff = kg * 192;              // bias rows of this lane half: [G][kg][48] floats
#define VCNF_LOAD_BIASF(G)                                                                \
  _Pragma("unroll") for (int b = 0; b < 3; ++b) {                                         \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                    \
      const floatx4 b4_ = *reinterpret_cast<const floatx4*>(biasf + (G) * 96 + kg * 48 + 16 * b + 4 * i_); \
      pa[b][4 * i_ + 0] = b4_[0]; pa[b][4 * i_ + 1] = b4_[1]; pa[b][4 * i_ + 2] = b4_[2]; pa[b][4 * i_ + 3] = b4_[3]; \
    }                                                                                     \
  }
    // per round this lane transforms positions 4 g + 2 kg + {0, 1} of the transformed half AND of the identity half
    // of sample 32 rp + c32: xq = {identity 0, identity 1, transformed 0, transformed 1}, byte offsets in xoq
    const int rowoff = (rp * 32 + c32) * (D * 4);
    // (the loaded chunk is only touched in the NEXT vector step: unpacking it right away would put the load's
    // whole latency - and that of every older store - into this step)
    floatx4 xc;
    int xoq[4];
#define VCNF_LOAD_XIN(G)                                                                  \
  {                                                                                       \
    const int j_ = 4 * (G) + 2 * kg;                                                      \
    xoq[0] = rowoff + 4 * idi[j_]; xoq[1] = rowoff + 4 * idi[j_ + 1];                     \
    xoq[2] = rowoff + 4 * tfi[j_]; xoq[3] = rowoff + 4 * tfi[j_ + 1];                     \
    if (VCNF_ABL == 32) { xc = floatx4{0.1f * kg, 0.01f * c32, -0.02f * c32, 0.3f}; } else \
    if (chunked) {                                                                        \
      xc = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(xr, min(xoq[0], xoq[2]), 0, VCNF_AUX_XIN)); \
    } else {                                                                              \
      _Pragma("unroll") for (int q = 0; q < 4; ++q)                                       \
        xc[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, xoq[q], 0, 0)); \
    }                                                                                     \
  }
    VCNF_LOAD_XIN(0)
    { VCNF_T(8) VCNF_SYNC(); VCNF_T(15) }                             // every wave has its operand fragments: the window may be written
    VCNF_STAGE_DMA(0)
    wait_vector_memory();
    { VCNF_T(9) VCNF_SYNC(); VCNF_T(15) }
    VCNF_LOAD_BIASF(0)
    float ykeep[VCNF_YFLUSH][4];
    int ykoff[VCNF_YFLUSH][4];
    static_assert(NG % VCNF_YFLUSH == 0, "whole flush periods");
#pragma unroll 1
    for (int g = 0; g < (VCNF_ABL == 3 ? 0 : NG); ++g) {
      {
        // ---- step M: 72 matrix instructions on the window, fragments read VCNF_AHEAD steps ahead
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
          if (VCNF_ABL != 12 && VCNF_ABL != 13 && VCNF_ABL < 16) {
          pa[b] = mfma32h(wh[u % RING], fhi[t], pa[b]);
          corr = mfma32h(wh[u % RING], flo[t], corr);
          corr = mfma32h(wl[u % RING], fhi[t], corr);
          } else if (t == 0 && VCNF_ABL < 17) { pa[b][0] += (float)wh[u % RING][0] + (float)wl[u % RING][1]; }
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
        // ---- step V: two spline evaluations per lane; the next window, bias and inputs travel meanwhile
        const bool more = g + 1 < NG;
        float yq[4], lad[2];
        // {identity 0, identity 1, transformed 0, transformed 1} of this round out of the chunk loaded a round ago
        float xq[4];
        if (chunked == 2) { xq[0] = xc[1]; xq[1] = xc[3]; xq[2] = xc[0]; xq[3] = xc[2]; }
        else if (chunked == 1) { xq[0] = xc[0]; xq[1] = xc[2]; xq[2] = xc[1]; xq[3] = xc[3]; }
        else { xq[0] = xc[0]; xq[1] = xc[1]; xq[2] = xc[2]; xq[3] = xc[3]; }
        int yo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) yo[q] = xoq[q];
        // the chunk is in registers (its wait is in front of this line: were the window pieces requested first,
        // the wait for the chunk - requests complete in order - would wait for them as well)
        asm volatile("" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]));
        __builtin_amdgcn_sched_barrier(0);
        if (more && VCNF_ABL != 2 && VCNF_ABL < 16) {
          VCNF_STAGE_DMA(g + 1)
          __builtin_amdgcn_sched_barrier(0);   // the window pieces stay the oldest requests of this step
        }
        {
          // identity pair through the unconditional spline's tables (branch-free: a point outside the interval is
          // evaluated at the left end and selected to the identity afterwards)
          float lsum = 0.f;
#pragma unroll
          for (int f2 = 0; f2 < 2; ++f2) {
            yq[f2] = xq[f2];
            if (shared && VCNF_ABL != 14) {
              const bool in_ = (xq[f2] >= c.lo_x) && (xq[f2] <= c.hi_x);
              float yv_, lad_;
              bool bad1 = false;
              rqs_point_table_inside<INV, K>(in_ ? xq[f2] : c.lo_x, tab + (4 * g + 2 * kg + f2) * TABW, yv_, lad_, bad1);
              bad = bad || (bad1 && in_);
              yq[f2] = in_ ? yv_ : xq[f2];
              lsum += in_ ? lad_ : 0.f;
            }
          }
          ld_acc += lsum;
        }
#pragma unroll
        for (int f2 = 0; f2 < 2; ++f2) {
          // logits of feature f2: accumulator entries 24 f2 + 0 .. 22 (entry v = register v % 16 of block v / 16)
          float lg[P];
#pragma unroll
          for (int tl = 0; tl < P; ++tl) lg[tl] = pa[(24 * f2 + tl) >> 4][(24 * f2 + tl) & 15];
          if (VCNF_ABL == 11 || VCNF_ABL == 13 || VCNF_ABL >= 16) { yq[2 + f2] = xq[2 + f2] + lg[0] + lg[22]; lad[f2] = lg[7] + lg[15]; }
          else rqs_lean_eval<INV>(xq[2 + f2], lg, lc, yq[2 + f2], lad[f2], bad);
        }
        if (more && VCNF_ABL < 18) {           // next round's chunk first: its vmcnt wait then does not cover this round's store
          VCNF_LOAD_XIN(g + 1)
        }
        // outputs wait in registers until VCNF_YFLUSH rounds are complete: with the alternating masks the two lane
        // halves of a row then hold 32 VCNF_YFLUSH contiguous bytes and write them back to back (a 128-byte line
        // of y written 16 bytes at a time over four rounds left L2 in pieces: 1.4 ms instead of 0.75)
        {
          const int slot = g % VCNF_YFLUSH;
#pragma unroll
          for (int u = 0; u < VCNF_YFLUSH; ++u) {
            if (u == slot) {
#pragma unroll
              for (int q = 0; q < 4; ++q) { ykeep[u][q] = yq[q]; ykoff[u][q] = yo[q]; }
            }
          }
          if (slot == VCNF_YFLUSH - 1 && VCNF_ABL != 31) {
#pragma unroll
            for (int u = 0; u < VCNF_YFLUSH; ++u) {
              if (chunked) {
                const floatx4 o4 = chunked == 1 ? floatx4{ykeep[u][0], ykeep[u][2], ykeep[u][1], ykeep[u][3]}
                                                : floatx4{ykeep[u][2], ykeep[u][0], ykeep[u][3], ykeep[u][1]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o4), yr, min(ykoff[u][0], ykoff[u][2]), 0, VCNF_AUX_Y);
              } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ykeep[u][q]), yr, ykoff[u][q], 0, 0);
              }
            }
          }
        }
        ld_acc += lad[0] + lad[1];
        if (more && VCNF_ABL < 18) {
          VCNF_LOAD_BIASF(g + 1)
        }
      }
      // this wave's part of the window must have landed before the other waves read it (vmcnt(0); the
      // loads were requested a whole vector step ago)
      if (g + 1 < NG && VCNF_ABL != 2 && VCNF_ABL < 16) {
        // younger than the window pieces: the next round's chunk and this round's store (1 + 1 or 4 + 4 requests)
        __builtin_amdgcn_sched_barrier(0);
        // younger than the window pieces: the next round's chunk (1 or 4 requests) and, in a flush round, the stores
        if (g % VCNF_YFLUSH == VCNF_YFLUSH - 1) { if (chunked) VCNF_WAIT_VM(1 + VCNF_YFLUSH); else VCNF_WAIT_VM(4 + 4 * VCNF_YFLUSH); }
        else { if (chunked) VCNF_WAIT_VM(1); else VCNF_WAIT_VM(4); }
      }
      { VCNF_T(11) VCNF_SYNC(); VCNF_T(15) }
    }
#undef VCNF_STAGE_DMA
#undef VCNF_LOAD_BIASF
#undef VCNF_LOAD_XIN

    // ---- per-sample log|det|: the two lane halves hold two pairs of features each of every group
    ld_acc += __shfl_xor(ld_acc, 32, 64);
    if (kg == 0) {
      const int mrow = rp * 32 + c32;
      if (mrow < rows) {
        const float o = a.ld_sign * ld_acc;
        a.logdet[b0 + mrow] = a.ld_mode ? a.logdet[b0 + mrow] + o : o;
      }
    }
  }
#undef VCNF_LOAD_ROWS
#undef VCNF_TILE_RSRC
#if VCNF_TIME
  VCNF_T(13)
  if (blockIdx.x == 0 && tid == 0) {
    // (the kernel's own stores of row 0 are done: same thread order not guaranteed across waves, timing builds only)
    for (int i = 0; i < 16; ++i) a.y[i] = (float)tacc[i];
    a.y[16] = (float)(clock64() - tstart);             // shader cycles of the whole kernel ...
    a.y[17] = (float)(wall_clock64() - rstart);        // ... and 100 MHz ticks: in-kernel clock = 100 MHz * y[16] / y[17]
  }
#endif
  if (INV && a.bad && bad) atomicAdd(a.bad, 1);
  if (a.sat && satm > 65504.f) atomicAdd(a.sat, 1);
}

template <int DI, int DT, int C, int H, int NBLK, int K>
static int launch_v7(const FusedArgs& a, int inverse, hipStream_t st) {
  constexpr int D = DI + DT;
  constexpr int TILE = 128;
  constexpr size_t ACT = (size_t)2 * (H / 16) * 4 * 64 * 16;       // activation fragments hi | lo (64 KB) >= one 48 KB window
  const size_t lds = ACT + (C > 0 ? 4 * 2 * 64 * 16 : 0) + (((DI * 3 * (K + 1) + 3) & ~3) + D + 4 + (DT / 4) * 96) * 4 + 64;
  static bool attr_set[2] = {false, false};
  if (!attr_set[inverse ? 1 : 0]) {
    hipError_t e;
    if (inverse)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_rqs_layer_v7_kernel<DI, DT, C, H, NBLK, K, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_rqs_layer_v7_kernel<DI, DT, C, H, NBLK, K, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return VCNF_ERR_LAUNCH;
    attr_set[inverse ? 1 : 0] = true;
  }
  const long long ntiles = (a.B + TILE - 1) / TILE;
  dim3 grid((unsigned)(ntiles < 512 ? ntiles : 512));              // two resident workgroups per CU
  if (inverse)
    hipLaunchKernelGGL((fused_rqs_layer_v7_kernel<DI, DT, C, H, NBLK, K, true>), grid, dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL((fused_rqs_layer_v7_kernel<DI, DT, C, H, NBLK, K, false>), grid, dim3(256), lds, st, a);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

// Shape family of the fused fp16 split-half kernel: (d_id = d_t, ctx, residual blocks) with H = 128, 8 bins.
template <int NBLK>
static int launch_v7_family(const FusedArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
#ifdef VCNF_DEV_ONLY
  return launch_v7<32, 32, 16, 128, NBLK, 8>(a, inverse, st);
#else
  if (d_id == 32) {
    return ctx_dim == 16 ? launch_v7<32, 32, 16, 128, NBLK, 8>(a, inverse, st)
                         : launch_v7<32, 32, 0, 128, NBLK, 8>(a, inverse, st);
  }
  return ctx_dim == 16 ? launch_v7<16, 16, 16, 128, NBLK, 8>(a, inverse, st)
                       : launch_v7<16, 16, 0, 128, NBLK, 8>(a, inverse, st);
#endif
}

// One translation unit per number of residual blocks (-DVCNF_V7_NBLK=1|2|3; build.py runs them in
// parallel); fused_layer.hip dispatches to launch_fused_v7_b<N>.
#ifndef VCNF_V7_NBLK
#define VCNF_V7_NBLK 2
#endif
#if VCNF_V7_NBLK == 1
int launch_fused_v7_b1(const FusedArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
  return launch_v7_family<1>(a, d_id, ctx_dim, inverse, st);
}
#elif VCNF_V7_NBLK == 2
int launch_fused_v7_b2(const FusedArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
  return launch_v7_family<2>(a, d_id, ctx_dim, inverse, st);
}
#else
int launch_fused_v7_b3(const FusedArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st) {
  return launch_v7_family<3>(a, d_id, ctx_dim, inverse, st);
}
#endif

}  // namespace vcnf
