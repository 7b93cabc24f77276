// Last conditioner layer + splines of an RQS coupling in one kernel, for ANY number of
// transformed features (hidden width 128, linear tails, 8, 10 (the reference's default) or 16 bins).
//
// The layer families of fused_layer*.hip run a whole coupling per launch.  Every other RQS
// coupling used the three-step path, whose cost is the conditioner's last Linear: it writes
// d_t * (3K-1) floats per sample to HBM only for the spline kernel to read them back (config C5,
// D = 1024, K = 16: 96 KB per sample and layer, and a 128 x 24064 fp32 GEMM).  Here the trunk of
// the conditioner stays on PyTorch-ROCm (its activations are [B, 128]) and this kernel takes the
// trunk output h, multiplies it with the last layer's weights on the fp16 split-half matrix path
// (fused_common.hpp: 22-bit operands, fp32 accumulation, 3 instructions per product) and
// evaluates the splines straight from the accumulators.  The logits never exist in memory.
//
// Work split - weight stationary over feature groups:
//   * a workgroup owns GW consecutive feature groups (4 features each; GW * P4 * 8 KB <= 96 KB of
//     fragments) and keeps their weights in LDS for the whole launch; blockIdx = sample block x
//     group block;
//   * its 8 waves then run free: a wave takes a 16-sample column block, reads those rows of h
//     straight from global memory into B-operand fragments (natural k order: 8 consecutive
//     hidden units per lane and k-step), and for each resident feature group issues
//     P4 * 4 * 3 matrix instructions against the LDS fragments and evaluates one spline per lane
//     (sample lane & 15, feature 4 g + (lane >> 4)), reading x and writing y in place in global
//     memory.  No barrier after the prologue, so matrix and vector work of different waves
//     overlap on their own;
//   * log|det|: a workgroup covers only its groups, so each writes a partial row
//     partial[group block][sample]; the host adds the rows (deterministic, no atomics).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"
#include "rqs_math.hpp"
#include "rqs_lean.hpp"
#include "fused_common.hpp"

namespace vcnf {

struct FinalArgs {
  const float* x;
  const float* h;
  float* y;
  float* partial;
  const int32_t* tf_idx;
  const float* wpack;          // [NG][P4][4][hi|lo][64][8 halves] as floats, then bias [NG][4][4 P4]
  long long B;
  int D, d_t, NG, gblocks, sblocks;
  int32_t* bad;
  RqsConst c;
};

constexpr int kFinBlock = 512;
constexpr int kFinH = 128;
constexpr int kFinNS = kFinH / 32;

template <int K>
struct FinalShape {
  static constexpr int P = 3 * K - 1;
  static constexpr int P4 = (P + 3) / 4;
  static constexpr int GFRAG = P4 * kFinNS * 2 * 64;               // 16-byte fragments per feature group
  static constexpr int GW = (96 * 1024) / (GFRAG * 16) >= 1 ? (96 * 1024) / (GFRAG * 16) : 1;
  static_assert(P4 % 2 == 0, "row blocks are processed in pairs");
};

// PRE: the trunk rows arrive already split (vcnf_resnet_trunk_split_f32: 128 hi halves | 128 lo halves per row), so a
// wave's B-operand fragments are its loads - the ~100 vector instructions per column block that split fp32 rows (in
// every one of the feature-group workgroups that read the same row) are gone.
template <int K, bool INV, bool PRE>
__global__ __launch_bounds__(kFinBlock, 2) void rqs_final_fused_kernel(const FinalArgs a) {
  using S = FinalShape<K>;
  constexpr int P4 = S::P4, GFRAG = S::GFRAG, GW = S::GW, NS = kFinNS;
  extern __shared__ __align__(16) uint4 wlds[];                    // [GW][GFRAG]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15;
  const int q = lane >> 4;
  const LeanConst lc = make_lean_const(a.c);
  const int gb = blockIdx.x % a.gblocks;     // neighbouring workgroups share their samples' h rows in L2
  const int sb = blockIdx.x / a.gblocks;
  const int g0 = gb * GW;

  {   // ---- this workgroup's weights: global -> LDS, once
    const uint4* src = reinterpret_cast<const uint4*>(a.wpack) + (long long)g0 * GFRAG;
    const int have = min(GW, a.NG - g0) * GFRAG;
    for (int i = tid; i < GW * GFRAG; i += kFinBlock) wlds[i] = i < have ? src[i] : make_uint4(0u, 0u, 0u, 0u);
  }
  // bias rows [group][q][4 P4] of the resident groups behind the weights in LDS; this lane's transformed
  // column of each resident group in a register: neither changes over the launch, and a global load
  // per group and column block sat exposed in front of every matrix phase
  float* blds = reinterpret_cast<float*>(wlds + GW * GFRAG);
  {
    const float* bsrc = a.wpack + (long long)a.NG * GFRAG * 4 + (long long)g0 * 16 * P4;
    const int have = min(GW, a.NG - g0) * 16 * P4;
    for (int i = tid; i < GW * 16 * P4; i += kFinBlock) blds[i] = i < have ? bsrc[i] : 0.f;
  }
  int tcol[GW];
#pragma unroll
  for (int gi = 0; gi < GW; ++gi) {
    const int f = 4 * (g0 + gi) + q;
    tcol[gi] = f < a.d_t ? a.tf_idx[f] : -1;
  }
  __syncthreads();

  bool bad = false;
  const long long ncb = (a.B + 15) / 16;
  // the trunk rows of the NEXT column block travel while the current one is processed
  const long long cstride = (long long)a.sblocks * 8;
  float4 nh[2 * NS];
  // fp32 rows: hidden units 32 s + 8 q .. + 7 are two float4; pre-split rows: the same units' hi halves are the 16 bytes
  // at half offset 32 s + 8 q, their lo halves 128 halves further
#define VCNF_FF_FETCH(CB)                                                                   \
  {                                                                                         \
    const long long r_ = (CB) * 16 + m16;                                                   \
    const float4* hr_ = reinterpret_cast<const float4*>(a.h) + (r_ < a.B ? r_ : 0) * (kFinH / 4) + (PRE ? q : 2 * q); \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                        \
      nh[2 * s] = hr_[PRE ? 4 * s : 8 * s];                                                 \
      nh[2 * s + 1] = hr_[PRE ? 4 * s + kFinH / 8 : 8 * s + 1];                             \
    }                                                                                       \
  }
  if ((long long)sb * 8 + wave < ncb) {
    VCNF_FF_FETCH((long long)sb * 8 + wave)
  }
  for (long long cb = (long long)sb * 8 + wave; cb < ncb; cb += cstride) {
    const long long row = cb * 16 + m16;
    const bool valid = row < a.B;
    // ---- this lane's B-operand fragments: hidden units 32 s + 8 q .. + 7 of sample `row`, split hi/lo
    half8 fhi[NS], flo[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if constexpr (PRE) {
        fhi[s] = __builtin_bit_cast(half8, nh[2 * s]);
        flo[s] = __builtin_bit_cast(half8, nh[2 * s + 1]);
      } else {
        const float4 v0 = nh[2 * s], v1 = nh[2 * s + 1];
        half4 h0, l0, h1, l1;
        split4<false>(floatx4{v0.x, v0.y, v0.z, v0.w}, h0, l0);
        split4<false>(floatx4{v1.x, v1.y, v1.z, v1.w}, h1, l1);
        fhi[s] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        flo[s] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
    if (cb + cstride < ncb) {
      VCNF_FF_FETCH(cb + cstride)
    }
    float ld = 0.f;
#pragma unroll
    for (int gi = 0; gi < GW; ++gi) {
      const int g = g0 + gi;
      if (g < a.NG) {
        const uint4* win = wlds + gi * GFRAG;
        floatx4 pa[P4];
#pragma unroll
        for (int b = 0; b < P4; ++b)
          pa[b] = *reinterpret_cast<const floatx4*>(blds + (gi * 4 + q) * (4 * P4) + 4 * b);
        // this lane's input element of the group: requested before the matrix work
        const bool live = valid && tcol[gi] >= 0;
        const long long at = live ? row * a.D + tcol[gi] : 0;
        const float xv = live ? a.x[at] : 0.f;
        // two row blocks advance together (six independent accumulator updates per k-step); the four
        // weight fragments of step u + 1 are read from LDS before the matrix instructions of step u
        {
          constexpr int NSTEP = (P4 / 2) * NS;
          half8 whi[2][2], wlo[2][2];          // [slot][row block of the pair]
          floatx4 mainv[2], corr[2];
#define VCNF_FF_READ(U)                                                                      \
  _Pragma("unroll") for (int bb = 0; bb < 2; ++bb) {                                         \
    whi[(U) & 1][bb] = __builtin_bit_cast(half8, win[(((2 * ((U) / NS) + bb) * NS + ((U) % NS)) * 2 + 0) * 64 + lane]); \
    wlo[(U) & 1][bb] = __builtin_bit_cast(half8, win[(((2 * ((U) / NS) + bb) * NS + ((U) % NS)) * 2 + 1) * 64 + lane]); \
  }
          VCNF_FF_READ(0)
#pragma unroll
          for (int u = 0; u < NSTEP; ++u) {
            const int bp = 2 * (u / NS), s = u % NS;
            if (u + 1 < NSTEP) {
              VCNF_FF_READ(u + 1)
            }
            if (s == 0) {
              mainv[0] = pa[bp];
              mainv[1] = pa[bp + 1];
              corr[0] = corr[1] = floatx4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) mainv[bb] = mfma16h(whi[u & 1][bb], fhi[s], mainv[bb]);
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) corr[bb] = mfma16h(whi[u & 1][bb], flo[s], corr[bb]);
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) corr[bb] = mfma16h(wlo[u & 1][bb], fhi[s], corr[bb]);
            if (s == NS - 1) {
#pragma unroll
              for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int r = 0; r < 4; ++r) pa[bp + bb][r] = fmaf(corr[bb][r], kLoUnscale, mainv[bb][r]);
            }
          }
#undef VCNF_FF_READ
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
          for (int u = 0; u + 1 < NSTEP; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        }
        // ---- one spline per lane: sample `row`, transformed feature 4 g + q.  The logits arrive pre-scaled
        // (vcnf_amd/fused_final.py::pack folds 1/sqrt(hidden) and the log2(e) factors into the packed rows), the
        // evaluation is rqs_lean.hpp's (exp-sum space, branch free: ~1/3 fewer vector instructions than rqs_select +
        // rqs_bin_eval, and the vector side is what this kernel waits for beside its matrix instructions)
        if (live) {
          float lg[S::P];
#pragma unroll
          for (int t = 0; t < S::P; ++t) lg[t] = pa[t >> 2][t & 3];
          float yv, lad;
          rqs_lean_eval<K, INV>(xv, lg, lc, yv, lad, bad);
          a.y[at] = yv;
          ld += lad;
        }
      }
    }
    ld += __shfl_xor(ld, 16, 64);
    ld += __shfl_xor(ld, 32, 64);
    if (q == 0 && valid) a.partial[(long long)gb * a.B + row] = ld;
  }
#undef VCNF_FF_FETCH
  if (INV && a.bad && bad) atomicAdd(a.bad, 1);
}

template <int K, bool PRE>
static int launch_final_pre(const FinalArgs& a, int inverse, hipStream_t st) {
  using S = FinalShape<K>;
  const size_t lds = (size_t)S::GW * S::GFRAG * 16 + (size_t)S::GW * 16 * S::P4 * sizeof(float);
  static bool attr_set[2] = {false, false};
  if (!attr_set[inverse ? 1 : 0]) {
    hipError_t e = inverse
        ? hipFuncSetAttribute(reinterpret_cast<const void*>(&rqs_final_fused_kernel<K, true, PRE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
        : hipFuncSetAttribute(reinterpret_cast<const void*>(&rqs_final_fused_kernel<K, false, PRE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return VCNF_ERR_LAUNCH;
    attr_set[inverse ? 1 : 0] = true;
  }
  dim3 grid((unsigned)(a.gblocks * a.sblocks));
  if (inverse)
    hipLaunchKernelGGL((rqs_final_fused_kernel<K, true, PRE>), grid, dim3(kFinBlock), lds, st, a);
  else
    hipLaunchKernelGGL((rqs_final_fused_kernel<K, false, PRE>), grid, dim3(kFinBlock), lds, st, a);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

template <int K>
static int launch_final(const FinalArgs& a, int inverse, bool pre, hipStream_t st) {
  return pre ? launch_final_pre<K, true>(a, inverse, st) : launch_final_pre<K, false>(a, inverse, st);
}

static bool final_shape_ok(int d_t, int hidden, int K, int tails) {
  return d_t >= 1 && hidden == kFinH && (K == 8 || K == 10 || K == 16) && tails == VCNF_TAILS_LINEAR;
}

static int final_gw(int K) { return K == 8 ? FinalShape<8>::GW : K == 10 ? FinalShape<10>::GW : FinalShape<16>::GW; }
static int final_gfrag(int K) { return K == 8 ? FinalShape<8>::GFRAG : K == 10 ? FinalShape<10>::GFRAG : FinalShape<16>::GFRAG; }
static int final_p4(int K) { return K == 8 ? FinalShape<8>::P4 : K == 10 ? FinalShape<10>::P4 : FinalShape<16>::P4; }

}  // namespace vcnf

using namespace vcnf;

extern "C" int vcnf_rqs_final_fused_supported(int32_t d_t, int32_t hidden, int32_t num_bins, int32_t tails) {
  return final_shape_ok(d_t, hidden, num_bins, tails) ? 1 : 0;
}

/* floats of the packed last-layer buffer (weight fragments of ceil(d_t / 4) feature groups + bias rows) */
extern "C" int64_t vcnf_rqs_final_fused_pack_floats(int32_t d_t, int32_t hidden, int32_t num_bins) {
  if (!final_shape_ok(d_t, hidden, num_bins, VCNF_TAILS_LINEAR)) return 0;
  const int64_t ng = (d_t + 3) / 4;
  return ng * final_gfrag(num_bins) * 4 + ng * 4 * 4 * final_p4(num_bins);
}

/* rows of the partial log-det buffer [rows, batch] the kernel writes (one per group block) */
extern "C" int64_t vcnf_rqs_final_fused_partial_rows(int32_t d_t, int32_t num_bins) {
  if (num_bins != 8 && num_bins != 10 && num_bins != 16) return 0;
  const int ng = (d_t + 3) / 4, gw = final_gw(num_bins);
  return (ng + gw - 1) / gw;
}

static int run_final(const float* x, const float* h, float* y, float* partial, int64_t batch, int32_t features,
                     const int32_t* transform_idx, int32_t d_t, int32_t hidden, const float* wpack, int64_t wpack_floats,
                     const vcnf_rqs_cfg* cfg, int inverse, int32_t* bad_disc, bool pre, void* stream) {
  if (!cfg) return VCNF_ERR_NULL;
  if (!final_shape_ok(d_t, hidden, cfg->num_bins, cfg->tails)) return VCNF_ERR_UNSUPPORTED;
  if (batch < 0 || features < d_t) return VCNF_ERR_SHAPE;
  if (wpack_floats != vcnf_rqs_final_fused_pack_floats(d_t, hidden, cfg->num_bins)) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!x || !h || !y || !partial || !transform_idx || !wpack) return VCNF_ERR_NULL;
  if ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(wpack)) & 15) return VCNF_ERR_ALIGN;
  const int K = cfg->num_bins;
  if ((double)cfg->min_bin_width * K > 1.0 || (double)cfg->min_bin_height * K > 1.0) return VCNF_ERR_VALUE;
  FinalArgs a;
  a.x = x; a.h = h; a.y = y; a.partial = partial; a.tf_idx = transform_idx; a.wpack = wpack;
  a.B = batch; a.D = features; a.d_t = d_t; a.NG = (d_t + 3) / 4; a.bad = bad_disc;
  a.gblocks = (int)vcnf_rqs_final_fused_partial_rows(d_t, K);
  // enough sample blocks to fill the chip (one 8-wave workgroup per CU), at most one per 128 samples
  const long long tiles = (batch + 127) / 128;
  long long sblocks = (256 * 2 + a.gblocks - 1) / a.gblocks;
  if (sblocks > tiles) sblocks = tiles;
  if (sblocks < 1) sblocks = 1;
  a.sblocks = (int)sblocks;
  RqsConst& c = a.c;
  c.K = K; c.tails = cfg->tails;
  c.lo_x = cfg->left; c.hi_x = cfg->right; c.span_x = (float)((double)cfg->right - (double)cfg->left);
  c.lo_y = cfg->bottom; c.hi_y = cfg->top; c.span_y = (float)((double)cfg->top - (double)cfg->bottom);
  c.min_w = cfg->min_bin_width; c.min_h = cfg->min_bin_height; c.min_d = cfg->min_derivative;
  c.free_w = (float)(1.0 - (double)cfg->min_bin_width * K);
  c.free_h = (float)(1.0 - (double)cfg->min_bin_height * K);
  c.wh_scale = cfg->wh_scale;
  c.edge_logit = (float)log(exp(1.0 - (double)cfg->min_derivative) - 1.0);
  hipStream_t st = (hipStream_t)stream;
  return K == 8 ? launch_final<8>(a, inverse, pre, st) : K == 10 ? launch_final<10>(a, inverse, pre, st)
                                                          : launch_final<16>(a, inverse, pre, st);
}

extern "C" int vcnf_rqs_final_fused_f32(const float* x, const float* h, float* y, float* partial,
                                        int64_t batch, int32_t features, const int32_t* transform_idx, int32_t d_t,
                                        int32_t hidden, const float* wpack, int64_t wpack_floats,
                                        const vcnf_rqs_cfg* cfg, int inverse, int32_t* bad_disc, void* stream) {
  return run_final(x, h, y, partial, batch, features, transform_idx, d_t, hidden, wpack, wpack_floats, cfg, inverse, bad_disc,
                   false, stream);
}

/* the same call on trunk rows that vcnf_resnet_trunk_split_f32 delivered already split into fp16 halves */
extern "C" int vcnf_rqs_final_fused_presplit_f32(const float* x, const float* h_split, float* y, float* partial,
                                                 int64_t batch, int32_t features, const int32_t* transform_idx,
                                                 int32_t d_t, int32_t hidden, const float* wpack, int64_t wpack_floats,
                                                 const vcnf_rqs_cfg* cfg, int inverse, int32_t* bad_disc, void* stream) {
  return run_final(x, h_split, y, partial, batch, features, transform_idx, d_t, hidden, wpack, wpack_floats, cfg, inverse,
                   bad_disc, true, stream);
}
