// RQS coupling kernels for MI355X (gfx950, wave64) and their C-ABI entry points.
//
// Data layout (HBM): x, y [B, D] fp32 row-major; params [B, d_t * P] fp32 row-major,
// P = 3K-1 logits per transformed feature laid out [K widths | K heights | K-1 derivs]
// (the conditioner output as the reference reshapes it, coupling.py:155, :310-312).
//
// rqs_coupling_kernel: a workgroup of 256 threads walks tiles of S = 256/G samples,
// G (power of two <= 64) lanes per sample.  Per tile it copies the x rows and the
// params rows HBM -> LDS with 16-byte coalesced loads (the params tile is one
// contiguous block of HBM), evaluates one spline per lane from LDS (stride-P reads:
// conflict free for odd P), writes results into an LDS y tile through the feature
// index maps (masks / permutations cost no HBM traffic), reduces log|det| over the
// G lanes of a sample with wave shuffles and streams the y tile back with 16-byte
// stores.  HBM traffic per sample-layer is the algorithmic minimum:
// 4D (x) + 4 d_t P (params) + 4D (y) + 4..8 (logdet).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/vcnf_hip.h"
#include "rqs_math.hpp"
#include "rqs_lean.hpp"

namespace vcnf {

constexpr int kBlock = 256;

struct CouplingArgs {
  const float* x;
  const float* params;
  const int32_t* tf_idx;
  const int32_t* id_idx;
  const float* sh_w;
  const float* sh_h;
  const float* sh_d;
  float* y;
  float* logdet;
  int32_t* bad;
  long long B;
  int D, d_t, d_id;
  int P;          // logits per transformed feature
  int Pd;         // derivative logits per feature (K-1 or K+1)
  int G, S;       // lanes per sample, samples per tile
  int JC;         // transformed features per params chunk (== d_t: whole rows)
  int vec_x;      // x / y tiles may use 16-byte accesses
  int vec_p;      // params chunks may use 16-byte accesses
  int sh_mode;    // shared (unconditional) spline: 0 absent, 1 knot tables in LDS, 2 logits read from HBM/L2
  int ld_mode;
  float ld_sign;
  RqsConst c;
};

__device__ __forceinline__ void copy_in(float* dst, const float* src, long long n, bool vec, int tid) {
  if (vec) {
    const long long n4 = n >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (long long i = tid; i < n4; i += kBlock) d4[i] = s4[i];
    for (long long i = (n4 << 2) + tid; i < n; i += kBlock) dst[i] = src[i];
  } else {
    for (long long i = tid; i < n; i += kBlock) dst[i] = src[i];
  }
}

template <int KT, bool INV>
__global__ __launch_bounds__(kBlock) void rqs_coupling_kernel(const CouplingArgs a) {
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x;
  const RqsConst& c = a.c;
  const int K = KT > 0 ? KT : c.K;
  const int tabw = 3 * (K + 1);
  // LDS carve-up (every region a multiple of 4 floats so 16-byte accesses stay aligned)
  const int n_x = a.S * a.D;
  const int n_p = ((a.S * a.JC * a.P + 3) >> 2) << 2;
  float* xt = smem;
  float* yt = xt + n_x;
  float* pt = yt + n_x;
  float* tab = pt + n_p;
  int* tfi = reinterpret_cast<int*>(tab + (((a.sh_mode == 1 ? a.d_id * tabw : 0) + 3) >> 2 << 2));
  int* idi = tfi + a.d_t;

  for (int i = tid; i < a.d_t; i += kBlock) tfi[i] = a.tf_idx[i];
  for (int i = tid; i < a.d_id; i += kBlock) idi[i] = a.id_idx[i];
  if (a.sh_mode == 1) {
    RqsConst cs = c;
    cs.K = K;
    if constexpr (KT > 0) {
      // one thread per (feature, table column), the column's logits requested before the first is used (one memory
      // latency per workgroup instead of one per loop iteration; bitwise the same table, rqs_math.hpp)
      for (int i = tid; i < 3 * a.d_id; i += kBlock) {
        const int f = i % a.d_id;
        SplitLogits p{a.sh_w + (long long)f * K, a.sh_h + (long long)f * K, a.sh_d + (long long)f * a.Pd,
                      K, 1.f, c.edge_logit, c.tails};
        rqs_build_table_part_k<(KT > 0 ? KT : 4)>(p, cs, tab + f * tabw, 1, i / a.d_id);
      }
    } else {
      for (int f = tid; f < a.d_id; f += kBlock) {
        SplitLogits p{a.sh_w + (long long)f * K, a.sh_h + (long long)f * K, a.sh_d + (long long)f * a.Pd,
                      K, 1.f, c.edge_logit, c.tails};
        rqs_build_table(p, cs, tab + f * tabw);
      }
    }
  }

  const int g = tid & (a.G - 1);
  const int s = tid / a.G;
  const long long rowlen = (long long)a.d_t * a.P;
  const long long ntiles = (a.B + a.S - 1) / a.S;
  bool bad = false;

  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long b0 = tile * a.S;
    const int rows = (int)min((long long)a.S, a.B - b0);
    __syncthreads();   // previous tile's y stores / table build done before LDS is reused
    copy_in(xt, a.x + b0 * a.D, (long long)rows * a.D, a.vec_x, tid);
    float acc = 0.f;

    for (int jc = 0; jc < a.d_t; jc += a.JC) {
      const int nj = min(a.JC, a.d_t - jc);
      if (jc > 0) __syncthreads();   // chunk buffer still being read
      if (a.JC == a.d_t) {
        copy_in(pt, a.params + b0 * rowlen, (long long)rows * rowlen, a.vec_p, tid);
      } else {
        const int seg = nj * a.P;      // floats of one row's chunk
        if (a.vec_p && (seg & 3) == 0) {
          const int seg4 = seg >> 2;
          for (int i = tid; i < rows * seg4; i += kBlock) {
            const int r = i / seg4, o = i - r * seg4;
            reinterpret_cast<float4*>(pt + r * seg)[o] =
                reinterpret_cast<const float4*>(a.params + (b0 + r) * rowlen + (long long)jc * a.P)[o];
          }
        } else {
          for (int i = tid; i < rows * seg; i += kBlock) {
            const int r = i / seg, o = i - r * seg;
            pt[r * seg + o] = a.params[(b0 + r) * rowlen + (long long)jc * a.P + o];
          }
        }
      }
      __syncthreads();
      if (s < rows) {
        const int seg = nj * a.P;
        for (int j = g; j < nj; j += a.G) {
          const int col = tfi[jc + j];
          const float xv = xt[s * a.D + col];
          PackedLogits p{pt + s * seg + j * a.P, K, c.wh_scale, c.edge_logit, c.tails};
          float yv, lad;
          rqs_point<KT, INV>(xv, p, c, yv, lad, bad);
          yt[s * a.D + col] = yv;
          acc += lad;
        }
      }
    }
    if (s < rows) {
      for (int j = g; j < a.d_id; j += a.G) {
        const int col = idi[j];
        const float xv = xt[s * a.D + col];
        float yv = xv, lad = 0.f;
        if (a.sh_mode == 1) {
          rqs_point_table<INV>(xv, tab + j * tabw, c, yv, lad, bad);
        } else if (a.sh_mode == 2) {
          SplitLogits p{a.sh_w + (long long)j * K, a.sh_h + (long long)j * K, a.sh_d + (long long)j * a.Pd,
                        K, 1.f, c.edge_logit, c.tails};
          rqs_point<KT, INV>(xv, p, c, yv, lad, bad);
        }
        yt[s * a.D + col] = yv;
        acc += lad;
      }
    }
    // per-sample log|det|: butterfly over the G lanes that share a sample
    for (int m = a.G >> 1; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (g == 0 && s < rows) {
      const float v = a.ld_sign * acc;
      a.logdet[b0 + s] = a.ld_mode ? a.logdet[b0 + s] + v : v;
    }
    __syncthreads();
    {
      const long long n = (long long)rows * a.D;
      float* dst = a.y + b0 * a.D;
      if (a.vec_x) {
        const long long n4 = n >> 2;
        for (long long i = tid; i < n4; i += kBlock)
          reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(yt)[i];
        for (long long i = (n4 << 2) + tid; i < n; i += kBlock) dst[i] = yt[i];
      } else {
        for (long long i = tid; i < n; i += kBlock) dst[i] = yt[i];
      }
    }
  }
  if (INV && a.bad && bad) atomicAdd(a.bad, 1);
}

// Fast path of the same kernel for tiles that are one contiguous, 16-byte aligned
// block of HBM (whole params rows per tile): the NEXT tile's x and params rows are
// requested into registers before the current tile is evaluated and written to LDS
// after the barrier that retires the current tile, so every workgroup keeps one
// tile (~25 KB for config C3) in flight while it computes.
template <int KT, bool INV, int NPF>
__global__ __launch_bounds__(kBlock) void rqs_coupling_pf_kernel(const CouplingArgs a) {
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x;
  const RqsConst& c = a.c;
  const int K = KT > 0 ? KT : c.K;
  const int tabw = 3 * (K + 1);
  const int n_x = a.S * a.D;
  const int n_p = ((a.S * a.d_t * a.P + 3) >> 2) << 2;
  float* xt = smem;
  float* yt = xt + n_x;
  float* pt = yt + n_x;
  float* tab = pt + n_p;
  int* tfi = reinterpret_cast<int*>(tab + (((a.sh_mode == 1 ? a.d_id * tabw : 0) + 3) >> 2 << 2));
  int* idi = tfi + a.d_t;

  for (int i = tid; i < a.d_t; i += kBlock) tfi[i] = a.tf_idx[i];
  for (int i = tid; i < a.d_id; i += kBlock) idi[i] = a.id_idx[i];
  if (a.sh_mode == 1) {
    RqsConst cs = c;
    cs.K = K;
    if constexpr (KT > 0) {
      // one thread per (feature, table column), the column's logits requested before the first is used (one memory
      // latency per workgroup instead of one per loop iteration; bitwise the same table, rqs_math.hpp)
      for (int i = tid; i < 3 * a.d_id; i += kBlock) {
        const int f = i % a.d_id;
        SplitLogits p{a.sh_w + (long long)f * K, a.sh_h + (long long)f * K, a.sh_d + (long long)f * a.Pd,
                      K, 1.f, c.edge_logit, c.tails};
        rqs_build_table_part_k<(KT > 0 ? KT : 4)>(p, cs, tab + f * tabw, 1, i / a.d_id);
      }
    } else {
      for (int f = tid; f < a.d_id; f += kBlock) {
        SplitLogits p{a.sh_w + (long long)f * K, a.sh_h + (long long)f * K, a.sh_d + (long long)f * a.Pd,
                      K, 1.f, c.edge_logit, c.tails};
        rqs_build_table(p, cs, tab + f * tabw);
      }
    }
  }

  const int g = tid & (a.G - 1);
  const int s = tid / a.G;
  const bool lean = c.tails == 1;
  const LeanConst lc = make_lean_const(c);
  const float s_wh = c.wh_scale * kLog2e;
  const int rowlen4 = (a.d_t * a.P) >> 2;     // float4 per params row (host guarantees divisibility)
  const int d4 = a.D >> 2;
  const long long ntiles = (a.B + a.S - 1) / a.S;
  bool bad = false;

  float4 rp[NPF];
  float4 rx[2];
#define VCNF_REQUEST(TILE)                                                                  \
  {                                                                                         \
    const long long rb0 = (TILE) * a.S;                                                     \
    const int rrows = (int)min((long long)a.S, a.B - rb0);                                  \
    const float4* sp = reinterpret_cast<const float4*>(a.params) + rb0 * rowlen4;           \
    const float4* sx = reinterpret_cast<const float4*>(a.x) + rb0 * d4;                     \
    const int rnp4 = rrows * rowlen4, rnx4 = rrows * d4;                                    \
    _Pragma("unroll") for (int i = 0; i < NPF; ++i) {                                       \
      const int idx = tid + i * kBlock;                                                     \
      rp[i] = idx < rnp4 ? sp[idx] : make_float4(0.f, 0.f, 0.f, 0.f);                       \
    }                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                         \
      const int idx = tid + i * kBlock;                                                     \
      rx[i] = idx < rnx4 ? sx[idx] : make_float4(0.f, 0.f, 0.f, 0.f);                       \
    }                                                                                       \
  }

  long long tile = blockIdx.x;
  if (tile < ntiles) VCNF_REQUEST(tile)
  for (; tile < ntiles; tile += gridDim.x) {
    const long long b0 = tile * a.S;
    const int rows = (int)min((long long)a.S, a.B - b0);
    const int np4 = rows * rowlen4, nx4 = rows * d4;
    __syncthreads();   // LDS tiles of the previous round fully consumed (and tables built)
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int idx = tid + i * kBlock;
      if (idx < np4) reinterpret_cast<float4*>(pt)[idx] = rp[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * kBlock;
      if (idx < nx4) reinterpret_cast<float4*>(xt)[idx] = rx[i];
    }
    if (tile + gridDim.x < ntiles) VCNF_REQUEST(tile + gridDim.x)   // stays in flight during the evaluation
    __syncthreads();

    float acc = 0.f;
    if (s < rows) {
      const int seg = a.d_t * a.P;
      for (int j = g; j < a.d_t; j += a.G) {
        const int col = tfi[j];
        const float xv = xt[s * a.D + col];
        float yv, lad;
        if constexpr (KT >= 4 && KT % 2 == 0) {
          if (lean) {
            // linear tails with a compile-time bin count: rqs_lean.hpp's evaluation (a third fewer vector instructions;
            // this kernel's time is its HBM stream plus what the vector unit does not hide behind it)
            float lg[3 * KT - 1];
            const float* q = pt + s * seg + j * a.P;
#pragma unroll
            for (int t = 0; t < 3 * KT - 1; ++t) lg[t] = q[t] * (t < 2 * KT ? s_wh : kLog2e);
            rqs_lean_eval<KT, INV>(xv, lg, lc, yv, lad, bad);
          } else {
            PackedLogits p{pt + s * seg + j * a.P, K, c.wh_scale, c.edge_logit, c.tails};
            rqs_point<KT, INV>(xv, p, c, yv, lad, bad);
          }
        } else {
          PackedLogits p{pt + s * seg + j * a.P, K, c.wh_scale, c.edge_logit, c.tails};
          rqs_point<KT, INV>(xv, p, c, yv, lad, bad);
        }
        yt[s * a.D + col] = yv;
        acc += lad;
      }
      for (int j = g; j < a.d_id; j += a.G) {
        const int col = idi[j];
        const float xv = xt[s * a.D + col];
        float yv = xv, lad = 0.f;
        if (a.sh_mode == 1) {
          rqs_point_table<INV>(xv, tab + j * tabw, c, yv, lad, bad);
        } else if (a.sh_mode == 2) {
          SplitLogits p{a.sh_w + (long long)j * K, a.sh_h + (long long)j * K, a.sh_d + (long long)j * a.Pd,
                        K, 1.f, c.edge_logit, c.tails};
          rqs_point<KT, INV>(xv, p, c, yv, lad, bad);
        }
        yt[s * a.D + col] = yv;
        acc += lad;
      }
    }
    for (int m = a.G >> 1; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (g == 0 && s < rows) {
      const float v = a.ld_sign * acc;
      a.logdet[b0 + s] = a.ld_mode ? a.logdet[b0 + s] + v : v;
    }
    __syncthreads();
    float4* dst = reinterpret_cast<float4*>(a.y) + b0 * d4;
    for (int i = tid; i < nx4; i += kBlock) dst[i] = reinterpret_cast<const float4*>(yt)[i];
  }
#undef VCNF_REQUEST
  if (INV && a.bad && bad) atomicAdd(a.bad, 1);
}

// ------------------------------------------------------------------ elementwise
struct ElemArgs {
  const float *x, *uw, *uh, *ud;
  long long ld_w, ld_h, ld_d;
  float *y, *lad;
  int32_t* bad;
  long long n;
  RqsConst c;
};

template <int KT, bool INV>
__global__ __launch_bounds__(kBlock) void rqs_elementwise_kernel(const ElemArgs a) {
  const int K = KT > 0 ? KT : a.c.K;
  bool bad = false;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (long long)gridDim.x * kBlock) {
    SplitLogits p{a.uw + i * a.ld_w, a.uh + i * a.ld_h, a.ud + i * a.ld_d, K, a.c.wh_scale, a.c.edge_logit, a.c.tails};
    float yv, lad;
    rqs_point<KT, INV>(a.x[i], p, a.c, yv, lad, bad);
    a.y[i] = yv;
    a.lad[i] = lad;
  }
  if (INV && a.bad && bad) atomicAdd(a.bad, 1);
}

// General addressing of the per-element logit rows (images, batch-shared rows):
//   r = period > 0 ? i % period : i;  outer = r / inner;  s = r % inner;
//   logit k of element i at  base + outer * row + s + k * ks   (row per tensor: row_w, row_h, row_d).
struct ElemStridedArgs {
  const float *x, *uw, *uh, *ud;
  long long row_w, row_h, row_d, inner, ks, period;
  float *y, *lad;
  int32_t* bad;
  long long n;
  RqsConst c;
};

template <int KT, bool INV>
__global__ __launch_bounds__(kBlock) void rqs_elementwise_strided_kernel(const ElemStridedArgs a) {
  const int K = KT > 0 ? KT : a.c.K;
  bool bad = false;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (long long)gridDim.x * kBlock) {
    const long long r = a.period > 0 ? i % a.period : i;
    const long long outer = r / a.inner, s = r - outer * a.inner;
    StridedLogits p{a.uw + outer * a.row_w + s, a.uh + outer * a.row_h + s, a.ud + outer * a.row_d + s, a.ks,
                    K, a.c.wh_scale, a.c.edge_logit, a.c.tails};
    float yv, lad;
    rqs_point<KT, INV>(a.x[i], p, a.c, yv, lad, bad);
    a.y[i] = yv;
    a.lad[i] = lad;
  }
  if (INV && a.bad && bad) atomicAdd(a.bad, 1);
}

template <bool INV>
static void launch_elem_strided(const ElemStridedArgs& a, int K, dim3 grid, hipStream_t st) {
  switch (K) {
    case 4: hipLaunchKernelGGL((rqs_elementwise_strided_kernel<4, INV>), grid, dim3(kBlock), 0, st, a); break;
    case 8: hipLaunchKernelGGL((rqs_elementwise_strided_kernel<8, INV>), grid, dim3(kBlock), 0, st, a); break;
    case 10: hipLaunchKernelGGL((rqs_elementwise_strided_kernel<10, INV>), grid, dim3(kBlock), 0, st, a); break;
    case 16: hipLaunchKernelGGL((rqs_elementwise_strided_kernel<16, INV>), grid, dim3(kBlock), 0, st, a); break;
    default: hipLaunchKernelGGL((rqs_elementwise_strided_kernel<0, INV>), grid, dim3(kBlock), 0, st, a); break;
  }
}

// Batch-shared per-position splines through knot tables: the K+K+nd logits of a position are turned
// into knots ONCE (build kernel, one thread per position, table row xk | yk | dk in global memory,
// L2-resident), the per-element kernel then only searches the row and evaluates the bin - no
// softmax per element (the logit form costs 2K exponentials per element: 514 us vs 60 us for the
// identity half of a C5 layer).
struct TableArgs {
  const float *x, *sw, *sh, *sd;
  float *tab, *y, *lad;
  int32_t* bad;
  long long n, period;
  int nd;
  RqsConst c;
};

__global__ __launch_bounds__(kBlock) void rqs_build_tables_kernel(const TableArgs a) {
  const int K = a.c.K;
  const long long f = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (f >= a.period) return;
  SplitLogits p{a.sw + f * K, a.sh + f * K, a.sd + f * a.nd, K, a.c.wh_scale, a.c.edge_logit, a.c.tails};
  rqs_build_table(p, a.c, a.tab + f * 3 * (K + 1));
}

template <int KT, bool INV>
__global__ __launch_bounds__(kBlock) void rqs_table_eval_kernel(const TableArgs a) {
  const int K = KT > 0 ? KT : a.c.K;
  const int tabw = 3 * (K + 1);
  bool bad = false;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (long long)gridDim.x * kBlock) {
    const long long f = i % a.period;
    float yv, lad;
    rqs_point_table<INV, KT>(a.x[i], a.tab + f * tabw, a.c, yv, lad, bad);
    a.y[i] = yv;
    a.lad[i] = lad;
  }
  if (INV && a.bad && bad) atomicAdd(a.bad, 1);
}

template <bool INV>
static void launch_table_eval(const TableArgs& a, dim3 grid, hipStream_t st) {
  switch (a.c.K) {
    case 4: hipLaunchKernelGGL((rqs_table_eval_kernel<4, INV>), grid, dim3(kBlock), 0, st, a); break;
    case 8: hipLaunchKernelGGL((rqs_table_eval_kernel<8, INV>), grid, dim3(kBlock), 0, st, a); break;
    case 10: hipLaunchKernelGGL((rqs_table_eval_kernel<10, INV>), grid, dim3(kBlock), 0, st, a); break;
    case 16: hipLaunchKernelGGL((rqs_table_eval_kernel<16, INV>), grid, dim3(kBlock), 0, st, a); break;
    default: hipLaunchKernelGGL((rqs_table_eval_kernel<0, INV>), grid, dim3(kBlock), 0, st, a); break;
  }
}

// ------------------------------------------------------------------ conditioner input
struct CondInArgs {
  const float* x;
  const int32_t* id_idx;
  const float* ctx;
  const float *sh_w, *sh_h, *sh_d;
  float* out;
  long long B;
  int D, d_id, C, Pd;
  int apply;      // 0 raw gather, 1 inverse shared spline via LDS tables, 2 via logits in HBM/L2
  RqsConst c;
};

// out[b, :d_id] = x[b, id_idx] (optionally through the inverse shared spline),
// out[b, d_id:] = ctx[b, :].  One thread per output element, rows of out are
// written contiguously; tables live in LDS.
__global__ __launch_bounds__(kBlock) void rqs_conditioner_input_kernel(const CondInArgs a) {
  extern __shared__ __align__(16) float smem[];
  const int K = a.c.K;
  const int tabw = 3 * (K + 1);
  float* tab = smem;
  int* idi = reinterpret_cast<int*>(smem + (a.apply == 1 ? a.d_id * tabw : 0));
  for (int i = threadIdx.x; i < a.d_id; i += kBlock) idi[i] = a.id_idx[i];
  if (a.apply == 1) {
    for (int f = threadIdx.x; f < a.d_id; f += kBlock) {
      SplitLogits p{a.sh_w + (long long)f * K, a.sh_h + (long long)f * K, a.sh_d + (long long)f * a.Pd,
                    K, 1.f, a.c.edge_logit, a.c.tails};
      rqs_build_table(p, a.c, tab + f * tabw);
    }
  }
  __syncthreads();
  const int W = a.d_id + a.C;
  const long long total = a.B * W;
  bool bad = false;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    const long long b = i / W;
    const int j = (int)(i - b * W);
    float v;
    if (j < a.d_id) {
      v = a.x[b * a.D + idi[j]];
      if (a.apply == 1) {
        float yv, lad;
        rqs_point_table<true>(v, tab + j * tabw, a.c, yv, lad, bad);
        v = yv;
      } else if (a.apply == 2) {
        SplitLogits p{a.sh_w + (long long)j * K, a.sh_h + (long long)j * K, a.sh_d + (long long)j * a.Pd,
                      K, 1.f, a.c.edge_logit, a.c.tails};
        float yv, lad;
        rqs_point<0, true>(v, p, a.c, yv, lad, bad);
        v = yv;
      }
    } else {
      v = a.ctx[b * a.C + (j - a.d_id)];
    }
    a.out[i] = v;
  }
}

// ------------------------------------------------------------------ host side
// ------------------------------------------------------------------ identity half of a coupling
// Everything a coupling layer does with its identity features, in one launch (callers: the last-layer kernel's host
// path, vcnf_amd/fused_final.py; reference coupling.py:76-116):
//     v        = x[b, id_idx[f]]
//     y        = S(v) / S^-1(v)  with the batch-shared spline of feature f (apply), or v
//     out[b, id_idx[f]] = y ;  cond_in[b, f] = sampling ? y : v ;  partial[chunk][b] = sum over the chunk's features
// It replaces a gather, the table build, the table evaluation (21 scattered global reads per element: 83 us for
// the 16384 x 512 identity half of a config-C5 layer) and an index_put.  A workgroup owns a chunk of 64 features and
// 128 rows: waves 0-2 build the chunk's knot tables into LDS, one COLUMN per feature ([entry][64]), so that with
// lane = feature every table read of a wave is conflict free; each wave then walks 32 rows.  The bin is found
// by bisection on the knots (they ascend, so the result is the count of knots <= x that rqs_point_table computes).
struct IdHalfArgs {
  const float* x;
  float *y, *cond_in, *partial;
  const int32_t* id_idx;
  const float *sw, *sh, *sd;
  int32_t* bad;
  long long B;
  int D, d_id, chunks, nd, apply, cond_out;
  int rows_wg;         // rows per workgroup (a multiple of 16)
  RqsConst c;
};


template <int K, bool INV>
__global__ __launch_bounds__(kBlock) void rqs_identity_half_kernel(const IdHalfArgs a) {
  __shared__ float tabT[3 * (K + 1) * 64];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int fc = blockIdx.x % a.chunks;
  const long long rb = blockIdx.x / a.chunks;
  const int f = fc * 64 + lane;
  const bool fok = f < a.d_id;
  const RqsConst& c = a.c;
  if (a.apply) {
    if (wave < 3 && fok) {                       // wave 0: x knots, 1: y knots, 2: derivatives
      SplitLogits p{a.sw + (long long)f * K, a.sh + (long long)f * K, a.sd + (long long)f * a.nd,
                    K, c.wh_scale, c.edge_logit, c.tails};
      rqs_build_table_part_k<K>(p, c, tabT + lane, 64, wave);      // logits requested first: one memory latency (the
                                                                    // run-time loops waited for a load per iteration)
    }
    __syncthreads();
  }
  const int col = fok ? a.id_idx[f] : 0;
  const float* t = tabT + lane;
  const float* key = t + (INV ? (K + 1) * 64 : 0);
  bool bad = false;
  const long long r0 = rb * a.rows_wg + wave * (a.rows_wg / 4);
  constexpr int U = 4;
#pragma unroll 1
  for (int i = 0; i < a.rows_wg / 4; i += U) {
    float v[U], yv[U], lad[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long row = r0 + i + u;
      v[u] = (fok && row < a.B) ? a.x[row * a.D + col] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      yv[u] = v[u];
      lad[u] = 0.f;
      if (a.apply) {
        const bool inside = c.tails == 0 || ((v[u] >= c.lo_x) && (v[u] <= c.hi_x));
        const float xi = inside ? v[u] : c.lo_x;
        int bin = 0;
        if constexpr ((K & (K - 1)) == 0) {
#pragma unroll
          for (int step = K / 2; step >= 1; step >>= 1) bin += (xi >= key[(bin + step) * 64]) ? step : 0;
        } else {
#pragma unroll
          for (int k = 1; k < K; ++k) bin += (xi >= key[k * 64]) ? 1 : 0;
        }
        RqsBin b;
        b.xl = t[bin * 64];
        b.w = t[(bin + 1) * 64] - b.xl;
        b.yl = t[(K + 1 + bin) * 64];
        b.h = t[(K + 2 + bin) * 64] - b.yl;
        b.d0 = t[(2 * K + 2 + bin) * 64];
        b.d1 = t[(2 * K + 3 + bin) * 64];
        float yy, ll;
        bool bb = false;
        rqs_bin_eval<INV>(xi, b, yy, ll, bb);
        yv[u] = inside ? yy : v[u];
        lad[u] = (inside && fok) ? ll : 0.f;
        bad = bad || (inside && fok && bb);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long row = r0 + i + u;
      if (fok && row < a.B) {
        a.y[row * a.D + col] = yv[u];
        if (a.cond_in) a.cond_in[row * a.d_id + f] = a.cond_out ? yv[u] : v[u];
      }
      if (a.apply) {
        float s = lad[u];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0 && row < a.B) a.partial[(long long)fc * a.B + row] = s;
      }
    }
  }
  if (INV && a.bad && bad) atomicAdd(a.bad, 1);
}

template <bool INV>
static bool launch_identity_half(const IdHalfArgs& a, dim3 grid, hipStream_t st) {
  switch (a.c.K) {
    case 4: hipLaunchKernelGGL((rqs_identity_half_kernel<4, INV>), grid, dim3(kBlock), 0, st, a); return true;
    case 8: hipLaunchKernelGGL((rqs_identity_half_kernel<8, INV>), grid, dim3(kBlock), 0, st, a); return true;
    case 10: hipLaunchKernelGGL((rqs_identity_half_kernel<10, INV>), grid, dim3(kBlock), 0, st, a); return true;
    case 16: hipLaunchKernelGGL((rqs_identity_half_kernel<16, INV>), grid, dim3(kBlock), 0, st, a); return true;
    case 32: hipLaunchKernelGGL((rqs_identity_half_kernel<32, INV>), grid, dim3(kBlock), 0, st, a); return true;
    default: return false;
  }
}

static int fill_const(const vcnf_rqs_cfg* cfg, RqsConst& c, int& n_deriv) {
  if (!cfg) return VCNF_ERR_NULL;
  const int K = cfg->num_bins;
  if (K < 1 || K > 1024) return VCNF_ERR_SHAPE;
  if (cfg->tails != VCNF_TAILS_NONE && cfg->tails != VCNF_TAILS_LINEAR && cfg->tails != VCNF_TAILS_CIRCULAR)
    return VCNF_ERR_UNSUPPORTED;
  if (cfg->tails == VCNF_TAILS_LINEAR && K < 2) return VCNF_ERR_SHAPE;
  // splines.py:104-107
  if ((double)cfg->min_bin_width * K > 1.0 || (double)cfg->min_bin_height * K > 1.0) return VCNF_ERR_VALUE;
  c.K = K;
  c.tails = cfg->tails;
  c.lo_x = cfg->left;
  c.hi_x = cfg->right;
  c.span_x = (float)((double)cfg->right - (double)cfg->left);
  c.lo_y = cfg->bottom;
  c.hi_y = cfg->top;
  c.span_y = (float)((double)cfg->top - (double)cfg->bottom);
  c.min_w = cfg->min_bin_width;
  c.min_h = cfg->min_bin_height;
  c.min_d = cfg->min_derivative;
  c.free_w = (float)(1.0 - (double)cfg->min_bin_width * K);
  c.free_h = (float)(1.0 - (double)cfg->min_bin_height * K);
  c.wh_scale = cfg->wh_scale;
  c.edge_logit = (float)log(exp(1.0 - (double)cfg->min_derivative) - 1.0);
  n_deriv = cfg->tails == VCNF_TAILS_LINEAR ? K - 1 : cfg->tails == VCNF_TAILS_CIRCULAR ? K : K + 1;
  return VCNF_OK;
}

static inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

template <bool INV>
static void launch_coupling(const CouplingArgs& a, int K, dim3 grid, size_t lds, hipStream_t st) {
  switch (K) {
    case 4: hipLaunchKernelGGL((rqs_coupling_kernel<4, INV>), grid, dim3(kBlock), lds, st, a); break;
    case 8: hipLaunchKernelGGL((rqs_coupling_kernel<8, INV>), grid, dim3(kBlock), lds, st, a); break;
    case 10: hipLaunchKernelGGL((rqs_coupling_kernel<10, INV>), grid, dim3(kBlock), lds, st, a); break;
    case 16: hipLaunchKernelGGL((rqs_coupling_kernel<16, INV>), grid, dim3(kBlock), lds, st, a); break;
    default: hipLaunchKernelGGL((rqs_coupling_kernel<0, INV>), grid, dim3(kBlock), lds, st, a); break;
  }
}

template <bool INV>
static void launch_coupling_pf(const CouplingArgs& a, int K, dim3 grid, size_t lds, hipStream_t st) {
  switch (K) {
    case 4: hipLaunchKernelGGL((rqs_coupling_pf_kernel<4, INV, 8>), grid, dim3(kBlock), lds, st, a); break;
    case 8: hipLaunchKernelGGL((rqs_coupling_pf_kernel<8, INV, 8>), grid, dim3(kBlock), lds, st, a); break;
    case 10: hipLaunchKernelGGL((rqs_coupling_pf_kernel<10, INV, 8>), grid, dim3(kBlock), lds, st, a); break;
    case 16: hipLaunchKernelGGL((rqs_coupling_pf_kernel<16, INV, 8>), grid, dim3(kBlock), lds, st, a); break;
    default: hipLaunchKernelGGL((rqs_coupling_pf_kernel<0, INV, 8>), grid, dim3(kBlock), lds, st, a); break;
  }
}

template <bool INV>
static void launch_elem(const ElemArgs& a, int K, dim3 grid, hipStream_t st) {
  switch (K) {
    case 4: hipLaunchKernelGGL((rqs_elementwise_kernel<4, INV>), grid, dim3(kBlock), 0, st, a); break;
    case 8: hipLaunchKernelGGL((rqs_elementwise_kernel<8, INV>), grid, dim3(kBlock), 0, st, a); break;
    case 10: hipLaunchKernelGGL((rqs_elementwise_kernel<10, INV>), grid, dim3(kBlock), 0, st, a); break;
    case 16: hipLaunchKernelGGL((rqs_elementwise_kernel<16, INV>), grid, dim3(kBlock), 0, st, a); break;
    default: hipLaunchKernelGGL((rqs_elementwise_kernel<0, INV>), grid, dim3(kBlock), 0, st, a); break;
  }
}

constexpr size_t kLdsBudget = 60 * 1024;     // per workgroup; leaves >= 2 workgroups per CU
constexpr size_t kTableBudget = 16 * 1024;   // knot tables of the shared spline; beyond: read logits directly

static int shared_mode(bool present, int d_id, int K) {
  if (!present || d_id == 0) return 0;
  return (size_t)d_id * 3 * (K + 1) * 4 <= kTableBudget ? 1 : 2;
}

// LDS bytes that do not depend on the params chunk: x tile, y tile, tables, index maps.
static size_t fixed_lds(int S, long long D, int d_id, int K, int sh_mode) {
  return (size_t)2 * S * D * 4 + (sh_mode == 1 ? (((size_t)d_id * 3 * (K + 1) + 3) & ~(size_t)3) * 4 : 0) +
         (size_t)D * 4 + 64;
}

// Lanes per sample: the power of two <= 64 that wastes the fewest lane-slots over
// both halves of the layer among the group sizes whose tiles (x, y and one pass of
// params for all S = 256/G samples) fit the LDS budget; ties go to the wider group
// (shorter per-lane loops).  Falls back to 64 lanes (smallest tiles).
static int pick_group(int d_t, int d_id, int P, int K, int sh_mode) {
  int best = 64;
  double best_cost = 1e300;
  for (int G = 1; G <= 64; G <<= 1) {
    const int S = kBlock / G;
    const size_t need = fixed_lds(S, (long long)d_t + d_id, d_id, K, sh_mode) +
                        (size_t)S * P * 4 * (d_t < G ? d_t : G);
    if (need > kLdsBudget) continue;
    const double slots = (double)((d_t + G - 1) / G + (d_id + G - 1) / G) * G;
    if (slots <= best_cost) {
      best_cost = slots;
      best = G;
    }
  }
  return best;
}

}  // namespace vcnf

using namespace vcnf;

extern "C" int vcnf_rqs_coupling_f32(const float* x, const float* params,
                                     const int32_t* transform_idx, int32_t d_t,
                                     const int32_t* identity_idx, int32_t d_id,
                                     const float* shared_w, const float* shared_h, const float* shared_d,
                                     float* y, float* logdet, int64_t batch,
                                     const vcnf_rqs_cfg* cfg, int inverse,
                                     int ld_mode, float ld_sign, int32_t* bad_disc, void* stream) {
  CouplingArgs a;
  int nd = 0;
  const int rc = fill_const(cfg, a.c, nd);
  if (rc != VCNF_OK) return rc;
  if (batch < 0 || d_t < 1 || d_id < 0) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!x || !params || !transform_idx || !y || !logdet || (d_id > 0 && !identity_idx)) return VCNF_ERR_NULL;
  const bool any_sh = shared_w || shared_h || shared_d;
  if (any_sh && !(shared_w && shared_h && shared_d)) return VCNF_ERR_NULL;
  if (ld_mode != VCNF_LD_STORE && ld_mode != VCNF_LD_ACCUM) return VCNF_ERR_UNSUPPORTED;
  const long long D = (long long)d_t + d_id;
  if (D > 16384) return VCNF_ERR_SHAPE;
  if (!aligned(x, 4) || !aligned(params, 4) || !aligned(y, 4) || !aligned(logdet, 4)) return VCNF_ERR_ALIGN;

  const int K = a.c.K;
  a.x = x; a.params = params; a.tf_idx = transform_idx; a.id_idx = identity_idx;
  a.sh_w = shared_w; a.sh_h = shared_h; a.sh_d = shared_d;
  a.sh_mode = shared_mode(any_sh, d_id, K);
  a.y = y; a.logdet = logdet; a.bad = bad_disc;
  a.B = batch; a.D = (int)D; a.d_t = d_t; a.d_id = d_id;
  a.P = 2 * K + nd; a.Pd = nd;
  a.G = pick_group(d_t, d_id, a.P, K, a.sh_mode);
  a.S = kBlock / a.G;
  a.ld_mode = ld_mode; a.ld_sign = ld_sign;

  // LDS plan: x tile + y tile + tables + indices are fixed; the params chunk takes the rest.
  const size_t fixed = fixed_lds(a.S, D, d_id, K, a.sh_mode);
  if (fixed + (size_t)a.S * a.P * 4 > 150 * 1024) return VCNF_ERR_SHAPE;
  const size_t budget = fixed + (size_t)a.S * a.P * 4 > kLdsBudget ? fixed + (size_t)a.S * a.P * 4 : kLdsBudget;
  long long jc = (long long)((budget - fixed) / ((size_t)a.S * a.P * 4));
  if (jc >= d_t) {
    jc = d_t;
  } else {
    // whole multiples of G keep every lane busy; multiples of 4 keep chunks 16-byte aligned
    if (jc >= a.G) jc -= jc % a.G;
    if (jc >= 4) jc -= jc % 4;
    if (jc < 1) jc = 1;
  }
  a.JC = (int)jc;
  const size_t n_p = (((size_t)a.S * a.JC * a.P + 3) & ~(size_t)3) * 4;
  const size_t lds = fixed + n_p;
  if (lds > 64 * 1024) return VCNF_ERR_SHAPE;   // default dynamic-LDS limit; D beyond ~1.5k needs column tiling

  const long long rowlen = (long long)d_t * a.P;
  a.vec_x = aligned(x, 16) && aligned(y, 16);           // tile start = b0*D floats, S % 4 == 0
  a.vec_p = aligned(params, 16) && (a.JC == d_t || ((rowlen & 3) == 0 && (((long long)a.JC * a.P) & 3) == 0));

  const long long ntiles = (batch + a.S - 1) / a.S;
  const long long max_grid = 256 * 8;
  dim3 grid((unsigned)(ntiles < max_grid ? ntiles : max_grid));
  hipStream_t st = (hipStream_t)stream;
  // prefetching fast path: whole rows per tile, everything 16-byte sized and aligned, tile within
  // the 8 + 2 float4 register slots a thread keeps in flight
  const bool pf = a.JC == d_t && a.vec_x && a.vec_p && (rowlen & 3) == 0 && (D & 3) == 0 &&
                  (long long)a.S * rowlen <= 8LL * 4 * kBlock && (long long)a.S * D <= 2LL * 4 * kBlock;
  if (pf) {
    if (inverse) launch_coupling_pf<true>(a, K, grid, lds, st);
    else launch_coupling_pf<false>(a, K, grid, lds, st);
  } else {
    if (inverse) launch_coupling<true>(a, K, grid, lds, st);
    else launch_coupling<false>(a, K, grid, lds, st);
  }
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

extern "C" int vcnf_rqs_elementwise_f32(const float* x, const float* uw, const float* uh, const float* ud,
                                        int64_t ld_w, int64_t ld_h, int64_t ld_d,
                                        float* y, float* logabsdet, int64_t n,
                                        const vcnf_rqs_cfg* cfg, int inverse, int32_t* bad_disc, void* stream) {
  ElemArgs a;
  int nd = 0;
  const int rc = fill_const(cfg, a.c, nd);
  if (rc != VCNF_OK) return rc;
  if (n < 0 || ld_w < 0 || ld_h < 0 || ld_d < 0) return VCNF_ERR_SHAPE;
  if (n == 0) return VCNF_OK;
  if (!x || !uw || !uh || !y || !logabsdet || (nd > 0 && !ud)) return VCNF_ERR_NULL;
  a.x = x; a.uw = uw; a.uh = uh; a.ud = ud; a.ld_w = ld_w; a.ld_h = ld_h; a.ld_d = ld_d;
  a.y = y; a.lad = logabsdet; a.bad = bad_disc; a.n = n;
  const long long blocks = (n + kBlock - 1) / kBlock;
  dim3 grid((unsigned)(blocks < 256 * 16 ? blocks : 256 * 16));
  hipStream_t st = (hipStream_t)stream;
  if (inverse) launch_elem<true>(a, a.c.K, grid, st);
  else launch_elem<false>(a, a.c.K, grid, st);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

extern "C" int vcnf_rqs_elementwise_strided_f32(const float* x, const float* uw, const float* uh, const float* ud,
                                                int64_t row_w, int64_t row_h, int64_t row_d, int64_t inner,
                                                int64_t k_stride, int64_t period,
                                                float* y, float* logabsdet, int64_t n,
                                                const vcnf_rqs_cfg* cfg, int inverse, int32_t* bad_disc, void* stream) {
  ElemStridedArgs a;
  int nd = 0;
  const int rc = fill_const(cfg, a.c, nd);
  if (rc != VCNF_OK) return rc;
  if (n < 0 || row_w < 0 || row_h < 0 || row_d < 0 || inner < 1 || k_stride < 1 || period < 0) return VCNF_ERR_SHAPE;
  if (n == 0) return VCNF_OK;
  if (!x || !uw || !uh || !y || !logabsdet || (nd > 0 && !ud)) return VCNF_ERR_NULL;
  a.x = x; a.uw = uw; a.uh = uh; a.ud = ud; a.row_w = row_w; a.row_h = row_h; a.row_d = row_d;
  a.inner = inner; a.ks = k_stride; a.period = period;
  a.y = y; a.lad = logabsdet; a.bad = bad_disc; a.n = n;
  const long long blocks = (n + kBlock - 1) / kBlock;
  dim3 grid((unsigned)(blocks < 256 * 16 ? blocks : 256 * 16));
  hipStream_t st = (hipStream_t)stream;
  if (inverse) launch_elem_strided<true>(a, a.c.K, grid, st);
  else launch_elem_strided<false>(a, a.c.K, grid, st);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

extern "C" int vcnf_rqs_shared_f32(const float* x, const float* sw, const float* sh, const float* sd, int64_t period,
                                   float* tables, float* y, float* logabsdet, int64_t n,
                                   const vcnf_rqs_cfg* cfg, int inverse, int32_t* bad_disc, void* stream) {
  TableArgs a;
  const int rc = fill_const(cfg, a.c, a.nd);
  if (rc != VCNF_OK) return rc;
  if (n < 0 || period < 1) return VCNF_ERR_SHAPE;
  if (n == 0) return VCNF_OK;
  if (!x || !sw || !sh || !y || !logabsdet || !tables || (a.nd > 0 && !sd)) return VCNF_ERR_NULL;
  a.x = x; a.sw = sw; a.sh = sh; a.sd = sd; a.tab = tables; a.y = y; a.lad = logabsdet; a.bad = bad_disc;
  a.n = n; a.period = period;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rqs_build_tables_kernel, dim3((unsigned)((period + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, a);
  const long long blocks = (n + kBlock - 1) / kBlock;
  dim3 grid((unsigned)(blocks < 256 * 16 ? blocks : 256 * 16));
  if (inverse) launch_table_eval<true>(a, grid, st);
  else launch_table_eval<false>(a, grid, st);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

extern "C" int vcnf_rqs_identity_half_supported(int32_t num_bins, int32_t tails) {
  return ((num_bins == 4 || num_bins == 8 || num_bins == 10 || num_bins == 16 || num_bins == 32) &&
          (tails == VCNF_TAILS_NONE || tails == VCNF_TAILS_LINEAR)) ? 1 : 0;
}

extern "C" int64_t vcnf_rqs_identity_half_partial_rows(int32_t d_id) { return d_id > 0 ? (d_id + 63) / 64 : 0; }

extern "C" int vcnf_rqs_identity_half_f32(const float* x, float* y, float* cond_in, float* partial, int64_t batch,
                                          int32_t features, const int32_t* identity_idx, int32_t d_id,
                                          const float* shared_w, const float* shared_h, const float* shared_d,
                                          const vcnf_rqs_cfg* cfg, int inverse, int cond_sees_output,
                                          int32_t* bad_disc, void* stream) {
  IdHalfArgs a;
  const bool any_sh = shared_w || shared_h || shared_d;
  a.nd = 0;
  if (any_sh) {
    const int rc = fill_const(cfg, a.c, a.nd);
    if (rc != VCNF_OK) return rc;
    if (!vcnf_rqs_identity_half_supported(a.c.K, a.c.tails)) return VCNF_ERR_UNSUPPORTED;
    if (!(shared_w && shared_h) || (a.nd > 0 && !shared_d)) return VCNF_ERR_NULL;
  } else {
    a.c = RqsConst{};
    a.c.K = 4;
  }
  if (batch < 0 || features < 1 || d_id < 0 || d_id > features) return VCNF_ERR_SHAPE;
  if (batch == 0 || d_id == 0) return VCNF_OK;
  if (!x || !y || !identity_idx || (any_sh && !partial)) return VCNF_ERR_NULL;
  a.x = x; a.y = y; a.cond_in = cond_in; a.partial = partial; a.id_idx = identity_idx;
  a.sw = shared_w; a.sh = shared_h; a.sd = shared_d; a.bad = bad_disc;
  a.B = batch; a.D = features; a.d_id = d_id; a.chunks = (d_id + 63) / 64;
  a.apply = any_sh ? 1 : 0;
  a.cond_out = cond_sees_output ? 1 : 0;
  a.rows_wg = 128;      // (512 rows per workgroup for large batches - fewer table builds - measured slower: 1.59 vs 1.47 ms at
                        // 524 288 x 512, the kernel is bound by its strided row accesses, not by the set-up)
  const long long rblocks = (batch + a.rows_wg - 1) / a.rows_wg;
  if (rblocks * a.chunks > 0x7fffffffLL) return VCNF_ERR_SHAPE;
  dim3 grid((unsigned)(rblocks * a.chunks));
  hipStream_t st = (hipStream_t)stream;
  const bool ok = inverse ? launch_identity_half<true>(a, grid, st) : launch_identity_half<false>(a, grid, st);
  if (!ok) return VCNF_ERR_UNSUPPORTED;
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

extern "C" int vcnf_rqs_conditioner_input_f32(const float* x, int64_t batch, int32_t features,
                                              const int32_t* identity_idx, int32_t d_id,
                                              const float* context, int32_t ctx_dim,
                                              const float* shared_w, const float* shared_h, const float* shared_d,
                                              const vcnf_rqs_cfg* cfg, int apply_inverse_shared,
                                              float* out, void* stream) {
  CondInArgs a;
  int nd = 0;
  const int rc = fill_const(cfg, a.c, nd);
  if (rc != VCNF_OK) return rc;
  if (batch < 0 || features < 1 || d_id < 0 || d_id > features || ctx_dim < 0) return VCNF_ERR_SHAPE;
  if (batch == 0 || d_id + ctx_dim == 0) return VCNF_OK;
  if (!x || !out || (d_id > 0 && !identity_idx) || (ctx_dim > 0 && !context)) return VCNF_ERR_NULL;
  if (apply_inverse_shared && !(shared_w && shared_h && shared_d)) return VCNF_ERR_NULL;
  a.x = x; a.id_idx = identity_idx; a.ctx = context; a.sh_w = shared_w; a.sh_h = shared_h; a.sh_d = shared_d;
  a.out = out; a.B = batch; a.D = features; a.d_id = d_id; a.C = ctx_dim; a.Pd = nd;
  a.apply = apply_inverse_shared ? shared_mode(true, d_id, a.c.K) : 0;
  const size_t lds = ((size_t)(a.apply == 1 ? d_id * 3 * (a.c.K + 1) : 0) + d_id) * 4 + 16;
  if (lds > 150 * 1024) return VCNF_ERR_SHAPE;
  const long long total = batch * (long long)(d_id + ctx_dim);
  const long long blocks = (total + kBlock - 1) / kBlock;
  dim3 grid((unsigned)(blocks < 256 * 8 ? blocks : 256 * 8));
  hipLaunchKernelGGL(rqs_conditioner_input_kernel, grid, dim3(kBlock), lds, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}
