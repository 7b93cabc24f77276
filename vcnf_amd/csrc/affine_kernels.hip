// Affine coupling, masked affine, per-channel affine, column gather and
// diagonal-Gaussian end caps for MI355X (gfx950, wave64), with their C-ABI entry
// points.  All of these are HBM-bound elementwise maps with a per-sample row
// reduction: one group of G lanes (power of two <= 64) owns a sample, strides over
// its row with coalesced accesses, and the log|det| partials are combined with
// wave shuffles - no second pass over the data and no atomics.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"

namespace vcnf {

constexpr int kBlock = 256;

// The kernels below that the reference's fp64 drivers reach (runadultvdeq.py:101-108,183: model.double() over
// MaskedAffineFlow / ActNorm / DiagGaussian stacks) are templates on the scalar type; the fp32 instantiations are the
// code they always were, the fp64 ones use the double-precision library functions.
template <typename T>
__device__ __forceinline__ T group_sum(T v, int G) {
  for (int m = G >> 1; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

template <typename T>
__device__ __forceinline__ void put_ld(T* ld, long long b, T v, int mode) {
  ld[b] = mode ? ld[b] + v : v;
}

__device__ __forceinline__ float exp_(float v) { return expf(v); }
__device__ __forceinline__ double exp_(double v) { return exp(v); }
__device__ __forceinline__ float log_(float v) { return logf(v); }
__device__ __forceinline__ double log_(double v) { return log(v); }

// lanes per sample for a row of n elements
static int pick_lanes(long long n) {
  int G = 1;
  while (G < 64 && G < n) G <<= 1;
  return G;
}

static dim3 grid_for(long long batch, int G) {
  const long long per_block = kBlock / G;
  long long blocks = (batch + per_block - 1) / per_block;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  return dim3((unsigned)blocks);
}

// torch.sigmoid and log(sigmoid) evaluated like the reference does
// (flows/affine/coupling.py:128-136): sigma = 1/(1+exp(-v)), then log(sigma).
__device__ __forceinline__ float sigmoid_f(float v) { return 1.f / (1.f + expf(-v)); }
__device__ __forceinline__ double sigmoid_f(double v) { return 1.0 / (1.0 + exp(-v)); }

// ------------------------------------------------------------------ affine coupling
template <typename T>
struct AffineArgsT {
  const T* z;
  const T* param;
  T* out;
  T* logdet;
  long long B;
  int C, inner, t_off, d_t;
  int scale_map, inverse, G, ld_mode;
  T ld_sign;
};
using AffineArgs = AffineArgsT<float>;

template <typename T>
__global__ __launch_bounds__(kBlock) void affine_coupling_kernel(const AffineArgsT<T> a) {
  const int g = threadIdx.x & (a.G - 1);
  const int per_block = kBlock / a.G;
  const long long row = (long long)a.C * a.inner;          // elements of one sample
  const long long t_lo = (long long)a.t_off * a.inner;     // transformed span inside a row
  const long long t_n = (long long)a.d_t * a.inner;
  const int npar = a.scale_map == VCNF_SCALE_NONE ? 1 : 2;
  for (long long b = (long long)blockIdx.x * per_block + threadIdx.x / a.G; b < a.B;
       b += (long long)gridDim.x * per_block) {
    const T* zr = a.z + b * row;
    T* orow = a.out + b * row;
    const T* pr = a.param + b * (long long)npar * t_n;
    T acc = 0;
    for (long long e = g; e < row; e += a.G) {
      T v = zr[e];
      const long long te = e - t_lo;
      if (te >= 0 && te < t_n) {
        if (npar == 1) {                                   // coupling.py:139-141 / :165-167
          const T p = pr[te];
          v = a.inverse ? v - p : v + p;
        } else {
          const long long c = te / a.inner, i = te - c * a.inner;
          const T shift = pr[(2 * c) * a.inner + i];   // param[:, 0::2]
          const T sc = pr[(2 * c + 1) * a.inner + i];  // param[:, 1::2]
          if (a.scale_map == VCNF_SCALE_EXP) {             // :124-126 / :150-152
            if (a.inverse) { v = (v - shift) * exp_(-sc); acc -= sc; }
            else { v = v * exp_(sc) + shift; acc += sc; }
          } else {
            const T sg = sigmoid_f(sc + T(2));
            const T lg = log_(sg);
            const bool divide = (a.scale_map == VCNF_SCALE_SIGMOID) != (a.inverse != 0);
            if (a.inverse) v = divide ? (v - shift) / sg : (v - shift) * sg;
            else v = divide ? v / sg + shift : v * sg + shift;
            acc += divide ? -lg : lg;
          }
        }
      }
      orow[e] = v;
    }
    acc = group_sum(acc, a.G);
    if (g == 0 && a.logdet) put_ld(a.logdet, b, a.ld_sign * acc, a.ld_mode);
  }
}

// ------------------------------------------------------------------ autoregressive affine (MAF)
// flows/affine/autoregressive.py:75-103: params [B, D, 2] = (unconstrained scale, shift) per feature from a MADE pass,
// scale = sigmoid(u + 2) + 1e-3; density direction y = scale x + shift, log|det| = sum log scale; the other direction
// y = (x - shift) / scale, log|det| = -sum log scale (called D times by the sampling loop, :29-36).
struct MafArgs {
  const float* x;
  const float* param;
  float* out;
  float* logdet;
  long long B;
  int D, inverse, G, ld_mode;
  float ld_sign;
};

__global__ __launch_bounds__(kBlock) void maf_affine_kernel(const MafArgs a) {
  const int g = threadIdx.x & (a.G - 1);
  const int per_block = kBlock / a.G;
  for (long long b = (long long)blockIdx.x * per_block + threadIdx.x / a.G; b < a.B;
       b += (long long)gridDim.x * per_block) {
    const float2* pr = reinterpret_cast<const float2*>(a.param) + b * a.D;
    float acc = 0.f;
    for (int j = g; j < a.D; j += a.G) {
      const float2 p = pr[j];
      const float scale = sigmoid_f(p.x + 2.f) + 1e-3f;
      const float v = a.x[b * a.D + j];
      a.out[b * a.D + j] = a.inverse ? (v - p.y) / scale : scale * v + p.y;
      acc += logf(scale);
    }
    acc = group_sum(acc, a.G);
    if (g == 0) put_ld(a.logdet, b, a.ld_sign * (a.inverse ? -acc : acc), a.ld_mode);
  }
}

// ------------------------------------------------------------------ masked affine
template <typename T>
struct MaskedArgsT {
  const T *z, *s, *t, *b;
  T* out;
  T* logdet;
  long long B;
  int D, inverse, G, ld_mode;
  T ld_sign;
};
using MaskedArgs = MaskedArgsT<float>;

template <typename T>
__global__ __launch_bounds__(kBlock) void masked_affine_kernel(const MaskedArgsT<T> a) {
  const int g = threadIdx.x & (a.G - 1);
  const int per_block = kBlock / a.G;
  const T nanv = (T)__builtin_nanf("");
  for (long long r = (long long)blockIdx.x * per_block + threadIdx.x / a.G; r < a.B;
       r += (long long)gridDim.x * per_block) {
    T acc = 0;
    for (int j = g; j < a.D; j += a.G) {
      const long long e = r * a.D + j;
      const T m = a.b[j];
      const T zv = a.z[e];
      T sc = a.s ? a.s[e] : T(0);
      T tr = a.t ? a.t[e] : T(0);
      sc = isfinite(sc) ? sc : nanv;                       // coupling.py:205-208
      tr = isfinite(tr) ? tr : nanv;
      const T zm = m * zv;
      const T om = T(1) - m;
      T v;
      if (a.inverse) v = zm + om * (zv - tr) * exp_(-sc);  // :220
      else v = zm + om * (zv * exp_(sc) + tr);             // :209
      a.out[e] = v;
      acc += om * sc;                                      // :210 / :221
    }
    acc = group_sum(acc, a.G);
    if (g == 0) put_ld(a.logdet, r, a.ld_sign * (a.inverse ? -acc : acc), a.ld_mode);
  }
}

// ------------------------------------------------------------------ per-channel affine, gather
template <typename T>
struct ConstArgsT {
  const T *z, *s, *t;
  T* out;
  long long total;
  int C, inner, inverse;
};
using ConstArgs = ConstArgsT<float>;

template <typename T>
__global__ __launch_bounds__(kBlock) void affine_const_kernel(const ConstArgsT<T> a) {
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < a.total; e += (long long)gridDim.x * kBlock) {
    const int c = (int)((e / a.inner) % a.C);
    const T s = a.s ? a.s[c] : T(0), t = a.t ? a.t[c] : T(0);
    const T v = a.z[e];
    a.out[e] = a.inverse ? (v - t) * exp_(-s) : v * exp_(s) + t;   // coupling.py:38 / :47
  }
}

template <typename T>
struct PermArgsT {
  const T* z;
  const int32_t* idx;
  T* out;
  long long total;
  int C, inner;
};
using PermArgs = PermArgsT<float>;

template <typename T>
__global__ __launch_bounds__(kBlock) void permute_kernel(const PermArgsT<T> a) {
  const long long row = (long long)a.C * a.inner;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < a.total; e += (long long)gridDim.x * kBlock) {
    const long long b = e / row;
    const long long r = e - b * row;
    const int c = (int)(r / a.inner);
    const int i = (int)(r - (long long)c * a.inner);
    a.out[e] = a.z[b * row + (long long)a.idx[c] * a.inner + i];
  }
}

// The channel partition of a coupling as two tensors and back (the training path's gather / scatter,
// flows/neural_spline/coupling.py:86-88 and :122-124): idx lists the columns of the first part, then those of the
// second.  One pass each way - no [B, C] intermediate, no slice copies, and each direction is the other's VJP.
template <typename T>
struct SplitArgsT {
  const T* z;          // split: source [B, C]; merge: unused
  T* out;              // merge: destination [B, C]; split: unused
  T* a;                // [B, na]
  T* b;                // [B, C - na]
  const int32_t* idx;  // split: column of z that gathered position c reads; merge: gathered position that column c reads
  long long total;     // B * C
  int C, na;
};

template <typename T, bool MERGE>
__global__ __launch_bounds__(kBlock) void split_merge_kernel(const SplitArgsT<T> a) {
  const int nb = a.C - a.na;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < a.total; e += (long long)gridDim.x * kBlock) {
    const long long r = e / a.C;
    const int c = (int)(e - r * a.C);
    if (MERGE) {
      const int p = a.idx[c];
      a.out[e] = p < a.na ? a.a[r * a.na + p] : a.b[r * nb + (p - a.na)];
    } else {
      const T v = a.z[r * a.C + a.idx[c]];
      if (c < a.na) a.a[r * a.na + c] = v;
      else a.b[r * nb + (c - a.na)] = v;
    }
  }
}

// ------------------------------------------------------------------ diagonal Gaussian
template <typename T>
struct GaussArgsT {
  const T *in, *loc, *log_scale;
  T* z;
  T* logp;
  long long B;
  int D, G, ld_mode, sample;
  T log_temp, ld_sign, norm;   // norm = -0.5 * D * log(2 pi)
};
using GaussArgs = GaussArgsT<float>;

template <typename T>
__global__ __launch_bounds__(kBlock) void diag_gaussian_kernel(const GaussArgsT<T> a) {
  const int g = threadIdx.x & (a.G - 1);
  const int per_block = kBlock / a.G;
  for (long long r = (long long)blockIdx.x * per_block + threadIdx.x / a.G; r < a.B;
       r += (long long)gridDim.x * per_block) {
    T acc = 0;
    for (int j = g; j < a.D; j += a.G) {
      const T ls = a.log_scale[j] + a.log_temp;
      const T v = a.in[r * a.D + j];
      if (a.sample) {                                      // base.py:639-641
        a.z[r * a.D + j] = a.loc[j] + exp_(ls) * v;
        acc += ls + T(0.5) * v * v;
      } else {                                             // base.py:649-651
        const T u = (v - a.loc[j]) / exp_(ls);
        acc += ls + T(0.5) * u * u;
      }
    }
    acc = group_sum(acc, a.G);
    if (g == 0) put_ld(a.logp, r, a.ld_sign * (a.norm - acc), a.ld_mode);
  }
}

// Same arithmetic for D <= NC * G (a lane owns at most NC fixed columns): the per-column constants - and the exp()
// of the scale - are computed once per lane instead of once per element (the loads from log_scale may alias the
// outputs, so the compiler cannot hoist them out of the row loop above), and two rows are in flight per lane group.
// 1M x 64: 201 -> 151 us per call (HBM floor 32 us); the four-columns-per-lane kernel below takes the aligned shapes.
template <int NC>
__global__ __launch_bounds__(kBlock) void diag_gaussian_cols_kernel(const GaussArgs a) {
  const int g = threadIdx.x & (a.G - 1);
  const int per_block = kBlock / a.G;
  float ls[NC], sc[NC], lc[NC];
  bool have[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int j = g + c * a.G;
    have[c] = j < a.D;
    ls[c] = have[c] ? a.log_scale[j] + a.log_temp : 0.f;
    sc[c] = expf(ls[c]);
    lc[c] = have[c] ? a.loc[j] : 0.f;
  }
  const long long step = (long long)gridDim.x * per_block;
  for (long long r = (long long)blockIdx.x * per_block + threadIdx.x / a.G; r < a.B; r += 2 * step) {
    const long long r2 = r + step;
    const bool two = r2 < a.B;
    float v0[NC], v1[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      v0[c] = have[c] ? a.in[r * a.D + g + c * a.G] : 0.f;
      v1[c] = (have[c] && two) ? a.in[r2 * a.D + g + c * a.G] : 0.f;
    }
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (have[c]) {
        if (a.sample) {
          a.z[r * a.D + g + c * a.G] = lc[c] + sc[c] * v0[c];
          if (two) a.z[r2 * a.D + g + c * a.G] = lc[c] + sc[c] * v1[c];
          acc0 += ls[c] + 0.5f * v0[c] * v0[c];
          acc1 += ls[c] + 0.5f * v1[c] * v1[c];
        } else {
          const float u0 = (v0[c] - lc[c]) / sc[c], u1 = (v1[c] - lc[c]) / sc[c];
          acc0 += ls[c] + 0.5f * u0 * u0;
          acc1 += ls[c] + 0.5f * u1 * u1;
        }
      }
    }
    acc0 = group_sum(acc0, a.G);
    acc1 = group_sum(acc1, a.G);
    if (g == 0) {
      put_ld(a.logp, r, a.ld_sign * (a.norm - acc0), a.ld_mode);
      if (two) put_ld(a.logp, r2, a.ld_sign * (a.norm - acc1), a.ld_mode);
    }
  }
}

// D % 4 == 0, D <= 256, 16-byte aligned rows: a lane owns FOUR ADJACENT columns (one 16-byte load / store per row), a
// row is shared by G = pow2 >= D / 4 lanes, so the log2(G) shuffle steps of the row sum are paid once per four elements
// (with one column per lane a 64-feature row spent more instructions in the reduction than in the density).
__global__ __launch_bounds__(kBlock) void diag_gaussian_vec4_kernel(const GaussArgs a) {
  const int g = threadIdx.x & (a.G - 1);
  const int per_block = kBlock / a.G;
  const bool have = 4 * g < a.D;
  float ls[4], sc[4], lc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    ls[c] = have ? a.log_scale[4 * g + c] + a.log_temp : 0.f;
    sc[c] = expf(ls[c]);
    lc[c] = have ? a.loc[4 * g + c] : 0.f;
  }
  const long long step = (long long)gridDim.x * per_block;
  const int d4 = a.D >> 2;
  for (long long r = (long long)blockIdx.x * per_block + threadIdx.x / a.G; r < a.B; r += 2 * step) {
    const long long r2 = r + step;
    const bool two = r2 < a.B;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 p0 = have ? reinterpret_cast<const float4*>(a.in)[r * d4 + g] : z4;
    const float4 p1 = (have && two) ? reinterpret_cast<const float4*>(a.in)[r2 * d4 + g] : z4;
    const float v0[4] = {p0.x, p0.y, p0.z, p0.w}, v1[4] = {p1.x, p1.y, p1.z, p1.w};
    float acc0 = 0.f, acc1 = 0.f;
    float o0[4], o1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (a.sample) {
        o0[c] = lc[c] + sc[c] * v0[c];
        o1[c] = lc[c] + sc[c] * v1[c];
        acc0 += ls[c] + 0.5f * v0[c] * v0[c];
        acc1 += ls[c] + 0.5f * v1[c] * v1[c];
      } else {
        const float u0 = (v0[c] - lc[c]) / sc[c], u1 = (v1[c] - lc[c]) / sc[c];
        acc0 += ls[c] + 0.5f * u0 * u0;
        acc1 += ls[c] + 0.5f * u1 * u1;
      }
    }
    if (a.sample && have) {
      reinterpret_cast<float4*>(a.z)[r * d4 + g] = make_float4(o0[0], o0[1], o0[2], o0[3]);
      if (two) reinterpret_cast<float4*>(a.z)[r2 * d4 + g] = make_float4(o1[0], o1[1], o1[2], o1[3]);
    }
    acc0 = group_sum(have ? acc0 : 0.f, a.G);
    acc1 = group_sum(have ? acc1 : 0.f, a.G);
    if (g == 0) {
      put_ld(a.logp, r, a.ld_sign * (a.norm - acc0), a.ld_mode);
      if (two) put_ld(a.logp, r2, a.ld_sign * (a.norm - acc1), a.ld_mode);
    }
  }
}

static inline bool ok_ld(int m) { return m == VCNF_LD_STORE || m == VCNF_LD_ACCUM; }
static inline int launched() { return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH; }

}  // namespace vcnf

using namespace vcnf;

extern "C" int vcnf_abi_version(void) { return VCNF_ABI_VERSION; }

extern "C" const char* vcnf_status_string(int status) {
  switch (status) {
    case VCNF_OK: return "ok";
    case VCNF_ERR_NULL: return "required pointer is NULL";
    case VCNF_ERR_SHAPE: return "inconsistent or unsupported sizes";
    case VCNF_ERR_ALIGN: return "buffer not 4-byte aligned";
    case VCNF_ERR_VALUE: return "minimal bin width/height too large for the number of bins";
    case VCNF_ERR_UNSUPPORTED: return "unsupported mode";
    case VCNF_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown status";
  }
}

extern "C" int vcnf_affine_coupling_f32(const float* z, const float* param, float* out, float* logdet,
                                        int64_t batch, int32_t channels, int32_t inner,
                                        int32_t t_off, int32_t d_t, int scale_map, int inverse,
                                        int ld_mode, float ld_sign, void* stream) {
  if (batch < 0 || channels < 1 || inner < 1 || t_off < 0 || d_t < 0 || t_off + d_t > channels) return VCNF_ERR_SHAPE;
  if (scale_map < VCNF_SCALE_EXP || scale_map > VCNF_SCALE_NONE) return VCNF_ERR_UNSUPPORTED;
  if (!ok_ld(ld_mode)) return VCNF_ERR_UNSUPPORTED;
  if (batch == 0) return VCNF_OK;
  if (!z || !out || (d_t > 0 && !param)) return VCNF_ERR_NULL;
  if (scale_map != VCNF_SCALE_NONE && !logdet) return VCNF_ERR_NULL;
  AffineArgs a{z, param, out, logdet, batch, channels, inner, t_off, d_t, scale_map, inverse ? 1 : 0,
               pick_lanes((long long)channels * inner), ld_mode, ld_sign};
  hipLaunchKernelGGL(affine_coupling_kernel<float>, grid_for(batch, a.G), dim3(kBlock), 0, (hipStream_t)stream, a);
  return launched();
}

extern "C" int vcnf_maf_affine_f32(const float* x, const float* params, float* out, float* logdet, int64_t batch,
                                   int32_t features, int inverse, int ld_mode, float ld_sign, void* stream) {
  if (batch < 0 || features < 1) return VCNF_ERR_SHAPE;
  if (!ok_ld(ld_mode)) return VCNF_ERR_UNSUPPORTED;
  if (batch == 0) return VCNF_OK;
  if (!x || !params || !out || !logdet) return VCNF_ERR_NULL;
  if (reinterpret_cast<uintptr_t>(params) & 7) return VCNF_ERR_ALIGN;
  MafArgs a{x, params, out, logdet, batch, features, inverse ? 1 : 0, pick_lanes(features), ld_mode, ld_sign};
  hipLaunchKernelGGL(maf_affine_kernel, grid_for(batch, a.G), dim3(kBlock), 0, (hipStream_t)stream, a);
  return launched();
}

extern "C" int vcnf_masked_affine_f32(const float* z, const float* s, const float* t, const float* b,
                                      float* out, float* logdet, int64_t batch, int32_t features,
                                      int inverse, int ld_mode, float ld_sign, void* stream) {
  if (batch < 0 || features < 1) return VCNF_ERR_SHAPE;
  if (!ok_ld(ld_mode)) return VCNF_ERR_UNSUPPORTED;
  if (batch == 0) return VCNF_OK;
  if (!z || !b || !out || !logdet) return VCNF_ERR_NULL;
  MaskedArgs a{z, s, t, b, out, logdet, batch, features, inverse ? 1 : 0, pick_lanes(features), ld_mode, ld_sign};
  hipLaunchKernelGGL(masked_affine_kernel<float>, grid_for(batch, a.G), dim3(kBlock), 0, (hipStream_t)stream, a);
  return launched();
}

extern "C" int vcnf_affine_const_f32(const float* z, const float* s, const float* t, float* out,
                                     int64_t batch, int32_t channels, int32_t inner, int inverse, void* stream) {
  if (batch < 0 || channels < 1 || inner < 1) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!z || !out) return VCNF_ERR_NULL;
  ConstArgs a{z, s, t, out, batch * (long long)channels * inner, channels, inner, inverse ? 1 : 0};
  long long blocks = (a.total + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(affine_const_kernel<float>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a);
  return launched();
}

extern "C" int vcnf_permute_f32(const float* z, const int32_t* idx, float* out,
                                int64_t batch, int32_t channels, int32_t inner, void* stream) {
  if (batch < 0 || channels < 1 || inner < 1) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!z || !idx || !out) return VCNF_ERR_NULL;
  PermArgs a{z, idx, out, batch * (long long)channels * inner, channels, inner};
  long long blocks = (a.total + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(permute_kernel<float>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a);
  return launched();
}

template <typename T, bool MERGE>
static int split_merge(const T* z, T* out, T* pa, T* pb, const int32_t* idx, int64_t batch, int32_t channels,
                       int32_t first, void* stream) {
  if (batch < 0 || channels < 1 || first < 0 || first > channels) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!(MERGE ? (const void*)out : (const void*)z) || !idx || (first > 0 && !pa) || (first < channels && !pb)) return VCNF_ERR_NULL;
  SplitArgsT<T> a{z, out, pa, pb, idx, batch * (long long)channels, channels, first};
  long long blocks = (a.total + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL((split_merge_kernel<T, MERGE>), dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a);
  return launched();
}

extern "C" int vcnf_split_columns_f32(const float* z, const int32_t* idx, float* first_part, float* second_part,
                                      int64_t batch, int32_t channels, int32_t first, void* stream) {
  return split_merge<float, false>(z, nullptr, first_part, second_part, idx, batch, channels, first, stream);
}
extern "C" int vcnf_merge_columns_f32(const float* first_part, const float* second_part, const int32_t* idx, float* out,
                                      int64_t batch, int32_t channels, int32_t first, void* stream) {
  return split_merge<float, true>(nullptr, out, const_cast<float*>(first_part), const_cast<float*>(second_part), idx,
                                  batch, channels, first, stream);
}
extern "C" int vcnf_split_columns_f64(const double* z, const int32_t* idx, double* first_part, double* second_part,
                                      int64_t batch, int32_t channels, int32_t first, void* stream) {
  return split_merge<double, false>(z, nullptr, first_part, second_part, idx, batch, channels, first, stream);
}
extern "C" int vcnf_merge_columns_f64(const double* first_part, const double* second_part, const int32_t* idx, double* out,
                                      int64_t batch, int32_t channels, int32_t first, void* stream) {
  return split_merge<double, true>(nullptr, out, const_cast<double*>(first_part), const_cast<double*>(second_part), idx,
                                   batch, channels, first, stream);
}

static int gauss(const float* in, const float* loc, const float* log_scale, float log_temperature,
                 float* z, float* logp, int64_t batch, int32_t features, int ld_mode, float ld_sign,
                 int sample, void* stream) {
  if (batch < 0 || features < 1) return VCNF_ERR_SHAPE;
  if (!ok_ld(ld_mode)) return VCNF_ERR_UNSUPPORTED;
  if (batch == 0) return VCNF_OK;
  if (!in || !loc || !log_scale || !logp || (sample && !z)) return VCNF_ERR_NULL;
  GaussArgs a{in, loc, log_scale, z, logp, batch, features, pick_lanes(features), ld_mode, sample,
              log_temperature, ld_sign, (float)(-0.5 * (double)features * log(2.0 * M_PI))};
  hipStream_t st = (hipStream_t)stream;
  const bool al16 = ((reinterpret_cast<uintptr_t>(in) | (sample ? reinterpret_cast<uintptr_t>(z) : 0)) & 15) == 0;
  if (features % 4 == 0 && features <= 256 && al16) {
    a.G = pick_lanes(features / 4);
    hipLaunchKernelGGL(diag_gaussian_vec4_kernel, grid_for(batch, a.G), dim3(kBlock), 0, st, a);
    return launched();
  }
  const int nc = (features + a.G - 1) / a.G;
  const dim3 grid = grid_for(batch, a.G);
  switch (nc) {
    case 1: hipLaunchKernelGGL(diag_gaussian_cols_kernel<1>, grid, dim3(kBlock), 0, st, a); break;
    case 2: hipLaunchKernelGGL(diag_gaussian_cols_kernel<2>, grid, dim3(kBlock), 0, st, a); break;
    case 3: hipLaunchKernelGGL(diag_gaussian_cols_kernel<3>, grid, dim3(kBlock), 0, st, a); break;
    case 4: hipLaunchKernelGGL(diag_gaussian_cols_kernel<4>, grid, dim3(kBlock), 0, st, a); break;
    default: hipLaunchKernelGGL(diag_gaussian_kernel<float>, grid, dim3(kBlock), 0, st, a); break;
  }
  return launched();
}

extern "C" int vcnf_diag_gaussian_log_prob_f32(const float* z, const float* loc, const float* log_scale,
                                               float log_temperature, float* logp, int64_t batch,
                                               int32_t features, int ld_mode, float ld_sign, void* stream) {
  return gauss(z, loc, log_scale, log_temperature, nullptr, logp, batch, features, ld_mode, ld_sign, 0, stream);
}

extern "C" int vcnf_diag_gaussian_sample_f32(const float* eps, const float* loc, const float* log_scale,
                                             float log_temperature, float* z, float* logp, int64_t batch,
                                             int32_t features, void* stream) {
  return gauss(eps, loc, log_scale, log_temperature, z, logp, batch, features, VCNF_LD_STORE, 1.f, 1, stream);
}

// ------------------------------------------------------------------ fp64 entry points (same contracts, T = double)
extern "C" int vcnf_affine_coupling_f64(const double* z, const double* param, double* out, double* logdet,
                                        int64_t batch, int32_t channels, int32_t inner,
                                        int32_t t_off, int32_t d_t, int scale_map, int inverse,
                                        int ld_mode, double ld_sign, void* stream) {
  if (batch < 0 || channels < 1 || inner < 1 || t_off < 0 || d_t < 0 || t_off + d_t > channels) return VCNF_ERR_SHAPE;
  if (scale_map < VCNF_SCALE_EXP || scale_map > VCNF_SCALE_NONE) return VCNF_ERR_UNSUPPORTED;
  if (!ok_ld(ld_mode)) return VCNF_ERR_UNSUPPORTED;
  if (batch == 0) return VCNF_OK;
  if (!z || !out || (d_t > 0 && !param)) return VCNF_ERR_NULL;
  if (scale_map != VCNF_SCALE_NONE && !logdet) return VCNF_ERR_NULL;
  AffineArgsT<double> a{z, param, out, logdet, batch, channels, inner, t_off, d_t, scale_map, inverse ? 1 : 0,
                        pick_lanes((long long)channels * inner), ld_mode, ld_sign};
  hipLaunchKernelGGL(affine_coupling_kernel<double>, grid_for(batch, a.G), dim3(kBlock), 0, (hipStream_t)stream, a);
  return launched();
}

extern "C" int vcnf_masked_affine_f64(const double* z, const double* s, const double* t, const double* b,
                                      double* out, double* logdet, int64_t batch, int32_t features,
                                      int inverse, int ld_mode, double ld_sign, void* stream) {
  if (batch < 0 || features < 1) return VCNF_ERR_SHAPE;
  if (!ok_ld(ld_mode)) return VCNF_ERR_UNSUPPORTED;
  if (batch == 0) return VCNF_OK;
  if (!z || !b || !out || !logdet) return VCNF_ERR_NULL;
  MaskedArgsT<double> a{z, s, t, b, out, logdet, batch, features, inverse ? 1 : 0, pick_lanes(features), ld_mode, ld_sign};
  hipLaunchKernelGGL(masked_affine_kernel<double>, grid_for(batch, a.G), dim3(kBlock), 0, (hipStream_t)stream, a);
  return launched();
}

extern "C" int vcnf_affine_const_f64(const double* z, const double* s, const double* t, double* out,
                                     int64_t batch, int32_t channels, int32_t inner, int inverse, void* stream) {
  if (batch < 0 || channels < 1 || inner < 1) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!z || !out) return VCNF_ERR_NULL;
  ConstArgsT<double> a{z, s, t, out, batch * (long long)channels * inner, channels, inner, inverse ? 1 : 0};
  long long blocks = (a.total + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(affine_const_kernel<double>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a);
  return launched();
}

extern "C" int vcnf_permute_f64(const double* z, const int32_t* idx, double* out,
                                int64_t batch, int32_t channels, int32_t inner, void* stream) {
  if (batch < 0 || channels < 1 || inner < 1) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!z || !idx || !out) return VCNF_ERR_NULL;
  PermArgsT<double> a{z, idx, out, batch * (long long)channels * inner, channels, inner};
  long long blocks = (a.total + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(permute_kernel<double>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a);
  return launched();
}

static int gauss64(const double* in, const double* loc, const double* log_scale, double log_temperature,
                   double* z, double* logp, int64_t batch, int32_t features, int ld_mode, double ld_sign,
                   int sample, void* stream) {
  if (batch < 0 || features < 1) return VCNF_ERR_SHAPE;
  if (!ok_ld(ld_mode)) return VCNF_ERR_UNSUPPORTED;
  if (batch == 0) return VCNF_OK;
  if (!in || !loc || !log_scale || !logp || (sample && !z)) return VCNF_ERR_NULL;
  GaussArgsT<double> a{in, loc, log_scale, z, logp, batch, features, pick_lanes(features), ld_mode, sample,
                       log_temperature, ld_sign, -0.5 * (double)features * log(2.0 * M_PI)};
  hipLaunchKernelGGL(diag_gaussian_kernel<double>, grid_for(batch, a.G), dim3(kBlock), 0, (hipStream_t)stream, a);
  return launched();
}

extern "C" int vcnf_diag_gaussian_log_prob_f64(const double* z, const double* loc, const double* log_scale,
                                               double log_temperature, double* logp, int64_t batch,
                                               int32_t features, int ld_mode, double ld_sign, void* stream) {
  return gauss64(z, loc, log_scale, log_temperature, nullptr, logp, batch, features, ld_mode, ld_sign, 0, stream);
}

extern "C" int vcnf_diag_gaussian_sample_f64(const double* eps, const double* loc, const double* log_scale,
                                             double log_temperature, double* z, double* logp, int64_t batch,
                                             int32_t features, void* stream) {
  return gauss64(eps, loc, log_scale, log_temperature, z, logp, batch, features, VCNF_LD_STORE, 1.0, 1, stream);
}
