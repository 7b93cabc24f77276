// A run of MaskedAffineFlow layers WITH their MLP conditioners, and the ActNorm / AffineConstFlow layers between
// them, in ONE launch, fp32 and fp64: the model of the reference's own drivers (/root/reference/run.py:58-68,
// runadultvdeq.py:101-108, rundiag.py:57-65: K x [MaskedAffineFlow(b, t, s), ActNorm] with s, t = MLP([D, H, D]),
// D = 2 .. 15, H = 2 D .. 8 D, 1024 - 2048 samples per call, .double()).  Layer by layer that is ~9 launches per
// (coupling, ActNorm) pair - two dense layers and an activation for each of s and t, the masked affine map, the
// per-feature map - of a microsecond of work each: the evaluation is launch-bound.
//
// Reference semantics (normflow/flows/affine/coupling.py:171-222, :22-61; nets/mlp.py:30-58; core.py:144-183):
//   masked affine   z' = b z + (1 - b) (z exp(s(b z)) + t(b z)),   log|det| = sum (1 - b) s      (forward)
//                   z' = b z + (1 - b) (z - t(b z)) exp(-s(b z)),  log|det| = -sum (1 - b) s     (inverse)
//                   non-finite s / t entries become NaN (:205-208)
//   per-feature     z' = z exp(s) + t | (z - t) exp(-s),           log|det| = +- sum s
//   s, t            Linear(D, H) - LeakyReLU(slope) - Linear(H, D)  (either may be absent)
//
// Work split: 16 lanes per sample, lane d owns feature d (D <= 16); hidden unit u of a conditioner belongs to lane
// u % 16 (H <= 64: at most four per lane); a layer gathers the masked inputs into every lane and reduce-scatters the
// units' contributions to the D outputs back to the owning lanes with wave shuffles.  No LDS, no barrier; the parameters (a few hundred values per layer) are read through the caches.  `table` is a device
// array of 12 int64 per layer, in application order:
//   [0] kind (0 masked affine, 1 per-feature)  [1] H  [2] slope (bits of a double)  [3] b
//   [4..7] s: W1 [H, D], b1 [H], W2 [D, H], b2 [D] (0 = no s)   [8..11] t likewise
//   per-feature: [4] s [D] or 0, [5] t [D] or 0
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"

namespace vcnf {

constexpr int kMsMaxD = 16;
constexpr int kMsG = 16;
constexpr int kMsMaxHL = 4;                 // hidden units per lane: H <= 64
constexpr int kMsEntry = 12;

template <typename T>
struct MaStackArgs {
  const T* z;
  T* out;
  T* logdet;
  const long long* table;
  long long B;
  int D, n_layers, inverse, ld_mode;
  T ld_sign;
};

__device__ __forceinline__ float ms_exp(float v) { return expf(v); }
__device__ __forceinline__ double ms_exp(double v) { return exp(v); }

template <typename T>
__device__ __forceinline__ T ms_group_sum(T v) {
#pragma unroll
  for (int m = kMsG >> 1; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// Sum of v[i] over the 16 lanes of a group for all 16 i at once, result i in lane i (reduce-scatter: 8 + 4 + 2 + 1
// exchanges instead of 16 x 4).
template <typename T>
__device__ __forceinline__ T ms_reduce_scatter(T (&v)[kMsMaxD], int g) {
#pragma unroll
  for (int m = kMsG >> 1; m > 0; m >>= 1) {
    const bool up = (g & m) != 0;                      // this lane keeps the upper half of what is left
#pragma unroll
    for (int i = 0; i < m; ++i) {
      const T keep = up ? v[m + i] : v[i];
      const T send = up ? v[i] : v[m + i];
      v[i] = keep + __shfl_xor(send, m, 64);
    }
  }
  return v[0];
}

// Lane d of a group holds out_d = b2[d] + sum_u W2[d, u] act(b1[u] + sum_d' W1[u, d'] in[d']); ``in`` is the whole
// input vector (identical in the 16 lanes), hidden unit u belongs to lane u % 16.
template <typename T>
__device__ __forceinline__ T ms_mlp(const T* w1, const T* b1, const T* w2, const T* b2, int D, int H, T slope, int g,
                                    const T (&in)[kMsMaxD]) {
  T part[kMsMaxD];
#pragma unroll
  for (int d = 0; d < kMsMaxD; ++d) part[d] = T(0);
#pragma unroll
  for (int m = 0; m < kMsMaxHL; ++m) {
    const int u = g + kMsG * m;
    if (u < H) {
      T acc = b1[u];
#pragma unroll
      for (int d = 0; d < kMsMaxD; ++d)
        if (d < D) acc += w1[u * D + d] * in[d];
      const T h = acc > T(0) ? acc : slope * acc;       // nn.LeakyReLU (mlp.py:33)
#pragma unroll
      for (int d = 0; d < kMsMaxD; ++d)
        if (d < D) part[d] += w2[d * H + u] * h;
    }
  }
  T tot = T(0);
  if (D <= 4) {
    // few outputs: one butterfly per output (4 D exchanges) beats the 15-step reduce-scatter over 16 slots
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (d < D) {
        const T sum = ms_group_sum(part[d]);
        tot = (g == d) ? sum : tot;
      }
    }
  } else {
    tot = ms_reduce_scatter<T>(part, g);
  }
  return g < D ? tot + b2[g] : T(0);
}

// Lane d of a 16-lane group owns feature d of the sample through the whole run (one exponential per lane and layer; as
// identical copies of the vector in every lane it was D of them - 0.41 ms for 16 pairs of layers at D = 15 in fp64).
// Per masked-affine layer: the masked inputs are gathered into every lane (D shuffles), the hidden units of s and t are
// spread over the lanes, their contributions to the D outputs are reduce-scattered back to the owning lanes.
template <typename T>
__global__ __launch_bounds__(256) void masked_affine_stack_kernel(const MaStackArgs<T> a) {
  const int g = threadIdx.x & (kMsG - 1);
  const int D = a.D;
  const T nanv = (T)__builtin_nanf("");
  const long long per_block = 256 / kMsG;
  const bool mine = g < D;
  for (long long r0 = (long long)blockIdx.x * per_block; r0 < a.B; r0 += (long long)gridDim.x * per_block) {
    const long long r = r0 + (threadIdx.x >> 4);
    const bool live = r < a.B;                       // a dead group runs along on zeros (shuffles stay wave-wide)
    T z = (live && mine) ? a.z[r * D + g] : T(0);
    T ld = T(0);                                     // this feature's share of the log|det|
    for (int l = 0; l < a.n_layers; ++l) {
      const long long* e = a.table + (long long)l * kMsEntry;
      if (e[0] == 0) {
        const int H = (int)e[1];
        const T slope = (T)__builtin_bit_cast(double, e[2]);
        const T bm = mine ? reinterpret_cast<const T*>(e[3])[g] : T(0);
        const T zm = bm * z;
        T zin[kMsMaxD];
#pragma unroll
        for (int d = 0; d < kMsMaxD; ++d) zin[d] = d < D ? __shfl(zm, d, kMsG) : T(0);
        T sc = T(0), tr = T(0);
        if (e[4])
          sc = ms_mlp<T>(reinterpret_cast<const T*>(e[4]), reinterpret_cast<const T*>(e[5]), reinterpret_cast<const T*>(e[6]),
                         reinterpret_cast<const T*>(e[7]), D, H, slope, g, zin);
        if (e[8])
          tr = ms_mlp<T>(reinterpret_cast<const T*>(e[8]), reinterpret_cast<const T*>(e[9]), reinterpret_cast<const T*>(e[10]),
                         reinterpret_cast<const T*>(e[11]), D, H, slope, g, zin);
        if (mine) {
          const T s_ = isfinite(sc) ? sc : nanv;         // coupling.py:205-208
          const T t_ = isfinite(tr) ? tr : nanv;
          const T om = T(1) - bm;
          if (a.inverse) { z = zm + om * (z - t_) * ms_exp(-s_); ld -= om * s_; }     // :220-221
          else { z = zm + om * (z * ms_exp(s_) + t_); ld += om * s_; }                  // :209-210
        }
      } else if (mine) {
        const T s_ = e[4] ? reinterpret_cast<const T*>(e[4])[g] : T(0);
        const T t_ = e[5] ? reinterpret_cast<const T*>(e[5])[g] : T(0);
        if (a.inverse) { z = (z - t_) * ms_exp(-s_); ld -= s_; }                       // coupling.py:47-53
        else { z = z * ms_exp(s_) + t_; ld += s_; }                                     // :38-45
      }
    }
    ld = ms_group_sum(ld);
    if (live) {
      if (mine) a.out[r * D + g] = z;
      if (g == 0) {
        const T v = a.ld_sign * ld;
        a.logdet[r] = a.ld_mode ? a.logdet[r] + v : v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------- backward
// Vector-Jacobian product of the same run (training: NormalizingFlow.forward_kld / reverse_kld through these
// layers, core.py:30-141).  Nothing but the run's OUTPUT is kept from the forward pass: a coupling layer is
// invertible and its conditioner only sees the features it leaves unchanged, so walking the layers backwards the
// kernel rebuilds each layer's input from its output (and the conditioner's hidden units from the unchanged
// features), exactly like the forward kernel but with the inverse map.  Per layer and sample:
//   upstream gz (w.r.t. the layer's output), gl (w.r.t. the run's summed log|det|)  ->
//   gs = om (gz dy/ds) +- om gl,  gt = om gz dy/dt,  gx = gz dy/dx + b (W1s^T ga_s + W1t^T ga_t)
// with ga = (W2^T g) act'(a) for each conditioner.  Parameter gradients are summed over the four samples of a wave
// with shuffles and added to the flat gradient buffer with hardware floating-point atomics (the summation order
// over the batch is not fixed: ~1e-16 relative in fp64, ~1e-7 in fp32).  goff[l]: element offset of layer l's block
// in the buffer - masked affine: [W1s | b1s | W2s | b2s | W1t | b1t | W2t | b2t] (absent conditioners take no room),
// per-feature: [s | t].
template <typename T>
struct MaBwdArgs {
  const T* zout;      // the run's output
  const T* gout;      // gradient w.r.t. it
  const T* gld;       // gradient w.r.t. the run's summed log|det| [B] (or NULL: 0)
  T* gin;             // gradient w.r.t. the run's input
  T* grads;           // flat parameter-gradient buffer (zeroed by the caller)
  const long long* table;
  const long long* goff;
  long long B;
  int D, n_layers, inverse;
};

template <typename T>
__device__ __forceinline__ void ms_accumulate(T* dst, T v, bool owner) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  if (owner) unsafeAtomicAdd(dst, v);
}

// One conditioner, backward: g = upstream gradient of its D outputs (identical copies in every lane), zin = its
// input.  Adds the parameter gradients, returns this lane's share of W1^T ga per input feature in gx (to be
// reduce-scattered by the caller together with the other conditioner's).
template <typename T>
__device__ __forceinline__ void ms_mlp_bwd(const T* w1, const T* b1, const T* w2, int D, int H, T slope, int g, bool first16,
                                           const T (&zin)[kMsMaxD], const T (&gin)[kMsMaxD], T* gw1, T* gb1, T* gw2,
                                           T (&gx)[kMsMaxD]) {
#pragma unroll
  for (int m = 0; m < kMsMaxHL; ++m) {
    const int u = g + kMsG * m;
    const bool ok = u < H;                              // (uniform over the four groups of a wave: shuffles inside)
    T acc = ok ? b1[u] : T(0);
#pragma unroll
    for (int d = 0; d < kMsMaxD; ++d)
      if (d < D && ok) acc += w1[u * D + d] * zin[d];
    const T h = acc > T(0) ? acc : slope * acc;
    T gh = T(0);
#pragma unroll
    for (int d = 0; d < kMsMaxD; ++d)
      if (d < D && ok) gh += w2[d * H + u] * gin[d];
    const T ga = gh * (acc > T(0) ? T(1) : slope);
    if (kMsG * m < H) {                                 // some lane of the group owns a unit in this round
#pragma unroll
      for (int d = 0; d < kMsMaxD; ++d) {
        if (d < D) {
          ms_accumulate<T>(gw2 + d * H + (ok ? u : 0), ok ? gin[d] * h : T(0), first16 && ok);
          ms_accumulate<T>(gw1 + (ok ? u : 0) * D + d, ok ? ga * zin[d] : T(0), first16 && ok);
          if (ok) gx[d] += w1[u * D + d] * ga;
        }
      }
      ms_accumulate<T>(gb1 + (ok ? u : 0), ok ? ga : T(0), first16 && ok);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void masked_affine_stack_bwd_kernel(const MaBwdArgs<T> a) {
  const int g = threadIdx.x & (kMsG - 1);
  const int D = a.D;
  const bool mine = g < D;
  const bool first16 = (threadIdx.x & 63) < kMsG;       // the lanes that issue the atomics of a wave
  const long long per_block = 256 / kMsG;
  for (long long r0 = (long long)blockIdx.x * per_block; r0 < a.B; r0 += (long long)gridDim.x * per_block) {
    const long long r = r0 + (threadIdx.x >> 4);
    const bool live = r < a.B;                          // dead groups run along on zeros
    T z = (live && mine) ? a.zout[r * D + g] : T(0);
    T gz = (live && mine) ? a.gout[r * D + g] : T(0);
    const T gl = (live && a.gld) ? a.gld[r] : T(0);
    for (int l = a.n_layers - 1; l >= 0; --l) {
      const long long* e = a.table + (long long)l * kMsEntry;
      T* gb = a.grads + a.goff[l];
      if (e[0] == 0) {
        const int H = (int)e[1];
        const T slope = (T)__builtin_bit_cast(double, e[2]);
        const T bm = mine ? reinterpret_cast<const T*>(e[3])[g] : T(0);
        const T om = mine ? T(1) - bm : T(0);
        const T zm = bm * z;                            // the conditioners' input: unchanged by the layer
        T zin[kMsMaxD];
#pragma unroll
        for (int d = 0; d < kMsMaxD; ++d) zin[d] = d < D ? __shfl(zm, d, kMsG) : T(0);
        T sc = T(0), tr = T(0);
        if (e[4])
          sc = ms_mlp<T>(reinterpret_cast<const T*>(e[4]), reinterpret_cast<const T*>(e[5]), reinterpret_cast<const T*>(e[6]),
                         reinterpret_cast<const T*>(e[7]), D, H, slope, g, zin);
        if (e[8])
          tr = ms_mlp<T>(reinterpret_cast<const T*>(e[8]), reinterpret_cast<const T*>(e[9]), reinterpret_cast<const T*>(e[10]),
                         reinterpret_cast<const T*>(e[11]), D, H, slope, g, zin);
        T gs, gt, x;
        if (!a.inverse) {                               // y = b x + (1 - b)(x e^s + t),  ld += (1 - b) s
          const T es = ms_exp(sc);
          const T den = bm + om * es;
          x = (z - om * tr) / den;
          gs = om * (gz * x * es + gl);
          gt = om * gz;
          gz = gz * den;
        } else {                                        // y = b x + (1 - b)(x - t) e^-s,  ld -= (1 - b) s
          const T es = ms_exp(-sc);
          const T den = bm + om * es;
          x = (z + om * tr * es) / den;
          gs = om * (-gz * (x - tr) * es - gl);
          gt = -om * gz * es;
          gz = gz * den;
        }
        T gx[kMsMaxD], gin[kMsMaxD];
#pragma unroll
        for (int d = 0; d < kMsMaxD; ++d) gx[d] = T(0);
        const int blk = H * D + H + D * H + D;          // elements of one conditioner's block
        T* gnet = gb;
        if (e[4]) {
#pragma unroll
          for (int d = 0; d < kMsMaxD; ++d) gin[d] = d < D ? __shfl(gs, d, kMsG) : T(0);
          ms_mlp_bwd<T>(reinterpret_cast<const T*>(e[4]), reinterpret_cast<const T*>(e[5]), reinterpret_cast<const T*>(e[6]),
                        D, H, slope, g, first16, zin, gin, gnet, gnet + H * D, gnet + H * D + H, gx);
          ms_accumulate<T>(gnet + H * D + H + D * H + (mine ? g : 0), mine ? gs : T(0), first16 && mine);
          gnet += blk;
        }
        if (e[8]) {
#pragma unroll
          for (int d = 0; d < kMsMaxD; ++d) gin[d] = d < D ? __shfl(gt, d, kMsG) : T(0);
          ms_mlp_bwd<T>(reinterpret_cast<const T*>(e[8]), reinterpret_cast<const T*>(e[9]), reinterpret_cast<const T*>(e[10]),
                        D, H, slope, g, first16, zin, gin, gnet, gnet + H * D, gnet + H * D + H, gx);
          ms_accumulate<T>(gnet + H * D + H + D * H + (mine ? g : 0), mine ? gt : T(0), first16 && mine);
        }
        const T gxm = ms_reduce_scatter<T>(gx, g);      // lane d: sum over units and both conditioners of W1[u, d] ga[u]
        gz += bm * gxm;
        z = mine ? x : T(0);
      } else {
        const T s_ = (mine && e[4]) ? reinterpret_cast<const T*>(e[4])[g] : T(0);
        const T t_ = (mine && e[5]) ? reinterpret_cast<const T*>(e[5])[g] : T(0);
        T gs, gt, x;
        if (!a.inverse) {                               // y = x e^s + t, ld += s
          const T es = ms_exp(s_);
          x = (z - t_) / es;
          gs = gz * x * es + gl;
          gt = gz;
          gz = gz * es;
        } else {                                        // y = (x - t) e^-s, ld -= s
          const T es = ms_exp(-s_);
          x = z / es + t_;
          gs = -gz * z - gl;
          gt = -gz * es;
          gz = gz * es;
        }
        if (e[4]) ms_accumulate<T>(gb + (mine ? g : 0), (mine && live) ? gs : T(0), first16 && mine);
        if (e[5]) ms_accumulate<T>(gb + (e[4] ? D : 0) + (mine ? g : 0), (mine && live) ? gt : T(0), first16 && mine);
        z = mine ? x : T(0);
      }
    }
    if (live && mine) a.gin[r * D + g] = gz;
  }
}

template <typename T>
static int ms_launch_bwd(const T* zout, const T* gout, const T* gld, T* gin, T* grads, const int64_t* table,
                         const int64_t* goff, int64_t batch, int32_t features, int32_t n_layers, int inverse, void* stream) {
  if (batch < 0 || features < 1 || features > kMsMaxD || n_layers < 1) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!zout || !gout || !gin || !grads || !table || !goff) return VCNF_ERR_NULL;
  MaBwdArgs<T> a{zout, gout, gld, gin, grads, reinterpret_cast<const long long*>(table),
                 reinterpret_cast<const long long*>(goff), batch, features, n_layers, inverse ? 1 : 0};
  long long blocks = (batch + 15) / 16;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(masked_affine_stack_bwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

template <typename T>
static int ms_launch(const T* z, T* out, T* logdet, const int64_t* table, int64_t batch, int32_t features,
                     int32_t n_layers, int inverse, int ld_mode, T ld_sign, void* stream) {
  if (batch < 0 || features < 1 || features > kMsMaxD || n_layers < 1) return VCNF_ERR_SHAPE;
  if (ld_mode != VCNF_LD_STORE && ld_mode != VCNF_LD_ACCUM) return VCNF_ERR_UNSUPPORTED;
  if (batch == 0) return VCNF_OK;
  if (!z || !out || !logdet || !table) return VCNF_ERR_NULL;
  MaStackArgs<T> a{z, out, logdet, reinterpret_cast<const long long*>(table), batch, features, n_layers, inverse ? 1 : 0,
                   ld_mode, ld_sign};
  long long blocks = (batch + 15) / 16;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(masked_affine_stack_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

}  // namespace vcnf

using namespace vcnf;

extern "C" int vcnf_masked_affine_stack_supported(int32_t features, int32_t hidden) {
  return (features >= 1 && features <= kMsMaxD && hidden >= 0 && hidden <= kMsG * kMsMaxHL) ? 1 : 0;
}

extern "C" int vcnf_masked_affine_stack_f32(const float* z, float* out, float* logdet, const int64_t* table,
                                            int64_t batch, int32_t features, int32_t n_layers, int inverse,
                                            int ld_mode, float ld_sign, void* stream) {
  return ms_launch<float>(z, out, logdet, table, batch, features, n_layers, inverse, ld_mode, ld_sign, stream);
}

extern "C" int vcnf_masked_affine_stack_f64(const double* z, double* out, double* logdet, const int64_t* table,
                                            int64_t batch, int32_t features, int32_t n_layers, int inverse,
                                            int ld_mode, double ld_sign, void* stream) {
  return ms_launch<double>(z, out, logdet, table, batch, features, n_layers, inverse, ld_mode, ld_sign, stream);
}

extern "C" int vcnf_masked_affine_stack_bwd_f32(const float* z_out, const float* g_out, const float* g_logdet, float* g_in,
                                                float* grads, const int64_t* table, const int64_t* grad_offsets,
                                                int64_t batch, int32_t features, int32_t n_layers, int inverse, void* stream) {
  return ms_launch_bwd<float>(z_out, g_out, g_logdet, g_in, grads, table, grad_offsets, batch, features, n_layers, inverse, stream);
}

extern "C" int vcnf_masked_affine_stack_bwd_f64(const double* z_out, const double* g_out, const double* g_logdet, double* g_in,
                                                double* grads, const int64_t* table, const int64_t* grad_offsets,
                                                int64_t batch, int32_t features, int32_t n_layers, int inverse, void* stream) {
  return ms_launch_bwd<double>(z_out, g_out, g_logdet, g_in, grads, table, grad_offsets, batch, features, n_layers, inverse, stream);
}
