// Vector-Jacobian product of the rational-quadratic spline (training path).
//
// Given upstream gradients (g_y, g_lad) of one spline evaluation y, lad = RQS(x; uw, uh, ud)
// the kernel re-evaluates the element (same arithmetic as rqs_math.hpp) and walks the
// computation backwards by hand: bin formulas -> selected knots -> cumulative widths ->
// softmax -> logits, and the two knot derivatives through softplus.  The reference gets
// these gradients from PyTorch autograd over ~40 ATen kernels (utils/splines.py:88-193);
// here it is one pass per element.  The sampling direction (inverse spline) uses the
// implicit-function identities on the forward map F(v) = u:
//     dv/du = 1/F_v,   dv/dtheta = -F_theta / F_v,   lad_inv = -lad_fwd(v).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"
#include "rqs_math.hpp"

namespace vcnf {

constexpr int kBwdBlock = 256;
constexpr int kBwdMaxK = 64;

// Addressing of the logit rows as in vcnf_rqs_elementwise_strided_f32: element i -> outer = i / inner,
// s = i % inner, logit k at base + outer * row + s + k * ks.  Gradients use their own row strides
// (grow_*) with the same inner / ks: dense rows (K, K, nd) or the layout of the logits themselves
// (packed conditioner output: the gradient is then directly the conditioner's upstream gradient).
// The upstream log-det gradient is read at i / lad_div (1: per element; d_t * inner: per sample).
struct BwdArgs {
  const float *x, *uw, *uh, *ud;
  long long row_w, row_h, row_d, inner, ks;
  const float *gy, *glad;
  long long lad_div;
  float *gx, *guw, *guh, *gud;
  long long grow_w, grow_h, grow_d;
  long long n;
  int nd;                          // derivative logits per element
  RqsConst c;
};

// gradient of softplus (beta 1, threshold 20)
__device__ __forceinline__ float softplus_grad(float v) {
  return v > 20.f ? 1.f : div_nr(1.f, 1.f + hw_exp2(-v * kLog2e));
}

// Reverse pass through rqs_bin_eval's forward branch: inputs the bin (xl, w, yl, h, d0, d1),
// the point x and upstream (gy, gl); outputs gradients w.r.t. all seven.
struct BinGrad {
  float gx, gxl, gw, gyl, gh, gd0, gd1;
};

// Adjoint of the bin-coordinate map  (t, s, h, d0, d1) -> (y - yl, lad):
//   y - yl = h (s t^2 + d0 t(1-t)) / Q,   Q = s + (d0 + d1 - 2 s) t(1-t)
//   lad    = log(s^2 (d1 t^2 + 2 s t(1-t) + d0 (1-t)^2)) - 2 log Q          (splines.py:179-191)
// gh is the direct dependence on h (through the numerator), not the one through s = h / w.
struct CoreGrad {
  float gt, gs, gh, gd0, gd1;
};

__device__ __forceinline__ CoreGrad bin_core_vjp(float t, float s, float h, float d0, float d1, float gy, float gl) {
  const float omt = 1.f - t;
  const float a = t * omt;
  const float e = d0 + d1 - 2.f * s;
  const float inner = s * t * t + d0 * a;
  const float N = h * inner;
  const float Q = s + e * a;
  const float M = d1 * t * t + 2.f * s * a + d0 * omt * omt;
  const float rQ = 1.f / Q;
  const float g_dn = gl / (s * s * M);
  const float g_Q = -2.f * gl * rQ - gy * N * rQ * rQ;
  const float g_N = gy * rQ;
  const float g_M = s * s * g_dn;
  const float g_in = h * g_N;
  const float g_e = a * g_Q;
  const float g_a = 2.f * s * g_M + e * g_Q + d0 * g_in;
  const float g_omt = 2.f * d0 * omt * g_M + t * g_a;
  CoreGrad r;
  r.gs = 2.f * s * M * g_dn + 2.f * a * g_M + g_Q + t * t * g_in - 2.f * g_e;
  r.gt = 2.f * d1 * t * g_M + 2.f * s * t * g_in + omt * g_a - g_omt;
  r.gh = inner * g_N;
  r.gd0 = omt * omt * g_M + a * g_in + g_e;
  r.gd1 = t * t * g_M + g_e;
  return r;
}

// Density direction: y = F(x), lad = log F'(x), with t = (x - xl) / w and s = h / w.
__device__ __forceinline__ BinGrad bin_forward_vjp(float x, const RqsBin& b, float gy, float gl) {
  const float rw = 1.f / b.w;
  const float s = b.h * rw;
  const float t = (x - b.xl) * rw;
  const CoreGrad c = bin_core_vjp(t, s, b.h, b.d0, b.d1, gy, gl);
  BinGrad r;
  r.gx = c.gt * rw;
  r.gxl = -r.gx;
  r.gw = -(c.gt * t + c.gs * s) * rw;
  r.gh = c.gh + c.gs * rw;
  r.gyl = gy;
  r.gd0 = c.gd0;
  r.gd1 = c.gd1;
  return r;
}

// Sampling direction: v = xl + w r with r the root of  h phi(r; s, d0, d1) = u - yl,  lad = -log F'(v).
// The root is differentiated implicitly IN BIN COORDINATES (dr/du = 1 / (h phi_r), dr/ds = -phi_s / phi_r,
// ...): written per unit x the width gradient is a difference of two terms of size L_t / w that
// agree to several digits in narrow bins; in r they never appear.  ``x`` is u.
__device__ __forceinline__ BinGrad bin_inverse_vjp(float u, float v0, const RqsBin& b, float gy, float gl) {
  const float rw = 1.f / b.w;
  const float s = b.h * rw;
  const float target = u - b.yl;
  float r = fminf(fmaxf((v0 - b.xl) * rw, 0.f), 1.f);
  // The closed-form root leaves a residual of up to tens of ulps of u in steep bins, which the
  // curvature of log F' magnifies; two Newton steps on the fp32 forward map remove it (the
  // value the forward call returned to the caller is not changed).
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const float omr = 1.f - r, a = r * omr;
    const float Q = s + (b.d0 + b.d1 - 2.f * s) * a;
    const float f = b.h * (s * r * r + b.d0 * a) / Q;
    const float fr = b.h * s * (b.d1 * r * r + 2.f * s * a + b.d0 * omr * omr) / (Q * Q);
    r = fminf(fmaxf(r - (f - target) / fr, 0.f), 1.f);
  }
  const CoreGrad F = bin_core_vjp(r, s, b.h, b.d0, b.d1, 1.f, 0.f);
  const CoreGrad L = bin_core_vjp(r, s, b.h, b.d0, b.d1, 0.f, 1.f);
  const float gr = gy * b.w - gl * L.gt;
  const float gu = gr / F.gt;
  const float gs = -gu * F.gs - gl * L.gs;
  BinGrad o;
  o.gx = gu;
  o.gyl = -gu;
  o.gxl = gy;
  o.gw = gy * r - gs * s * rw;
  o.gh = -gu * F.gh + gs * rw;
  o.gd0 = -gu * F.gd0 - gl * L.gd0;
  o.gd1 = -gu * F.gd1 - gl * L.gd1;
  return o;
}

// PACKED: logits and their gradients are rows of P contiguous floats in the same layout (conditioner
// output, inner = 1).  A thread's row is 4 P bytes from its neighbour's, so direct accesses are 64
// scattered 4-byte transfers per instruction (2.6 TB/s measured).  In this mode the workgroup's 256
// rows - one contiguous block - travel through LDS with coalesced accesses in both directions: the
// logits are staged, every thread reads and then overwrites ITS row in LDS (odd P: conflict free), and
// the block of gradients leaves with coalesced stores.
template <int KT, bool INV, bool PACKED>
__global__ __launch_bounds__(kBwdBlock) void rqs_elementwise_bwd_kernel(const BwdArgs a) {
  extern __shared__ float stage[];                 // PACKED: [kBwdBlock][P]
  const RqsConst& c = a.c;
  const int K = KT > 0 ? KT : c.K;
  constexpr int KA = KT > 0 ? KT : kBwdMaxK;
  const int P = 2 * K + a.nd;
  for (long long base = (long long)blockIdx.x * kBwdBlock; base < a.n; base += (long long)gridDim.x * kBwdBlock) {
    const long long i = base + threadIdx.x;
    const bool active = i < a.n;
    if (PACKED) {
      const long long lim = (a.n - base < kBwdBlock ? a.n - base : kBwdBlock) * P;
      const float* src = a.uw + base * P;
      __syncthreads();                              // previous block of gradients has left
      for (long long e = threadIdx.x; e < lim; e += kBwdBlock) stage[e] = src[e];
      __syncthreads();
    }
    if (active) do {
    const float x = a.x[i];
    const float gy = a.gy[i], gl = a.glad[a.lad_div == 1 ? i : i / a.lad_div];
    const long long outer = a.inner == 1 ? i : i / a.inner, inn = a.inner == 1 ? 0 : i - outer * a.inner;
    const long long ks = PACKED ? 1 : a.ks;
    const float* uw = PACKED ? stage + threadIdx.x * P : a.uw + outer * a.row_w + inn;
    const float* uh = PACKED ? uw + K : a.uh + outer * a.row_h + inn;
    const float* ud = PACKED ? uw + 2 * K : a.ud + outer * a.row_d + inn;
    float* guw = PACKED ? stage + threadIdx.x * P : a.guw + outer * a.grow_w + inn;
    float* guh = PACKED ? guw + K : a.guh + outer * a.grow_h + inn;
    float* gud = PACKED ? guw + 2 * K : a.gud + outer * a.grow_d + inn;
    if (c.tails != 0 && !((x >= c.lo_x) && (x <= c.hi_x))) {     // identity outside: dy/dx = 1
      a.gx[i] = gy;
      for (int k = 0; k < K; ++k) { guw[k * ks] = 0.f; guh[k * ks] = 0.f; }
      for (int k = 0; k < a.nd; ++k) gud[k * ks] = 0.f;
      continue;
    }
    // ---- forward pieces: softmax probabilities and knots
    float pw[KA], ph[KA];
    float mw = -INFINITY, mh = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) { mw = fmaxf(mw, uw[k * ks]); mh = fmaxf(mh, uh[k * ks]); }
    const float sc2 = c.wh_scale * kLog2e;
    float sw = 0.f, sh = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      pw[k] = hw_exp2((uw[k * ks] - mw) * sc2);
      ph[k] = hw_exp2((uh[k * ks] - mh) * sc2);
      sw += pw[k];
      sh += ph[k];
    }
    const float rsw = div_nr(1.f, sw), rsh = div_nr(1.f, sh);
    float cw = 0.f, ch = 0.f, xl = c.lo_x, yl = c.lo_y;
    RqsBin b;
    int bin = 0;
    b.xl = xl; b.yl = yl; b.w = 1.f; b.h = 1.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      pw[k] *= rsw;
      ph[k] *= rsh;
      cw += fmaf(pw[k], c.free_w, c.min_w);
      ch += fmaf(ph[k], c.free_h, c.min_h);
      const float xr = (k == K - 1) ? c.hi_x : fmaf(c.span_x, cw, c.lo_x);
      const float yr = (k == K - 1) ? c.hi_y : fmaf(c.span_y, ch, c.lo_y);
      const bool take = (k == 0) || (INV ? (x >= yl) : (x >= xl));
      if (take) { bin = k; b.xl = xl; b.w = xr - xl; b.yl = yl; b.h = yr - yl; }
      xl = xr; yl = yr;
    }
    // knot derivatives of the bin: logit index (or boundary constant)
    const int i0 = c.tails == 1 ? bin - 1 : bin;            // logit of the left knot (-1: boundary)
    // logit of the right knot (linear: nd = boundary constant; circular: knot K shares logit 0)
    const int i1 = c.tails == 1 ? bin : (c.tails == 2 && bin + 1 == K) ? 0 : bin + 1;
    const bool has0 = i0 >= 0, has1 = i1 < a.nd;
    const float l0 = has0 ? ud[i0 * ks] : c.edge_logit, l1 = has1 ? ud[i1 * ks] : c.edge_logit;
    b.d0 = c.min_d + softplus_f(l0);
    b.d1 = c.min_d + softplus_f(l1);

    BinGrad g;
    if (!INV) {
      g = bin_forward_vjp(x, b, gy, gl);
    } else {
      // v = F^-1(u): solve as the forward kernel does, then differentiate the root
      float v, lad_inv;
      bool bad = false;
      rqs_bin_eval<true>(x, b, v, lad_inv, bad);
      g = bin_inverse_vjp(x, v, b, gy, gl);
    }
    a.gx[i] = g.gx;
    // ---- knots -> cumulative widths -> softmax -> logits
    // X_bin carries (gxl - gw), X_bin+1 carries gw; the end knots are constants.
    const float gXl = (bin >= 1) ? (g.gxl - g.gw) : 0.f;
    const float gXr = (bin + 1 <= K - 1) ? g.gw : 0.f;
    const float gYl = (bin >= 1) ? (g.gyl - g.gh) : 0.f;
    const float gYr = (bin + 1 <= K - 1) ? g.gh : 0.f;
    // dX_k/dW_i = span for i < k  ->  gW_i = span * (gXl [i < bin] + gXr [i < bin+1])
    float dotw = 0.f, doth = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float gWk = c.span_x * c.free_w * ((k < bin ? gXl : 0.f) + (k < bin + 1 ? gXr : 0.f));
      const float gHk = c.span_y * c.free_h * ((k < bin ? gYl : 0.f) + (k < bin + 1 ? gYr : 0.f));
      dotw += pw[k] * gWk;
      doth += ph[k] * gHk;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float gWk = c.span_x * c.free_w * ((k < bin ? gXl : 0.f) + (k < bin + 1 ? gXr : 0.f));
      const float gHk = c.span_y * c.free_h * ((k < bin ? gYl : 0.f) + (k < bin + 1 ? gYr : 0.f));
      guw[k * ks] = c.wh_scale * pw[k] * (gWk - dotw);
      guh[k * ks] = c.wh_scale * ph[k] * (gHk - doth);
    }
    for (int k = 0; k < a.nd; ++k) gud[k * ks] = 0.f;
    if (has0) gud[i0 * ks] += g.gd0 * softplus_grad(l0);
    if (has1) gud[i1 * ks] += g.gd1 * softplus_grad(l1);        // circular, one bin: both knots share logit 0
    } while (0);
    if (PACKED) {
      const long long lim = (a.n - base < kBwdBlock ? a.n - base : kBwdBlock) * P;
      float* dst = a.guw + base * P;
      __syncthreads();                              // every row of the block holds its gradient
      for (long long e = threadIdx.x; e < lim; e += kBwdBlock) dst[e] = stage[e];
    }
  }
}

// ------------------------------------------------------------------ logits shared by the batch
// Backward of the unconditional per-position spline (PiecewiseRationalQuadraticCDF,
// coupling.py:211-240): x [B, period], one logit row per position.  A thread owns ONE position
// for a strided subset of the samples: its knot table lives in registers (built once), every
// sample costs a bin search + the bin's reverse pass, and the knot adjoints of all its samples
// are accumulated in a private LDS strip (dynamic bin index -> ds_add_f32 on the thread's own
// words, so the summation order is fixed).  At the end the thread maps the accumulated knot
// adjoints through cumsum / floor / softmax / softplus ONCE and writes one partial row
// [group][position][P]; the host adds the few thousand partial rows.
struct SharedBwdArgs {
  const float *x, *sw, *sh, *sd;
  const float *gy, *glad;
  long long lad_div;
  float *gx, *partial;
  long long B, period, groups;     // samples, positions per sample, sample groups (= partial rows per position)
  int nd;
  RqsConst c;
};

template <int KT, bool INV>
__global__ __launch_bounds__(kBwdBlock) void rqs_shared_bwd_kernel(const SharedBwdArgs a) {
  constexpr int K = KT;
  constexpr int NA = 3 * (K + 1);                    // adjoints of X_k, Y_k, D_k, k = 0..K
  extern __shared__ float acc_lds[];                 // [NA][kBwdBlock]: word j of thread t at j * block + t
  const RqsConst& c = a.c;
  const long long t = (long long)blockIdx.x * kBwdBlock + threadIdx.x;
  float* acc = acc_lds + threadIdx.x;
#pragma unroll
  for (int j = 0; j < NA; ++j) acc[j * kBwdBlock] = 0.f;
  const long long f = t % a.period, sg = t / a.period;
  if (sg >= a.groups) return;                        // no barrier below: early exit is safe
  const float* uw = a.sw + f * K;
  const float* uh = a.sh + f * K;
  const float* ud = a.sd + f * a.nd;
  // ---- this position's spline: softmax probabilities, knots, knot derivatives (registers)
  float pw[K], ph[K], xk[K + 1], yk[K + 1], dk[K + 1], dl[K + 1];
  {
    float mw = -INFINITY, mh = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) { mw = fmaxf(mw, uw[k]); mh = fmaxf(mh, uh[k]); }
    const float sc2 = c.wh_scale * kLog2e;
    float sw = 0.f, sh = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      pw[k] = hw_exp2((uw[k] - mw) * sc2);
      ph[k] = hw_exp2((uh[k] - mh) * sc2);
      sw += pw[k];
      sh += ph[k];
    }
    const float rsw = div_nr(1.f, sw), rsh = div_nr(1.f, sh);
    float cw = 0.f, ch = 0.f;
    xk[0] = c.lo_x;
    yk[0] = c.lo_y;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      pw[k] *= rsw;
      ph[k] *= rsh;
      cw += fmaf(pw[k], c.free_w, c.min_w);
      ch += fmaf(ph[k], c.free_h, c.min_h);
      xk[k + 1] = (k == K - 1) ? c.hi_x : fmaf(c.span_x, cw, c.lo_x);
      yk[k + 1] = (k == K - 1) ? c.hi_y : fmaf(c.span_y, ch, c.lo_y);
    }
#pragma unroll
    for (int k = 0; k <= K; ++k) {
      // logit of knot k: linear tails fix knots 0 and K, circular ties knot K to logit 0
      const int j = c.tails == 1 ? k - 1 : (c.tails == 2 && k == K) ? 0 : k;
      dl[k] = (c.tails == 1 && (k == 0 || k == K)) ? c.edge_logit : ud[j];
      dk[k] = c.min_d + softplus_f(dl[k]);
    }
  }
  for (long long smp = sg; smp < a.B; smp += a.groups) {
    const long long i = smp * a.period + f;
    const float x = a.x[i];
    const float gy = a.gy[i], gl = a.glad[a.lad_div == 1 ? i : i / a.lad_div];
    if (c.tails != 0 && !((x >= c.lo_x) && (x <= c.hi_x))) {
      a.gx[i] = gy;
      continue;
    }
    int bin = 0;
#pragma unroll
    for (int k = 1; k < K; ++k) bin += (x >= (INV ? yk[k] : xk[k])) ? 1 : 0;
    RqsBin b;
    float xr = xk[1], yr = yk[1];
    b.xl = xk[0]; b.yl = yk[0]; b.d0 = dk[0]; b.d1 = dk[1];
#pragma unroll
    for (int k = 1; k < K; ++k) {
      if (bin == k) { b.xl = xk[k]; xr = xk[k + 1]; b.yl = yk[k]; yr = yk[k + 1]; b.d0 = dk[k]; b.d1 = dk[k + 1]; }
    }
    b.w = xr - b.xl;
    b.h = yr - b.yl;
    BinGrad g;
    if (!INV) {
      g = bin_forward_vjp(x, b, gy, gl);
    } else {
      float v, lad_inv;
      bool bad = false;
      rqs_bin_eval<true>(x, b, v, lad_inv, bad);
      g = bin_inverse_vjp(x, v, b, gy, gl);
    }
    a.gx[i] = g.gx;
    // knot adjoints: X_bin += gxl - gw, X_bin+1 += gw (same for Y), D_bin += gd0, D_bin+1 += gd1
    acc[(bin) * kBwdBlock] += g.gxl - g.gw;
    acc[(bin + 1) * kBwdBlock] += g.gw;
    acc[(K + 1 + bin) * kBwdBlock] += g.gyl - g.gh;
    acc[(K + 1 + bin + 1) * kBwdBlock] += g.gh;
    acc[(2 * (K + 1) + bin) * kBwdBlock] += g.gd0;
    acc[(2 * (K + 1) + bin + 1) * kBwdBlock] += g.gd1;
  }
  // ---- knot adjoints -> logits, once per thread
  float* out = a.partial + (sg * a.period + f) * (2 * K + a.nd);
  {
    // dX_k / dW_i = span for i < k (interior knots k = 1..K-1; knots 0 and K are constants):
    // g_W[i] = span * free * sum_{k > i, k < K} aX[k];  g_uw = scale * p (g_W - <p, g_W>)
    float gW[K], gH[K];
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int i = K - 1; i >= 0; --i) {
      gW[i] = c.span_x * c.free_w * sx;
      gH[i] = c.span_y * c.free_h * sy;
      if (i >= 1) {
        sx += acc[i * kBwdBlock];
        sy += acc[(K + 1 + i) * kBwdBlock];
      }
    }
    float dotw = 0.f, doth = 0.f;
#pragma unroll
    for (int i = 0; i < K; ++i) { dotw += pw[i] * gW[i]; doth += ph[i] * gH[i]; }
#pragma unroll
    for (int i = 0; i < K; ++i) {
      out[i] = c.wh_scale * pw[i] * (gW[i] - dotw);
      out[K + i] = c.wh_scale * ph[i] * (gH[i] - doth);
    }
    for (int j = 0; j < a.nd; ++j) out[2 * K + j] = 0.f;
#pragma unroll
    for (int k = 0; k <= K; ++k) {
      const int j = c.tails == 1 ? k - 1 : (c.tails == 2 && k == K) ? 0 : k;
      if (!(c.tails == 1 && (k == 0 || k == K)))
        out[2 * K + j] += acc[(2 * (K + 1) + k) * kBwdBlock] * softplus_grad(dl[k]);
    }
  }
}

template <bool INV>
static int launch_shared_bwd(const SharedBwdArgs& a, dim3 grid, hipStream_t st) {
  const size_t lds = (size_t)3 * (a.c.K + 1) * kBwdBlock * sizeof(float);
  switch (a.c.K) {
    case 4: hipLaunchKernelGGL((rqs_shared_bwd_kernel<4, INV>), grid, dim3(kBwdBlock), lds, st, a); break;
    case 8: hipLaunchKernelGGL((rqs_shared_bwd_kernel<8, INV>), grid, dim3(kBwdBlock), lds, st, a); break;
    case 10: hipLaunchKernelGGL((rqs_shared_bwd_kernel<10, INV>), grid, dim3(kBwdBlock), lds, st, a); break;
    case 16: hipLaunchKernelGGL((rqs_shared_bwd_kernel<16, INV>), grid, dim3(kBwdBlock), lds, st, a); break;
    default: return VCNF_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

template <bool INV, bool PACKED>
static void launch_bwd(const BwdArgs& a, dim3 grid, hipStream_t st) {
  const size_t lds = PACKED ? (size_t)kBwdBlock * (2 * a.c.K + a.nd) * sizeof(float) : 0;
  switch (a.c.K) {
    case 4: hipLaunchKernelGGL((rqs_elementwise_bwd_kernel<4, INV, PACKED>), grid, dim3(kBwdBlock), lds, st, a); break;
    case 8: hipLaunchKernelGGL((rqs_elementwise_bwd_kernel<8, INV, PACKED>), grid, dim3(kBwdBlock), lds, st, a); break;
    case 10: hipLaunchKernelGGL((rqs_elementwise_bwd_kernel<10, INV, PACKED>), grid, dim3(kBwdBlock), lds, st, a); break;
    case 16: hipLaunchKernelGGL((rqs_elementwise_bwd_kernel<16, INV, PACKED>), grid, dim3(kBwdBlock), lds, st, a); break;
    default: hipLaunchKernelGGL((rqs_elementwise_bwd_kernel<0, INV, PACKED>), grid, dim3(kBwdBlock), lds, st, a); break;
  }
}

}  // namespace vcnf

using namespace vcnf;

static int bwd_common(const vcnf_rqs_cfg* cfg, BwdArgs& a, int64_t n) {
  if (!cfg) return VCNF_ERR_NULL;
  const int K = cfg->num_bins;
  if (K < 1 || K > kBwdMaxK) return VCNF_ERR_SHAPE;
  if (cfg->tails != VCNF_TAILS_NONE && cfg->tails != VCNF_TAILS_LINEAR && cfg->tails != VCNF_TAILS_CIRCULAR)
    return VCNF_ERR_UNSUPPORTED;
  if (cfg->tails == VCNF_TAILS_LINEAR && K < 2) return VCNF_ERR_SHAPE;
  if ((double)cfg->min_bin_width * K > 1.0 || (double)cfg->min_bin_height * K > 1.0) return VCNF_ERR_VALUE;
  if (n < 0) return VCNF_ERR_SHAPE;
  a.n = n;
  a.nd = cfg->tails == VCNF_TAILS_LINEAR ? K - 1 : cfg->tails == VCNF_TAILS_CIRCULAR ? K : K + 1;
  RqsConst& c = a.c;
  c.K = K; c.tails = cfg->tails;
  c.lo_x = cfg->left; c.hi_x = cfg->right; c.span_x = (float)((double)cfg->right - (double)cfg->left);
  c.lo_y = cfg->bottom; c.hi_y = cfg->top; c.span_y = (float)((double)cfg->top - (double)cfg->bottom);
  c.min_w = cfg->min_bin_width; c.min_h = cfg->min_bin_height; c.min_d = cfg->min_derivative;
  c.free_w = (float)(1.0 - (double)cfg->min_bin_width * K);
  c.free_h = (float)(1.0 - (double)cfg->min_bin_height * K);
  c.wh_scale = cfg->wh_scale;
  c.edge_logit = (float)log(exp(1.0 - (double)cfg->min_derivative) - 1.0);
  return VCNF_OK;
}

static int bwd_launch(const BwdArgs& a, int inverse, void* stream) {
  const long long blocks = (a.n + kBwdBlock - 1) / kBwdBlock;
  dim3 grid((unsigned)(blocks < 256 * 16 ? blocks : 256 * 16));
  // rows of logits and of gradients packed alike and small enough for LDS: staged, coalesced variant
  const long long P = 2 * a.c.K + a.nd;
  const bool packed = a.inner == 1 && a.ks == 1 && a.row_w == P && a.row_h == P && a.row_d == P &&
                      a.grow_w == P && a.grow_h == P && a.grow_d == P && a.uh == a.uw + a.c.K &&
                      a.ud == a.uw + 2 * a.c.K && a.guh == a.guw + a.c.K && a.gud == a.guw + 2 * a.c.K &&
                      (size_t)kBwdBlock * P * sizeof(float) <= 48 * 1024;
  hipStream_t st = (hipStream_t)stream;
  if (packed) {
    if (inverse) launch_bwd<true, true>(a, grid, st);
    else launch_bwd<false, true>(a, grid, st);
  } else {
    if (inverse) launch_bwd<true, false>(a, grid, st);
    else launch_bwd<false, false>(a, grid, st);
  }
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

extern "C" int vcnf_rqs_elementwise_bwd_f32(const float* x, const float* uw, const float* uh, const float* ud,
                                            int64_t ld_w, int64_t ld_h, int64_t ld_d,
                                            const float* g_y, const float* g_logabsdet,
                                            float* g_x, float* g_uw, float* g_uh, float* g_ud, int64_t n,
                                            const vcnf_rqs_cfg* cfg, int inverse, void* stream) {
  BwdArgs a;
  const int rc = bwd_common(cfg, a, n);
  if (rc != VCNF_OK) return rc;
  if (ld_w < 0 || ld_h < 0 || ld_d < 0) return VCNF_ERR_SHAPE;
  if (n == 0) return VCNF_OK;
  if (!x || !uw || !uh || !ud || !g_y || !g_logabsdet || !g_x || !g_uw || !g_uh || !g_ud) return VCNF_ERR_NULL;
  a.x = x; a.uw = uw; a.uh = uh; a.ud = ud; a.row_w = ld_w; a.row_h = ld_h; a.row_d = ld_d; a.inner = 1; a.ks = 1;
  a.gy = g_y; a.glad = g_logabsdet; a.lad_div = 1;
  a.gx = g_x; a.guw = g_uw; a.guh = g_uh; a.gud = g_ud;
  a.grow_w = a.c.K; a.grow_h = a.c.K; a.grow_d = a.nd;
  return bwd_launch(a, inverse, stream);
}

extern "C" int vcnf_rqs_packed_bwd_f32(const float* x, const float* params, int64_t inner, int64_t lad_div,
                                       const float* g_y, const float* g_logabsdet,
                                       float* g_x, float* g_params, int64_t n,
                                       const vcnf_rqs_cfg* cfg, int inverse, void* stream) {
  BwdArgs a;
  const int rc = bwd_common(cfg, a, n);
  if (rc != VCNF_OK) return rc;
  if (inner < 1 || lad_div < 1) return VCNF_ERR_SHAPE;
  if (n == 0) return VCNF_OK;
  if (!x || !params || !g_y || !g_logabsdet || !g_x || !g_params) return VCNF_ERR_NULL;
  const long long K = a.c.K, P = 2 * K + a.nd;
  a.x = x; a.uw = params; a.uh = params + K * inner; a.ud = params + 2 * K * inner;
  a.row_w = a.row_h = a.row_d = P * inner; a.inner = inner; a.ks = inner;
  a.gy = g_y; a.glad = g_logabsdet; a.lad_div = lad_div;
  a.gx = g_x; a.guw = g_params; a.guh = g_params + K * inner; a.gud = g_params + 2 * K * inner;
  a.grow_w = a.grow_h = a.grow_d = P * inner;
  return bwd_launch(a, inverse, stream);
}

/* Number of partial rows per position vcnf_rqs_shared_bwd_f32 writes for a given problem. */
extern "C" int64_t vcnf_rqs_shared_bwd_groups(int64_t batch, int64_t period) {
  if (batch < 1 || period < 1) return 0;
  long long g = (256LL * 1024) / period;
  if (g < 1) g = 1;
  if (g > batch) g = batch;
  return g;
}

extern "C" int vcnf_rqs_shared_bwd_f32(const float* x, const float* sw, const float* sh, const float* sd,
                                       int64_t batch, int64_t period, int64_t lad_div,
                                       const float* g_y, const float* g_logabsdet,
                                       float* g_x, float* partial, int64_t groups,
                                       const vcnf_rqs_cfg* cfg, int inverse, void* stream) {
  BwdArgs tmp;
  const int rc = bwd_common(cfg, tmp, batch);
  if (rc != VCNF_OK) return rc;
  if (period < 1 || lad_div < 1 || groups != vcnf_rqs_shared_bwd_groups(batch, period)) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!x || !sw || !sh || !sd || !g_y || !g_logabsdet || !g_x || !partial) return VCNF_ERR_NULL;
  SharedBwdArgs a;
  a.x = x; a.sw = sw; a.sh = sh; a.sd = sd; a.gy = g_y; a.glad = g_logabsdet; a.lad_div = lad_div;
  a.gx = g_x; a.partial = partial; a.B = batch; a.period = period; a.groups = groups; a.nd = tmp.nd; a.c = tmp.c;
  const long long threads = groups * period;
  dim3 grid((unsigned)((threads + kBwdBlock - 1) / kBwdBlock));
  return inverse ? launch_shared_bwd<true>(a, grid, (hipStream_t)stream)
                 : launch_shared_bwd<false>(a, grid, (hipStream_t)stream);
}
