// Device-side rational-quadratic spline arithmetic for gfx950.
//
// One spline evaluation = (1) turn K width logits and K height logits into knot
// positions (softmax, floor, running sum, affine map onto the interval, exact end
// knots), (2) find the bin of the input by comparing it with the left knots while
// they are generated, (3) evaluate the rational-quadratic map or its inverse and
// log|dy/dx| in that bin.  Follows normflow/utils/splines.py:88-193 of the
// reference in operation order; nothing is materialised per bin: the K+1 knots live
// in registers for one loop trip each and only the selected bin survives.
//
// Floating-point contraction is switched off in this header and every fused
// multiply-add is written out (fmaf): the same source then produces the same bits
// in every kernel it is inlined into (the sampling direction evaluates the shared
// spline in two kernels and the results must agree exactly).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#pragma clang fp contract(off)

namespace vcnf {

struct RqsConst {
  int K;
  int tails;              // 0 none (K+1 derivative logits), 1 linear (K-1, identity outside),
                          // 2 circular (K: the last knot shares the first knot's logit, identity outside)
  float lo_x, hi_x, span_x;
  float lo_y, hi_y, span_y;
  float min_w, min_h, min_d;
  float free_w, free_h;   // 1 - min*K, rounded from double like the reference's Python scalar
  float wh_scale;
  float edge_logit;       // log(exp(1 - min_d) - 1), splines.py:38
};

// Hardware transcendentals (v_exp_f32 = 2^x, v_log_f32 = log2, v_rcp_f32, v_sqrt_f32: 1 ulp
// each).  The coupling kernel stays HBM-bound only if the ~30 transcendental calls per
// element are one or two instructions each; the IEEE-exact library forms cost ~10 apiece
// (measured: 1.07 ms -> 0.78 ms per 1M x 64 layer).
constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;
__device__ __forceinline__ float hw_exp2(float v) { return __builtin_amdgcn_exp2f(v); }
__device__ __forceinline__ float hw_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float hw_log(float v) { return __builtin_amdgcn_logf(v) * kLn2; }
__device__ __forceinline__ float hw_sqrt(float v) { return __builtin_amdgcn_sqrtf(v); }
// a / b to ~1 ulp: reciprocal estimate plus one Newton correction of the quotient
__device__ __forceinline__ float div_nr(float a, float b) {
  const float r = hw_rcp(b);
  const float q = a * r;
  return fmaf(fmaf(-b, q, a), r, q);
}

// torch.nn.functional.softplus (beta 1, threshold 20), splines.py:121.  log1p(e) is
// evaluated as log(u) * e / (u - 1), u = 1 + e, which keeps full relative accuracy for
// tiny e (derivatives near the 1e-3 floor are the ill-conditioned part of the spline).
__device__ __forceinline__ float softplus_f(float v) {
  const float e = hw_exp2(v * kLog2e);
  const float u = 1.f + e;
  const float um1 = u - 1.f;
  const float l = (um1 == 0.f) ? e : hw_log(u) * div_nr(e, um1);
  return v > 20.f ? v : l;
}

// Selected-bin quantities: left x knot, width, left y knot, height, knot derivatives.
// b^2 - 4ac of a monotone rational-quadratic bin is positive, but next to a bin edge it equals (h d)^2 while its two
// terms are of size (2 s h)^2: with a floor-level derivative d ~ 1e-3 beside a steep bin (s ~ 30) the true value is
// ~1e-9 of the terms - below fp32 resolution - and the rounding of a, b, c (each a few 1e-7 relative) leaves a result of
// either sign at a few 1e-7 .. 1e-6 of (b^2 + |4ac|).  Seen in the 524 288 x 1024 x 24 shard of config C5 (1.3e10
// evaluations): one element of layer 18 whose sign depends on the accumulation order of the conditioner's last layer.
// The reference asserts disc >= 0 on its own fp32 rounding (splines.py:163-164) and would stop on such an element if
// its rounding fell the same way; here a negative value within 2^-14 of (b^2 + |4ac|) IS the double root (0, the
// point sits on the bin edge) and is not counted; invalid spline parameters give a discriminant of the order of
// -(b^2 + |4ac|), which is counted and yields NaN as before.
__device__ __forceinline__ float rqs_rounding_level_zero(float disc, float qb, float p) {
  const float scale = fmaf(qb, qb, __builtin_fabsf(p));
  return (disc < 0.f && disc >= -6.103515625e-5f * scale) ? 0.f : disc;
}

struct RqsBin {
  float xl, w, yl, h, d0, d1;
};

// splines.py:179-193 (forward) and :152-177 (inverse) on the selected bin.
template <bool INV>
__device__ __forceinline__ void rqs_bin_eval(float x, const RqsBin& b, float& y, float& lad, bool& bad) {
  const float s = div_nr(b.h, b.w);                // :144
  const float e = fmaf(-2.f, s, b.d0 + b.d1);      // d0 + d1 - 2 s
  if (!INV) {
    const float t = div_nr(x - b.xl, b.w);         // :179
    const float omt = 1.f - t;
    const float tt = t * omt;
    const float num = b.h * fmaf(s * t, t, b.d0 * tt);
    const float den = fmaf(e, tt, s);
    y = b.yl + div_nr(num, den);                   // :186
    const float dn = (s * s) * fmaf(b.d1 * t, t, fmaf(2.f * s, tt, (b.d0 * omt) * omt));
    lad = fmaf(-2.f, hw_log(den), hw_log(dn));     // :191
  } else {
    const float dy = x - b.yl;
    const float qa = fmaf(dy, e, b.h * (s - b.d0));   // :153-156
    const float qb = fmaf(-dy, e, b.h * b.d0);        // :157-160
    const float qc = -s * dy;                         // :161
    // b^2 - 4ac with the rounding error of the product 4ac recovered by an fma (Kahan):
    // the discriminant is where the inverse loses digits when b^2 ~ 4ac
    const float fa = 4.f * qa;
    const float p = fa * qc;
    const float perr = fmaf(fa, qc, -p);
    const float disc = rqs_rounding_level_zero(fmaf(qb, qb, -p) - perr, qb, p);    // :163
    bad = bad || !(disc >= 0.f);                   // :164 (the reference asserts)
    const float r = div_nr(2.f * qc, -qb - hw_sqrt(disc));   // :166
    y = fmaf(r, b.w, b.xl);                        // :167
    const float omr = 1.f - r;
    const float rr = r * omr;
    const float den = fmaf(e, rr, s);
    const float dn = (s * s) * fmaf(b.d1 * r, r, fmaf(2.f * s, rr, (b.d0 * omr) * omr));
    lad = fmaf(2.f, hw_log(den), -hw_log(dn));     // :175-177
  }
}

// Generate the knots of one element and select the bin of x.  P supplies the raw
// logits: w(k), h(k) for k < K (NOT yet multiplied by the 1/sqrt(hidden) scale: the
// scale and log2(e) are folded into one factor of the exp2 argument, ``sc2``), d(k)
// for k <= K (boundary logits included).  KT > 0: K known at compile time, the
// exponentials are kept in registers; KT == 0: runtime K, recomputed per pass.
template <int KT, bool INV, class P>
__device__ __forceinline__ void rqs_select(float x, const P& p, const RqsConst& c, float sc2, RqsBin& sel) {
  const int K = KT > 0 ? KT : c.K;
  constexpr int KR = KT > 0 ? KT : 1;
  float ew[KR], eh[KR];
  float mw = -INFINITY, mh = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    mw = fmaxf(mw, p.w(k));
    mh = fmaxf(mh, p.h(k));
  }
  float sw = 0.f, sh = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float a = hw_exp2((p.w(k) - mw) * sc2), b = hw_exp2((p.h(k) - mh) * sc2);
    if (KT > 0) { ew[k] = a; eh[k] = b; }
    sw += a;
    sh += b;
  }
  // softmax normaliser folded with (1 - min*K): width_k = min + e_k * gw   (:109-110)
  const float gw = div_nr(c.free_w, sw), gh = div_nr(c.free_h, sh);
  float cw = 0.f, ch = 0.f;
  float xl = c.lo_x, yl = c.lo_y, dl = p.d(0);
  sel.xl = xl; sel.yl = yl; sel.w = 1.f; sel.h = 1.f; sel.d0 = dl; sel.d1 = dl;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float a = KT > 0 ? ew[k] : hw_exp2((p.w(k) - mw) * sc2);
    const float b = KT > 0 ? eh[k] : hw_exp2((p.h(k) - mh) * sc2);
    cw += fmaf(a, gw, c.min_w);                    // :110-111 / :124-125
    ch += fmaf(b, gh, c.min_h);
    const float xr = (k == K - 1) ? c.hi_x : fmaf(c.span_x, cw, c.lo_x);   // :116-118
    const float yr = (k == K - 1) ? c.hi_y : fmaf(c.span_y, ch, c.lo_y);   // :130-132
    const float dr = p.d(k + 1);
    // searchsorted (:12-17): last left knot that is <= x; bin 0 is the floor.
    const bool take = (k == 0) || (INV ? (x >= yl) : (x >= xl));
    if (take) {
      sel.xl = xl; sel.w = xr - xl;                // :119, :140-141
      sel.yl = yl; sel.h = yr - yl;                // :133, :143, :150
      sel.d0 = dl; sel.d1 = dr;                    // :147-148 (still logits)
    }
    xl = xr; yl = yr; dl = dr;
  }
  sel.d0 = c.min_d + softplus_f(sel.d0);           // :121
  sel.d1 = c.min_d + softplus_f(sel.d1);
}

// rqs_select<K, INV> cut into stages with the same arithmetic in the same order, so that a kernel
// can place other work (matrix instructions) between the stages: maxima + exponentials of bins
// [k0, k1), normalisers, knots + bin search over [k0, k1), derivatives.
template <int K, bool INV>
struct RqsStaged {
  float ew[K], eh[K];
  float mw, mh, sw, sh, gw, gh, cw, ch, xl, yl, dl;
  RqsBin sel;
  template <class P>
  __device__ __forceinline__ void maxima(const P& p) {
    mw = -INFINITY;
    mh = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      mw = fmaxf(mw, p.w(k));
      mh = fmaxf(mh, p.h(k));
    }
    sw = 0.f;
    sh = 0.f;
  }
  // bins [k0, k1): the bounds are constants at every (inlined, unrolled) call site
  template <class P>
  __device__ __forceinline__ void exps(const P& p, float sc2, int k0, int k1) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k >= k0 && k < k1) {
        ew[k] = hw_exp2((p.w(k) - mw) * sc2);
        eh[k] = hw_exp2((p.h(k) - mh) * sc2);
        sw += ew[k];
        sh += eh[k];
      }
    }
  }
  template <class P>
  __device__ __forceinline__ void normalise(const P& p, const RqsConst& c) {
    gw = div_nr(c.free_w, sw);
    gh = div_nr(c.free_h, sh);
    cw = 0.f;
    ch = 0.f;
    xl = c.lo_x;
    yl = c.lo_y;
    dl = p.d(0);
    sel.xl = xl; sel.yl = yl; sel.w = 1.f; sel.h = 1.f; sel.d0 = dl; sel.d1 = dl;
  }
  template <class P>
  __device__ __forceinline__ void knots(float x, const P& p, const RqsConst& c, int k0, int k1) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k < k0 || k >= k1) continue;
      cw += fmaf(ew[k], gw, c.min_w);
      ch += fmaf(eh[k], gh, c.min_h);
      const float xr = (k == K - 1) ? c.hi_x : fmaf(c.span_x, cw, c.lo_x);
      const float yr = (k == K - 1) ? c.hi_y : fmaf(c.span_y, ch, c.lo_y);
      const float dr = p.d(k + 1);
      const bool take = (k == 0) || (INV ? (x >= yl) : (x >= xl));
      // plain selects (no branch: the stages share scheduling regions with matrix instructions)
      sel.xl = take ? xl : sel.xl;
      sel.w = take ? xr - xl : sel.w;
      sel.yl = take ? yl : sel.yl;
      sel.h = take ? yr - yl : sel.h;
      sel.d0 = take ? dl : sel.d0;
      sel.d1 = take ? dr : sel.d1;
      xl = xr; yl = yr; dl = dr;
    }
  }
  __device__ __forceinline__ void derivatives(const RqsConst& c) {
    sel.d0 = c.min_d + softplus_f(sel.d0);
    sel.d1 = c.min_d + softplus_f(sel.d1);
  }
};

// Full evaluation with tails handling (splines.py:30-43: outside -> identity, 0).
template <int KT, bool INV, class P>
__device__ __forceinline__ void rqs_point(float x, const P& p, const RqsConst& c,
                                          float& y, float& lad, bool& bad) {
  if (c.tails != 0 && !((x >= c.lo_x) && (x <= c.hi_x))) {
    y = x;
    lad = 0.f;
    return;
  }
  RqsBin sel;
  rqs_select<KT, INV>(x, p, c, p.scale * kLog2e, sel);
  rqs_bin_eval<INV>(x, sel, y, lad, bad);
}

// Logits of one element stored as [K widths | K heights | derivative logits]
// (the reference's transform_params[..., :] layout, coupling.py:310-312).
struct PackedLogits {
  const float* q;
  int K;
  float scale, edge;
  int tails;
  __device__ __forceinline__ float w(int k) const { return q[k]; }
  __device__ __forceinline__ float h(int k) const { return q[K + k]; }
  __device__ __forceinline__ float d(int k) const {
    if (tails == 1) return (k == 0 || k == K) ? edge : q[2 * K + k - 1];   // :37-40
    if (tails == 2) return q[2 * K + (k == K ? 0 : k)];                      // :45-46
    return q[2 * K + k];
  }
};

// Three separately strided logit rows (functional API, splines.py:20-29).
struct SplitLogits {
  const float *pw, *ph, *pd;
  int K;
  float scale, edge;
  int tails;
  __device__ __forceinline__ float w(int k) const { return pw[k]; }
  __device__ __forceinline__ float h(int k) const { return ph[k]; }
  __device__ __forceinline__ float d(int k) const {
    if (tails == 1) return (k == 0 || k == K) ? edge : pd[k - 1];
    if (tails == 2) return pd[k == K ? 0 : k];
    return pd[k];
  }
};

// Logit rows with an element stride between consecutive k (image layout: the conditioner emits
// [B, C*P, H, W], so the P logits of one pixel are H*W floats apart; coupling.py:148-151).
struct StridedLogits {
  const float *pw, *ph, *pd;
  long long ks;
  int K;
  float scale, edge;
  int tails;
  __device__ __forceinline__ float w(int k) const { return pw[k * ks]; }
  __device__ __forceinline__ float h(int k) const { return ph[k * ks]; }
  __device__ __forceinline__ float d(int k) const {
    if (tails == 1) return (k == 0 || k == K) ? edge : pd[(k - 1) * ks];
    if (tails == 2) return pd[(k == K ? 0 : k) * ks];
    return pd[k * ks];
  }
};

// Per-feature knot table for batch-shared logits (PiecewiseRationalQuadraticCDF,
// coupling.py:165-246): xk[K+1] | yk[K+1] | dk[K+1], built once per workgroup with
// the same arithmetic as rqs_select, so table and direct evaluation agree bitwise.
// `stride`: distance in floats between consecutive table entries (1 = the row layout above; 64 = one
// LDS column per feature of a 64-feature chunk, rqs_identity_half_kernel).  The three columns are independent of each
// other: rqs_build_table_part builds one (0 x knots, 1 y knots, 2 derivatives), so three waves can share a table.
template <class P>
__device__ __forceinline__ void rqs_build_table_part(const P& p, const RqsConst& c, float* tab, int stride, int part) {
  const int K = c.K;
  float* col = tab + part * (K + 1) * stride;
  if (part == 2) {
    for (int k = 0; k <= K; ++k) col[k * stride] = c.min_d + softplus_f(p.d(k));
    return;
  }
  const bool xs = part == 0;
  float m = -INFINITY;
  for (int k = 0; k < K; ++k) m = fmaxf(m, xs ? p.w(k) : p.h(k));
  const float sc2 = p.scale * kLog2e;
  float sum = 0.f;
  for (int k = 0; k < K; ++k) sum += hw_exp2(((xs ? p.w(k) : p.h(k)) - m) * sc2);
  const float g = div_nr(xs ? c.free_w : c.free_h, sum);
  const float mn = xs ? c.min_w : c.min_h, span = xs ? c.span_x : c.span_y, lo = xs ? c.lo_x : c.lo_y;
  float cum = 0.f;
  col[0] = lo;
  for (int k = 0; k < K; ++k) {
    cum += fmaf(hw_exp2(((xs ? p.w(k) : p.h(k)) - m) * sc2), g, mn);
    col[(k + 1) * stride] = (k == K - 1) ? (xs ? c.hi_x : c.hi_y) : fmaf(span, cum, lo);
  }
}

// rqs_build_table_part with the bin count known at compile time, in two pieces: the column's logits are fetched first
// (independent loads: ONE memory latency - the run-time loops above wait for a load in every iteration, 7 us per
// workgroup measured in the fused layer kernels, profiles/r03_small_batch.md), then the same operations in the same
// order: bitwise the same table.  v: K width / height logits (parts 0 / 1) or the K + 1 derivative logits (part 2).
template <int K, class P>
__device__ __forceinline__ void rqs_table_column_load(const P& p, int part, float (&v)[K + 1]) {
#pragma unroll
  for (int k = 0; k <= K; ++k) v[k] = part == 2 ? p.d(k) : (k < K ? (part == 0 ? p.w(k) : p.h(k)) : 0.f);
}

template <int K>
__device__ __forceinline__ void rqs_table_column_build(const float (&v)[K + 1], float scale, const RqsConst& c, float* tab,
                                                       int stride, int part) {
  float* col = tab + part * (K + 1) * stride;
  if (part == 2) {
#pragma unroll
    for (int k = 0; k <= K; ++k) col[k * stride] = c.min_d + softplus_f(v[k]);
    return;
  }
  const bool xs = part == 0;
  float e[K];
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) m = fmaxf(m, v[k]);
  const float sc2 = scale * kLog2e;
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    e[k] = hw_exp2((v[k] - m) * sc2);
    sum += e[k];
  }
  const float g = div_nr(xs ? c.free_w : c.free_h, sum);
  const float mn = xs ? c.min_w : c.min_h, span = xs ? c.span_x : c.span_y, lo = xs ? c.lo_x : c.lo_y;
  float cum = 0.f;
  col[0] = lo;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    cum += fmaf(e[k], g, mn);
    col[(k + 1) * stride] = (k == K - 1) ? (xs ? c.hi_x : c.hi_y) : fmaf(span, cum, lo);
  }
}

template <int K, class P>
__device__ __forceinline__ void rqs_build_table_part_k(const P& p, const RqsConst& c, float* tab, int stride, int part) {
  float v[K + 1];
  rqs_table_column_load<K>(p, part, v);
  rqs_table_column_build<K>(v, p.scale, c, tab, stride, part);
}

template <class P>
__device__ __forceinline__ void rqs_build_table(const P& p, const RqsConst& c, float* tab, int stride = 1) {
#pragma unroll 1
  for (int part = 0; part < 3; ++part) rqs_build_table_part(p, c, tab, stride, part);
}

// KT > 0: bin count known at compile time (the knot scan unrolls and its LDS reads batch).
template <bool INV, int KT = 0>
__device__ __forceinline__ void rqs_point_table(float x, const float* tab, const RqsConst& c,
                                                float& y, float& lad, bool& bad) {
  if (c.tails != 0 && !((x >= c.lo_x) && (x <= c.hi_x))) {
    y = x;
    lad = 0.f;
    return;
  }
  const int K = KT > 0 ? KT : c.K;
  const float* xk = tab;
  const float* yk = tab + (K + 1);
  const float* dk = tab + 2 * (K + 1);
  int bin = 0;
  const float* key = INV ? yk : xk;
  if (KT > 0) {
#pragma unroll
    for (int k = 1; k < KT; ++k) bin += (x >= key[k]) ? 1 : 0;   // knots ascend: count = last hit
  } else {
    for (int k = 1; k < K; ++k) bin += (x >= key[k]) ? 1 : 0;
  }
  RqsBin b;
  b.xl = xk[bin];
  b.w = xk[bin + 1] - b.xl;
  b.yl = yk[bin];
  b.h = yk[bin + 1] - b.yl;
  b.d0 = dk[bin];
  b.d1 = dk[bin + 1];
  rqs_bin_eval<INV>(x, b, y, lad, bad);
}

// rqs_point_table for a point already known to lie inside the interval (no tails branch): the caller
// selects the identity for outside points afterwards.  KT > 0 only.
template <bool INV, int KT>
__device__ __forceinline__ void rqs_point_table_inside(float x, const float* tab, float& y, float& lad, bool& bad) {
  const float* xk = tab;
  const float* yk = tab + (KT + 1);
  const float* dk = tab + 2 * (KT + 1);
  const float* key = INV ? yk : xk;
  int bin = 0;
#pragma unroll
  for (int k = 1; k < KT; ++k) bin += (x >= key[k]) ? 1 : 0;
  RqsBin b;
  b.xl = xk[bin];
  b.w = xk[bin + 1] - b.xl;
  b.yl = yk[bin];
  b.h = yk[bin + 1] - b.yl;
  b.d0 = dk[bin];
  b.d1 = dk[bin + 1];
  rqs_bin_eval<INV>(x, b, y, lad, bad);
}

}  // namespace vcnf
