// Device-side rational-quadratic spline arithmetic for gfx950.
//
// One spline evaluation = (1) turn K width logits and K height logits into knot
// positions (softmax, floor, running sum, affine map onto the interval, exact end
// knots), (2) find the bin of the input by comparing it with the left knots while
// they are generated, (3) evaluate the rational-quadratic map or its inverse and
// log|dy/dx| in that bin.  Follows normflow/utils/splines.py:88-193 of the
// reference in operation order; nothing is materialised per bin: the K+1 knots live
// in registers for one loop trip each and only the selected bin survives.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace vcnf {

struct RqsConst {
  int K;
  int tails;              // 0 none (K+1 derivative logits), 1 linear (K-1, identity outside)
  float lo_x, hi_x, span_x;
  float lo_y, hi_y, span_y;
  float min_w, min_h, min_d;
  float free_w, free_h;   // 1 - min*K, rounded from double like the reference's Python scalar
  float wh_scale;
  float edge_logit;       // log(exp(1 - min_d) - 1), splines.py:38
};

// torch.nn.functional.softplus (beta 1, threshold 20), splines.py:121
__device__ __forceinline__ float softplus_f(float v) { return v > 20.f ? v : log1pf(expf(v)); }

// Selected-bin quantities: left x knot, width, left y knot, height, knot derivatives.
struct RqsBin {
  float xl, w, yl, h, d0, d1;
};

// splines.py:179-193 (forward) and :152-177 (inverse) on the selected bin.
template <bool INV>
__device__ __forceinline__ void rqs_bin_eval(float x, const RqsBin& b, float& y, float& lad, bool& bad) {
  const float s = b.h / b.w;                       // :144
  const float e = b.d0 + b.d1 - 2.f * s;
  if (!INV) {
    const float t = (x - b.xl) / b.w;              // :179
    const float tt = t * (1.f - t);
    const float num = b.h * (s * t * t + b.d0 * tt);
    const float den = s + e * tt;
    y = b.yl + num / den;                          // :186
    const float omt = 1.f - t;
    const float dn = s * s * (b.d1 * t * t + 2.f * s * tt + b.d0 * omt * omt);
    lad = logf(dn) - 2.f * logf(den);              // :191
  } else {
    const float dy = x - b.yl;
    const float qa = dy * e + b.h * (s - b.d0);    // :153-156
    const float qb = b.h * b.d0 - dy * e;          // :157-160
    const float qc = -s * dy;                      // :161
    const float disc = qb * qb - 4.f * qa * qc;    // :163
    bad = bad || !(disc >= 0.f);                   // :164 (the reference asserts)
    const float r = (2.f * qc) / (-qb - sqrtf(disc));   // :166
    y = r * b.w + b.xl;                            // :167
    const float rr = r * (1.f - r);
    const float den = s + e * rr;
    const float omr = 1.f - r;
    const float dn = s * s * (b.d1 * r * r + 2.f * s * rr + b.d0 * omr * omr);
    lad = -(logf(dn) - 2.f * logf(den));           // :175-177
  }
}

// Generate the knots of one element and select the bin of x.  P supplies the raw
// logits: w(k), h(k) for k < K (already multiplied by wh_scale), d(k) for k <= K
// (boundary logits included).  KT > 0: K known at compile time, the exponentials
// are kept in registers; KT == 0: runtime K, exponentials recomputed per pass.
template <int KT, bool INV, class P>
__device__ __forceinline__ void rqs_select(float x, const P& p, const RqsConst& c, RqsBin& sel) {
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
    const float a = expf(p.w(k) - mw), b = expf(p.h(k) - mh);
    if (KT > 0) { ew[k] = a; eh[k] = b; }
    sw += a;
    sh += b;
  }
  const float rw = 1.f / sw, rh = 1.f / sh;
  float cw = 0.f, ch = 0.f;
  float xl = c.lo_x, yl = c.lo_y, dl = p.d(0);
  sel.xl = xl; sel.yl = yl; sel.w = 1.f; sel.h = 1.f; sel.d0 = dl; sel.d1 = dl;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float a = KT > 0 ? ew[k] : expf(p.w(k) - mw);
    const float b = KT > 0 ? eh[k] : expf(p.h(k) - mh);
    cw += c.min_w + c.free_w * (a * rw);           // :110-111 / :124-125
    ch += c.min_h + c.free_h * (b * rh);
    const float xr = (k == K - 1) ? c.hi_x : c.span_x * cw + c.lo_x;   // :116-118
    const float yr = (k == K - 1) ? c.hi_y : c.span_y * ch + c.lo_y;   // :130-132
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

// Full evaluation with tails handling (splines.py:30-43: outside -> identity, 0).
template <int KT, bool INV, class P>
__device__ __forceinline__ void rqs_point(float x, const P& p, const RqsConst& c,
                                          float& y, float& lad, bool& bad) {
  if (c.tails == 1 && !((x >= c.lo_x) && (x <= c.hi_x))) {
    y = x;
    lad = 0.f;
    return;
  }
  RqsBin sel;
  rqs_select<KT, INV>(x, p, c, sel);
  rqs_bin_eval<INV>(x, sel, y, lad, bad);
}

// Logits of one element stored as [K widths | K heights | derivative logits]
// (the reference's transform_params[..., :] layout, coupling.py:310-312).
struct PackedLogits {
  const float* q;
  int K;
  float scale, edge;
  int tails;
  __device__ __forceinline__ float w(int k) const { return q[k] * scale; }
  __device__ __forceinline__ float h(int k) const { return q[K + k] * scale; }
  __device__ __forceinline__ float d(int k) const {
    if (tails == 1) return (k == 0 || k == K) ? edge : q[2 * K + k - 1];   // :37-40
    return q[2 * K + k];
  }
};

// Three separately strided logit rows (functional API, splines.py:20-29).
struct SplitLogits {
  const float *pw, *ph, *pd;
  int K;
  float scale, edge;
  int tails;
  __device__ __forceinline__ float w(int k) const { return pw[k] * scale; }
  __device__ __forceinline__ float h(int k) const { return ph[k] * scale; }
  __device__ __forceinline__ float d(int k) const {
    if (tails == 1) return (k == 0 || k == K) ? edge : pd[k - 1];
    return pd[k];
  }
};

// Per-feature knot table for batch-shared logits (PiecewiseRationalQuadraticCDF,
// coupling.py:165-246): xk[K+1] | yk[K+1] | dk[K+1], built once per workgroup with
// the same arithmetic as rqs_select, so table and direct evaluation agree bitwise.
template <class P>
__device__ __forceinline__ void rqs_build_table(const P& p, const RqsConst& c, float* tab) {
  const int K = c.K;
  float mw = -INFINITY, mh = -INFINITY;
  for (int k = 0; k < K; ++k) {
    mw = fmaxf(mw, p.w(k));
    mh = fmaxf(mh, p.h(k));
  }
  float sw = 0.f, sh = 0.f;
  for (int k = 0; k < K; ++k) {
    sw += expf(p.w(k) - mw);
    sh += expf(p.h(k) - mh);
  }
  const float rw = 1.f / sw, rh = 1.f / sh;
  float cw = 0.f, ch = 0.f;
  float* xk = tab;
  float* yk = tab + (K + 1);
  float* dk = tab + 2 * (K + 1);
  xk[0] = c.lo_x;
  yk[0] = c.lo_y;
  dk[0] = c.min_d + softplus_f(p.d(0));
  for (int k = 0; k < K; ++k) {
    cw += c.min_w + c.free_w * (expf(p.w(k) - mw) * rw);
    ch += c.min_h + c.free_h * (expf(p.h(k) - mh) * rh);
    xk[k + 1] = (k == K - 1) ? c.hi_x : c.span_x * cw + c.lo_x;
    yk[k + 1] = (k == K - 1) ? c.hi_y : c.span_y * ch + c.lo_y;
    dk[k + 1] = c.min_d + softplus_f(p.d(k + 1));
  }
}

template <bool INV>
__device__ __forceinline__ void rqs_point_table(float x, const float* tab, const RqsConst& c,
                                                float& y, float& lad, bool& bad) {
  if (c.tails == 1 && !((x >= c.lo_x) && (x <= c.hi_x))) {
    y = x;
    lad = 0.f;
    return;
  }
  const int K = c.K;
  const float* xk = tab;
  const float* yk = tab + (K + 1);
  const float* dk = tab + 2 * (K + 1);
  int bin = 0;
  const float* key = INV ? yk : xk;
  for (int k = 1; k < K; ++k) bin += (x >= key[k]) ? 1 : 0;   // knots ascend: count = last hit
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
