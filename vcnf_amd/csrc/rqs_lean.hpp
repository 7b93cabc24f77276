// Rational-quadratic spline evaluation for the fused layer kernel's vector steps: the same map as
// rqs_math.hpp::rqs_select + rqs_bin_eval (reference: normflow/utils/splines.py:88-193, linear tails :30-43)
// with fewer vector instructions (~190 against ~290 unpacked), because the vector steps of that kernel are
// bound by vector-instruction issue beside the partner wave's matrix instructions
// (profiles/r02_spline_eval_microbench.md).
//
// The logits arrive PRE-SCALED by the host (vcnf_amd/fused.py folds the factors into the last layer's rows):
//   width / height logits:  w' = w * (1/sqrt(hidden)) * log2(e)   (coupling.py:314-316, softmax as 2^x)
//   derivative logits:      d' = d * log2(e)
// and the evaluation works in "exp-sum space".  With e_k = 2^(w'_k - max), S = sum e_k and the floor m = min_w:
//   width_k = m + (1 - K m) e_k / S = (e_k + S m / (1 - K m)) * (1 - K m) / S          (splines.py:109-110)
// so, in units of (1 - K m) (right - left) / S, bin k is e_k + S mf wide (mf = m / (1 - K m)) and its left
// edge is prefix_k = sum_{j<k} e_j + k S mf.  The bin search compares u = (x - left) S / ((right - left)(1 - K m))
// with the prefix sums (no division before the search, no knots), only the selected bin is mapped back.
// Rounding differs from the reference's cumsum order by a few ulp of the knots; the error against fp64 is the
// same as rqs_math.hpp's (table in the profile note above).  Branch-free, no packed-f32 forms.
#pragma once
#include "rqs_math.hpp"

namespace vcnf {

struct LeanConst {
  float lo_x, hi_x, lo_y, hi_y;
  float kx, ky;        // 1 / ((right - left) (1 - K min_w)), same for the y side
  float mfw, mfh;      // min / (1 - K min)
  float sfx, sfy;      // (right - left) (1 - K min_w), y side
  float min_d, edge2;  // edge derivative logit (splines.py:38) in log2 units
  int tails;
};

__device__ __forceinline__ LeanConst make_lean_const(const RqsConst& c) {
  LeanConst l;
  l.lo_x = c.lo_x; l.hi_x = c.hi_x; l.lo_y = c.lo_y; l.hi_y = c.hi_y;
  l.sfx = c.span_x * c.free_w;
  l.sfy = c.span_y * c.free_h;
  l.kx = 1.f / l.sfx;
  l.ky = 1.f / l.sfy;
  l.mfw = c.min_w / c.free_w;
  l.mfh = c.min_h / c.free_h;
  l.min_d = c.min_d;
  l.edge2 = c.edge_logit * kLog2e;
  l.tails = c.tails;
  return l;
}

// min_d + softplus(v) for v2 = v log2(e):  ln2 (max(v2, 0) + log2(1 + 2^-|v2|)); the rounding error of 1 + e
// is added back (first order), which keeps full relative accuracy when the result is tiny (splines.py:121)
__device__ __forceinline__ float lean_derivative(float v2, float min_d) {
  const float e = __builtin_amdgcn_exp2f(-__builtin_fabsf(v2));
  const float u = 1.f + e;
  const float cc = e - (u - 1.f);
  const float l2 = __builtin_amdgcn_logf(u);
  return min_d + fmaf(kLn2, fmaxf(v2, 0.f) + l2, cc);
}

// lg: K width logits, K height logits, K - 1 derivative logits (pre-scaled, see above); linear tails: a point
// outside [lo, hi] maps to itself with log|det| 0 (evaluated at the left end, selected away afterwards).
template <int K, bool INV>
__device__ __forceinline__ void rqs_lean_eval(float x, const float (&lg)[3 * K - 1], const LeanConst& lc,
                                              float& yv, float& lad, bool& bad) {
  static_assert(K >= 4 && K % 2 == 0, "bin count");
  const bool inside = (x >= lc.lo_x) && (x <= lc.hi_x);
  const float xi = inside ? x : lc.lo_x;
  float ew[K], eh[K];
  {
    float mw = fmaxf(fmaxf(lg[0], lg[1]), lg[2]), mh = fmaxf(fmaxf(lg[K], lg[K + 1]), lg[K + 2]);   // v_max3_f32
#pragma unroll
    for (int k = 3; k + 1 < K; k += 2) {
      mw = fmaxf(fmaxf(mw, lg[k]), lg[k + 1]);
      mh = fmaxf(fmaxf(mh, lg[K + k]), lg[K + k + 1]);
    }
    mw = fmaxf(mw, lg[K - 1]); mh = fmaxf(mh, lg[2 * K - 1]);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      ew[k] = __builtin_amdgcn_exp2f(lg[k] - mw);
      eh[k] = __builtin_amdgcn_exp2f(lg[K + k] - mh);
    }
  }
  float cw[K], ch[K];                       // prefix sums of the softmax numerators
  cw[0] = ew[0]; ch[0] = eh[0];
#pragma unroll
  for (int k = 1; k < K; ++k) { cw[k] = cw[k - 1] + ew[k]; ch[k] = ch[k - 1] + eh[k]; }
  const float Sw = cw[K - 1], Sh = ch[K - 1];
  const float mSw = Sw * lc.mfw, mSh = Sh * lc.mfh;
  // a = searched side (x side in the density direction, y side in the sampling direction), b = the other
  const float* ca = INV ? ch : cw; const float* cb = INV ? cw : ch;
  const float* ea = INV ? eh : ew; const float* eb = INV ? ew : eh;
  const float mSa = INV ? mSh : mSw, mSb = INV ? mSw : mSh;
  const float Sa = INV ? Sh : Sw, Sb = INV ? Sw : Sh;
  const float u = (xi - (INV ? lc.lo_y : lc.lo_x)) * (Sa * (INV ? lc.ky : lc.kx));
  float selca = 0.f, selea = ea[0], selcb = 0.f, seleb = eb[0];
  float d0 = lc.edge2, d1 = lg[2 * K];
#pragma unroll
  for (int k = 1; k < K; ++k) {             // searchsorted (splines.py:12-17): last left edge that is <= u
    const float ca_l = fmaf((float)k, mSa, ca[k - 1]);
    const float cb_l = fmaf((float)k, mSb, cb[k - 1]);
    const bool take = u >= ca_l;
    selca = take ? ca_l : selca;
    selea = take ? ea[k] : selea;
    selcb = take ? cb_l : selcb;
    seleb = take ? eb[k] : seleb;
    d0 = take ? lg[2 * K + k - 1] : d0;
    d1 = take ? (k == K - 1 ? lc.edge2 : lg[2 * K + k]) : d1;
  }
  const float wa = selea + mSa, wb = seleb + mSb;      // bin extents in exp-sum units
  const float D0 = lean_derivative(d0, lc.min_d), D1 = lean_derivative(d1, lc.min_d);
  // The selected bin in x / y units, by the same operations in both directions: unit of the x side gx = sfx / Sw,
  // of the y side gy = sfy / Sh, knots lo + prefix * unit, extents e' * unit.  (An earlier version divided in
  // exp-sum units, (u - prefix) / e': one division less, but the two directions then saw bin edges that differ in
  // the last bits, and inverse(forward(x)) was 2-3x less exact than the reference's round trip, which uses
  // bit-identical knots in both directions - splines.py:109-133.)
  const float rSw = hw_rcp(Sw), rSh = hw_rcp(Sh);
  float gx = lc.sfx * rSw; gx = fmaf(fmaf(-Sw, gx, lc.sfx), rSw, gx);
  float gy = lc.sfy * rSh; gy = fmaf(fmaf(-Sh, gy, lc.sfy), rSh, gy);
  const float w = (INV ? wb : wa) * gx, h = (INV ? wa : wb) * gy;
  const float xl = fmaf(INV ? selcb : selca, gx, lc.lo_x), yl = fmaf(INV ? selca : selcb, gy, lc.lo_y);
  const float rw = hw_rcp(w);
  float s = h * rw; s = fmaf(fmaf(-w, s, h), rw, s);                                  // splines.py:144
  if (!INV) {
    const float dt = xi - xl;                                                          // :179
    float t = dt * rw; t = fmaf(fmaf(-w, t, dt), rw, t);
    const float omt = 1.f - t, tt = t * omt;
    const float e = fmaf(-2.f, s, D0 + D1);
    const float den = fmaf(e, tt, s);
    const float num = h * fmaf(s * t, t, D0 * tt);
    const float rden = hw_rcp(den);
    float qn = num * rden; qn = fmaf(fmaf(-den, qn, num), rden, qn);
    const float dn = (s * s) * fmaf(D1 * t, t, fmaf(2.f * s, tt, (D0 * omt) * omt));
    const float l = kLn2 * fmaf(-2.f, __builtin_amdgcn_logf(den), __builtin_amdgcn_logf(dn));
    yv = inside ? yl + qn : x;
    lad = inside ? l : 0.f;
  } else {
    const float dy = xi - yl;                                                          // :153-177
    const float e = fmaf(-2.f, s, D0 + D1);
    const float qa = fmaf(dy, e, h * (s - D0));
    const float qb = fmaf(-dy, e, h * D0);
    const float qc = -s * dy;
    const float fa = 4.f * qa;              // b^2 - 4ac with the rounding error of 4ac recovered (Kahan)
    const float p = fa * qc;
    const float perr = fmaf(fa, qc, -p);
    const float disc = rqs_rounding_level_zero(fmaf(qb, qb, -p) - perr, qb, p);   // rqs_math.hpp: rounding-level negatives are the double root
    bad = bad || (inside && !(disc >= 0.f));              // :164 (the reference asserts)
    const float dd = -qb - hw_sqrt(disc);
    const float rdd = hw_rcp(dd);
    const float n2 = 2.f * qc;
    float r = n2 * rdd; r = fmaf(fmaf(-dd, r, n2), rdd, r);
    const float omr = 1.f - r, rr = r * omr;
    const float den = fmaf(e, rr, s);
    const float dn = (s * s) * fmaf(D1 * r, r, fmaf(2.f * s, rr, (D0 * omr) * omr));
    const float l = kLn2 * fmaf(2.f, __builtin_amdgcn_logf(den), -__builtin_amdgcn_logf(dn));
    yv = inside ? fmaf(r, w, xl) : x;
    lad = inside ? l : 0.f;
  }
}

}  // namespace vcnf
