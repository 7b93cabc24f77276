// Rational-quadratic spline in double precision: the elementwise spline of vcnf_rqs_elementwise_f32 for models
// converted with .double() (the reference's drivers do: /root/reference run.py:114, runadultvdeq.py:183; its Flow
// contract is dtype-agnostic).  One thread per element, the reference's operation order as written in
// normflow/utils/splines.py:20-85 (tails) and :88-193 (spline): softmax -> floor -> cumsum -> affine map -> exact end
// knots -> sizes by differencing; compare-and-count bin search with the last knot bumped by 1e-6; the stable root form
// 2c / (-b - sqrt(disc)).  Library exp / log / sqrt in fp64: this path is for parity with fp64 models, not for speed
// (fp64 vector rate is 1/2 .. 1/4 of fp32 and nothing here is tuned); the hot path is fp32.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"

namespace vcnf {

constexpr int kMaxBins64 = 64;

struct Rqs64Args {
  const double *x, *uw, *uh, *ud;
  long long ld_w, ld_h, ld_d;
  double *y, *lad;
  long long n;
  int K, tails, inverse;
  double left, right, bottom, top, min_w, min_h, min_d, wh_scale, edge;
  int32_t* bad;
};

// knots of one side: cum[0..K] from K logits (splines.py:109-119 / :123-133)
__device__ __forceinline__ void partition64(const double* lg, int K, double scale, double lo, double hi, double floor_,
                                            double* cum) {
  double m = -INFINITY;
  for (int k = 0; k < K; ++k) m = fmax(m, lg[k] * scale);
  double s = 0.0;
  for (int k = 0; k < K; ++k) s += exp(lg[k] * scale - m);
  double run = 0.0;
  cum[0] = lo;
  for (int k = 0; k < K; ++k) {
    const double p = floor_ + (1.0 - floor_ * K) * (exp(lg[k] * scale - m) / s);
    run += p;
    cum[k + 1] = (hi - lo) * run + lo;
  }
  cum[K] = hi;
}

__device__ __forceinline__ double softplus64(double v) { return v > 20.0 ? v : log1p(exp(v)); }   // F.softplus (threshold 20)

__global__ __launch_bounds__(256) void rqs_elementwise_f64_kernel(const Rqs64Args a) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x) {
    const double x = a.x[i];
    if (a.tails != VCNF_TAILS_NONE && !(x >= a.left && x <= a.right)) {   // splines.py:30-49: identity outside
      a.y[i] = x;
      a.lad[i] = 0.0;
      continue;
    }
    const int K = a.K;
    double xk[kMaxBins64 + 1], yk[kMaxBins64 + 1];
    partition64(a.uw + i * a.ld_w, K, a.wh_scale, a.left, a.right, a.min_w, xk);
    partition64(a.uh + i * a.ld_h, K, a.wh_scale, a.bottom, a.top, a.min_h, yk);
    // bin: number of knots <= value, last knot bumped by eps (splines.py:12-17)
    const double* kn = a.inverse ? yk : xk;
    int bin = -1;
    for (int k = 0; k <= K; ++k) bin += (x >= (k == K ? kn[k] + 1e-6 : kn[k])) ? 1 : 0;
    bin = bin < 0 ? 0 : (bin > K - 1 ? K - 1 : bin);
    // derivative logits of the bin's two knots (padding per tails mode, splines.py:36-49)
    const double* ud = a.ud + i * a.ld_d;
    auto dlogit = [&](int k) -> double {
      if (a.tails == VCNF_TAILS_LINEAR) return (k == 0 || k == K) ? a.edge : ud[k - 1];
      if (a.tails == VCNF_TAILS_CIRCULAR) return k == K ? ud[0] : ud[k];
      return ud[k];
    };
    const double d0 = a.min_d + softplus64(dlogit(bin));           // :121
    const double d1 = a.min_d + softplus64(dlogit(bin + 1));
    const double x_lo = xk[bin], w = xk[bin + 1] - xk[bin];
    const double y_lo = yk[bin], h = yk[bin + 1] - yk[bin];
    const double s = h / w;                                          // :144
    double out, lad;
    if (a.inverse) {
      const double dy = x - y_lo;
      const double e = d0 + d1 - 2.0 * s;
      const double qa = dy * e + h * (s - d0);                       // :153-161
      const double qb = h * d0 - dy * e;
      const double qc = -s * dy;
      const double disc = qb * qb - 4.0 * qa * qc;                   // :163
      if (!(disc >= 0.0) && a.bad) atomicAdd(a.bad, 1);              // :164 (the reference asserts)
      const double r = (2.0 * qc) / (-qb - sqrt(disc));              // :166
      out = r * w + x_lo;
      const double rr = r * (1.0 - r);
      const double den = s + e * rr;
      const double dnum = s * s * (d1 * r * r + 2.0 * s * rr + d0 * (1.0 - r) * (1.0 - r));
      lad = -(log(dnum) - 2.0 * log(den));                           // :175-177
    } else {
      const double t = (x - x_lo) / w;                               // :179
      const double tt = t * (1.0 - t);
      const double num = h * (s * t * t + d0 * tt);
      const double den = s + (d0 + d1 - 2.0 * s) * tt;
      out = y_lo + num / den;                                        // :186
      const double dnum = s * s * (d1 * t * t + 2.0 * s * tt + d0 * (1.0 - t) * (1.0 - t));
      lad = log(dnum) - 2.0 * log(den);                              // :191
    }
    a.y[i] = out;
    a.lad[i] = lad;
  }
}

}  // namespace vcnf

using namespace vcnf;

extern "C" int vcnf_rqs_elementwise_f64(const double* x, const double* uw, const double* uh, const double* ud,
                                        int64_t ld_w, int64_t ld_h, int64_t ld_d,
                                        double* y, double* logabsdet, int64_t n,
                                        const vcnf_rqs_cfg_f64* cfg, int inverse, int32_t* bad_disc, void* stream) {
  if (!cfg) return VCNF_ERR_NULL;
  if (n < 0 || cfg->num_bins < 1 || cfg->num_bins > kMaxBins64) return VCNF_ERR_SHAPE;
  if (cfg->tails < VCNF_TAILS_NONE || cfg->tails > VCNF_TAILS_CIRCULAR) return VCNF_ERR_UNSUPPORTED;
  const int K = cfg->num_bins;
  if (cfg->min_bin_width * K > 1.0 || cfg->min_bin_height * K > 1.0) return VCNF_ERR_VALUE;
  if (n == 0) return VCNF_OK;
  if (!x || !uw || !uh || !ud || !y || !logabsdet) return VCNF_ERR_NULL;
  Rqs64Args a;
  a.x = x; a.uw = uw; a.uh = uh; a.ud = ud; a.ld_w = ld_w; a.ld_h = ld_h; a.ld_d = ld_d;
  a.y = y; a.lad = logabsdet; a.n = n; a.K = K; a.tails = cfg->tails; a.inverse = inverse ? 1 : 0;
  a.left = cfg->left; a.right = cfg->right; a.bottom = cfg->bottom; a.top = cfg->top;
  a.min_w = cfg->min_bin_width; a.min_h = cfg->min_bin_height; a.min_d = cfg->min_derivative;
  a.wh_scale = cfg->wh_scale;
  a.edge = log(exp(1.0 - cfg->min_derivative) - 1.0);
  a.bad = bad_disc;
  long long blocks = (n + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(rqs_elementwise_f64_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}
