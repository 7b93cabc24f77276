// cycles per spline evaluation (logits in registers, as in the fused kernel's vector step)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <algorithm>
#include "rqs_math.hpp"
#include "fused_common.hpp"
#include "alt.hpp"
using namespace vcnf;

__host__ __device__ inline void gen_inputs(unsigned tid, unsigned bid, float* p24, float& x) {
  unsigned s = tid * 2654435761u + bid * 40503u + 12345u;
  for (int t = 0; t < 24; ++t) { s = s * 1664525u + 1013904223u; p24[t] = ((s >> 8) * (1.f / 16777216.f) - 0.5f) * (t < 16 ? 40.f : 8.f); }
  s = s * 1664525u + 1013904223u;
  x = ((s >> 8) * (1.f / 16777216.f) - 0.5f) * 7.f;
}

template <int VAR, bool INV>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* cyc, int iters, int nwaves, RqsConst c, int mm) {
  const int wave = threadIdx.x >> 6;
  floatx4 pa[6];
  float p24[24], x;
  gen_inputs(threadIdx.x, blockIdx.x, p24, x);
  for (int t = 0; t < 24; ++t) pa[t >> 2][t & 3] = p24[t];
  if (VAR >= 1) {
    const float sc2 = c.wh_scale * kLog2e;
    for (int t = 0; t < 16; ++t) pa[t >> 2][t & 3] *= sc2;
    for (int t = 16; t < 24; ++t) pa[t >> 2][t & 3] *= kLog2e;
  }
  float ysum = 0.f, lsum = 0.f, ylast = 0.f, llast = 0.f;
  bool bad = false;
  half8 a8, b8; for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(i * 0.1f); b8[i] = (_Float16)(0.3f); }
  floatx4 acc[4] = {};
  typedef float floatx16 __attribute__((ext_vector_type(16)));
  floatx16 acc2[2] = {};
  __syncthreads();
  const long long t0 = clock64();
  if (wave < nwaves) {
    for (int it = 0; it < iters; ++it) {
      for (int b = 0; b < 6; ++b) asm volatile("" : "+v"(pa[b]));
      asm volatile("" : "+v"(x));
      float yv, lad;
      if (VAR == 0) {
        if (c.tails == 1 && !((x >= c.lo_x) && (x <= c.hi_x))) { yv = x; lad = 0.f; }
        else { RegLogits<8, 6> p{pa, c.wh_scale, c.edge_logit}; RqsBin sel; rqs_select<8, INV>(x, p, c, c.wh_scale * kLog2e, sel); rqs_bin_eval<INV>(x, sel, yv, lad, bad); }
      } else {
        alt_eval<VAR, 8, INV>(x, pa, c, yv, lad, bad);
      }
      ysum += yv; lsum += lad; ylast = yv; llast = lad;
    }
  } else if (mm == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 72; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[u & 3], 0, 0, 0);
    }
  } else if (mm == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 36; ++u) acc2[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc2[u & 1], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  out[(blockIdx.x * 512 + threadIdx.x) * 2] = ylast + 1e-30f * ysum + acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] + acc2[0][0] + acc2[1][0];
  out[(blockIdx.x * 512 + threadIdx.x) * 2 + 1] = llast + 1e-30f * lsum + (bad ? 1e9f : 0.f);
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

static float h[256 * 512 * 2];
static double refd[2][256 * 512 * 2];
static void ref64(bool inv) {
  const int K = 8; const double lo = -3, hi = 3, mn = 1e-3;
  for (int bid = 0; bid < 256; ++bid) for (int tid = 0; tid < 512; ++tid) {
    float p[24], xf; gen_inputs(tid, bid, p, xf);
    double x = xf, sc = 1.0 / sqrt(128.0);
    double* o = &refd[inv][(bid * 512 + tid) * 2];
    if (!(x >= lo && x <= hi)) { o[0] = x; o[1] = 0; continue; }
    double w[8], hh[8], d[9], sw = 0, sh = 0, mw = -1e300, mh = -1e300;
    for (int k = 0; k < K; ++k) { mw = fmax(mw, p[k] * sc); mh = fmax(mh, p[8 + k] * sc); }
    for (int k = 0; k < K; ++k) { w[k] = exp(p[k] * sc - mw); hh[k] = exp(p[8 + k] * sc - mh); sw += w[k]; sh += hh[k]; }
    double xk[9], yk[9]; xk[0] = lo; yk[0] = lo; double cw = 0, chh = 0;
    for (int k = 0; k < K; ++k) { w[k] = mn + (1 - mn * K) * w[k] / sw; hh[k] = mn + (1 - mn * K) * hh[k] / sh; cw += w[k]; chh += hh[k]; xk[k + 1] = lo + (hi - lo) * cw; yk[k + 1] = lo + (hi - lo) * chh; }
    xk[K] = hi; yk[K] = hi;
    const double edge = log(exp(1 - mn) - 1);
    for (int k = 0; k <= K; ++k) { double v = (k == 0 || k == K) ? edge : (double)p[16 + k - 1]; d[k] = mn + (v > 30 ? v : log1p(exp(v))); }
    int bin = 0; const double* key = inv ? yk : xk;
    for (int k = 1; k < K; ++k) if (x >= key[k]) bin = k;
    const double xl = xk[bin], ww = xk[bin + 1] - xk[bin], yl = yk[bin], hgt = yk[bin + 1] - yk[bin], s = hgt / ww, d0 = d[bin], d1 = d[bin + 1];
    if (!inv) {
      const double t = (x - xl) / ww, tt = t * (1 - t), den = s + (d0 + d1 - 2 * s) * tt;
      o[0] = yl + hgt * (s * t * t + d0 * tt) / den;
      o[1] = log(s * s * (d1 * t * t + 2 * s * tt + d0 * (1 - t) * (1 - t))) - 2 * log(den);
    } else {
      const double dy = x - yl, e = d0 + d1 - 2 * s, a = dy * e + hgt * (s - d0), b = hgt * d0 - dy * e, cc = -s * dy;
      const double r = 2 * cc / (-b - sqrt(b * b - 4 * a * cc));
      o[0] = r * ww + xl; const double rr = r * (1 - r), den = s + e * rr;
      o[1] = -(log(s * s * (d1 * r * r + 2 * s * rr + d0 * (1 - r) * (1 - r))) - 2 * log(den));
    }
  }
}
template <int VAR, bool INV>
void run(const char* name, float* ref, bool keep) {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 8); (void)hipMalloc(&cyc, 256 * 8 * 8);
  RqsConst c;
  c.K = 8; c.tails = 1; c.lo_x = -3.f; c.hi_x = 3.f; c.span_x = 6.f; c.lo_y = -3.f; c.hi_y = 3.f; c.span_y = 6.f;
  c.min_w = c.min_h = c.min_d = 1e-3f; c.free_w = c.free_h = (float)(1.0 - 8e-3); c.wh_scale = 1.f / sqrtf(128.f);
  c.edge_logit = (float)log(exp(1.0 - 1e-3) - 1.0);
  const int iters = 64;
  for (int cfg = 0; cfg < 4; ++cfg) {
    const int nw = cfg == 1 ? 8 : 4, mm = cfg == 2 ? 1 : cfg == 3 ? 2 : 0;
    for (int rep = 0; rep < 2; ++rep) k<VAR, INV><<<256, 512>>>(out, cyc, iters, nw, c, mm);
    (void)hipDeviceSynchronize();
    long long hc[256 * 8];
    (void)hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
    double m = 0, mo = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < nw ? m : mo) += hc[b * 8 + w];
    printf("%-14s %s %-24s %7.1f cycles/eval/wave", name, INV ? "inv" : "fwd", cfg == 0 ? "1 wave/SIMD" : cfg == 1 ? "2 waves/SIMD" : cfg == 2 ? "beside 16x16x32 MFMAs" : "beside 32x32x16 MFMAs", m / 256 / nw / iters);
    if (mm) printf("   (MFMA wave %7.1f cyc/iter for 1152 cyc of matrix work)", mo / 256 / 4 / iters);
    printf("\n");
    if (cfg == 0) {
      (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
      {
        double ey = 0, el = 0, sy = 0, sl = 0; int n = 0;
        static double els[256 * 256];
        for (int i = 0; i < 256 * 512; ++i) {
          if ((i >> 6) % 8 >= 4) continue;
          const double dy_ = fabs(h[2 * i] - refd[INV][2 * i]), dl_ = fabs(h[2 * i + 1] - refd[INV][2 * i + 1]);
          ey = fmax(ey, dy_); el = fmax(el, dl_); sy += dy_; sl += dl_; els[n++] = dl_;
        }
        std::sort(els, els + n);
        printf("    vs fp64: y err mean %.3e max %.3e | lad err mean %.3e p99 %.3e p99.9 %.3e max %.3e\n", sy / n, ey, sl / n, els[(int)(n * 0.99)], els[(int)(n * 0.999)], el);
      }
    }
  }
  (void)hipFree(out); (void)hipFree(cyc);
}
static float ref_f[256 * 512 * 2], ref_i[256 * 512 * 2];
int main() {
  ref64(false); ref64(true);
  run<0, false>("base", ref_f, true);
  run<0, true>("base", ref_i, true);
  ALT_RUNS
  return 0;
}
