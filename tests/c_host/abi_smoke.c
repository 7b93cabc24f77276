/* C host program against the C ABI of libvcnf_hip.so (no Python, no PyTorch): allocates device
 * buffers with the HIP runtime, evaluates the rational-quadratic spline through
 * vcnf_rqs_elementwise_f32 in both directions and checks it against a scalar double-precision
 * restatement of the same formulas (utils/splines.py:88-193 with linear tails :30-43) written here.
 * Built and run by tests/test_c_host.py on the GPU box:
 *     gcc -std=c11 -D__HIP_PLATFORM_AMD__ tests/c_host/abi_smoke.c -I include -I /opt/rocm/include \
 *         -L vcnf_amd/csrc -L /opt/rocm/lib -lvcnf_hip -lamdhip64 -lm -o abi_smoke
 * Exit code 0 and a line "abi_smoke ok ..." on success. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "vcnf_hip.h"

#define K 8
#define N 4096

static double softplus(double v) { return v > 20.0 ? v : log1p(exp(v)); }

/* one element, linear tails, tail bound T; returns y, writes log|dy/dx| */
static double ref_spline(double x, const float* uw, const float* uh, const float* ud, double T, int inverse,
                         double* lad) {
  const double min_w = 1e-3, min_h = 1e-3, min_d = 1e-3;
  if (!(x >= -T && x <= T)) { *lad = 0.0; return x; }
  double w[K], h[K], xk[K + 1], yk[K + 1], d[K + 1];
  double mw = -1e300, mh = -1e300, sw = 0.0, sh = 0.0;
  for (int k = 0; k < K; ++k) { if (uw[k] > mw) mw = uw[k]; if (uh[k] > mh) mh = uh[k]; }
  for (int k = 0; k < K; ++k) { w[k] = exp(uw[k] - mw); h[k] = exp(uh[k] - mh); sw += w[k]; sh += h[k]; }
  xk[0] = -T; yk[0] = -T;
  double cw = 0.0, ch = 0.0;
  for (int k = 0; k < K; ++k) {
    cw += min_w + (1.0 - min_w * K) * w[k] / sw;
    ch += min_h + (1.0 - min_h * K) * h[k] / sh;
    xk[k + 1] = 2.0 * T * cw - T;
    yk[k + 1] = 2.0 * T * ch - T;
  }
  xk[K] = T; yk[K] = T;
  const double edge = log(exp(1.0 - min_d) - 1.0);
  for (int k = 0; k <= K; ++k) d[k] = min_d + softplus((k == 0 || k == K) ? edge : (double)ud[k - 1]);
  const double* key = inverse ? yk : xk;
  int b = 0;
  for (int k = 1; k < K; ++k) if (x >= key[k]) b = k;
  const double bw = xk[b + 1] - xk[b], bh = yk[b + 1] - yk[b], s = bh / bw, d0 = d[b], d1 = d[b + 1];
  if (!inverse) {
    const double t = (x - xk[b]) / bw, tt = t * (1.0 - t);
    const double den = s + (d0 + d1 - 2.0 * s) * tt;
    *lad = log(s * s * (d1 * t * t + 2.0 * s * tt + d0 * (1.0 - t) * (1.0 - t))) - 2.0 * log(den);
    return yk[b] + bh * (s * t * t + d0 * tt) / den;
  }
  const double dy = x - yk[b];
  const double qa = dy * (d0 + d1 - 2.0 * s) + bh * (s - d0), qb = bh * d0 - dy * (d0 + d1 - 2.0 * s), qc = -s * dy;
  const double r = 2.0 * qc / (-qb - sqrt(qb * qb - 4.0 * qa * qc));
  const double rr = r * (1.0 - r), den = s + (d0 + d1 - 2.0 * s) * rr;
  *lad = -(log(s * s * (d1 * r * r + 2.0 * s * rr + d0 * (1.0 - r) * (1.0 - r))) - 2.0 * log(den));
  return r * bw + xk[b];
}

#define CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
  fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_)); return 2; } } while (0)

int main(void) {
  if (vcnf_abi_version() != 1) { fprintf(stderr, "unexpected ABI version %d\n", vcnf_abi_version()); return 3; }
  const double T = 3.0;
  float *x = malloc(N * sizeof(float)), *uw = malloc(N * K * sizeof(float)), *uh = malloc(N * K * sizeof(float));
  float *ud = malloc(N * (K - 1) * sizeof(float)), *y = malloc(N * sizeof(float)), *lad = malloc(N * sizeof(float));
  unsigned s = 12345u;
#define RND() (s = s * 1664525u + 1013904223u, (float)((s >> 8) & 0xFFFF) / 65535.0f * 2.0f - 1.0f)
  for (int i = 0; i < N; ++i) x[i] = 3.6f * RND();
  for (int i = 0; i < N * K; ++i) { uw[i] = 1.5f * RND(); uh[i] = 1.5f * RND(); }
  for (int i = 0; i < N * (K - 1); ++i) ud[i] = 1.5f * RND();
  float *dx, *dw, *dh, *dd, *dy, *dl;
  int32_t* dbad;
  CHECK(hipMalloc((void**)&dx, N * sizeof(float)));
  CHECK(hipMalloc((void**)&dw, N * K * sizeof(float)));
  CHECK(hipMalloc((void**)&dh, N * K * sizeof(float)));
  CHECK(hipMalloc((void**)&dd, N * (K - 1) * sizeof(float)));
  CHECK(hipMalloc((void**)&dy, N * sizeof(float)));
  CHECK(hipMalloc((void**)&dl, N * sizeof(float)));
  CHECK(hipMalloc((void**)&dbad, sizeof(int32_t)));
  CHECK(hipMemset(dbad, 0, sizeof(int32_t)));
  CHECK(hipMemcpy(dx, x, N * sizeof(float), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dw, uw, N * K * sizeof(float), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dh, uh, N * K * sizeof(float), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dd, ud, N * (K - 1) * sizeof(float), hipMemcpyHostToDevice));
  vcnf_rqs_cfg cfg = {K, VCNF_TAILS_LINEAR, (float)-T, (float)T, (float)-T, (float)T, 1e-3f, 1e-3f, 1e-3f, 1.0f};
  double worst[2] = {0.0, 0.0}, mean[2] = {0.0, 0.0};
  for (int inverse = 0; inverse < 2; ++inverse) {
    const int rc = vcnf_rqs_elementwise_f32(dx, dw, dh, dd, K, K, K - 1, dy, dl, N, &cfg, inverse, dbad, NULL);
    if (rc != VCNF_OK) { fprintf(stderr, "vcnf_rqs_elementwise_f32: %s\n", vcnf_status_string(rc)); return 4; }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(y, dy, N * sizeof(float), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(lad, dl, N * sizeof(float), hipMemcpyDeviceToHost));
    for (int i = 0; i < N; ++i) {
      double rl;
      const double ry = ref_spline(x[i], uw + i * K, uh + i * K, ud + i * (K - 1), T, inverse, &rl);
      const double e = fabs(y[i] - ry) / (1.0 + fabs(ry)) + fabs(lad[i] - rl) / (1.0 + fabs(rl));
      if (e > worst[inverse]) worst[inverse] = e;
      mean[inverse] += e / N;
    }
  }
  /* NULL cfg and a negative count are rejected on the host */
  if (vcnf_rqs_elementwise_f32(dx, dw, dh, dd, K, K, K - 1, dy, dl, N, NULL, 0, NULL, NULL) != VCNF_ERR_NULL) return 5;
  if (vcnf_rqs_elementwise_f32(dx, dw, dh, dd, K, K, K - 1, dy, dl, -1, &cfg, 0, NULL, NULL) != VCNF_ERR_SHAPE) return 6;
  int32_t bad = 0;
  CHECK(hipMemcpy(&bad, dbad, sizeof(int32_t), hipMemcpyDeviceToHost));
  printf("abi_smoke ok: forward mean %.2e max %.2e, inverse mean %.2e max %.2e, bad discriminants %d\n", mean[0], worst[0],
         mean[1], worst[1], bad);
  /* fp32 kernel against a double restatement on ill-conditioned random splines: tight on average */
  return (mean[0] < 2e-5 && mean[1] < 2e-5 && worst[0] < 5e-2 && worst[1] < 5e-2 && bad == 0) ? 0 : 1;
}
