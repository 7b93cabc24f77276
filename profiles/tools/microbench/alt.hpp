#pragma once
#include "rqs_lean.hpp"
namespace vcnf {
template <int VAR, int K, bool INV>
__device__ __forceinline__ void alt_eval(float x, const floatx4 (&pa)[6], const RqsConst& c, float& yv, float& lad, bool& bad) {
  const LeanConst lc = make_lean_const(c);
  float lg[23];
#pragma unroll
  for (int t = 0; t < 23; ++t) lg[t] = pa[t >> 2][t & 3];
  rqs_lean_eval<INV>(x, lg, lc, yv, lad, bad);
}
}
#define ALT_RUNS run<1, false>("lean", ref_f, false); run<1, true>("lean", ref_i, false);
