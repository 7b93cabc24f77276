// Empirical operand / result layout of v_mfma_f32_32x32x16_f16 on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
// A[i][k] = 1 if (i == ia && k-slot (lane group, idx) == (ga, xa)); B[k][j] likewise; D gets a single 1 at (ia, jb) iff the slots match
__global__ void probe(float* out, int la, int xa, int lb, int xb) {
  const int l = threadIdx.x;
  half8 a = {}, b = {};
  if (l == la) a[xa] = (_Float16)1.f;
  if (l == lb) b[xb] = (_Float16)1.f;
  floatx16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}
int main() {
  float* d; (void)hipMalloc(&d, 64 * 16 * 4);
  float h[64 * 16];
  // 1) D layout: A lane la (row la%32, group la/32) idx 0, B lane lb same group idx 0 -> where does the 1 land?
  printf("D layout probe: (A lane, B lane) -> (lane, reg) holding the product\n");
  const int tests[][2] = {{0, 0}, {1, 0}, {4, 0}, {5, 0}, {8, 0}, {9, 0}, {31, 0}, {0, 1}, {0, 31}, {7, 13}, {32 + 3, 32 + 5}, {32 + 20, 32 + 31}};
  for (auto& t : tests) {
    probe<<<1, 64>>>(d, t[0], 0, t[1], 0);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("  A lane %2d B lane %2d:", t[0], t[1]);
    for (int i = 0; i < 64 * 16; ++i) if (h[i] != 0.f) printf(" lane %d reg %d (=%g)", i / 16, i % 16, h[i]);
    printf("   expected row %d col %d -> lane %d(+32*((row/4)%%2)) reg %d\n", t[0] % 32, t[1] % 32, t[1] % 32 + 32 * ((t[0] % 32 / 4) % 2), 4 * (t[0] % 32 / 8) + t[0] % 4);
  }
  // 2) which (group, idx) slots of A pair with which of B: for A slot (ga, xa) find the B slots that give a nonzero
  printf("k-slot pairing: A (group, idx) pairs with B (group, idx):\n");
  for (int ga = 0; ga < 2; ++ga) for (int xa = 0; xa < 8; ++xa) {
    printf("  A(%d,%d) ->", ga, xa);
    for (int gb = 0; gb < 2; ++gb) for (int xb = 0; xb < 8; ++xb) {
      probe<<<1, 64>>>(d, 32 * ga, xa, 32 * gb, xb);
      (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      bool nz = false; for (int i = 0; i < 64 * 16; ++i) nz |= h[i] != 0.f;
      if (nz) printf(" B(%d,%d)", gb, xb);
    }
    printf("\n");
  }
  return 0;
}
