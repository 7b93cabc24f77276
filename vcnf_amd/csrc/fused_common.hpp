// Shared pieces of the fused RQS-layer kernels (fused_layer.hip: exact fp32 matrix path;
// fused_layer_v6.hip: fp16x3 split-half matrix path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rqs_math.hpp"

namespace vcnf {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kFBlock = 256;     // 4 waves; each wave owns kCB 16-sample column blocks of the tile
constexpr int kFusedTile = 128;  // samples per tile of the large-batch fused layer kernels
constexpr int kFusedFlagRows = 32;   // samples per range flag (FusedArgs::redo) = tile of the small-batch kernel

struct FusedArgs {
  const float* x;
  const float* ctx;
  float* y;
  float* logdet;
  const int32_t* tf_idx;
  const int32_t* id_idx;
  const float *sh_w, *sh_h, *sh_d;       // shared (unconditional) spline logits or NULL
  const float* wpack;                    // packed conditioner weights, layout below
  unsigned wpack_bytes;
  int32_t* bad;
  int32_t* sat;                          // fp16 split-half path: tiles with a value beyond the fp16 range (or NULL)
  int32_t* redo;                         // [ceil(B / kFusedFlagRows)] or NULL.  Split-half kernels: OUT - 1 for the rows
                                         // of a tile that held a non-finite input or a value beyond +-65504 (nothing of
                                         // that tile is written; a 128-sample tile sets its four entries), else 0.
                                         // Exact fp32 kernel: IN - only flagged rows are evaluated and written.
  long long B;
  int ld_mode;
  float ld_sign;
  RqsConst c;
};

// A run of coupling layers of ONE shape evaluated by one launch (small batches: the tile stays in LDS from the first
// layer to the last, reference loop: normflow/core.py:144-183).  Per layer: index vectors, the unconditional
// spline's logits, the packed conditioner weights.  FusedArgs::tf_idx / id_idx / sh_* / wpack are those of layer 0.
constexpr int kMaxStackLayers = 16;
struct FusedLayerDesc {
  const int32_t* tf_idx;
  const int32_t* id_idx;
  const float *sh_w, *sh_h, *sh_d;
  const float* wpack;
};
struct FusedStackArgs {
  FusedArgs a;
  int n_layers;
  FusedLayerDesc lay[kMaxStackLayers];
};

__device__ __forceinline__ floatx4 mfma4(float a, float b, floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-byte load through a buffer descriptor: address = base + voff (per lane) + soff
// (wave-uniform, normally a compile-time constant).  All weight traffic goes through
// one descriptor with the lane part fixed at lane*16, so no load needs 64-bit address
// arithmetic in vector registers (thousands of fully unrolled loads otherwise spill
// their precomputed addresses).
__device__ __forceinline__ floatx4 wload(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
  return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

// 16 bytes per lane straight from a buffer into LDS (buffer_load_dwordx4 ... lds): lane l lands at
// lds_base + 16 l, lds_base wave-uniform; completion is counted by vmcnt like any other load.
__device__ __forceinline__ void dma16_to_lds(__amdgpu_buffer_rsrc_t rsrc, void* lds_base, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void wait_vector_memory() { __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt(0)

// Packed weight buffer of one layer, in floats (host side: vcnf_amd/fused.py::pack_layer):
//   W0 [NB][NS0/4][64][4] | b0 [H] | per block: WA [NB][NSH/4][64][4] | ba [H] |
//   WB [NB][NSH/4][64][4] | bb [H] | (WC [NB][NSC/4][64][4] | bc [H] if C > 0) |
//   WF [NG][P4][NSH/4][64][4] | bf [NG][4][4*P4]
template <int DI, int DT, int C, int H, int NBLK, int K>
struct PackLayout {
  static constexpr int NB = H / 16, NS0 = (DI + C) / 4, NSH = H / 4, NSC = C / 4;
  static constexpr int P4 = (3 * K - 1 + 3) / 4, NG = DT / 4;
  static constexpr int W0 = 0;
  static constexpr int B0 = W0 + NB * NS0 * 64;
  static constexpr int BLK0 = B0 + H;
  static constexpr int WA = 0, BA = WA + NB * NSH * 64, WB = BA + H, BB = WB + NB * NSH * 64;
  static constexpr int WC = BB + H, BC = WC + NB * NSC * 64;
  static constexpr int BLK = (C > 0) ? BC + H : WC;           // floats per residual block
  static constexpr int WF = BLK0 + NBLK * BLK;
  static constexpr int BF = WF + NG * P4 * NSH * 64;
  static constexpr int TOTAL = BF + NG * 4 * 4 * P4;
};

// Logits of one element taken straight from accumulator registers: v[t], t = 4 b + r.
template <int K, int P4>
struct RegLogits {
  const floatx4 (&v)[P4];
  float scale, edge;
  __device__ __forceinline__ float at(int t) const { return v[t >> 2][t & 3]; }
  __device__ __forceinline__ float w(int k) const { return at(k); }
  __device__ __forceinline__ float h(int k) const { return at(K + k); }
  __device__ __forceinline__ float d(int k) const { return (k == 0 || k == K) ? edge : at(2 * K + k - 1); }
};

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr float kLoScale = 2048.f, kLoUnscale = 1.f / 2048.f;

__device__ __forceinline__ floatx4 mfma16h(half8 a, half8 b, floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// hi/lo halves of 4 consecutive accumulator registers (optionally ReLU'd first)
template <bool RELU>
__device__ __forceinline__ void split4(const floatx4 v, half4& hi, half4& lo) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    // ReLU and saturation at the fp16 range in one v_med3_f32 (no inf after the conversion)
    const float x = __builtin_amdgcn_fmed3f(v[r], RELU ? 0.f : -65504.f, 65504.f);
    const _Float16 hv = (_Float16)x;
    hi[r] = hv;
    lo[r] = (_Float16)((x - (float)hv) * kLoScale);
  }
}

// same with the lower clamp as a run-time value (0: ReLU, -65504: saturation only)
__device__ __forceinline__ void split4v(const floatx4 v, float lower, half4& hi, half4& lo) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float x = __builtin_amdgcn_fmed3f(v[r], lower, 65504.f);
    const _Float16 hv = (_Float16)x;
    hi[r] = hv;
    lo[r] = (_Float16)((x - (float)hv) * kLoScale);
  }
}


// Packed weight buffer of the fp16 split-half kernel (fused_layer_v6.hip; host side:
// vcnf_amd/fused.py::pack_layer_h3), in floats; every matrix is stored as A fragments of
// v_mfma_f32_32x32x16_f16: [row block of 32][k-step of 16][hi | lo][lane][8 halves] (256 floats per fragment),
// every bias in accumulator order [row block][lane half][16].  Same total size as PackLayout.
//   W0 | b0 | per block: WA | ba | WB | bb | (WC | bc if C > 0) |
//   WF [group of 4 features][3 row blocks][8 k-steps][hi | lo][lane][8] | bf [group][lane half][48]
template <int DI, int DT, int C, int H, int NBLK, int K>
struct PackLayout6 {
  static constexpr int NB = H / 32, NT0 = (DI + C) / 16, NTH = H / 16, NTC = C / 16, NG = DT / 4;
  static constexpr int W0 = 0;
  static constexpr int B0 = W0 + NB * NT0 * 512;
  static constexpr int BLK0 = B0 + H;
  static constexpr int WA = 0, BA = WA + NB * NTH * 512, WB = BA + H, BB = WB + NB * NTH * 512;
  static constexpr int WC = BB + H, BC = WC + NB * NTC * 512;
  static constexpr int BLK = (C > 0) ? BC + H : WC;
  static constexpr int WF = BLK0 + NBLK * BLK;
  static constexpr int BF = WF + NG * 3 * NTH * 512;
  static constexpr int TOTAL = BF + NG * 96;
  static_assert(TOTAL == PackLayout<DI, DT, C, H, NBLK, K>::TOTAL, "both matrix paths take a buffer of the same size");
};

// defined in fused_layer.hip built with -DVCNF_F32_NBLK=1 / 3 (exact fp32 kernel, one / three residual blocks)
int launch_fused_f32_b1(const FusedStackArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st);
int launch_fused_f32_b3(const FusedStackArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st);

// defined in fused_layer_v6.hip
int launch_fused_v6_b1(const FusedArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st);
int launch_fused_v6_b2(const FusedArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st);
int launch_fused_v6_b3(const FusedArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st);

// defined in fused_layer_v6s.hip (32-sample tiles: small batches)
int launch_fused_v6s_b1(const FusedStackArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st);
int launch_fused_v6s_b2(const FusedStackArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st);
int launch_fused_v6s_b3(const FusedStackArgs& a, int d_id, int ctx_dim, int inverse, hipStream_t st);

}  // namespace vcnf
