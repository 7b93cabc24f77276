// ResidualNet trunk (everything in front of final_layer) of an RQS coupling's conditioner in ONE launch, for layer
// shapes outside the one-kernel families (config C5: 512 identity features -> hidden 128, 2 blocks):
//     h = W0 x + b0 ;  per block  h += Wb relu(Wa relu(h) + ba) + bb            (nets/resnet.py:92-106, :42-57)
// The last layer and the splines run in csrc/fused_final.hip on this kernel's output h[B, 128].  On PyTorch-ROCm the
// trunk was ~9 launches per layer and direction (3 GEMMs + elementwise kernels): config C5 at its 16 384-sample
// micro-batch spent 14 of 31 ms per step there.
//
// Work split as in fused_affine.hip: a wave owns 16 samples from input to output, nothing is exchanged between
// waves, no barrier.  Every layer runs on v_mfma_f32_16x16x4_f32 (exact fp32 products and accumulation: the result
// differs from the reference's GEMMs only in summation order) with the weights as the A operand.  A layer's
// accumulators ARE the next layer's B operand (register r of row block pb = unit 16 pb + 4 q + r of sample lane & 15
// in lane group q = lane >> 4 = k-step 4 pb + r); the first layer reads its input with one 16-byte load per lane and
// four k-steps in the same k order (k = 16 j + 4 q + c), so all five weight matrices are packed alike on the host
// (vcnf_amd/fused_final.py::pack_trunk).  Weights stream from L2 with one 16-byte buffer load per four matrix
// instructions (0.5 MB per layer for config C5).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"
#include "fused_common.hpp"

namespace vcnf {

struct TrunkArgs {
  const float* x;        // [B, d_in] contiguous: (identity features | context)
  float* h;              // [B, 128]
  const float* wpack;
  unsigned wpack_bytes;
  long long B;
  int d_in;              // multiple of 16
};

constexpr int kTrunkBlock = 256;     // 4 waves x 16 samples
constexpr int kTH = 128, kTNB = kTH / 16;

// packed floats: W0 [8][d_in/16][64][4] | b0 [128] | per block: Wa [8][8][64][4] | ba [128] | Wb [8][8][64][4] | bb [128]
template <int NBLK>
__global__ __launch_bounds__(kTrunkBlock) void resnet_trunk_kernel(const TrunkArgs a) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15;
  const int q = lane >> 4;
  const __amdgpu_buffer_rsrc_t wr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpack), 0, a.wpack_bytes, 0x00020000);
  const int voff = lane * 16;
  const int qoff = q * 16;
  const int J = a.d_in / 16;
  const int off_b0 = kTNB * J * 256;
  const int off_blk = off_b0 + kTH;
  constexpr int kBlkFloats = 2 * (kTNB * kTNB * 256 + kTH);
  const long long nwt = (a.B + 15) / 16;
  const long long wstride = (long long)gridDim.x * (kTrunkBlock / 64);
  for (long long wt = (long long)blockIdx.x * (kTrunkBlock / 64) + wave; wt < nwt; wt += wstride) {
    const long long b0 = wt * 16;
    const long long left = a.B - b0;
    // rows of this tile through a bounds-checked descriptor: rows past the batch read 0
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x) + b0 * a.d_in, 0, (int)(min(left, 16LL) * a.d_in * 4), 0x00020000);
    // ---- first layer
    floatx4 h[kTNB];
#pragma unroll
    for (int nb = 0; nb < kTNB; ++nb) h[nb] = wload(wr, qoff, 4 * (off_b0 + 16 * nb));
    floatx4 xv = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(xr, (m16 * a.d_in + 4 * q) * 4, 0, 0));
    for (int j = 0; j < J; ++j) {
      const floatx4 xc = xv;
      if (j + 1 < J)
        xv = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(xr, (m16 * a.d_in + 16 * (j + 1) + 4 * q) * 4, 0, 0));
#pragma unroll
      for (int nb = 0; nb < kTNB; ++nb) {
        const floatx4 w = wload(wr, voff, 4 * ((nb * J + j) * 256));
#pragma unroll
        for (int r = 0; r < 4; ++r) h[nb] = mfma4(w[r], xc[r], h[nb]);
      }
    }
    // ---- residual blocks
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      const int base = off_blk + blk * kBlkFloats;
      floatx4 t[kTNB], u[kTNB];
#pragma unroll
      for (int nb = 0; nb < kTNB; ++nb) {
        floatx4 acc = wload(wr, qoff, 4 * (base + kTNB * kTNB * 256 + 16 * nb));
#pragma unroll
        for (int pb = 0; pb < kTNB; ++pb) {
          const floatx4 w = wload(wr, voff, 4 * (base + (nb * kTNB + pb) * 256));
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = mfma4(w[r], fmaxf(h[pb][r], 0.f), acc);
        }
        t[nb] = acc;
      }
      const int base2 = base + kTNB * kTNB * 256 + kTH;
#pragma unroll
      for (int nb = 0; nb < kTNB; ++nb) {
        floatx4 acc = wload(wr, qoff, 4 * (base2 + kTNB * kTNB * 256 + 16 * nb));
#pragma unroll
        for (int pb = 0; pb < kTNB; ++pb) {
          const floatx4 w = wload(wr, voff, 4 * (base2 + (nb * kTNB + pb) * 256));
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = mfma4(w[r], fmaxf(t[pb][r], 0.f), acc);
        }
        u[nb] = acc;
      }
#pragma unroll
      for (int nb = 0; nb < kTNB; ++nb) h[nb] += u[nb];
    }
    // ---- out: lane holds units 16 nb + 4 q .. + 3 of sample m16
    if (m16 < left) {
      float* dst = a.h + (b0 + m16) * kTH + 4 * q;
#pragma unroll
      for (int nb = 0; nb < kTNB; ++nb) *reinterpret_cast<floatx4*>(dst + 16 * nb) = h[nb];
    }
  }
}

template <int NBLK>
static int launch_trunk(const TrunkArgs& a, hipStream_t st) {
  const long long ntiles = (a.B + 63) / 64;
  const long long cap = 256 * 8;
  dim3 grid((unsigned)(ntiles < cap ? ntiles : cap));
  hipLaunchKernelGGL((resnet_trunk_kernel<NBLK>), grid, dim3(kTrunkBlock), 0, st, a);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

}  // namespace vcnf

using namespace vcnf;

extern "C" int vcnf_resnet_trunk_supported(int32_t d_in, int32_t hidden, int32_t num_blocks) {
  return (hidden == 128 && d_in >= 16 && d_in <= 4096 && d_in % 16 == 0 && num_blocks >= 1 && num_blocks <= 3) ? 1 : 0;
}

extern "C" int64_t vcnf_resnet_trunk_pack_floats(int32_t d_in, int32_t hidden, int32_t num_blocks) {
  if (!vcnf_resnet_trunk_supported(d_in, hidden, num_blocks)) return 0;
  return (int64_t)kTNB * (d_in / 16) * 256 + kTH + (int64_t)num_blocks * 2 * (kTNB * kTNB * 256 + kTH);
}

extern "C" int vcnf_resnet_trunk_f32(const float* x, float* h, int64_t batch, int32_t d_in, int32_t hidden,
                                     int32_t num_blocks, const float* wpack, int64_t wpack_floats, void* stream) {
  if (!vcnf_resnet_trunk_supported(d_in, hidden, num_blocks)) return VCNF_ERR_UNSUPPORTED;
  if (batch < 0) return VCNF_ERR_SHAPE;
  if (wpack_floats != vcnf_resnet_trunk_pack_floats(d_in, hidden, num_blocks)) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!x || !h || !wpack) return VCNF_ERR_NULL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(h)) & 15) return VCNF_ERR_ALIGN;
  TrunkArgs a;
  a.x = x; a.h = h; a.wpack = wpack; a.wpack_bytes = (unsigned)(wpack_floats * 4); a.B = batch; a.d_in = d_in;
  hipStream_t st = (hipStream_t)stream;
  if (num_blocks == 1) return launch_trunk<1>(a, st);
  if (num_blocks == 2) return launch_trunk<2>(a, st);
  return launch_trunk<3>(a, st);
}
