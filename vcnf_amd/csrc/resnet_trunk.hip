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
  float* h;              // [B, 128] fp32, or (SPLIT) [B][hi 128 halves | lo 128 halves] for csrc/fused_final.hip
  int32_t* sat;          // SPLIT: counts workgroups that clamped a value at the fp16 range
  const float* wpack;
  unsigned wpack_bytes;
  long long B;
  int d_in;              // multiple of 16
};

constexpr int kTrunkBlock = 256;     // 4 waves x 16 samples
constexpr int kTH = 128, kTNB = kTH / 16;

// packed floats: W0 [8][d_in/16][64][4] | b0 [128] | per block: Wa [8][8][64][4] | ba [128] | Wb [8][8][64][4] | bb [128]
// NT: 16-sample column tiles per wave.  Every weight fragment is a 1 KB load per wave for 4 NT matrix instructions;
// with one tile the four waves of a CU ask the vector cache for ~128 B per clock, twice what it delivers, so large
// batches run two tiles per wave (same fragment, two accumulator sets).
template <int NBLK, int NT, bool SPLIT>
__global__ __launch_bounds__(kTrunkBlock) void resnet_trunk_kernel(const TrunkArgs a) {
  float satm = 0.f;
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
  constexpr int TS = 16 * NT;                    // samples per wave tile
  const long long nwt = (a.B + TS - 1) / TS;
  const long long wstride = (long long)gridDim.x * (kTrunkBlock / 64);
  for (long long wt = (long long)blockIdx.x * (kTrunkBlock / 64) + wave; wt < nwt; wt += wstride) {
    const long long b0 = wt * TS;
    const long long left = a.B - b0;
    // rows of this tile through a bounds-checked descriptor: rows past the batch read 0
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x) + b0 * a.d_in, 0, (int)(min(left, (long long)TS) * a.d_in * 4), 0x00020000);
    // ---- first layer
    floatx4 h[NT][kTNB];
#pragma unroll
    for (int nb = 0; nb < kTNB; ++nb) {
      const floatx4 b = wload(wr, qoff, 4 * (off_b0 + 16 * nb));
#pragma unroll
      for (int c = 0; c < NT; ++c) h[c][nb] = b;
    }
    floatx4 xv[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c)
      xv[c] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(xr, ((16 * c + m16) * a.d_in + 4 * q) * 4, 0, 0));
    for (int j = 0; j < J; ++j) {
      floatx4 xc[NT];
#pragma unroll
      for (int c = 0; c < NT; ++c) xc[c] = xv[c];
      if (j + 1 < J) {
#pragma unroll
        for (int c = 0; c < NT; ++c)
          xv[c] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  xr, ((16 * c + m16) * a.d_in + 16 * (j + 1) + 4 * q) * 4, 0, 0));
      }
#pragma unroll
      for (int nb = 0; nb < kTNB; ++nb) {
        const floatx4 w = wload(wr, voff, 4 * ((nb * J + j) * 256));
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < NT; ++c) h[c][nb] = mfma4(w[r], xc[c][r], h[c][nb]);
      }
    }
    // ---- residual blocks
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      const int base = off_blk + blk * kBlkFloats;
      floatx4 t[NT][kTNB];
#pragma unroll
      for (int nb = 0; nb < kTNB; ++nb) {
        const floatx4 b = wload(wr, qoff, 4 * (base + kTNB * kTNB * 256 + 16 * nb));
        floatx4 acc[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c] = b;
#pragma unroll
        for (int pb = 0; pb < kTNB; ++pb) {
          const floatx4 w = wload(wr, voff, 4 * (base + (nb * kTNB + pb) * 256));
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = mfma4(w[r], fmaxf(h[c][pb][r], 0.f), acc[c]);
        }
#pragma unroll
        for (int c = 0; c < NT; ++c) t[c][nb] = acc[c];
      }
      const int base2 = base + kTNB * kTNB * 256 + kTH;
      floatx4 u[NT][kTNB];
#pragma unroll
      for (int nb = 0; nb < kTNB; ++nb) {
        const floatx4 b = wload(wr, qoff, 4 * (base2 + kTNB * kTNB * 256 + 16 * nb));
        floatx4 acc[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c] = b;
#pragma unroll
        for (int pb = 0; pb < kTNB; ++pb) {
          const floatx4 w = wload(wr, voff, 4 * (base2 + (nb * kTNB + pb) * 256));
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = mfma4(w[r], fmaxf(t[c][pb][r], 0.f), acc[c]);
        }
#pragma unroll
        for (int c = 0; c < NT; ++c) u[c][nb] = acc[c];
      }
#pragma unroll
      for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int nb = 0; nb < kTNB; ++nb) h[c][nb] += u[c][nb];
    }
    // ---- out: lane holds units 16 nb + 4 q .. + 3 of sample 16 c + m16
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      if (16 * c + m16 < left) {
        if constexpr (!SPLIT) {
          float* dst = a.h + (b0 + 16 * c + m16) * kTH + 4 * q;
#pragma unroll
          for (int nb = 0; nb < kTNB; ++nb) *reinterpret_cast<floatx4*>(dst + 16 * nb) = h[c][nb];
        } else {
          // the consumer (last layer + spline kernel) multiplies h as fp16 hi + lo 2^-11 halves: split here, once,
          // instead of in every one of its feature-group workgroups (same arithmetic as fused_common.hpp::split4)
          _Float16* row = reinterpret_cast<_Float16*>(a.h + (b0 + 16 * c + m16) * kTH);
#pragma unroll
          for (int nb = 0; nb < kTNB; ++nb) {
            half4 hi, lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) satm = fmaxf(satm, __builtin_fabsf(h[c][nb][r]));
            split4<false>(h[c][nb], hi, lo);
            *reinterpret_cast<half4*>(row + 16 * nb + 4 * q) = hi;
            *reinterpret_cast<half4*>(row + kTH + 16 * nb + 4 * q) = lo;
          }
        }
      }
    }
  }
  if (SPLIT && a.sat && satm > 65504.f) atomicAdd(a.sat, 1);
}

template <int NBLK, int NT>
static int launch_trunk_nt(const TrunkArgs& a, bool split, hipStream_t st) {
  const long long ntiles = (a.B + 64 * NT - 1) / (64 * NT);
  const long long cap = 256 * 8;
  dim3 grid((unsigned)(ntiles < cap ? ntiles : cap));
  if (split) hipLaunchKernelGGL((resnet_trunk_kernel<NBLK, NT, true>), grid, dim3(kTrunkBlock), 0, st, a);
  else hipLaunchKernelGGL((resnet_trunk_kernel<NBLK, NT, false>), grid, dim3(kTrunkBlock), 0, st, a);
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}

// two tiles per wave once that still gives every SIMD of the chip a wave (256 CUs x 4 waves x 32 samples)
template <int NBLK>
static int launch_trunk(const TrunkArgs& a, bool split, hipStream_t st) {
  return a.B >= 2 * 256 * 4 * 32 ? launch_trunk_nt<NBLK, 2>(a, split, st) : launch_trunk_nt<NBLK, 1>(a, split, st);
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

static int run_trunk(const float* x, float* h, int64_t batch, int32_t d_in, int32_t hidden, int32_t num_blocks,
                     const float* wpack, int64_t wpack_floats, bool split, int32_t* sat, void* stream) {
  if (!vcnf_resnet_trunk_supported(d_in, hidden, num_blocks)) return VCNF_ERR_UNSUPPORTED;
  if (batch < 0) return VCNF_ERR_SHAPE;
  if (wpack_floats != vcnf_resnet_trunk_pack_floats(d_in, hidden, num_blocks)) return VCNF_ERR_SHAPE;
  if (batch == 0) return VCNF_OK;
  if (!x || !h || !wpack) return VCNF_ERR_NULL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(h)) & 15) return VCNF_ERR_ALIGN;
  TrunkArgs a;
  a.x = x; a.h = h; a.sat = sat; a.wpack = wpack; a.wpack_bytes = (unsigned)(wpack_floats * 4); a.B = batch; a.d_in = d_in;
  hipStream_t st = (hipStream_t)stream;
  if (num_blocks == 1) return launch_trunk<1>(a, split, st);
  if (num_blocks == 2) return launch_trunk<2>(a, split, st);
  return launch_trunk<3>(a, split, st);
}

extern "C" int vcnf_resnet_trunk_f32(const float* x, float* h, int64_t batch, int32_t d_in, int32_t hidden,
                                     int32_t num_blocks, const float* wpack, int64_t wpack_floats, void* stream) {
  return run_trunk(x, h, batch, d_in, hidden, num_blocks, wpack, wpack_floats, false, nullptr, stream);
}

/* same trunk, output already split for vcnf_rqs_final_fused_presplit_f32: row b of `h_split` (128 floats wide) holds 128
 * fp16 hi halves followed by 128 fp16 lo halves (h ~ hi + lo / 2048, clamped at +-65504 and counted in sat_count) */
extern "C" int vcnf_resnet_trunk_split_f32(const float* x, float* h_split, int64_t batch, int32_t d_in, int32_t hidden,
                                           int32_t num_blocks, const float* wpack, int64_t wpack_floats,
                                           int32_t* sat_count, void* stream) {
  return run_trunk(x, h_split, batch, d_in, hidden, num_blocks, wpack, wpack_floats, true, sat_count, stream);
}
