// Dense layer of the conditioner at TRAINING batch sizes on the fp16 split-half matrix path:
//   y[b, n] = sum_k x[b, k] * A[n, k] (+ bias[n]),   A[n, k] = w[n * ldn + k * ldk]
// which is nn.Linear's forward (w = weight [N, K]: ldn = K, ldk = 1; normflow/nets/resnet.py:42-57, 92-106) and its
// input gradient g_x = g_y W (w = weight [K', N']: n runs over the layer's inputs, ldn = 1, ldk = N').  The weights
// are read in their NATURAL fp32 layout and split in registers - a training step changes them every step, a packed
// copy per layer and direction would cost ~200 extra launches per step.  Same arithmetic as the fused inference
// kernels (split_half.hpp): hi*hi + (hi*lo + lo*hi) * 2^-11 on v_mfma_f32_32x32x16_f16, fp32 accumulation, plus
// lo*lo for reductions of at most 48 terms (tests/test_gpu_gemm_error.py: error against fp64 at or below an fp32
// GEMM's).  Values beyond +-65504 are clamped and counted (sat), as in the other training-side split-half kernels.
//
// The library's fp32 GEMMs run these shapes at 40-90 TFLOP/s (48 us for 131072 x 128 x 128, 276 us for the last
// layer's input gradient); the layer is memory-bound on this path: x read once, y written once.
//
// Work split: 256 threads, tile = 64 samples (two 32-sample column blocks); the tile's rows are staged once per 128-deep
// chunk of k as ready-made B fragments (hi | lo) in LDS; wave w owns the 32-row blocks nb = w, w + 4, ... of A with the
// block's fragments in registers for the chunk (each used for both column blocks); results leave through a per-wave
// LDS strip so that every store instruction writes whole 128-byte lines.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcnf_hip.h"
#include "fused_common.hpp"
#include "split_half.hpp"

namespace vcnf {

struct ShLinearArgs {
  const float* x;       // [B, K]
  const float* w;       // see above
  const float* bias;    // [N] or NULL
  float* y;             // [B, N]
  long long B;
  int K, N;
  long long ldn, ldk;
  int relu_in, relu_out;   // ReLU on x while it is staged / on y before it is stored (nets/resnet.py:42, :46)
  const float* mask;       // [B, N] or NULL: y *= (mask > 0)   - the ReLU's backward on an input gradient
  const float* addend;     // [B, N] or NULL: y += addend        - the skip connection's gradient (after the mask)
  int32_t* sat;
};

constexpr int kShTile = 64;
constexpr int kShChunk = 128;          // k values staged per pass (8 k-steps of 16)

// WROWS: ldk == 1 (the k values of a row of A are contiguous: nn.Linear's forward) - a lane's eight values are two
// 16-byte loads; one 4-byte load each otherwise (input gradient: then the 32 rows of a block are contiguous instead
// and the loads coalesce across lanes).  As eight 4-byte loads from 64 different lines the forward was bound by the
// address path: 79 us for 131 072 x 128 x 128 against the library's 51.
template <bool LOLO, bool WROWS>
__global__ __launch_bounds__(256, (!LOLO && !WROWS) ? 3 : 2) void linear_f16x3_kernel(const ShLinearArgs a) {
  // the input-gradient form requests the mask / addend values of a column block's four row groups together (below);
  // the forward forms keep the plain epilogue and their register allocation (measured: with the batched epilogue and
  // three workgroups per compute unit forced, 67 -> 74 us per forward launch of the C3 training step, 59 -> 54 us per
  // input-gradient launch)
  constexpr bool EPI = !WROWS;
  constexpr int NCB = kShTile / 32;
  constexpr int NT = kShChunk / 16;
  extern __shared__ __align__(16) float smem[];
  uint4* fhi = reinterpret_cast<uint4*>(smem);              // [t][cb][lane]
  uint4* flo = fhi + NT * NCB * 64;
  float* strip = reinterpret_cast<float*>(flo + NT * NCB * 64);   // [4 waves][32 samples][36]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c32 = lane & 31, kg = lane >> 5;
  const int K = a.K, N = a.N;
  const int nblocks = (N + 31) / 32;
  const int nchunks = (K + kShChunk - 1) / kShChunk;
  const long long b0 = (long long)blockIdx.x * kShTile;
  const int rows = (int)min((long long)kShTile, a.B - b0);
  float satm = 0.f;
  float* mystrip = strip + wave * 32 * 36;

  // K > 128: one row block per wave, accumulators live across the chunks; K <= 128: the wave walks its row blocks
  floatx16 mainv[NCB], corr[NCB], corr2[NCB];
  for (int ch = 0; ch < nchunks; ++ch) {
    const int k0 = ch * kShChunk;
    const int nt = min(NT, (K - k0) / 16);
    __syncthreads();
    // ---- stage x[b0 .. b0 + 63][k0 .. k0 + 16 nt): thread = (row, 8 consecutive k) -> one hi and one lo fragment entry
    // (a thread's up to four segments are requested two at a time, then split and stored: two round trips per chunk
    // instead of four; all four at once costs 22 registers and the third wave per SIMD)
    constexpr int SI = 2;
    for (int i0 = tid; i0 < kShTile * nt * 2; i0 += 256 * SI) {
      float4 xp[SI], xq[SI];
#pragma unroll
      for (int u = 0; u < SI; ++u) {
        const int i = i0 + 256 * u;
        const int r = i / (nt * 2), j = i - r * (nt * 2);       // j: 8-float segment of the chunk
        xp[u] = xq[u] = float4{0.f, 0.f, 0.f, 0.f};
        if (i < kShTile * nt * 2 && r < rows) {
          const float4* src = reinterpret_cast<const float4*>(a.x + (b0 + r) * K + k0 + 8 * j);
          xp[u] = src[0];
          xq[u] = src[1];
        }
      }
#pragma unroll
      for (int u = 0; u < SI; ++u) {
        const int i = i0 + 256 * u;
        if (i >= kShTile * nt * 2) continue;
        const int r = i / (nt * 2), j = i - r * (nt * 2);
        const float4 p = xp[u], q = xq[u];
        float v8[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) satm = fmaxf(satm, v8[e] == v8[e] ? 0.f : __builtin_inff());   // NaN inputs count
        half8 h8, l8;
        if (a.relu_in) split8<true>(v8, h8, l8, satm);
        else split8<false>(v8, h8, l8, satm);
        const int at = ((j >> 1) * NCB + (r >> 5)) * 64 + (r & 31) + 32 * (j & 1);
        fhi[at] = __builtin_bit_cast(uint4, h8);
        flo[at] = __builtin_bit_cast(uint4, l8);
      }
    }
    __syncthreads();
    for (int nb = wave; nb < nblocks; nb += 4) {
      const int n = min(nb * 32 + c32, N - 1);               // rows past N repeat the last one (never stored)
      if (ch == 0) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int nr = nb * 32 + 8 * (r >> 2) + 4 * kg + (r & 3);
            mainv[cb][r] = (a.bias && nr < N) ? a.bias[nr] : 0.f;
          }
          corr[cb] = floatx16{};
          corr2[cb] = floatx16{};
        }
      }
      const float* wrow = a.w + (long long)n * a.ldn + (long long)(k0 + 8 * kg) * a.ldk;
      // (requesting the whole chunk's A values before the first use costs 80 registers and the third wave per SIMD:
      // 52 instead of 42 us for 131 072 x 128 x 128 - the k-step loop below is what is shipped)
      for (int t = 0; t < nt; ++t) {
        float av[8];
        if (WROWS) {
          const float4* src = reinterpret_cast<const float4*>(wrow + 16 * t);
          const float4 p = src[0], q = src[1];
          av[0] = p.x; av[1] = p.y; av[2] = p.z; av[3] = p.w; av[4] = q.x; av[5] = q.y; av[6] = q.z; av[7] = q.w;
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) av[i] = wrow[(long long)(16 * t + i) * a.ldk];
        }
        half8 ah, al;
        split8<false>(av, ah, al, satm);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const half8 bh = __builtin_bit_cast(half8, fhi[(t * NCB + cb) * 64 + lane]);
          const half8 bl = __builtin_bit_cast(half8, flo[(t * NCB + cb) * 64 + lane]);
          mainv[cb] = mfma32h(ah, bh, mainv[cb]);
          corr[cb] = mfma32h(ah, bl, corr[cb]);
          corr[cb] = mfma32h(al, bh, corr[cb]);
          if (LOLO) corr2[cb] = mfma32h(al, bl, corr2[cb]);
        }
      }
      if (ch + 1 == nchunks) {
        // ---- results: register r of lane (c32, kg) is row 8 (r / 4) + 4 kg + r % 4 of the block, column c32;
        // through the wave's strip [sample][36] so that a store instruction writes whole rows of 32 floats
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            floatx4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int r = 4 * j + q;
              v[q] = LOLO ? fmaf(fmaf(corr2[cb][r], kLoUnscale, corr[cb][r]), kLoUnscale, mainv[cb][r])
                          : fmaf(corr[cb][r], kLoUnscale, mainv[cb][r]);
              if (a.relu_out) v[q] = fmaxf(v[q], 0.f);
            }
            *reinterpret_cast<floatx4*>(mystrip + c32 * 36 + 8 * j + 4 * kg) = v;
          }
          // (LDS operations of one wave execute in order: no barrier between the writes above and the reads below)
          // input-gradient form: the four row groups' mask / addend values are requested together, ahead of the stores
          // that use them (one round trip instead of four, eight with both)
          floatx4 mk[4], ad[4];
          if (EPI) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = cb * 32 + 8 * i + (lane >> 3), col = nb * 32 + 4 * (lane & 7);
              const bool whole = row < rows && col + 3 < N;
              mk[i] = floatx4{1.f, 1.f, 1.f, 1.f};
              ad[i] = floatx4{0.f, 0.f, 0.f, 0.f};
              if (a.mask && whole) mk[i] = *reinterpret_cast<const floatx4*>(a.mask + (b0 + row) * N + col);
              if (a.addend && whole) ad[i] = *reinterpret_cast<const floatx4*>(a.addend + (b0 + row) * N + col);
            }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int s = 8 * i + (lane >> 3), piece = lane & 7;
            const floatx4 v = *reinterpret_cast<const floatx4*>(mystrip + s * 36 + 4 * piece);
            const int row = cb * 32 + s, col = nb * 32 + 4 * piece;
            if (row < rows) {
              float* dst = a.y + (b0 + row) * N + col;
              if (col + 3 < N) {
                floatx4 o = v;
                if (a.mask) {
                  const floatx4 m = EPI ? mk[i] : *reinterpret_cast<const floatx4*>(a.mask + (b0 + row) * N + col);
#pragma unroll
                  for (int q = 0; q < 4; ++q) o[q] = m[q] > 0.f ? o[q] : 0.f;
                }
                if (a.addend) o += EPI ? ad[i] : *reinterpret_cast<const floatx4*>(a.addend + (b0 + row) * N + col);
                *reinterpret_cast<floatx4*>(dst) = o;
              } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  if (col + q < N) {
                    float o = v[q];
                    if (a.mask) o = a.mask[(b0 + row) * N + col + q] > 0.f ? o : 0.f;
                    if (a.addend) o += a.addend[(b0 + row) * N + col + q];
                    dst[q] = o;
                  }
                }
              }
            }
          }
        }
      }
    }
  }
  if (a.sat && satm > 65504.f) atomicAdd(a.sat, 1);
}

}  // namespace vcnf

using namespace vcnf;

extern "C" int vcnf_linear_f16x3_supported(int32_t k, int32_t n) {
  // whole k-steps of 16; a reduction longer than one staged chunk keeps its accumulators in registers: one row block
  // per wave then (n <= 128)
  if (k < 16 || k % 16 || n < 1 || (n % 4)) return 0;
  if (k > kShChunk && n > 128) return 0;
  return 1;
}

extern "C" int vcnf_linear_f16x3_f32(const float* x, const float* w, const float* bias, float* y, int64_t batch,
                                     int32_t k, int32_t n, int64_t ldn, int64_t ldk, int relu_input, int relu_output,
                                     const float* mask, const float* addend, int32_t* sat_count, void* stream) {
  if (!x || !w || !y) return VCNF_ERR_NULL;
  if (batch < 0) return VCNF_ERR_SHAPE;
  if (!vcnf_linear_f16x3_supported(k, n)) return VCNF_ERR_UNSUPPORTED;
  if (batch == 0) return VCNF_OK;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return VCNF_ERR_ALIGN;
  if ((reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(addend)) & 15) return VCNF_ERR_ALIGN;
  ShLinearArgs a{x, w, bias, y, (long long)batch, k, n, (long long)ldn, (long long)ldk, relu_input ? 1 : 0,
                 relu_output ? 1 : 0, mask, addend, sat_count};
  const size_t lds = (size_t)2 * (kShChunk / 16) * (kShTile / 32) * 64 * 16 + 4 * 32 * 36 * 4;
  dim3 grid((unsigned)((batch + kShTile - 1) / kShTile));
  hipStream_t st = (hipStream_t)stream;
  const bool wrows = ldk == 1 && (ldn % 4) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0;
  if (k <= 48) {
    if (wrows) hipLaunchKernelGGL((linear_f16x3_kernel<true, true>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((linear_f16x3_kernel<true, false>), grid, dim3(256), lds, st, a);
  } else {
    if (wrows) hipLaunchKernelGGL((linear_f16x3_kernel<false, true>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((linear_f16x3_kernel<false, false>), grid, dim3(256), lds, st, a);
  }
  return hipGetLastError() == hipSuccess ? VCNF_OK : VCNF_ERR_LAUNCH;
}
