/*
 * vcnf_hip.h - C ABI of the MI355X (gfx950) coupling-flow transform engine.
 *
 * The reference (telegraphroad/VCNF, a fork of normflow 1.2) is pure Python: it
 * has no FFI.  Its hot path sits behind Python classes
 * (normflow/flows/base.py:6-21, Flow.forward / Flow.inverse).  This header is
 * the boundary a binding for that path attaches to: every entry point replaces
 * the arithmetic of the reference function cited above it, on device buffers
 * the caller owns.  Conventions shared by all entry points:
 *
 *   - plain pointers + sizes, fp32 row-major dense buffers in HBM, int32 index
 *     vectors on the device; no allocation, no host sync, no host callbacks;
 *   - `stream` is a hipStream_t (NULL = default stream); the call only enqueues;
 *   - return value: VCNF_OK or a VCNF_ERR_* code (vcnf_status_string());
 *     arguments are validated on the host BEFORE anything is launched;
 *   - `logdet` handling: ld_mode VCNF_LD_STORE writes sign*sum, VCNF_LD_ACCUM
 *     adds sign*sum to what is there (the `log_q += / -= log_det` of
 *     normflow/core.py:153-155, :179-181 folded into the kernel);
 *   - direction names follow the reference's nsf code: `inverse == 0` is the
 *     density direction of a spline (search x knots), `inverse != 0` the
 *     sampling direction (search y knots, quadratic root).
 *
 * See INTEGRATION.md for the ctypes binding that ships in vcnf_amd/_lib.py.
 */
#ifndef VCNF_HIP_H
#define VCNF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VCNF_ABI_VERSION 1

enum {
  VCNF_OK = 0,
  VCNF_ERR_NULL = 1,        /* a required pointer is NULL */
  VCNF_ERR_SHAPE = 2,       /* sizes inconsistent / out of the supported range */
  VCNF_ERR_ALIGN = 3,       /* a buffer is not 4-byte aligned */
  VCNF_ERR_VALUE = 4,       /* e.g. min_bin_width * K > 1 (splines.py:104-107) */
  VCNF_ERR_UNSUPPORTED = 5, /* unknown enum value (tails, scale map) */
  VCNF_ERR_LAUNCH = 6       /* hipGetLastError() after the launch */
};

enum { VCNF_LD_STORE = 0, VCNF_LD_ACCUM = 1 };
enum { VCNF_TAILS_NONE = 0, VCNF_TAILS_LINEAR = 1, VCNF_TAILS_CIRCULAR = 2 };
enum { VCNF_PREC_F32 = 0, VCNF_PREC_F16X3 = 1 };   /* matrix path of the fused layer kernel */
enum { VCNF_SCALE_EXP = 0, VCNF_SCALE_SIGMOID = 1, VCNF_SCALE_SIGMOID_INV = 2, VCNF_SCALE_NONE = 3 };

int vcnf_abi_version(void);
const char* vcnf_status_string(int status);

/* Spline constants.  Replaces the keyword arguments of
 * unconstrained_rational_quadratic_spline / rational_quadratic_spline
 * (normflow/utils/splines.py:20-29, :88-96). */
typedef struct vcnf_rqs_cfg {
  int32_t num_bins;        /* K */
  int32_t tails;           /* VCNF_TAILS_* ; LINEAR: K-1 derivative logits, identity outside;
                              CIRCULAR: K logits (last knot = first knot, splines.py:44-49), identity outside */
  float left, right;       /* x interval; LINEAR uses [-tail_bound, tail_bound] */
  float bottom, top;       /* y interval */
  float min_bin_width;     /* splines.py:6 */
  float min_bin_height;    /* splines.py:7 */
  float min_derivative;    /* splines.py:8 */
  float wh_scale;          /* multiplies width/height logits: 1/sqrt(hidden_features),
                              flows/neural_spline/coupling.py:314-316; 1 for none */
} vcnf_rqs_cfg;

/* Elementwise spline on n independent elements with per-element parameters.
 * Replaces splines.py:20-85 (tails=LINEAR) and :88-193 (tails=NONE).
 * uw/uh/ud: element i reads uw[i*ld_w + k], k<K; uh[i*ld_h + k]; ud[i*ld_d + k],
 * k < K-1 (LINEAR) or K+1 (NONE).  bad_disc (optional, device int32) counts
 * negative discriminants, the condition the reference asserts on (splines.py:164). */
int vcnf_rqs_elementwise_f32(const float* x, const float* uw, const float* uh, const float* ud,
                             int64_t ld_w, int64_t ld_h, int64_t ld_d,
                             float* y, float* logabsdet, int64_t n,
                             const vcnf_rqs_cfg* cfg, int inverse, int32_t* bad_disc, void* stream);

/* Same spline with general addressing of the logit rows (image-shaped couplings,
 * coupling.py:148-151, and rows shared by the whole batch, coupling.py:211-240):
 *   r = period > 0 ? i % period : i;  outer = r / inner;  s = r % inner;
 *   logit k of element i = u?[outer * row_? + s + k * k_stride].
 * Conditioner output [B, C*P, H, W] of a 4-D coupling: inner = H*W, k_stride = H*W,
 * row_* = P*H*W, uh = uw + K*H*W, ud = uw + 2*K*H*W.  Per-pixel logits [C,H,W,K] shared by
 * the batch: inner = 1, k_stride = 1, row_* = K (K-1 | K | K+1 for ud), period = C*H*W. */
int vcnf_rqs_elementwise_strided_f32(const float* x, const float* uw, const float* uh, const float* ud,
                                     int64_t row_w, int64_t row_h, int64_t row_d, int64_t inner,
                                     int64_t k_stride, int64_t period,
                                     float* y, float* logabsdet, int64_t n,
                                     const vcnf_rqs_cfg* cfg, int inverse, int32_t* bad_discriminant,
                                     void* stream);

/* Whole AffineCouplingBlock on x[B, features] in one launch when the conditioner is an MLP with two
 * hidden layers of equal width: Linear(c_in, hidden), LeakyReLU, Linear(hidden, hidden), LeakyReLU,
 * Linear(hidden, n_out), n_out = 2*d_t (interleaved shift, scale) or d_t (scale map NONE).
 * Replaces flows/affine/coupling.py:247-258 with :113-168, reshape.py:25-29, :50-55 and
 * nets/mlp.py:30-58 for that case.  The conditioner reads x[:, cond_off : cond_off + c_in], the
 * features [t_off, t_off + d_t) are transformed, the rest is copied.  hidden in {32, 64, 128},
 * c_in <= 64, n_out <= 128.  wpack: vcnf_affine_layer_fused_pack_floats(...) floats in the
 * fragment order documented in vcnf_amd/fused_affine.py.  in_gather / out_gather (int32[features],
 * may be NULL) fold a neighbouring Permute (flows/mixing.py:32-54) into the layer: the block is
 * applied to x[:, in_gather] and the result returned as result[:, out_gather]. */
int vcnf_affine_layer_fused_supported(int32_t c_in, int32_t hidden, int32_t n_out, int32_t features);
int64_t vcnf_affine_layer_fused_pack_floats(int32_t c_in, int32_t hidden, int32_t n_out);
int vcnf_affine_layer_fused_f32(const float* x, float* y, float* logdet, int64_t batch, int32_t features,
                                int32_t cond_off, int32_t c_in, int32_t t_off, int32_t d_t,
                                int32_t hidden, float leaky_slope, int scale_map,
                                const float* wpack, int64_t wpack_floats,
                                const int32_t* in_gather, const int32_t* out_gather,
                                int inverse, int ld_mode, float ld_sign, void* stream);

/* Per-position splines whose logits are shared by the whole batch (coupling.py:211-240): x [n] with
 * position i % period, logits sw, sh [period, K], sd [period, K-1 | K | K+1].  `tables` is a caller
 * workspace of period * 3 * (K + 1) floats: the knots of every position are generated once into it,
 * the elements then search their row (no softmax per element). */
int vcnf_rqs_shared_f32(const float* x, const float* sw, const float* sh, const float* sd, int64_t period,
                        float* tables, float* y, float* logabsdet, int64_t n,
                        const vcnf_rqs_cfg* cfg, int inverse, int32_t* bad_discriminant, void* stream);

/* Vector-Jacobian product of vcnf_rqs_elementwise_f32 (training path; the reference
 * obtains it from autograd over utils/splines.py:88-193).  Inputs as the forward call
 * plus the upstream gradients g_y[n], g_logabsdet[n]; outputs g_x[n] and dense
 * g_uw[n,K], g_uh[n,K], g_ud[n, K-1 | K+1].  K <= 64. */
int vcnf_rqs_elementwise_bwd_f32(const float* x, const float* uw, const float* uh, const float* ud,
                                 int64_t ld_w, int64_t ld_h, int64_t ld_d,
                                 const float* g_y, const float* g_logabsdet,
                                 float* g_x, float* g_uw, float* g_uh, float* g_ud, int64_t n,
                                 const vcnf_rqs_cfg* cfg, int inverse, void* stream);

/* VJP of the spline with the logits read from (and their gradient written in) the conditioner's
 * own output layout: params / g_params [rows, P, inner] with P = 2K + (K-1 | K | K+1), x [rows*inner]
 * (2-D coupling: inner = 1, rows = B*d_t; image coupling: inner = H*W, rows = B*C_t).  g_logabsdet
 * is read at i / lad_div (per-sample log-det: lad_div = elements per sample). */
int vcnf_rqs_packed_bwd_f32(const float* x, const float* params, int64_t inner, int64_t lad_div,
                            const float* g_y, const float* g_logabsdet,
                            float* g_x, float* g_params, int64_t n,
                            const vcnf_rqs_cfg* cfg, int inverse, void* stream);

/* VJP of the batch-shared spline (coupling.py:211-240): x [batch, period], logits sw, sh [period, K],
 * sd [period, nd].  Writes g_x and `groups` partial gradient rows per position,
 * partial [groups, period, 2K+nd] (sum over the first axis = gradient of the logits, row layout
 * w | h | d); groups must equal vcnf_rqs_shared_bwd_groups(batch, period).  K in {4, 8, 10, 16}. */
int64_t vcnf_rqs_shared_bwd_groups(int64_t batch, int64_t period);
int vcnf_rqs_shared_bwd_f32(const float* x, const float* sw, const float* sh, const float* sd,
                            int64_t batch, int64_t period, int64_t lad_div,
                            const float* g_y, const float* g_logabsdet,
                            float* g_x, float* partial, int64_t groups,
                            const vcnf_rqs_cfg* cfg, int inverse, void* stream);

/* Last conditioner layer + splines of an RQS coupling for any number of transformed features:
 * h [B, hidden] is the conditioner trunk's output (input of nets/resnet.py:105 final_layer), wpack the
 * final layer's weights and bias in matrix-core fragment order (vcnf_amd/fused_final.py).  Writes the
 * transformed columns y[:, transform_idx] (the caller has filled the identity columns) and the
 * per-group-block log|det| rows partial[vcnf_rqs_final_fused_partial_rows(d_t, K), B] whose sum over
 * rows is the transform half's log|det|.  Replaces nets/resnet.py:105, coupling.py:147-159 and
 * :309-343 without materialising the [B, d_t * (3K-1)] logits.  hidden = 128, linear tails, K in {8, 10, 16}. */
int vcnf_rqs_final_fused_supported(int32_t d_t, int32_t hidden, int32_t num_bins, int32_t tails);
int64_t vcnf_rqs_final_fused_pack_floats(int32_t d_t, int32_t hidden, int32_t num_bins);
int64_t vcnf_rqs_final_fused_partial_rows(int32_t d_t, int32_t num_bins);
int vcnf_rqs_final_fused_f32(const float* x, const float* h, float* y, float* partial,
                             int64_t batch, int32_t features, const int32_t* transform_idx, int32_t d_t,
                             int32_t hidden, const float* wpack, int64_t wpack_floats,
                             const vcnf_rqs_cfg* cfg, int inverse, int32_t* bad_discriminant, void* stream);

/* Trunk of a ResidualNet conditioner in one launch: h[B, 128] = everything in front of final_layer
 * (nets/resnet.py:92-105: initial_layer, then per block h += linear_1(relu(linear_0(relu(h)))), ReLU, no batch
 * norm, dropout 0, no context gate) for x[B, d_in] = the (identity | context) input, d_in a multiple of 16,
 * hidden 128, 1-3 blocks; exact fp32 matrix instructions.  Feeds vcnf_rqs_final_fused_f32 (config C5).
 * wpack: vcnf_resnet_trunk_pack_floats floats, every weight matrix as [row block of 16][k block of 16][lane][4]
 * fragments (k = 16 kb + 4 (lane >> 4) + c), biases in natural order (vcnf_amd/fused_final.py::pack_trunk). */
int vcnf_resnet_trunk_supported(int32_t d_in, int32_t hidden, int32_t num_blocks);
int64_t vcnf_resnet_trunk_pack_floats(int32_t d_in, int32_t hidden, int32_t num_blocks);
int vcnf_resnet_trunk_f32(const float* x, float* h, int64_t batch, int32_t d_in, int32_t hidden,
                          int32_t num_blocks, const float* wpack, int64_t wpack_floats, void* stream);

/* The pair with the trunk output handed over already split for the matrix path of the last-layer kernel: row b of
 * h_split (128 floats wide) holds 128 fp16 hi halves followed by 128 fp16 lo halves (h ~ hi + lo / 2048, values clamped
 * at +-65504 and counted in sat_count) - the split then happens once per sample instead of once per sample and
 * feature-group workgroup.  Same results as the fp32 hand-over, bit for bit. */
int vcnf_resnet_trunk_split_f32(const float* x, float* h_split, int64_t batch, int32_t d_in, int32_t hidden,
                                int32_t num_blocks, const float* wpack, int64_t wpack_floats, int32_t* sat_count,
                                void* stream);
int vcnf_rqs_final_fused_presplit_f32(const float* x, const float* h_split, float* y, float* partial,
                                      int64_t batch, int32_t features, const int32_t* transform_idx, int32_t d_t,
                                      int32_t hidden, const float* wpack, int64_t wpack_floats,
                                      const vcnf_rqs_cfg* cfg, int inverse, int32_t* bad_discriminant, void* stream);


/* One RQS coupling layer on x[B,D] -> y[B,D].
 * Replaces Coupling.forward / .inverse (flows/neural_spline/coupling.py:70-96 /
 * :98-125) minus the conditioner call, PiecewiseCoupling._coupling_transform
 * (:147-159), _piecewise_cdf (:309-343), the unconditional per-feature spline
 * (PiecewiseRationalQuadraticCDF._spline :211-240) and sum_except_batch
 * (utils/nn.py:131-134).
 *   params[B, d_t*P] dense, P = 3K-1 (LINEAR) or 3K+1 (NONE): conditioner output.
 *   transform_idx[d_t], identity_idx[d_id]: feature indices (buffers
 *     transform_features / identity_features, coupling.py:45-46), d_t+d_id == D.
 *   shared_w/h[d_id,K], shared_d[d_id, K-1 | K+1]: unconditional spline logits or
 *     all NULL when apply_unconditional_transform is off; never scaled by wh_scale.
 *   logdet[B]: per-sample sum over all D features, see ld_mode / sign above. */
int vcnf_rqs_coupling_f32(const float* x, const float* params,
                          const int32_t* transform_idx, int32_t d_t,
                          const int32_t* identity_idx, int32_t d_id,
                          const float* shared_w, const float* shared_h, const float* shared_d,
                          float* y, float* logdet, int64_t batch,
                          const vcnf_rqs_cfg* cfg, int inverse,
                          int ld_mode, float ld_sign, int32_t* bad_disc, void* stream);

/* Conditioner input for one RQS coupling layer: out[B, d_id + ctx_dim] =
 * cat(identity_split, context).  With apply_inverse_shared != 0 the identity
 * columns first go through the INVERSE unconditional spline, which is what the
 * sampling direction feeds its conditioner (coupling.py:110-114); otherwise the
 * raw gather of coupling.py:78 (and the concat of nets/resnet.py:100-102). */
int vcnf_rqs_conditioner_input_f32(const float* x, int64_t batch, int32_t features,
                                   const int32_t* identity_idx, int32_t d_id,
                                   const float* context, int32_t ctx_dim,
                                   const float* shared_w, const float* shared_h, const float* shared_d,
                                   const vcnf_rqs_cfg* cfg, int apply_inverse_shared,
                                   float* out, void* stream);

/* Per-pixel channel mixing of an image batch: y[b, o, p] = sum_c matrix[o, c] x[b, c, p] + shift[o] for x, y
 * [batch, channels, inner] (NCHW, inner = H W); matrix [channels, channels] row-major and shift [channels] on the device.
 * One launch for Glow's invertible 1x1 convolution (flows/mixing.py:57-128) composed with the neighbouring ActNorm
 * (flows/normalization.py:8-38, flows/affine/coupling.py:37-53); the caller composes matrix / shift from the layer
 * parameters and adds the constant log|det| (vcnf_amd/flows/affine/glow.py).  Exact fp32 products on
 * v_mfma_f32_16x16x4_f32.  channels: multiple of 4, 4..64 (vcnf_channel_mix_supported). */
int vcnf_channel_mix_supported(int32_t channels);
int vcnf_channel_mix_f32(const float* x, float* y, const float* matrix, const float* shift,
                         int64_t batch, int32_t channels, int64_t inner, void* stream);

/* 1x1 convolution of an NCHW batch with the bias adds and LeakyReLUs around it in one pass (nets/cnn.py:20-52, the
 * middle layer of the Glow conditioner flows/affine/glow.py:37-47):
 *   y[b, o, p] = act_out( sum_c W[o, c] * act_in(x[b, c, p] + in_bias[c]) + out_bias[o] ),  act(t) = t >= 0 ? t : slope t
 * x [batch, c_in, inner], y [batch, c_out, inner]; in_bias / out_bias may be NULL; in_act / out_act switch the
 * activations.  The caller runs the preceding convolution without its bias and passes that bias as in_bias.
 * Matrix path: fp16 split-half operands (hi + lo 2^-11, fp32 accumulation, 3 instructions per product); values beyond
 * +-65504 are clamped and counted in sat_count (device int32, may be NULL).  wpack: vcnf_conv1x1_pack_floats(c_in, c_out)
 * floats = W as A fragments of v_mfma_f32_32x32x16_f16, [8 row blocks][c_in / 16][hi | lo][64 lanes][8 halves]
 * (vcnf_amd/nets/cnn.py::pack_conv1x1).  c_in multiple of 16 up to 256, c_out up to 256. */
int vcnf_conv1x1_supported(int32_t c_in, int32_t c_out);
int64_t vcnf_conv1x1_pack_floats(int32_t c_in, int32_t c_out);
int vcnf_conv1x1_f16x3_f32(const float* x, float* y, const float* wpack, int64_t wpack_floats,
                           const float* in_bias, const float* out_bias, int64_t batch, int32_t c_in, int32_t c_out,
                           int64_t inner, int in_act, float in_slope, int out_act, float out_slope,
                           int32_t* sat_count, void* stream);

/* Weight and bias gradient of a dense conditioner layer y = x W^T + b for large batches (training path; the reference
 * gets them from autograd over nets/resnet.py:92-106):  dw[o, i] (+)= sum_b dy[b, o] x[b, i],  db[o] (+)= sum_b dy[b, o]
 * with x [batch, in_features], dy [batch, out_features] row-major; db may be NULL.  The reduction over the batch is
 * split into vcnf_linear_wgrad_slices(...) slices (exact fp32 matrix instructions), whose partial results go through
 * `workspace` (>= slices * (out * in + out) floats) and are added in a fixed order.  accumulate != 0 adds to dw / db.
 * in_features: multiple of 16 up to 128. */
int vcnf_linear_wgrad_supported(int32_t in_features, int32_t out_features);
int64_t vcnf_linear_wgrad_slices(int64_t batch, int32_t in_features, int32_t out_features);
int vcnf_linear_wgrad_f32(const float* x, const float* dy, float* dw, float* db, float* workspace,
                          int64_t workspace_floats, int64_t batch, int32_t in_features, int32_t out_features,
                          int accumulate, void* stream);
/* The same gradients with the partial products on the fp16 split-half matrix path (v_mfma_f32_32x32x16_f16, both
 * operands split in registers, fp32 accumulation; arithmetic of VCNF_PREC_F16X3): the exact-fp32 kernel is bound by
 * its matrix instructions, this one runs the 736-row last layer of config C3 in a third of the time.  Same workspace,
 * same deterministic slice order.  relu_input: the layer's input is relu(x) (applied on load).  Values beyond +-65504
 * are clamped and counted in sat_count (may be NULL). */
int vcnf_linear_wgrad_f16x3_f32(const float* x, const float* dy, float* dw, float* db, float* workspace,
                                int64_t workspace_floats, int64_t batch, int32_t in_features,
                                int32_t out_features, int accumulate, int relu_input, int32_t* sat_count,
                                void* stream);

/* First two layers of the Glow conditioner in one launch (nets/cnn.py:20-52): y = act2(W2 act1(conv3x3(x; W1, padding 1)
 * + b1) + b2) with x [batch, c_in, height, width], 256 hidden and 256 output channels, act = LeakyReLU(slope); the
 * hidden activation between the two layers stays on chip.  Matrix path and saturation counter as
 * vcnf_conv1x1_f16x3_f32.  w1pack: the [256, 9 c_in] patch matrix (k = ci * 9 + ky * 3 + kx, zero padded to a multiple of
 * 16) as A fragments, vcnf_conv3x3_1x1_pack_floats(c_in) floats; w2pack: vcnf_conv1x1_pack_floats(256, 256) floats
 * (vcnf_amd/nets/cnn.py::pack_conv1x1).  c_in <= 24. */
int vcnf_conv3x3_1x1_supported(int32_t c_in, int32_t hidden, int32_t c_out);
int64_t vcnf_conv3x3_1x1_pack_floats(int32_t c_in);
int vcnf_conv3x3_1x1_f16x3_f32(const float* x, float* y, const float* w1pack, int64_t w1pack_floats,
                               const float* w2pack, int64_t w2pack_floats, const float* b1, const float* b2,
                               int64_t batch, int32_t c_in, int32_t height, int32_t width, float slope1, float slope2,
                               int32_t* sat_count, void* stream);

/* The whole Glow conditioner Conv3x3(c_in -> 256), LeakyReLU, Conv1x1(256 -> 256), LeakyReLU, Conv3x3(256 -> c_out)
 * (nets/cnn.py:20-52, flows/affine/glow.py:37-47) in two launches, neither 256-channel activation ever in memory:
 *   vcnf_convnet3_taps_f16x3_f32: the first two layers as vcnf_conv3x3_1x1_f16x3_f32, then the last layer's nine taps as
 *       1x1 convolutions  z[b, t * c_out + o, p] = sum_c W3[o, c, tap t] y[b, c, p],  t = ky * 3 + kx  (no shift);
 *       w3pack: the tap matrix [9 c_out, 256] as A fragments, vcnf_convnet3_w3_pack_floats(c_out) floats
 *       (vcnf_amd/nets/cnn.py::pack_conv1x1 with ceil(9 c_out / 32) row blocks);
 *   vcnf_col2im3x3_f32: out[b, o, py, px] = bias[o] + sum over the taps inside the image of
 *       z[b, t * channels + o, py + dy - 1, px + dx - 1]  (the convolution's zero padding).
 * c_in <= 24, c_out <= 56.  Matrix path and saturation counter as vcnf_conv1x1_f16x3_f32. */
int vcnf_convnet3_supported(int32_t c_in, int32_t hidden, int32_t c_out);
int64_t vcnf_convnet3_w3_pack_floats(int32_t c_out);
int vcnf_convnet3_taps_f16x3_f32(const float* x, float* z, const float* w1pack, int64_t w1pack_floats,
                                 const float* w2pack, int64_t w2pack_floats, const float* w3pack, int64_t w3pack_floats,
                                 const float* b1, const float* b2, int64_t batch, int32_t c_in, int32_t c_out,
                                 int32_t height, int32_t width, float slope1, float slope2, int32_t* sat_count,
                                 void* stream);
int vcnf_col2im3x3_f32(const float* z, const float* bias, float* out, int64_t batch, int32_t channels,
                       int32_t height, int32_t width, void* stream);

/* Fused elementwise maps of a residual block of the conditioner and of its backward (training path;
 * nets/resnet.py:38-57 under autograd), n contiguous floats each:
 *   op 0: out0 = a + b * sigmoid(c)                          (block output: input + second Linear gated by the context)
 *   op 1: out0 = a * s, out1 = a * b * s * (1 - s), s = sigmoid(c)     (gradients of the gated product)
 *   op 2: out0 = a where b > 0, else 0                       (ReLU backward, b = the ReLU's output; c unused)
 *   op 3: out0 = c + (a where b > 0, else 0)                 (ReLU backward joined with the skip connection's gradient) */
int vcnf_resblock_elementwise_f32(int op, const float* a, const float* b, const float* c, float* out0, float* out1,
                                  int64_t n, void* stream);

/* Identity half of one RQS coupling layer in one launch (coupling.py:76-116): for f < d_id and v = x[b, identity_idx[f]]
 *   y = S_f(v) (inverse = 0) or S_f^-1(v) (inverse = 1) with the batch-shared unconditional spline of feature f
 *       (PiecewiseRationalQuadraticCDF, coupling.py:165-246), or y = v when shared_w/h/d are all NULL;
 *   out[b, identity_idx[f]] = y;   cond_in[b, f] = cond_sees_output ? y : v   (cond_in may be NULL): the conditioner
 *       sees the raw identity features in the density direction (:78-80) and S^-1 of them when sampling (:110-114);
 *   partial[c][b] = sum of log|dy/dv| over features 64 c .. 64 c + 63, c < vcnf_rqs_identity_half_partial_rows(d_id)
 *       (written only with shared logits; the caller adds the rows - deterministic, no atomics).
 * Bin counts 4, 8, 10, 16, 32; tails none / linear (vcnf_rqs_identity_half_supported).  Results are bitwise those of
 * vcnf_rqs_shared_f32 on the gathered columns. */
int vcnf_rqs_identity_half_supported(int32_t num_bins, int32_t tails);
int64_t vcnf_rqs_identity_half_partial_rows(int32_t d_id);
int vcnf_rqs_identity_half_f32(const float* x, float* y, float* cond_in, float* partial, int64_t batch,
                               int32_t features, const int32_t* identity_idx, int32_t d_id,
                               const float* shared_w, const float* shared_h, const float* shared_d,
                               const vcnf_rqs_cfg* cfg, int inverse, int cond_sees_output,
                               int32_t* bad_disc, void* stream);

/* Packed conditioner weights of one fused RQS coupling layer: ONE device buffer of
 * vcnf_rqs_layer_fused_pack_floats(d_id, d_t, ctx_dim, num_blocks) floats (0: shape not supported) holding the ResidualNet's nn.Linear
 * weights (nets/resnet.py:78-90) re-ordered into matrix-core fragments, in the order
 *   W0 | b0 | per block: WA | ba | WB | bb | (WC | bc if ctx_dim > 0) | WF | bf
 * (exact fragment order: vcnf_amd/fused.py::pack_layer, csrc/fused_common.hpp PackLayout / PackLayout6). */
int64_t vcnf_rqs_layer_fused_pack_floats(int32_t d_id, int32_t d_t, int32_t ctx_dim, int32_t num_blocks);

/* 1 if vcnf_rqs_layer_fused_f32 has a kernel for this layer shape, else 0. */
int vcnf_rqs_layer_fused_supported(int32_t d_id, int32_t d_t, int32_t ctx_dim, int32_t hidden,
                                   int32_t num_blocks, int32_t num_bins, int32_t tails);

/* One RQS coupling layer INCLUDING its ResidualNet conditioner in a single kernel:
 * Coupling.forward / .inverse (flows/neural_spline/coupling.py:70-96 / :98-125) with
 * transform_net = ResidualNet(ReLU, no batch norm, dropout 0; nets/resnet.py:92-106)
 * evaluated on the fp32 matrix cores.  Same results contract as
 * vcnf_rqs_conditioner_input_f32 + the dense layers + vcnf_rqs_coupling_f32; the
 * conditioner output never touches HBM.  context[B, ctx_dim] may be NULL when
 * ctx_dim = 0.  x, y, context must be 16-byte aligned.
 * precision: VCNF_PREC_F32 - every dense layer on v_mfma_f32_16x16x4_f32 (exact fp32
 * fma chains); VCNF_PREC_F16X3 - every dense layer on v_mfma_f32_32x32x16_f16 with both
 * operands split into hi + lo*2^-11 fp16 halves (3 instructions per product, 4 in the
 * 16- and 48-deep layers, fp32 accumulation; GEMM error at or below the F32 path's on
 * every layer shape, tests/test_gpu_gemm_error.py).  The halves cannot carry a value
 * beyond +-65504 or a non-finite input.  redo_tiles (device int32, one entry per
 * vcnf_rqs_layer_fused_tile_rows() = 32 consecutive samples, ceil(batch / 32) entries, may
 * be NULL) makes the pair of calls below range-safe WITHOUT a host round trip:
 *   F16X3 call: redo_tiles is OUTPUT - 1 for the rows of a tile that held such a value
 *     (the kernel's tile is 32 samples for small batches, 128 = four entries otherwise),
 *     and then no y row and no logdet entry of that tile is written; 0 otherwise.
 *   F32 call with the same arguments (and the F32 packing of the same weights):
 *     redo_tiles is INPUT - only the flagged rows are evaluated and written.
 * After both, every sample has fp32-range results (the reference is plain fp32,
 * nets/resnet.py:92-106).  With redo_tiles == NULL the F16X3 call clamps at +-65504 and
 * stores.  sat_count (device int32, may be NULL) is incremented once per out-of-range tile
 * in either case.  wpack must have been packed for the same precision; the F16X3 buffer
 * has the 1/sqrt(hidden) logit scale (coupling.py:314-316) and the log2(e) factors of the
 * softmax / softplus / sigmoid exponentials folded in (vcnf_amd/fused.py::pack_layer_h3),
 * cfg->wh_scale must be the folded value. */
int32_t vcnf_rqs_layer_fused_tile_rows(void);
/* F16X3 path: batches of up to `rows` samples run on 32-sample tiles (one workgroup per 32
 * samples: the batch sizes of the reference's drivers, /root/reference/run.py:45-47, fill the
 * chip), larger ones on 128-sample tiles.  Same results contract; y is bitwise the same on
 * both, logdet differs by the order of its per-sample sum.  Process-wide; rows < 0 only
 * queries.  Returns the previous value (default 16384). */
int64_t vcnf_rqs_layer_fused_small_batch_rows(int64_t rows);
int vcnf_rqs_layer_fused_f32(const float* x, const float* context, float* y, float* logdet,
                             int64_t batch, const int32_t* transform_idx, int32_t d_t,
                             const int32_t* identity_idx, int32_t d_id, int32_t ctx_dim,
                             int32_t hidden, int32_t num_blocks, int32_t precision,
                             const float* wpack, int64_t wpack_floats,
                             const float* shared_w, const float* shared_h, const float* shared_d,
                             const vcnf_rqs_cfg* cfg, int inverse,
                             int ld_mode, float ld_sign, int32_t* bad_disc, int32_t* sat_count,
                             int32_t* redo_tiles, void* stream);

/* A run of MaskedAffineFlow layers INCLUDING their MLP conditioners s, t = Linear(D, H) - LeakyReLU - Linear(H, D), and
 * of the per-feature affine layers (AffineConstFlow / ActNorm) between them, in one launch: the loop body of
 * NormalizingFlow.log_prob / sample (core.py:144-183) over the models of the reference's own drivers
 * (/root/reference/run.py:58-68: K x [MaskedAffineFlow(b, t, s), ActNorm], D = 2 .. 15, 1024 - 2048 samples, fp64).
 * Layer arithmetic: flows/affine/coupling.py:171-222 (non-finite s / t -> NaN), :22-61, nets/mlp.py:30-58.
 * z, out [batch, features], features <= 16, hidden <= 64; logdet receives / accumulates ld_sign * the summed log|det|
 * of the run (inverse: every layer inverted, in the order given).  table: DEVICE array of 12 int64 per layer in
 * application order - [0] kind (0 masked affine, 1 per-feature), [1] hidden width, [2] LeakyReLU slope (bits of a
 * double), [3] b [features]; [4..7] s: W1 [H, D], b1 [H], W2 [D, H], b2 [D] (0: no s); [8..11] t likewise;
 * per-feature kind: [4] s [features] or 0, [5] t [features] or 0.  All pointers of the scalar type of the call. */
int vcnf_masked_affine_stack_supported(int32_t features, int32_t hidden);
int vcnf_masked_affine_stack_f32(const float* z, float* out, float* logdet, const int64_t* table,
                                 int64_t batch, int32_t features, int32_t n_layers, int inverse,
                                 int ld_mode, float ld_sign, void* stream);
int vcnf_masked_affine_stack_f64(const double* z, double* out, double* logdet, const int64_t* table,
                                 int64_t batch, int32_t features, int32_t n_layers, int inverse,
                                 int ld_mode, double ld_sign, void* stream);

/* Vector-Jacobian product of vcnf_masked_affine_stack_* (training through these layers: core.py:30-141 over
 * coupling.py:171-222 / :22-61 / mlp.py:30-58).  z_out = the forward call's output (nothing else is kept: the
 * kernel walks the layers backwards and rebuilds every layer's input from its output), g_out / g_logdet = upstream
 * gradients of the output and of the run's summed log|det| (g_logdet may be NULL), g_in = gradient of the run's
 * input.  Parameter gradients are ADDED (hardware floating-point atomics, batch order not fixed) into the flat
 * buffer `grads`, zeroed by the caller; grad_offsets[l] = element offset of layer l's block - masked affine:
 * [W1s | b1s | W2s | b2s | W1t | b1t | W2t | b2t] (absent conditioners take no room), per-feature: [s | t]. */
int vcnf_masked_affine_stack_bwd_f32(const float* z_out, const float* g_out, const float* g_logdet, float* g_in,
                                     float* grads, const int64_t* table, const int64_t* grad_offsets,
                                     int64_t batch, int32_t features, int32_t n_layers, int inverse, void* stream);
int vcnf_masked_affine_stack_bwd_f64(const double* z_out, const double* g_out, const double* g_logdet, double* g_in,
                                     double* grads, const int64_t* table, const int64_t* grad_offsets,
                                     int64_t batch, int32_t features, int32_t n_layers, int inverse, void* stream);

/* Dense layer at training batch sizes on the fp16 split-half matrix path (csrc/linear_f16x3.hip):
 *   y[b, n] = sum_k x[b, k] * A[n, k] (+ bias[n]),  A[n, k] = w[n * ldn + k * ldk],  x [batch, k], y [batch, n]
 * = nn.Linear's forward (w = weight [n, k]: ldn = k, ldk = 1; nets/resnet.py:42-57, 92-106) and its input gradient
 * g_x = g_y W (w = weight [k, n]: ldn = 1, ldk = n), i.e. what autograd computes for the conditioner's dense layers
 * in NormalizingFlow.forward_kld (core.py:30-65).  The weights are read in their natural fp32 layout and split in
 * registers; arithmetic as VCNF_PREC_F16X3 of vcnf_rqs_layer_fused_f32 (lo*lo kept for k <= 48).  k % 16 == 0,
 * n % 4 == 0, and n <= 128 when k > 128.  x, y 16-byte aligned.  relu_input / relu_output: ReLU applied to x while it
 * is read / to y before it is stored (the residual block's activations, nets/resnet.py:42, :46, without their own
 * passes over memory); mask / addend [batch, n] (16-byte aligned, may be NULL): y = addend + y * (mask > 0) - a ReLU's
 * backward pass and the skip connection's gradient applied to an input gradient as it is stored (autograd of
 * nets/resnet.py:42-57).  Values beyond +-65504 (or NaN inputs) are clamped and counted in sat_count (device int32,
 * may be NULL) once per 64-sample tile. */
int vcnf_linear_f16x3_supported(int32_t k, int32_t n);
int vcnf_linear_f16x3_f32(const float* x, const float* w, const float* bias, float* y, int64_t batch,
                          int32_t k, int32_t n, int64_t ldn, int64_t ldk, int relu_input, int relu_output,
                          const float* mask, const float* addend, int32_t* sat_count, void* stream);

/* A run of n_layers (<= vcnf_rqs_stack_fused_max_layers() = 16) RQS coupling layers of ONE shape and one spline
 * configuration in a single launch: the body of NormalizingFlow.log_prob / sample over consecutive
 * CoupledRationalQuadraticSpline layers (core.py:144-183 around coupling.py:70-125).  `layers` is a HOST array in
 * the order in which the layers are applied to x (its entries are copied into the launch); every entry carries the
 * layer's index vectors, its packed conditioner weights for `precision` and the unconditional spline's logits
 * (all layers with them, or none).  y = the last layer's output; logdet receives / accumulates ld_sign * the SUM of
 * the layers' log|det| (one rounding of the sum instead of one per layer).  Everything else as
 * vcnf_rqs_layer_fused_f32.  F16X3: every layer runs on 32-sample tiles that stay in LDS from the first layer to
 * the last (meant for the small batches at which one launch per layer is launch-bound: the reference's drivers use
 * 1024 - 2048 samples, /root/reference/run.py:45-47); a tile in which ANY layer met a value the fp16 halves cannot
 * carry is left unwritten and flagged in redo_tiles, and the same call with VCNF_PREC_F32 and the F32 packings
 * evaluates exactly the flagged rows through all layers on exact fp32 matrix instructions. */
typedef struct vcnf_rqs_stack_layer {
  const int32_t* transform_idx;
  const int32_t* identity_idx;
  const float* wpack;
  const float* shared_w;
  const float* shared_h;
  const float* shared_d;
} vcnf_rqs_stack_layer;
int32_t vcnf_rqs_stack_fused_max_layers(void);
int vcnf_rqs_stack_fused_f32(const float* x, const float* context, float* y, float* logdet,
                             int64_t batch, const vcnf_rqs_stack_layer* layers, int32_t n_layers,
                             int32_t d_t, int32_t d_id, int32_t ctx_dim, int32_t hidden,
                             int32_t num_blocks, int32_t precision, int64_t wpack_floats,
                             const vcnf_rqs_cfg* cfg, int inverse, int ld_mode, float ld_sign,
                             int32_t* bad_disc, int32_t* sat_count, int32_t* redo_tiles, void* stream);

/* Affine coupling on z[B, C, inner] (inner = H*W, 1 for 2-D inputs).
 * Replaces AffineCoupling.forward / .inverse (flows/affine/coupling.py:113-142 /
 * :144-168) together with the channel Split / Merge around it
 * (flows/reshape.py:25-29, :50-55; AffineCouplingBlock :247-258): channels
 * [t_off, t_off+d_t) are transformed, all other channels are copied.
 * param[B, n_par*d_t, inner]: channel 2c = shift, 2c+1 = scale logit
 * (coupling.py:122-123); scale_map NONE: param[B, d_t, inner] is the shift. */
int vcnf_affine_coupling_f32(const float* z, const float* param, float* out, float* logdet,
                             int64_t batch, int32_t channels, int32_t inner,
                             int32_t t_off, int32_t d_t, int scale_map, int inverse,
                             int ld_mode, float ld_sign, void* stream);

/* A run of AffineCouplingBlocks with ONE conditioner shape (c_in, hidden, d_t as in
 * vcnf_affine_layer_fused_f32) and the column permutations between them in a single launch:
 * the reference's loop over [AffineCouplingBlock, Permute] pairs of NormalizingFlow.log_prob /
 * .sample (core.py:150-155, :176-183; flows/affine/coupling.py:225-258; flows/mixing.py:32-54)
 * for config C1 / C2 style stacks.  ``layers`` (HOST array, at most 16) is in EXECUTION order of the
 * requested direction; gather_before = row of ``gathers`` [n_gather_rows][features] (device int32) whose
 * permutation the layer's input columns go through (the layer sees z[:, row]), -1 for none;
 * gather_after likewise for the result.  wpack = the layers' vcnf_affine_layer_fused_pack_floats
 * buffers back to back, in the same order.  log|det| of all layers is added to / stored in logdet. */
typedef struct {
  int32_t cond_off, t_off, d_t;
  int32_t gather_before;
} vcnf_affine_stack_layer;
/* 1 if vcnf_affine_stack_fused_f32 takes this layer shape at this feature count (the layer kernel's limits plus
 * the stack kernel's two LDS strips per wave: features <= 128), else 0 - then run the layers one launch each. */
int vcnf_affine_stack_fused_supported(int32_t c_in, int32_t hidden, int32_t n_out, int32_t features);
int vcnf_affine_stack_fused_f32(const float* x, float* y, float* logdet, int64_t batch, int32_t features,
                                int32_t n_layers, const vcnf_affine_stack_layer* layers, int32_t gather_after,
                                int32_t c_in, int32_t hidden, float leaky_slope, int scale_map,
                                const float* wpack, int64_t wpack_floats,
                                const int32_t* gathers, int32_t n_gather_rows,
                                int inverse, int ld_mode, float ld_sign, void* stream);

/* The same run with the two hidden-deep dense layers of every conditioner MLP (nets/mlp.py:30-35: hidden -> hidden,
 * hidden -> parameters) on the fp16 split-half matrix path (v_mfma_f32_16x16x32_f16, hi + lo halves of both operands,
 * fp32 accumulation; GEMM error at or below the fp32 chain's at depth >= 32, tests/test_gpu_gemm_error.py); the first
 * layer (raw inputs, c_in deep) stays on exact fp32 matrix instructions.  wpack as for vcnf_affine_stack_fused_f32
 * (first layer, biases; its second / third layer weights serve the range fallback), wpack_h3 =
 * vcnf_affine_layer_fused_h3_pack_floats floats per layer (vcnf_amd/fused_affine.py::pack_h3).  A wave whose hidden
 * activations leave the fp16 range (|h| > 65504) evaluates that layer with the exact fp32 body instead - nothing is
 * clamped - and bumps redo_count (device int32, may be NULL).  hidden in {32, 64, 128}. */
int64_t vcnf_affine_layer_fused_h3_pack_floats(int32_t c_in, int32_t hidden, int32_t n_out);
int vcnf_affine_stack_fused_f16x3_f32(const float* x, float* y, float* logdet, int64_t batch, int32_t features,
                                      int32_t n_layers, const vcnf_affine_stack_layer* layers,
                                      int32_t gather_after, int32_t c_in, int32_t hidden, float leaky_slope,
                                      int scale_map, const float* wpack, int64_t wpack_floats,
                                      const float* wpack_h3, int64_t wpack_h3_floats,
                                      const int32_t* gathers, int32_t n_gather_rows,
                                      int inverse, int ld_mode, float ld_sign, int32_t* redo_count, void* stream);

/* Elementwise map of the masked affine autoregressive flow (flows/affine/autoregressive.py:75-103): params
 * [batch, features, 2] = (unconstrained scale u, shift) per feature from one MADE pass (:96-103),
 * scale = sigmoid(u + 2) + 1e-3.  inverse = 0: y = scale x + shift, logdet = sum log scale (:75-81, the flow's
 * one-pass direction); inverse != 0: y = (x - shift) / scale, logdet = -sum log scale (:83-89, applied D times by the
 * sequential loop :29-36).  params must be 8-byte aligned. */
int vcnf_maf_affine_f32(const float* x, const float* params, float* out, float* logdet, int64_t batch,
                        int32_t features, int inverse, int ld_mode, float ld_sign, void* stream);

/* MaskedAffineFlow.forward / .inverse (flows/affine/coupling.py:202-211 /
 * :213-222) on z[B,D]; s,t [B,D] are the scale / shift net outputs (NULL =
 * zeros, coupling.py:192-200); b[D] the 0/1 float mask.  Non-finite s/t become
 * NaN (coupling.py:205-208). */
int vcnf_masked_affine_f32(const float* z, const float* s, const float* t, const float* b,
                           float* out, float* logdet, int64_t batch, int32_t features,
                           int inverse, int ld_mode, float ld_sign, void* stream);

/* AffineConstFlow / ActNorm arithmetic (flows/affine/coupling.py:37-53) on
 * z[B, C, inner] with per-channel s[C], t[C] (either may be NULL = zeros);
 * the parameter-only log_det (a scalar) is left to the host. */
int vcnf_affine_const_f32(const float* z, const float* s, const float* t, float* out,
                          int64_t batch, int32_t channels, int32_t inner, int inverse, void* stream);

/* Column gather out[b, j, :] = z[b, idx[j], :] for z[B, C, inner].
 * Replaces Permute.forward / .inverse (flows/mixing.py:32-54) and
 * _Permutation._permute (:219-228). */
int vcnf_permute_f32(const float* z, const int32_t* idx, float* out,
                     int64_t batch, int32_t channels, int32_t inner, void* stream);

/* The channel partition of a coupling as two tensors, and back (training path: the gather
 * inputs[:, identity_features] / inputs[:, transform_features] and the scatter outputs[:, ...] = ... of
 * flows/neural_spline/coupling.py:86-88, :122-124 - each the other's VJP).  z / out [B, C] row-major;
 * first_part [B, first], second_part [B, C - first].
 * split:  first_part[b, c] = z[b, idx[c]] (c < first), second_part[b, c - first] = z[b, idx[c]] (c >= first).
 * merge:  out[b, c] = p < first ? first_part[b, p] : second_part[b, p - first] with p = idx[c]
 *         (idx of merge = the inverse permutation of the idx of split). */
int vcnf_split_columns_f32(const float* z, const int32_t* idx, float* first_part, float* second_part,
                           int64_t batch, int32_t channels, int32_t first, void* stream);
int vcnf_merge_columns_f32(const float* first_part, const float* second_part, const int32_t* idx, float* out,
                           int64_t batch, int32_t channels, int32_t first, void* stream);

/* DiagGaussian.log_prob (normflow/distributions/base.py:644-652).
 * loc, log_scale [D]; log_temperature = log(T) or 0; logp[B] per ld_mode. */
int vcnf_diag_gaussian_log_prob_f32(const float* z, const float* loc, const float* log_scale,
                                    float log_temperature, float* logp, int64_t batch,
                                    int32_t features, int ld_mode, float ld_sign, void* stream);

/* DiagGaussian.forward (base.py:632-642) with the standard-normal draw eps[B,D]
 * supplied by the caller: z = loc + exp(log_scale) * eps, logp[B]. */
int vcnf_diag_gaussian_sample_f32(const float* eps, const float* loc, const float* log_scale,
                                  float log_temperature, float* z, float* logp, int64_t batch,
                                  int32_t features, void* stream);

/* ---- fp64 variants.  The reference's Flow contract is dtype-agnostic and its drivers convert
 * the model with .double() (/root/reference/run.py:114, rundiag.py:90, runadultvdeq.py:183 over MaskedAffineFlow,
 * ActNorm and Gaussian bases).  Same contracts as the _f32 entry points of the same name with every float buffer and
 * scalar a double; elementwise HBM-bound maps only - the fused / matrix-core kernels stay fp32. */
typedef struct vcnf_rqs_cfg_f64 {
  int32_t num_bins;
  int32_t tails;
  double left, right, bottom, top;
  double min_bin_width, min_bin_height, min_derivative, wh_scale;
} vcnf_rqs_cfg_f64;
int vcnf_rqs_elementwise_f64(const double* x, const double* uw, const double* uh, const double* ud,
                             int64_t ld_w, int64_t ld_h, int64_t ld_d,
                             double* y, double* logabsdet, int64_t n,
                             const vcnf_rqs_cfg_f64* cfg, int inverse, int32_t* bad_disc, void* stream);
int vcnf_affine_coupling_f64(const double* z, const double* param, double* out, double* logdet,
                             int64_t batch, int32_t channels, int32_t inner,
                             int32_t t_off, int32_t d_t, int scale_map, int inverse,
                             int ld_mode, double ld_sign, void* stream);
int vcnf_masked_affine_f64(const double* z, const double* s, const double* t, const double* b,
                           double* out, double* logdet, int64_t batch, int32_t features,
                           int inverse, int ld_mode, double ld_sign, void* stream);
int vcnf_affine_const_f64(const double* z, const double* s, const double* t, double* out,
                          int64_t batch, int32_t channels, int32_t inner, int inverse, void* stream);
int vcnf_permute_f64(const double* z, const int32_t* idx, double* out,
                     int64_t batch, int32_t channels, int32_t inner, void* stream);
int vcnf_split_columns_f64(const double* z, const int32_t* idx, double* first_part, double* second_part,
                           int64_t batch, int32_t channels, int32_t first, void* stream);
int vcnf_merge_columns_f64(const double* first_part, const double* second_part, const int32_t* idx, double* out,
                           int64_t batch, int32_t channels, int32_t first, void* stream);
int vcnf_diag_gaussian_log_prob_f64(const double* z, const double* loc, const double* log_scale,
                                    double log_temperature, double* logp, int64_t batch,
                                    int32_t features, int ld_mode, double ld_sign, void* stream);
int vcnf_diag_gaussian_sample_f64(const double* eps, const double* loc, const double* log_scale,
                                  double log_temperature, double* z, double* logp, int64_t batch,
                                  int32_t features, void* stream);

/* Diagnostic, not on any product path: ONE dense layer y[B, N] = x[B, K] W[N, K]^T + b (nn.Linear,
 * nets/resnet.py:78-106) evaluated with the arithmetic of one of the fused RQS layer kernels' matrix paths, so that
 * the GEMM-level error of each path can be measured against an fp64 product (tests/test_gpu_gemm_error.py):
 * VCNF_PROBE_F32 - v_mfma_f32_16x16x4_f32 chain as in vcnf_rqs_layer_fused_f32(VCNF_PREC_F32);
 * VCNF_PROBE_F16X3 - split-half operands, hi*hi + (hi*lo + lo*hi) 2^-11 (hidden / last layers of VCNF_PREC_F16X3);
 * VCNF_PROBE_F16X3_LL - the same plus the lo*lo term (its first layer).  relu_input != 0 applies max(x, 0) to the
 * input first (the hidden layers see relu(h), resnet.py:42-46).  K % 16 == 0, N % 32 == 0.  sat_count as in
 * vcnf_rqs_layer_fused_f32 (may be NULL). */
enum { VCNF_PROBE_F32 = 0, VCNF_PROBE_F16X3 = 1, VCNF_PROBE_F16X3_LL = 2 };
int vcnf_linear_probe_f32(const float* x, const float* weight, const float* bias, float* y, int64_t batch,
                          int32_t in_features, int32_t out_features, int mode, int relu_input,
                          int32_t* sat_count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VCNF_HIP_H */
