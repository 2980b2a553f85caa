/*
 * miseg.h -- C ABI of libmiseg.so: the MI355X (gfx950) device kernels behind the
 * segmentation-network forward pass that the reference builds at
 * utils/model.py:145-236 and runs at utils/predict.py:159.
 *
 * The reference has no FFI of its own: every arithmetic step of this path is a
 * tf.keras layer call.  Each entry point below therefore cites the Keras layer
 * family (and the reference line that instantiates it) it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, a negative MISEG_E_* code otherwise;
 *     miseg_last_error() returns a thread-local message for the last failure;
 *   - every pointer is a DEVICE pointer owned by the caller; nothing here
 *     allocates or frees device memory;
 *   - `stream` is a hipStream_t passed as void*;
 *   - activations are NHWC; "strides" are in ELEMENTS of the tensor's dtype and
 *     let a tensor be a channel slice of a wider buffer (concatenation without
 *     copies) or a strided/offset view (transpose-conv phases);
 *   - bf16 = upper 16 bits of an IEEE float32, round-to-nearest-even.
 */
#ifndef MISEG_H
#define MISEG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MISEG_OK 0
#define MISEG_E_INVALID_ARG (-1)
#define MISEG_E_UNSUPPORTED_SHAPE (-2)
#define MISEG_E_HIP (-3)
#define MISEG_E_NO_DEVICE (-4)

#define MISEG_DT_BF16 0 /* the call's 16-bit storage format: bf16, or fp16 when fmt = MISEG_FMT_F16 */
#define MISEG_DT_F32 1

/*
 * 16-bit storage format of activations and convolution weights.  The training path and the
 * default of every entry point is bf16.  The inference kernels exist a second time for IEEE
 * fp16 (same kernels, same matrix-instruction rate, same bytes; values saturate at +-65504):
 * 11 significand bits instead of 8, which is what it takes for the strict-argmax masks of
 * utils/utils.py:50-68 to agree with an fp32 run of model(batch, training=False)
 * (utils/predict.py:159) on >= 99.9 % of the pixels.  fp32 accumulation in both.
 */
#define MISEG_FMT_BF16 0
#define MISEG_FMT_F16 1

/* epilogue of the main output */
#define MISEG_ACT_NONE 0
#define MISEG_ACT_RELU 1
#define MISEG_ACT_SOFTMAX 2 /* over Cout (<= 4), fp32 output, optional u8 collapse */
#define MISEG_ACT_SIGMOID 3

#define MISEG_CONV_DGRAD_T 2 /* miseg_conv_desc.transposed: data gradient of the transpose convolution */
#define MISEG_TILE_NO_DMA 16
#define MISEG_TILE_NO_HALO 32
#define MISEG_TILE_SHARED_W 64  /* halo kernels: block-shared weight tiles instead of wave-private rings */
#define MISEG_RES_ADD 0
#define MISEG_RES_BN_RELU_MASK 1
#define MISEG_TILE_NO_PERSIST 128 /* Cin = 32 layers, the stem and the 64 -> 33..64 channel 3x3 layers: the kernel
                                     with one patch per block instead of the persistent one */

const char* miseg_last_error(void);
int miseg_version(void);
/* 0 if the current HIP device is a gfx950; MISEG_E_NO_DEVICE otherwise. */
int miseg_device_check(void);

/*
 * Convolution as implicit GEMM on bf16 MFMA with fp32 accumulation.
 * Replaces keras Conv2D / ZeroPadding2D+Conv2D / Conv2DTranspose(4,2,'same')
 * and the BatchNormalization+Activation('relu')+Add that surround them in
 * classification_models ResNet, keras_applications DenseNet and the
 * segmentation_models decoders (reached from utils/model.py:176,188,200).
 *
 *   acc[n,oy,ox,co] = sum_{dy,dx,ci} A[n, oy*stride_y - pad_y + dy,
 *                                        ox*stride_x - pad_x + dx, ci] * Wt[co,dy,dx,ci]
 *   A = in, or relu(in*in_scale[ci] + in_shift[ci]) when in_scale != NULL
 *       (taps outside the image contribute zero in both cases);
 *   v    = acc*scale[co] + shift[co] (+ residual[n,oy,ox,co]);
 *   out  = act(v)                           (dtype out_dtype; skipped if out == NULL;
 *                                            with res_after_act the residual is added to act(v) instead)
 *   out2 = relu(v*scale2[co] + shift2[co])  (bf16; skipped if out2 == NULL)
 *
 * transposed == 1 selects Conv2DTranspose(kernel 4, stride 2, 'same'): the call
 * computes the four output phases (a,b) in {0,1}^2 as 2x2 convolutions,
 * out[n, 2y+a, 2x+b, co]; `weight` then holds [4][Cout][2][2][Cin] and
 * KH = KW = 2, strides 1, Ho = H, Wo = W; the out, out2 and res strides describe the
 * full-resolution (2H x 2W) tensor.
 *
 * transposed == MISEG_CONV_DGRAD_T is the data gradient of that layer: `in` is the
 * gradient at its output (N, 2H, 2W, Cin; the in strides describe that tensor),
 * H, W, Ho, Wo are the dimensions of the result, KH = KW = 2, strides 1, and
 *   acc[n,i,j,co] = sum_{a,b,dy,dx,ci} in[n, 2(i+1-a-dy)+a, 2(j+1-b-dx)+b, ci]
 *                                      * Wt[(a,b)][co][dy][dx][ci]
 * with `weight` = [4][Cout][2][2][Cin] (the forward layer's phase weights with its
 * two channel axes exchanged).  Cin % 64 == 0; the epilogue is that of a plain convolution.
 *
 * weight layout: bf16 [Cout][KH][KW][Cin], Cin contiguous.  Cin % 32 == 0.
 */
typedef struct miseg_conv_desc {
  const void* in;
  const void* weight;
  void* out;
  void* out2;
  const void* residual;
  const float* scale;
  const float* shift;
  const float* scale2;
  const float* shift2;
  const float* in_scale;
  const float* in_shift;
  void* collapse_out; /* MISEG_ACT_SOFTMAX only: u8 [N,Ho,Wo,Cout] strict-argmax mask, or NULL */
  int32_t N, H, W, Cin;
  int32_t Ho, Wo, Cout;
  int32_t KH, KW, stride_y, stride_x, pad_y, pad_x;
  int32_t in_pix_stride, in_row_stride;
  int64_t in_img_stride;
  int32_t out_pix_stride, out_row_stride;
  int64_t out_img_stride;
  int32_t out2_pix_stride, out2_row_stride;
  int64_t out2_img_stride;
  int32_t res_pix_stride, res_row_stride;
  int64_t res_img_stride;
  int32_t out_dtype; /* MISEG_DT_* */
  int32_t res_dtype; /* MISEG_DT_* */
  int32_t act;       /* MISEG_ACT_* */
  int32_t transposed;
  float temperature;       /* MISEG_ACT_SOFTMAX: logits are divided by it */
  int32_t collapse_value;  /* value written for the winning channel (127) */
  int32_t tile_hint;       /* 0 = choose automatically; low 4 bits = tile config 1..15,
                              | MISEG_TILE_NO_DMA forces the register-staged path,
                              | MISEG_TILE_NO_HALO / _SHARED_W / _NO_PERSIST keep a layer off the
                              halo-resident / wave-private-ring / persistent kernels (tests compare
                              the variants with each other) */
  /* Optional per-channel statistics of the stored bf16 output for a following training-mode
   * BatchNormalization (fused into the epilogue instead of a second pass over the tensor):
   * double[stats_replicas + 1][2][stats_stride], zeroed by the caller; the kernel ADDS sum(v) to
   * [r][0][c] and sum(v*v) to [r][1][c], r = block index mod stats_replicas (the last slot is
   * spare).  Needs a bf16
   * `out`, Cout % 8 == 0 and 16-byte aligned output strides.  miseg_bn_finalize reduces it. */
  void* stats;
  int32_t stats_replicas;
  int32_t stats_stride;
  int32_t fmt;   /* MISEG_FMT_*: format of in, weight, out2 and of out / residual where their dtype is
                    MISEG_DT_BF16 (read: "16-bit"). */
  int32_t res_after_act; /* non-zero: out = act(v) + residual with v = acc*scale + shift, i.e. the
                            Add([relu(bn(conv)), skip]) of the LinkNet decoder blocks
                            (segmentation_models DecoderBlock, reached from utils/model.py:188-198),
                            instead of out = act(v + residual); needs residual, no out2 */
  int32_t res_upsample_shift; /* s > 0: the residual tensor has extent (Ho >> s, Wo >> s) and is read at
                                 [n, oy >> s, ox >> s, co]: UpSampling2D(2^s, 'nearest') + Add in the
                                 epilogue, the top-down path of the FPN blocks (segmentation_models
                                 FPNBlock, reached from utils/model.py:200-211) */
  int32_t out2_linear;        /* non-zero: the second output without its ReLU, out2 = v * scale2 + shift2 (a
                                 biased copy of the result: the engine's split form of FPN's segm_stage2a) */
  int32_t in_valid_c;         /* 0, or a multiple of 8 below Cin: only the first in_valid_c channels of an input
                                 pixel exist in memory (in_pix_stride may then be as small as in_valid_c), the rest
                                 of the Cin channels read as zero.  The training path's gradient at the logits is
                                 stored 8 channels wide, and the head convolution's data gradient reads it so.
                                 Taken by 3x3 stride-1 layers with Cin = 32, Cout <= 32 and 16-bit output without
                                 residual or second output; MISEG_E_UNSUPPORTED_SHAPE otherwise */
  int32_t res_mode;           /* MISEG_RES_ADD (0): the residual is added, as described above.
                                 MISEG_RES_BN_RELU_MASK: this call is the data gradient of a layer that reads
                                 a = relu(x * scale[c] + shift[c]) (training-mode BatchNormalization + ReLU, the
                                 pairs classification_models / segmentation_models put in front of their
                                 convolutions; backward of the model.fit step, model_train.py:284-329).  `residual`
                                 is x (16-bit, the BatchNormalization's input, at the output's positions), scale /
                                 shift are the BatchNormalization's forward coefficients and are NOT applied to the
                                 result.  For output channels c < res_mask_c:
                                     out = acc  where x * scale[c] + shift[c] > 0,  else 0
                                 (the gradient behind the ReLU), and `stats`, when given, receives sum(out) in
                                 [r][0][c] and sum(out * x) in [r][1][c] of the stored values: the two reductions of
                                 the BatchNormalization's backward pass (miseg_bn_bwd_sums finishes them), which
                                 then needs no pass of its own over x and the gradient.  Channels from res_mask_c on
                                 (a skip tensor concatenated behind a) are stored as they are.  Needs 16-bit out
                                 and residual, Cout % 8 == 0, res_mask_c % 8 == 0, 16-byte aligned strides, no
                                 activation, no second output */
  int32_t res_mask_c;
} miseg_conv_desc;

int miseg_conv2d_fwd(const miseg_conv_desc* d, void* stream);
/* Name of the device kernel miseg_conv2d_fwd launches for `d` (as a profiler prints it, without the
 * namespace), into buf: lets a caller attribute its own HIP-event timings to kernels. */
int miseg_conv_kernel_name(const miseg_conv_desc* d, char* buf, int32_t buflen);

/*
 * The tail of the U-Net the reference builds at utils/model.py:176-186, as one kernel:
 * `decoder_stage4b` (Conv2D 3x3 32 -> 32 + BatchNorm + ReLU, given as `body`: in, weight,
 * scale, shift, act; its out pointers are ignored) followed by `final_conv` + activation
 * (Conv2D 3x3 32 -> classes <= 4 with bias, softmax / sigmoid, given as `head`: weight,
 * scale, shift, act, temperature, fp32 out, optional collapse_out / collapse_value as in
 * utils/utils.py:50-68).  The 32-channel tensor between them stays on chip.  Same results as
 * the two miseg_conv2d_fwd calls up to the summation order inside the head's matrix
 * instruction.
 */
int miseg_conv_c32_head_fwd(const miseg_conv_desc* body, const miseg_conv_desc* head, void* stream);

/*
 * One layer of a DenseNet block as one kernel (keras_applications `conv_block`, the encoder the
 * reference selects at utils/model.py:188-211): <name>_0_bn + _0_relu applied to the first Cin
 * channels of the concatenation buffer, <name>_1_conv (1x1, Cin -> 128), _1_bn + _1_relu,
 * <name>_2_conv (3x3, 128 -> 32, zero padding 1), whose 32 output channels are appended to the
 * same buffer (_concat).  `c1` describes the 1x1 convolution as for miseg_conv2d_fwd: in and its
 * strides, Cin % 32 == 0 (at most 1792 with bf16 storage), weight [128][Cin], in_scale / in_shift
 * (the first BatchNorm), scale / shift (the second), act = MISEG_ACT_RELU; its out pointers are
 * ignored.  `c2` describes the 3x3 convolution: weight [32][3][3][128], 16-bit out with its
 * strides (usually the same buffer at channel offset Cin), optional scale / shift, no activation.
 * The 128-channel tensor between the two stays on chip.  Same results as the two
 * miseg_conv2d_fwd calls up to the rounding of the input BatchNorm (evaluated in the storage
 * format's own arithmetic when that is fp16).
 */
int miseg_dense_layer_fwd(const miseg_conv_desc* c1, const miseg_conv_desc* c2, void* stream);

/*
 * The tail of the LinkNet the reference builds at utils/model.py:188-198 as one kernel:
 * segmentation_models DecoderBlock `decoder_stage4` -- `a`: Conv2D 1x1 64 -> 16, `b`:
 * Conv2DTranspose(4, 2, 'same') 16 -> 16, `c`: Conv2D 1x1 16 -> 16, each with BatchNorm + ReLU
 * (folded into scale/shift vectors of 16 floats) -- followed by `final_conv` (Conv2D 3x3
 * 16 -> classes <= 4 with bias), softmax / sigmoid and the collapse_probabilities mask
 * (utils/utils.py:50-68).  The three 16-channel full-resolution tensors stay on chip.
 *   in : 16-bit [N][H][W][>= 64 channels] (strides in elements), the decoder's 64-channel
 *        tensor at half resolution;
 *   wa : [16][64]                      (cout, cin)
 *   wb : [4][16][2][2][16]             (phase a*2+b, cout, dy, dx, cin) with
 *        wb[(a,b)][co][dy][dx][ci] = kernel[3-a-2dy][3-b-2dx][co][ci] of the Keras layer
 *   wc : [16][32]                      (cout, cin; cin >= 16 zero)
 *   wd : [16][160]                     (class, tap*16 + cin; rows >= classes and k >= 144 zero)
 *   out: fp32 [N][2H][2W][classes] probabilities (strides in elements);
 *   collapse_out: u8 [N][2H][2W][classes] or NULL.
 */
typedef struct miseg_linknet_tail_desc {
  const void* in;
  const void *wa, *wb, *wc, *wd;
  const float *sa, *ta, *sb, *tb, *sc, *tc;
  const float* bias; /* [classes] or NULL */
  float* out;
  void* collapse_out;
  int32_t N, H, W;
  int32_t in_pix_stride, in_row_stride;
  int64_t in_img_stride;
  int32_t out_pix_stride, out_row_stride;
  int64_t out_img_stride;
  int32_t classes;
  int32_t act; /* MISEG_ACT_SOFTMAX or MISEG_ACT_SIGMOID */
  float temperature;
  int32_t collapse_value;
  int32_t fmt; /* MISEG_FMT_* of in and the weights */
} miseg_linknet_tail_desc;

int miseg_linknet_tail_fwd(const miseg_linknet_tail_desc* d, void* stream);

/*
 * A whole LinkNet decoder block whose skip tensor has 64 channels as one kernel (segmentation_models
 * DecoderBlock `decoder_stage3` of the networks the reference builds at utils/model.py:188-198): `a`: Conv2D
 * 1x1 cin -> m = cin / 4, `b`: Conv2DTranspose(4, 2, 'same') m -> m, `c`: Conv2D 1x1 m -> 64, each with
 * BatchNorm + ReLU (folded into scale/shift vectors), then Add([c, skip]); cin = 64 (ResNet encoders) or 128
 * (DenseNet encoders).  The two m-channel tensors stay on chip.  Same values as the three miseg_conv2d_fwd
 * calls (the last with res_after_act) up to the order of the fp32 sums.
 *   in  : 16-bit [N][H][W][>= cin channels] (strides in elements)
 *   wa  : [m][cin]; wb: [4][m][2][2][m] (phase, cout, dy, dx, cin) as in miseg_linknet_tail_desc;
 *   wc  : [64][32] (cout, cin; cin >= m zero)
 *   sa, ta, sb, tb: m floats each; sc, tc: 64 floats
 *   skip, out: 16-bit [N][2H][2W][>= 64 channels] (strides in elements, multiples of 8; 16-byte aligned)
 */
typedef struct miseg_linknet_block_desc {
  const void* in;
  const void *wa, *wb, *wc;
  const float *sa, *ta, *sb, *tb, *sc, *tc;
  const void* skip;
  void* out;
  int32_t N, H, W;
  int32_t in_pix_stride, in_row_stride;
  int64_t in_img_stride;
  int32_t skip_pix_stride, skip_row_stride;
  int64_t skip_img_stride;
  int32_t out_pix_stride, out_row_stride;
  int64_t out_img_stride;
  int32_t fmt; /* MISEG_FMT_* of in, the weights, skip and out */
  int32_t cin; /* 64 or 128 */
} miseg_linknet_block_desc;

int miseg_linknet_block_fwd(const miseg_linknet_block_desc* d, void* stream);

/*
 * The end of the FPN the reference builds at utils/model.py:200-211 as one kernel:
 * `final_upsampling` (UpSampling2D(2, interpolation='bilinear'): half-pixel centres, clamped
 * edges) + `head_conv` (Conv2D 3x3 128 -> classes <= 4 with bias, zero padding) + softmax /
 * sigmoid + the collapse_probabilities mask (utils/utils.py:50-68).  The up-sampled 128-channel
 * tensor stays on chip; its values are rounded to the storage format exactly as
 * miseg_upsample_bilinear2x_fwd would store them.
 *   in : 16-bit [N][H][W][>= 128 channels] (strides in elements);
 *   wd : [16][9][128]  (class, tap = dy*3 + dx, cin; rows >= classes zero)
 *   out: fp32 [N][2H][2W][classes]; collapse_out: u8 [N][2H][2W][classes] or NULL.
 */
typedef struct miseg_fpn_head_desc {
  const void* in;
  const void* wd;
  const float* bias; /* [classes] or NULL */
  float* out;
  void* collapse_out;
  int32_t N, H, W;
  int32_t in_pix_stride, in_row_stride;
  int64_t in_img_stride;
  int32_t out_pix_stride, out_row_stride;
  int64_t out_img_stride;
  int32_t classes;
  int32_t act; /* MISEG_ACT_SOFTMAX or MISEG_ACT_SIGMOID */
  float temperature;
  int32_t collapse_value;
  int32_t fmt;
} miseg_fpn_head_desc;

int miseg_fpn_head_fwd(const miseg_fpn_head_desc* d, void* stream);

/*
 * The same head with its channel contraction done at half resolution first.  UpSampling2D(2, 'bilinear') +
 * Conv2D(3x3, 128 -> classes) (segmentation_models build_fpn: final_upsampling, head_conv; reached from
 * /root/reference utils/model.py:200-211) is linear in its input, so the nine taps' contractions
 * y[tap * classes + c] = W[tap][:, c] . x can run as ONE 1x1 convolution with 9 * classes output channels
 * (miseg_conv2d_fwd, fp32 output) on the half-resolution tensor; this entry point finishes the layer: per
 * output pixel, for every tap position inside the image (the convolution's zero padding), the bilinear
 * sample (half-pixel centres, clamped) of that tap's values, their sum + bias, softmax (with temperature) or
 * sigmoid, and the collapse_probabilities mask (utils/utils.py:50-68).
 *   y: fp32 [N][h][w][y_pix_stride]; probs: fp32 [N][2h][2w][classes]; collapse_out: u8, same extents, or NULL.
 *   class_major != 0: y's channels are ordered c * 9 + tap instead, i.e. the 1x1 convolution used the rows of the
 *   head's [class][3][3][cin] weight tensor as they lie (what the training path does: no re-packing per step).
 */
int miseg_head_gather_fwd(const float* y, const float* bias, float* probs, void* collapse_out, int32_t N,
                          int32_t h, int32_t w, int32_t classes, int32_t y_pix_stride, int32_t act,
                          float temperature, int32_t collapse_value, int32_t class_major, void* stream);

/*
 * Backward of that step (the adjoint of sampling + tap summation), for training the same head without the
 * full-resolution 128-channel tensor: dz = d loss / d logits (bf16 [N][2h][2w][dz_pix_stride], first `classes`
 * channels) -> dy = d loss / d y (bf16 [N][h][w][dy_pix_stride], channel c * 9 + tap, zeros from 9 * classes on).
 * The 1x1 convolution's own data and weight gradients (miseg_conv2d_fwd with the transposed weights,
 * miseg_conv2d_wgrad) then run at half resolution.  Replaces the backward of UpSampling2D + Conv2D that
 * Keras derives inside model.fit (/root/reference utils/model.py:200-211, model_train.py:284-329).
 */
int miseg_head_gather_bwd(const void* dz, int32_t dz_pix_stride, void* dy, int32_t N, int32_t h, int32_t w,
                          int32_t classes, int32_t dy_pix_stride, void* stream);

/*
 * Input contract of the encoders: fp32 NHWC image with 3 channels ->
 * bf16 [N][H+6][W+8][4] with the image at rows 3.., columns 3.., channel 3 and
 * the border zero; value = x*scale[c] + shift[c] (ResNet `bn_data`, a
 * BatchNormalization(scale=False) on the raw input followed by
 * ZeroPadding2D(3); identity scale for DenseNet).  Feeds the 7x7 stride-2 stem.
 */
int miseg_prep_input(const float* in, void* out, const float* scale, const float* shift,
                     int32_t N, int32_t H, int32_t W, void* stream);
/* Training variant: the fourth, otherwise dead, channel holds 1 inside the image, so that the stem's
 * weight gradient for that channel is the per-tap sum of dY from which the beta gradient of the
 * input BatchNormalization (`bn_data`) follows without a data-gradient pass through the stem. */
int miseg_prep_input_indicator(const float* in, void* out, const float* scale, const float* shift,
                               int32_t N, int32_t H, int32_t W, void* stream);

/*
 * ZeroPadding2D(1) + MaxPooling2D(3, strides 2) on a non-negative bf16 tensor,
 * optionally followed by y = relu(y*scale[c] + shift[c]) (the first ResNet
 * unit's bn1+relu1).  C % 8 == 0.
 */
int miseg_maxpool3x3s2_fwd(const void* in, void* out, const float* scale, const float* shift,
                           int32_t N, int32_t H, int32_t W, int32_t C,
                           int32_t in_pix_stride, int32_t out_pix_stride, void* stream);

/* AveragePooling2D(2, strides 2) on bf16 (DenseNet transition `poolK_pool`). */
int miseg_avgpool2x2_fwd(const void* in, void* out, int32_t N, int32_t H, int32_t W, int32_t C,
                         int32_t in_pix_stride, int32_t out_pix_stride, void* stream);

/*
 * out[n,y,x,c] = in[n, y>>shift, x>>shift, c] (+ addend[n,y,x,c] if addend != NULL)
 * UpSampling2D nearest x2/x4/x8 (+ Add) of the FPN top-down path and the
 * `upsampling_stage{3,4,5}` layers writing into the aggregation buffer.
 * H, W are the INPUT dims.  bf16, C % 8 == 0.
 */
int miseg_upsample_nearest_fwd(const void* in, const void* addend, void* out,
                               int32_t N, int32_t H, int32_t W, int32_t C, int32_t shift,
                               int32_t in_pix_stride, int32_t add_pix_stride,
                               int32_t out_pix_stride, void* stream);

/* UpSampling2D(2, 'bilinear') with half-pixel centres (FPN `final_upsampling`). */
int miseg_upsample_bilinear2x_fwd(const void* in, void* out, int32_t N, int32_t H, int32_t W,
                                  int32_t C, int32_t in_pix_stride, int32_t out_pix_stride,
                                  void* stream);

/* y = relu(x*scale[c] + shift[c]) on bf16 (a stand-alone BatchNormalization+ReLU). */
int miseg_scale_shift_relu_fwd(const void* in, void* out, const float* scale, const float* shift,
                               int64_t pixels, int32_t C, int32_t in_pix_stride,
                               int32_t out_pix_stride, void* stream);

/*
 * The six layers above on fp16 storage (MISEG_FMT_F16); same arguments.  Inference only.
 */
int miseg_prep_input_f16(const float* in, void* out, const float* scale, const float* shift,
                         int32_t N, int32_t H, int32_t W, void* stream);
int miseg_maxpool3x3s2_fwd_f16(const void* in, void* out, const float* scale, const float* shift,
                               int32_t N, int32_t H, int32_t W, int32_t C,
                               int32_t in_pix_stride, int32_t out_pix_stride, void* stream);
int miseg_avgpool2x2_fwd_f16(const void* in, void* out, int32_t N, int32_t H, int32_t W, int32_t C,
                             int32_t in_pix_stride, int32_t out_pix_stride, void* stream);
int miseg_upsample_nearest_fwd_f16(const void* in, const void* addend, void* out,
                                   int32_t N, int32_t H, int32_t W, int32_t C, int32_t shift,
                                   int32_t in_pix_stride, int32_t add_pix_stride,
                                   int32_t out_pix_stride, void* stream);
int miseg_upsample_bilinear2x_fwd_f16(const void* in, void* out, int32_t N, int32_t H, int32_t W,
                                      int32_t C, int32_t in_pix_stride, int32_t out_pix_stride,
                                      void* stream);
int miseg_scale_shift_relu_fwd_f16(const void* in, void* out, const float* scale, const float* shift,
                                   int64_t pixels, int32_t C, int32_t in_pix_stride,
                                   int32_t out_pix_stride, void* stream);

/*
 * collapse_probabilities (utils/utils.py:50-68) on device: fp32 probabilities
 * [pixels][C] -> u8 [pixels][C]; channels are decided in order and channel i is
 * compared with the already-collapsed values of the channels before it, as the
 * reference's in-place loop does.
 */
int miseg_collapse_u8(const float* probs, uint8_t* out, int64_t pixels, int32_t C,
                      int32_t value, void* stream);

/*
 * Sums for cce_dice_loss / f1-score / iou_score (segmentation_models
 * base.functional, selected at model_train.py:172-179) over fp32 [pixels][C]
 * tensors, C <= 8.  sums (double[4*C], zeroed by the call) receives, per class,
 * sum(gt*pr), sum(pr), sum(gt), sum(gt*log(clip(pr/sum_c pr))).
 */
/*
 * Second half of miseg_bn_stats_fwd for statistics gathered by a convolution's epilogue
 * (miseg_conv_desc.stats): reduces the replicas, then mean / biased variance, scale / shift and
 * the moving-statistics update exactly as miseg_bn_stats_fwd does (tf.keras BatchNormalization
 * in training mode, model_train.py:284-329).
 */
int miseg_bn_finalize(const double* stats, int32_t replicas, int32_t stride, int64_t pixels, int32_t C,
                      const float* gamma, const float* beta, float* moving_mean, float* moving_var,
                      float momentum, float eps, float* mean, float* var, float* scale, float* shift,
                      void* stream);

int miseg_loss_sums(const float* gt, const float* pr, double* sums, int64_t pixels, int32_t C,
                    void* stream);

/* ------------------------------------------------------------------ training ---
 * The backward half of Model.fit (model_train.py:284-329): the kernels below, with
 * miseg_conv2d_fwd reused for the data gradients (a data gradient is a convolution of
 * the output gradient with repacked weights), replace the gradient ops TensorFlow
 * generates for the same Keras layers.
 */

/*
 * Weight gradient of Conv2D (and, per phase, of Conv2DTranspose):
 *   dw[co][ky][kx][ci] += sum_{n,oy,ox} dy[n,oy,ox,co] *
 *                         x[n, oy*stride_y - pad_y + ky, ox*stride_x - pad_x + kx, ci]
 * x, dy: bf16 NHWC views (strides in elements); dw: fp32 [Cout][KH][KW][Cin], accumulated
 * with float atomics (the caller zeroes it).  Cin % 8 == 0 and Cout % 8 == 0.
 */
typedef struct miseg_wgrad_desc {
  const void* x;
  const void* dy;
  float* dw;
  int32_t N, H, W, Cin;
  int32_t Ho, Wo, Cout;
  int32_t KH, KW, stride_y, stride_x, pad_y, pad_x;
  int32_t x_pix, x_row;
  int64_t x_img;
  int32_t dy_pix, dy_row;
  int64_t dy_img;
  int32_t tile_hint;   /* 0 = automatic; 1 = per-tap split-K kernel; 2 = halo-resident kernel
                          (stride 1, square filter up to 3x3, Cin % 32 == 0) */
  /* X prologue: when set, the layer's input is relu(x * x_scale[ci] + x_shift[ci]) -- the BatchNormalization +
     ReLU in front of the layer whose forward pass ran with miseg_conv_desc.in_scale / in_shift, so that the
     activated tensor was never stored.  Both or neither, 16-byte aligned fp32 [Cin]; built for 3x3 stride-1
     layers with Cin, Cout <= 32 (MISEG_E_UNSUPPORTED_SHAPE elsewhere: store the activated tensor then). */
  const float* x_scale;
  const float* x_shift;
} miseg_wgrad_desc;
int miseg_conv2d_wgrad(const miseg_wgrad_desc* d, void* stream);

/*
 * BatchNormalization in training mode, forward statistics.  x: bf16 [pixels][C] view.
 * Writes the batch mean and biased variance, y = x*scale + shift coefficients
 * (scale = gamma*rsqrt(var+eps), shift = beta - mean*scale; gamma == NULL means 1), and
 * updates the moving statistics in place: moving = momentum*moving + (1-momentum)*batch,
 * with the UNBIASED variance for moving_var (Keras' fused batch norm).
 * workspace: double[2*C], zeroed by the call.
 */
int miseg_bn_stats_fwd(const void* x, int64_t pixels, int32_t C, int32_t pix_stride,
                       const float* gamma, const float* beta, float* moving_mean,
                       float* moving_var, float momentum, float eps, float* mean, float* var,
                       float* scale, float* shift, double* workspace, void* stream);
/*
 * miseg_bn_stats_fwd without the workspace argument and without any allocation.  It has no
 * counterpart in the reference: it exists for callers that cannot allocate device memory (code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  The double[2*C]
 * partial sums live in a 64 KB array built into the library (one copy per GPU, held until the
 * process ends), so C <= 4096: larger C is rejected with MISEG_E_INVALID_ARG.  Every call on a GPU
 * shares that GPU's array: two calls that overlap on one GPU (different streams or threads) race;
 * calls on one stream are safe.  The first call on a device looks the array's address up
 * (miseg_device_check does it too: call that once per device before capturing a graph).
 */
int miseg_bn_stats_fwd_noalloc(const void* x, int64_t pixels, int32_t C, int32_t pix_stride,
                               const float* gamma, const float* beta, float* moving_mean,
                               float* moving_var, float momentum, float eps, float* mean, float* var,
                               float* scale, float* shift, void* stream);
/* The same for the fp32 3-channel network input (ResNet `bn_data`, scale=False). */
int miseg_input_stats_fwd(const float* x, int64_t pixels, const float* beta, float* moving_mean,
                          float* moving_var, float momentum, float eps, float* mean, float* var,
                          float* scale, float* shift, double* workspace, void* stream);

/*
 * Backward of y = relu(x*scale + shift) (BatchNormalization + ReLU, batch statistics).
 * Pass 1 (reduce): g = dy * [x*scale + shift > 0] (relu == 0: g = dy);
 *   sums[c] += sum g, sums[C + c] += sum g * xhat, xhat = (x - mean) * rsqrt(var + eps).
 *   dbeta = sums[0:C], dgamma = sums[C:2C].  sums: double[2*C], zeroed by the call.
 * Pass 2 (apply): dx = scale * (g - sums[c]/pixels - xhat * sums[C+c]/pixels)   (bf16),
 *   plus addend[p][c] when addend != NULL: the gradient the layer's input already received from
 *   its other consumers (a residual branch), so fan-out accumulation costs no extra pass.
 */
int miseg_bn_relu_bwd_reduce(const void* x, const void* dy, int64_t pixels, int32_t C,
                             int32_t x_pix_stride, int32_t dy_pix_stride, const float* scale,
                             const float* shift, const float* mean, const float* var, float eps,
                             int32_t relu, double* sums, void* stream);
int miseg_bn_relu_bwd_apply(const void* x, const void* dy, void* dx, int64_t pixels, int32_t C,
                            int32_t x_pix_stride, int32_t dy_pix_stride, int32_t dx_pix_stride,
                            const float* scale, const float* shift, const float* mean,
                            const float* var, float eps, int32_t relu, const double* sums,
                            const void* addend, int32_t addend_pix_stride, void* stream);

/* Pass 1 of the above from sums a data-gradient convolution gathered while it wrote g (miseg_conv_desc.res_mode =
 * MISEG_RES_BN_RELU_MASK; tf.keras computes both in FusedBatchNormGrad, reached from model.fit at
 * model_train.py:284-329): stats = double[replicas + 1][2][stride] with [r][0][c] = sum g, [r][1][c] = sum g * x.
 *   sums[c] = sum g, sums[C + c] = (sum g * x - mean[c] * sum g) * rsqrt(var[c] + eps), in fp64.
 * Pass 2 then runs with relu = 0 on the stored g. */
int miseg_bn_bwd_sums(const double* stats, int32_t replicas, int32_t stride, int32_t C, const float* mean,
                      const float* var, float eps, double* sums, void* stream);

/* dx of ZeroPadding2D(1)+MaxPooling2D(3,2): the gradient goes to the first maximum of each
 * window in row-major scan order.  x: pool input [N,H,W,C], dy: [N,H/2,W/2,C], bf16.  addend (may be NULL):
 * a gradient the pool's input already holds (same extents as x), added to the result. */
int miseg_maxpool3x3s2_bwd(const void* x, const void* dy, void* dx, int32_t N, int32_t H, int32_t W,
                           int32_t C, int32_t x_pix_stride, int32_t dy_pix_stride,
                           int32_t dx_pix_stride, const void* addend, int32_t addend_pix_stride, void* stream);
/* MaxPooling2D(2, strides=2) of the keras_applications VGG encoders (the block*_pool layers the reference
 * reaches through sm.Unet/Linknet/FPN(backbone_name="vgg16"), /root/reference utils/model.py:176-211).
 * in [N,H,W,C] -> out [N,H/2,W/2,C], 16-bit in storage format fmt; H, W even. */
int miseg_maxpool2x2_fwd(const void* in, void* out, int32_t N, int32_t H, int32_t W, int32_t C,
                         int32_t in_pix_stride, int32_t out_pix_stride, int32_t fmt, void* stream);
/* dx of miseg_maxpool2x2_fwd: the gradient goes to the first maximum of each window in row-major
 * order.  x: pool input [N,H,W,C], dy: [N,H/2,W/2,C], bf16. */
int miseg_maxpool2x2_bwd(const void* x, const void* dy, void* dx, int32_t N, int32_t H, int32_t W,
                         int32_t C, int32_t x_pix_stride, int32_t dy_pix_stride,
                         int32_t dx_pix_stride, void* stream);
/* 3x3 neighbourhoods of a 3-channel fp32 image as 32 channels per pixel (element (ky*3+kx)*3+c, zero
 * outside the image and above 26): VGG's block1_conv1 (3 -> 64, 3x3 'same') then runs as a 1x1
 * convolution.  taps = 9: as described; taps = 1: the centre tap only, i.e. the image zero-padded to 32
 * channels (the training path's form, whose weight gradient then has the layer's own layout).
 * in fp32 [N,H,W,3] -> out 16-bit [N,H,W,32] in storage format fmt. */
int miseg_im2col3x3_fwd(const float* in, void* out, int32_t N, int32_t H, int32_t W, int32_t taps,
                        int32_t fmt, void* stream);
/* Backward of a ReLU that follows a convolution directly (Conv2D(activation="relu"), the VGG layers):
 * dx = (y > 0 ? dy : 0) + addend (addend may be NULL); y is the ReLU's output.  bf16. */
int miseg_relu_bwd(const void* y, const void* dy, const void* addend, void* dx, int64_t pixels, int32_t C,
                   int32_t y_pix_stride, int32_t dy_pix_stride, int32_t addend_pix_stride,
                   int32_t dx_pix_stride, void* stream);
/* dx[n,y,x,c] = dy[n,y/2,x/2,c] / 4.  H, W are the pool INPUT dims. */
int miseg_avgpool2x2_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                         int32_t dy_pix_stride, int32_t dx_pix_stride, void* stream);
/* dx[n,y,x,c] = sum of dy over the 2^shift x 2^shift block it was copied to.  H, W: input dims. */
int miseg_upsample_nearest_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                               int32_t shift, int32_t dy_pix_stride, int32_t dx_pix_stride,
                               void* stream);
/* Transpose of the half-pixel bilinear x2 interpolation.  H, W: input dims. */
int miseg_upsample_bilinear2x_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W,
                                  int32_t C, int32_t dy_pix_stride, int32_t dx_pix_stride,
                                  void* stream);
/* out[p][0:C] = in[p][0:C] (+ addend[p][0:C]); bf16 channel-slice copy / accumulate
 * (Concatenate forward and backward, gradient accumulation at fan-out). */
int miseg_copy_channels(const void* in, const void* addend, void* out, int64_t pixels, int32_t C,
                        int32_t in_pix_stride, int32_t add_pix_stride, int32_t out_pix_stride,
                        void* stream);
/* per-channel sum of a bf16 [pixels][C] view into double sums[C] (bias gradients); zeroed here. */
int miseg_channel_sums(const void* x, int64_t pixels, int32_t C, int32_t pix_stride, double* sums,
                       void* stream);

/*
 * Gradient of loss_scale * cce_dice_loss (segmentation_models cce_dice_loss, selected at
 * model_train.py:173) with respect to the LOGITS of the softmax head, per pixel:
 *   g_c  = d loss / d pr_c  (cross-entropy term masked where clip(pr) saturates; Dice term
 *          from the per-class sums that miseg_loss_sums produced over the same tensors)
 *   dz_c = pr_c * (g_c - sum_j g_j pr_j) / temperature
 * gt, pr: fp32 [pixels][C]; dlogits: bf16 [pixels][out_stride]; when
 * out_stride is a multiple of 8 the channels >= C are written as zeros (the zero padding the
 * convolutions that consume it expect), otherwise they are left untouched.
 * use_cce / use_dice select the terms.
 */
int miseg_loss_grad(const float* gt, const float* pr, const double* sums, void* dlogits,
                    int64_t pixels, int32_t C, int32_t out_stride, float loss_scale,
                    float temperature, int32_t use_cce, int32_t use_dice, void* stream);

/*
 * The same for every loss the reference can select (model_train.py:172-177, utils/model.py:92-95):
 * flags = 1 (categorical_crossentropy) | 2 (dice_loss) | 4 (focal_loss: -gt alpha (1-pr)^gamma
 * log(clip(pr)), mean over all elements; segmentation_models' defaults are alpha 0.25, gamma 2).
 * C == 1 is the sigmoid head (classes == 1, utils/model.py:180): dz = g pr (1 - pr).
 */
#define MISEG_LOSS_CCE 1
#define MISEG_LOSS_DICE 2
#define MISEG_LOSS_FOCAL 4
int miseg_loss_grad_ex(const float* gt, const float* pr, const double* sums, void* dlogits,
                       int64_t pixels, int32_t C, int32_t out_stride, float loss_scale,
                       float temperature, int32_t flags, float alpha, float gamma, void* stream);

/* miseg_loss_sums plus a fifth row, sum(gt * alpha * (1 - q)^gamma * log q), q = clip(pr, 1e-7,
 * 1 - 1e-7): sums is double[5*C]. */
int miseg_loss_sums_focal(const float* gt, const float* pr, double* sums, int64_t pixels, int32_t C,
                          float alpha, float gamma, void* stream);

/*
 * Keras-form Adam on flat fp32 buffers:  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
 * p -= lr_t * m / (sqrt(v) + eps),  lr_t = lr*sqrt(1-b2^t)/(1-b1^t) (computed by the caller);
 * grad_scale multiplies g first.  Also writes the bf16 working copy of p when p_bf16 != NULL.
 */
int miseg_adam_step(float* p, float* m, float* v, const float* g, void* p_bf16, int64_t count,
                    float lr_t, float beta1, float beta2, float eps, float grad_scale, void* stream);

/* ------------------------------------------------------- the steps either side of the network ---
 * Input contract and post-processing of utils/data.py and utils/predict.py on the device.  Images
 * are NHWC with interleaved channels; `dtype` 0 = uint8, 1 = float32.
 */

/* Per-image minimum and maximum over all pixels and channels (normalize_image, utils/data.py:75-76).
 * minmax: float[2*N] = (min_0, max_0, min_1, ...). */
int miseg_image_minmax(const void* img, int32_t dtype, int32_t N, int64_t elems_per_image,
                       float* minmax, void* stream);
/* out = (img - min_n) / (max_n - min_n) as float32 (utils/data.py:78). */
int miseg_image_normalize(const void* img, int32_t dtype, int32_t N, int64_t elems_per_image,
                          const float* minmax, float* out, void* stream);
/* tf.image.resize(image, (Ho, Wo), method="bilinear") (utils/data.py:138): half-pixel centres, no
 * antialiasing; uint8 or float32 in, float32 out. */
int miseg_resize_bilinear(const void* in, int32_t dtype, float* out, int32_t N, int32_t H, int32_t W,
                          int32_t C, int32_t Ho, int32_t Wo, void* stream);
/* Nearest-neighbour resize of uint8 images.  mode 0: tf.image.resize(method="nearest")
 * (utils/data.py:136,164; source = floor((dst + 0.5) * scale)); mode 1: cv2.resize(...,
 * INTER_NEAREST) (utils/predict.py:166; source = floor(dst * scale)). */
int miseg_resize_nearest_u8(const uint8_t* in, uint8_t* out, int32_t N, int32_t H, int32_t W, int32_t C,
                            int32_t Ho, int32_t Wo, int32_t mode, void* stream);
/* tf.one_hot(mask, depth=C, axis=2) as float32 (utils/data.py:51-54): ids uint8 [pixels] ->
 * float32 [pixels][C]; an id >= C gives an all-zero pixel. */
int miseg_one_hot(const uint8_t* ids, float* out, int64_t pixels, int32_t C, void* stream);
/*
 * augment_dataset (utils/data.py:240-295) with the random draws made by the caller: per image
 * params[8] = {flip_left_right, flip_up_down, brightness delta, contrast factor, hue delta,
 * saturation factor, contrast on, hue/saturation on}.  img: float32 RGB [N,H,W,3]; mask (optional):
 * float32 [N,H,W,MC], flipped with the image.  Order: flips, brightness (x + delta), contrast
 * ((x - mean_c) * f + mean_c, mean per image and channel), hue, saturation (through HSV).
 * workspace: double[3*N].  Outputs must not alias inputs.
 */
int miseg_augment(const float* img, const float* mask, float* img_out, float* mask_out, int32_t N,
                  int32_t H, int32_t W, int32_t MC, const float* params, double* workspace,
                  void* stream);

/*
 * Gradient all-reduce of data-parallel training: what tf.distribute.MultiWorkerMirroredStrategy does
 * for the reference at model_train.py:272-301 (one replica per GPU, gradients summed over replicas
 * every step), on RCCL over xGMI, one process per GPU.  RCCL is loaded at run time on the first of
 * these calls.  Rank 0 obtains a 128-byte id with miseg_comm_unique_id and hands it to the other
 * ranks by any means; every rank then calls miseg_comm_init with the device it will use current.
 * miseg_allreduce_async sums `count` floats in place over all ranks, asynchronously on `stream`
 * (launch it on a communication stream as soon as a bucket of the flat gradient buffer is final, so
 * that it overlaps the rest of the backward pass); all ranks must issue the same sequence of calls.
 */
typedef struct miseg_comm miseg_comm;
int miseg_comm_unique_id(void* id128);
int miseg_comm_init(miseg_comm** comm, int32_t rank, int32_t world, const void* id128);
int miseg_allreduce_async(miseg_comm* comm, float* buf, int64_t count, void* stream);
int miseg_comm_destroy(miseg_comm* comm);

/*
 * A plan is a recorded list of the calls above with fixed arguments, replayed
 * with one call per forward pass (the analogue of the compiled tf.function
 * behind model(batch, training=False), utils/predict.py:159).
 */
typedef struct miseg_plan miseg_plan;
miseg_plan* miseg_plan_create(void);
void miseg_plan_destroy(miseg_plan* p);
int miseg_plan_num_ops(const miseg_plan* p);
/* Storage format (MISEG_FMT_*) of every op of the plan; call before the first miseg_plan_add_*.
 * Convolution descriptors added afterwards must carry the same `fmt`. */
int miseg_plan_set_format(miseg_plan* p, int32_t fmt);
int miseg_plan_add_conv(miseg_plan* p, const miseg_conv_desc* d);
int miseg_plan_add_conv_c32_head(miseg_plan* p, const miseg_conv_desc* body, const miseg_conv_desc* head);
int miseg_plan_add_dense_layer(miseg_plan* p, const miseg_conv_desc* c1, const miseg_conv_desc* c2);
int miseg_plan_add_linknet_tail(miseg_plan* p, const miseg_linknet_tail_desc* d);
int miseg_plan_add_linknet_block(miseg_plan* p, const miseg_linknet_block_desc* d);
int miseg_plan_add_fpn_head(miseg_plan* p, const miseg_fpn_head_desc* d);
int miseg_plan_add_prep_input(miseg_plan* p, const float* in, void* out, const float* scale,
                              const float* shift, int32_t N, int32_t H, int32_t W);
int miseg_plan_add_maxpool3x3s2(miseg_plan* p, const void* in, void* out, const float* scale,
                                const float* shift, int32_t N, int32_t H, int32_t W, int32_t C,
                                int32_t in_pix_stride, int32_t out_pix_stride);
int miseg_plan_add_avgpool2x2(miseg_plan* p, const void* in, void* out, int32_t N, int32_t H,
                              int32_t W, int32_t C, int32_t in_pix_stride,
                              int32_t out_pix_stride);
int miseg_plan_add_maxpool2x2(miseg_plan* p, const void* in, void* out, int32_t N, int32_t H,
                              int32_t W, int32_t C, int32_t in_pix_stride,
                              int32_t out_pix_stride);
int miseg_plan_add_im2col3x3(miseg_plan* p, const float* in, void* out, int32_t N, int32_t H, int32_t W);
int miseg_plan_add_head_gather(miseg_plan* p, const float* y, const float* bias, float* probs, void* collapse_out,
                               int32_t N, int32_t h, int32_t w, int32_t classes, int32_t y_pix_stride,
                               int32_t act, float temperature, int32_t collapse_value);
int miseg_plan_add_upsample_nearest(miseg_plan* p, const void* in, const void* addend, void* out,
                                    int32_t N, int32_t H, int32_t W, int32_t C, int32_t shift,
                                    int32_t in_pix_stride, int32_t add_pix_stride,
                                    int32_t out_pix_stride);
int miseg_plan_add_upsample_bilinear2x(miseg_plan* p, const void* in, void* out, int32_t N,
                                       int32_t H, int32_t W, int32_t C, int32_t in_pix_stride,
                                       int32_t out_pix_stride);
int miseg_plan_add_scale_shift_relu(miseg_plan* p, const void* in, void* out, const float* scale,
                                    const float* shift, int64_t pixels, int32_t C,
                                    int32_t in_pix_stride, int32_t out_pix_stride);
/* Launch every recorded op, in order, on `stream`. */
int miseg_plan_run(const miseg_plan* p, void* stream);
/*
 * Launch ops [first, last) and, when ms != NULL, time them with HIP events
 * recorded on `stream` (synchronises the stream): *ms = elapsed milliseconds.
 */
int miseg_plan_run_range(const miseg_plan* p, int first, int last, void* stream, float* ms);
/* Algorithmic multiply-accumulates of conv op `index` (0 for other ops). */
int64_t miseg_plan_op_macs(const miseg_plan* p, int index);
/* Name of the device kernel op `index` launches (as a profiler prints it), into buf. */
int miseg_plan_op_kernel(const miseg_plan* p, int index, char* buf, int32_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* MISEG_H */
