// C ABI of libmiseg.so (declared in include/miseg.h) and the plan recorder/replayer.
#include <string.h>

#include <string>
#include <vector>

#include "common.h"

namespace miseg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

#include "inference_decls.inc"
int launch_collapse_u8(const float*, uint8_t*, long long, int, int, hipStream_t);
int launch_head_gather(const float*, const float*, float*, uint8_t*, int, int, int, int, int, int, float, int, int,
                       hipStream_t);
int launch_head_gather_bwd(const void*, int, void*, int, int, int, int, int, hipStream_t);
int launch_loss_sums(const float*, const float*, double*, long long, int, hipStream_t);
int launch_wgrad(const miseg_wgrad_desc* d, hipStream_t stream);
int launch_bn_finalize(const double*, int, int, long long, int, const float*, const float*, float*, float*,
                       float, float, float*, float*, float*, float*, hipStream_t);
int launch_bn_stats(const void*, long long, int, int, const float*, const float*, float*, float*, float,
                    float, float*, float*, float*, float*, double*, hipStream_t);
int launch_bn_stats_noalloc(const void*, long long, int, int, const float*, const float*, float*, float*,
                            float, float, float*, float*, float*, float*, hipStream_t);
int noalloc_setup(double** out);
int launch_input_stats(const float*, long long, const float*, float*, float*, float, float, float*,
                       float*, float*, float*, double*, hipStream_t);
int launch_bn_relu_bwd_reduce(const void*, const void*, long long, int, int, int, const float*,
                              const float*, const float*, const float*, float, int, double*,
                              hipStream_t);
int launch_bn_relu_bwd_apply(const void*, const void*, void*, long long, int, int, int, int,
                             const float*, const float*, const float*, const float*, float, int,
                             const double*, const void*, int, hipStream_t);
int launch_bn_bwd_sums(const double*, int, int, int, const float*, const float*, float, double*, hipStream_t);
int launch_channel_sums(const void*, long long, int, int, double*, hipStream_t);
int launch_image_minmax(const void*, int, int, long long, float*, hipStream_t);
int launch_image_normalize(const void*, int, int, long long, const float*, float*, hipStream_t);
int launch_resize_bilinear(const void*, int, float*, int, int, int, int, int, int, hipStream_t);
int launch_resize_nearest_u8(const unsigned char*, unsigned char*, int, int, int, int, int, int, int,
                             hipStream_t);
int launch_one_hot(const unsigned char*, float*, long long, int, hipStream_t);
int launch_augment(const float*, const float*, float*, float*, int, int, int, int, const float*, double*,
                   hipStream_t);
int launch_maxpool3x3s2_bwd(const void*, const void*, void*, int, int, int, int, int, int, int, const void*,
                            int, hipStream_t);
int launch_avgpool2x2_bwd(const void*, void*, int, int, int, int, int, int, hipStream_t);
int launch_maxpool2x2_bwd(const void*, const void*, void*, int, int, int, int, int, int, int, hipStream_t);
int launch_relu_bwd(const void*, const void*, const void*, void*, long long, int, int, int, int, int,
                    hipStream_t);
int launch_upsample_nearest_bwd(const void*, void*, int, int, int, int, int, int, int, hipStream_t);
int launch_upsample_bilinear2x_bwd(const void*, void*, int, int, int, int, int, int, hipStream_t);
int launch_copy_channels(const void*, const void*, void*, long long, int, int, int, int, hipStream_t);
int launch_loss_grad(const float*, const float*, const double*, void*, long long, int, int, float,
                     float, int, int, hipStream_t);
int launch_loss_grad_ex(const float*, const float*, const double*, void*, long long, int, int, float, float,
                        int, float, float, hipStream_t);
int launch_loss_sums_focal(const float*, const float*, double*, long long, int, float, float, hipStream_t);
int launch_adam(float*, float*, float*, const float*, void*, long long, float, float, float, float,
                float, hipStream_t);

}  // namespace miseg
namespace miseg_f16 {
#include "inference_decls.inc"
}
namespace miseg {

enum OpKind { OP_CONV, OP_PREP, OP_MAXPOOL, OP_AVGPOOL, OP_UPNEAR, OP_UPBIL, OP_SSR, OP_CONV_HEAD, OP_DENSE, OP_LTAIL, OP_FPNHEAD,
              OP_MAXPOOL2, OP_IM2COL, OP_HEADGATHER, OP_LBLOCK };

struct Op {
  OpKind kind;
  miseg_conv_desc conv;
  miseg_linknet_tail_desc tail;   // OP_LTAIL
  miseg_linknet_block_desc block; // OP_LBLOCK
  miseg_fpn_head_desc fhead;      // OP_FPNHEAD
  miseg_conv_desc conv2;   // OP_CONV_HEAD: the head convolution; OP_DENSE: the 3x3 convolution
  const void* in;
  const void* in2;
  void* out;
  const float* scale;
  const float* shift;
  long long pixels;
  int N, H, W, C, shift_bits, ips, aps, ops;
  int act, cval;        // OP_HEADGATHER
  float temperature;
};

static int run_op_f16(const Op& o, hipStream_t s) {
  namespace F = ::miseg_f16;
  switch (o.kind) {
    case OP_CONV: return F::launch_conv(&o.conv, s);
    case OP_CONV_HEAD: return F::launch_conv_c32_head(&o.conv, &o.conv2, s);
    case OP_DENSE: return F::launch_dense_layer(&o.conv, &o.conv2, s);
    case OP_LTAIL: return F::launch_linknet_tail(&o.tail, s);
    case OP_LBLOCK: return F::launch_linknet_block(&o.block, s);
    case OP_FPNHEAD: return F::launch_fpn_head(&o.fhead, s);
    case OP_PREP: return F::launch_prep_input((const float*)o.in, o.out, o.scale, o.shift, o.N, o.H, o.W, s);
    case OP_MAXPOOL:
      return F::launch_maxpool3x3s2(o.in, o.out, o.scale, o.shift, o.N, o.H, o.W, o.C, o.ips, o.ops, s);
    case OP_AVGPOOL: return F::launch_avgpool2x2(o.in, o.out, o.N, o.H, o.W, o.C, o.ips, o.ops, s);
    case OP_MAXPOOL2: return F::launch_maxpool2x2(o.in, o.out, o.N, o.H, o.W, o.C, o.ips, o.ops, s);
    case OP_IM2COL: return F::launch_im2col3x3((const float*)o.in, o.out, o.N, o.H, o.W, 9, s);
    case OP_HEADGATHER:
      return ::miseg::launch_head_gather((const float*)o.in, o.scale, (float*)o.out, (uint8_t*)const_cast<void*>(o.in2), o.N,
                                         o.H, o.W, o.C, o.ips, o.act, o.temperature, o.cval, 0, s);
    case OP_UPNEAR:
      return F::launch_upsample_nearest(o.in, o.in2, o.out, o.N, o.H, o.W, o.C, o.shift_bits, o.ips,
                                        o.aps, o.ops, s);
    case OP_UPBIL: return F::launch_upsample_bilinear2x(o.in, o.out, o.N, o.H, o.W, o.C, o.ips, o.ops, s);
    case OP_SSR:
      return F::launch_scale_shift_relu(o.in, o.out, o.scale, o.shift, o.pixels, o.C, o.ips, o.ops, s);
  }
  set_error("plan: unknown op kind");
  return MISEG_E_INVALID_ARG;
}

static int run_op(const Op& o, hipStream_t s, int fmt) {
  if (fmt == MISEG_FMT_F16) return run_op_f16(o, s);
  switch (o.kind) {
    case OP_CONV: return launch_conv(&o.conv, s);
    case OP_CONV_HEAD: return launch_conv_c32_head(&o.conv, &o.conv2, s);
    case OP_DENSE: return launch_dense_layer(&o.conv, &o.conv2, s);
    case OP_LTAIL: return launch_linknet_tail(&o.tail, s);
    case OP_LBLOCK: return launch_linknet_block(&o.block, s);
    case OP_FPNHEAD: return launch_fpn_head(&o.fhead, s);
    case OP_PREP: return launch_prep_input((const float*)o.in, o.out, o.scale, o.shift, o.N, o.H, o.W, s);
    case OP_MAXPOOL:
      return launch_maxpool3x3s2(o.in, o.out, o.scale, o.shift, o.N, o.H, o.W, o.C, o.ips, o.ops, s);
    case OP_AVGPOOL: return launch_avgpool2x2(o.in, o.out, o.N, o.H, o.W, o.C, o.ips, o.ops, s);
    case OP_MAXPOOL2: return launch_maxpool2x2(o.in, o.out, o.N, o.H, o.W, o.C, o.ips, o.ops, s);
    case OP_IM2COL: return launch_im2col3x3((const float*)o.in, o.out, o.N, o.H, o.W, 9, s);
    case OP_HEADGATHER:
      return ::miseg::launch_head_gather((const float*)o.in, o.scale, (float*)o.out, (uint8_t*)const_cast<void*>(o.in2), o.N,
                                         o.H, o.W, o.C, o.ips, o.act, o.temperature, o.cval, 0, s);
    case OP_UPNEAR:
      return launch_upsample_nearest(o.in, o.in2, o.out, o.N, o.H, o.W, o.C, o.shift_bits, o.ips,
                                     o.aps, o.ops, s);
    case OP_UPBIL: return launch_upsample_bilinear2x(o.in, o.out, o.N, o.H, o.W, o.C, o.ips, o.ops, s);
    case OP_SSR:
      return launch_scale_shift_relu(o.in, o.out, o.scale, o.shift, o.pixels, o.C, o.ips, o.ops, s);
  }
  set_error("plan: unknown op kind");
  return MISEG_E_INVALID_ARG;
}

}  // namespace miseg

struct miseg_plan {
  std::vector<miseg::Op> ops;
  int fmt = MISEG_FMT_BF16;
};

#define MISEG_CHECK_FMT(fmt) MISEG_CHECK_ARG((fmt) == MISEG_FMT_BF16 || (fmt) == MISEG_FMT_F16, "unknown storage format %d", (int)(fmt))

using namespace miseg;

extern "C" {

const char* miseg_last_error(void) { return g_err; }

int miseg_version(void) { return 100; }

int miseg_device_check(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    set_error("no HIP device");
    return MISEG_E_NO_DEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    set_error("hipGetDeviceProperties failed");
    return MISEG_E_NO_DEVICE;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s, this library is built for gfx950 only", dev, prop.gcnArchName);
    return MISEG_E_NO_DEVICE;
  }
  // set-up of the allocation-free entry point: look up this device's scratch array now, so that
  // miseg_bn_stats_fwd_noalloc makes no runtime query later (e.g. under stream capture)
  return noalloc_setup(nullptr);
}

int miseg_conv2d_fwd(const miseg_conv_desc* d, void* stream) {
  MISEG_CHECK_ARG(d, "conv: null descriptor");
  MISEG_CHECK_FMT(d->fmt);
  if (d->fmt == MISEG_FMT_F16) return miseg_f16::launch_conv(d, (hipStream_t)stream);
  return launch_conv(d, (hipStream_t)stream);
}

int miseg_conv_kernel_name(const miseg_conv_desc* d, char* buf, int32_t buflen) {
  MISEG_CHECK_ARG(d && buf && buflen > 0, "conv_kernel_name: null argument");
  if (d->fmt == MISEG_FMT_F16)
    miseg_f16::conv_kernel_name(d, buf, buflen);
  else
    conv_kernel_name(d, buf, buflen);
  return MISEG_OK;
}

int miseg_conv_c32_head_fwd(const miseg_conv_desc* body, const miseg_conv_desc* head, void* stream) {
  MISEG_CHECK_ARG(body && head, "conv_c32_head: null descriptor");
  MISEG_CHECK_FMT(body->fmt);
  MISEG_CHECK_ARG(body->fmt == head->fmt, "conv_c32_head: body and head must use one storage format");
  if (body->fmt == MISEG_FMT_F16) return miseg_f16::launch_conv_c32_head(body, head, (hipStream_t)stream);
  return launch_conv_c32_head(body, head, (hipStream_t)stream);
}

int miseg_dense_layer_fwd(const miseg_conv_desc* c1, const miseg_conv_desc* c2, void* stream) {
  MISEG_CHECK_ARG(c1 && c2, "dense_layer: null descriptor");
  MISEG_CHECK_FMT(c1->fmt);
  MISEG_CHECK_ARG(c1->fmt == c2->fmt, "dense_layer: the two convolutions must use one storage format");
  if (c1->fmt == MISEG_FMT_F16) return miseg_f16::launch_dense_layer(c1, c2, (hipStream_t)stream);
  return launch_dense_layer(c1, c2, (hipStream_t)stream);
}

int miseg_linknet_tail_fwd(const miseg_linknet_tail_desc* d, void* stream) {
  MISEG_CHECK_ARG(d, "linknet_tail: null descriptor");
  MISEG_CHECK_FMT(d->fmt);
  if (d->fmt == MISEG_FMT_F16) return miseg_f16::launch_linknet_tail(d, (hipStream_t)stream);
  return launch_linknet_tail(d, (hipStream_t)stream);
}

int miseg_linknet_block_fwd(const miseg_linknet_block_desc* d, void* stream) {
  MISEG_CHECK_ARG(d, "linknet_block: null descriptor");
  MISEG_CHECK_FMT(d->fmt);
  if (d->fmt == MISEG_FMT_F16) return miseg_f16::launch_linknet_block(d, (hipStream_t)stream);
  return launch_linknet_block(d, (hipStream_t)stream);
}

int miseg_fpn_head_fwd(const miseg_fpn_head_desc* d, void* stream) {
  MISEG_CHECK_ARG(d, "fpn_head: null descriptor");
  MISEG_CHECK_FMT(d->fmt);
  if (d->fmt == MISEG_FMT_F16) return miseg_f16::launch_fpn_head(d, (hipStream_t)stream);
  return launch_fpn_head(d, (hipStream_t)stream);
}

// fp16-storage twins of the layers between the convolutions (inference only)
int miseg_prep_input_f16(const float* in, void* out, const float* scale, const float* shift, int32_t N,
                         int32_t H, int32_t W, void* stream) {
  return miseg_f16::launch_prep_input(in, out, scale, shift, N, H, W, (hipStream_t)stream);
}
int miseg_maxpool3x3s2_fwd_f16(const void* in, void* out, const float* scale, const float* shift,
                               int32_t N, int32_t H, int32_t W, int32_t C, int32_t ips, int32_t ops,
                               void* stream) {
  return miseg_f16::launch_maxpool3x3s2(in, out, scale, shift, N, H, W, C, ips, ops, (hipStream_t)stream);
}
int miseg_avgpool2x2_fwd_f16(const void* in, void* out, int32_t N, int32_t H, int32_t W, int32_t C,
                             int32_t ips, int32_t ops, void* stream) {
  return miseg_f16::launch_avgpool2x2(in, out, N, H, W, C, ips, ops, (hipStream_t)stream);
}
int miseg_upsample_nearest_fwd_f16(const void* in, const void* addend, void* out, int32_t N, int32_t H,
                                   int32_t W, int32_t C, int32_t shift, int32_t ips, int32_t aps,
                                   int32_t ops, void* stream) {
  return miseg_f16::launch_upsample_nearest(in, addend, out, N, H, W, C, shift, ips, aps, ops,
                                            (hipStream_t)stream);
}
int miseg_upsample_bilinear2x_fwd_f16(const void* in, void* out, int32_t N, int32_t H, int32_t W,
                                      int32_t C, int32_t ips, int32_t ops, void* stream) {
  return miseg_f16::launch_upsample_bilinear2x(in, out, N, H, W, C, ips, ops, (hipStream_t)stream);
}
int miseg_scale_shift_relu_fwd_f16(const void* in, void* out, const float* scale, const float* shift,
                                   int64_t pixels, int32_t C, int32_t ips, int32_t ops, void* stream) {
  return miseg_f16::launch_scale_shift_relu(in, out, scale, shift, pixels, C, ips, ops, (hipStream_t)stream);
}

int miseg_prep_input(const float* in, void* out, const float* scale, const float* shift, int32_t N,
                     int32_t H, int32_t W, void* stream) {
  return launch_prep_input(in, out, scale, shift, N, H, W, (hipStream_t)stream);
}

int miseg_prep_input_indicator(const float* in, void* out, const float* scale, const float* shift,
                               int32_t N, int32_t H, int32_t W, void* stream) {
  return launch_prep_input(in, out, scale, shift, N, H, W, (hipStream_t)stream, true);
}

int miseg_maxpool3x3s2_fwd(const void* in, void* out, const float* scale, const float* shift,
                           int32_t N, int32_t H, int32_t W, int32_t C, int32_t ips, int32_t ops,
                           void* stream) {
  return launch_maxpool3x3s2(in, out, scale, shift, N, H, W, C, ips, ops, (hipStream_t)stream);
}

int miseg_avgpool2x2_fwd(const void* in, void* out, int32_t N, int32_t H, int32_t W, int32_t C,
                         int32_t ips, int32_t ops, void* stream) {
  return launch_avgpool2x2(in, out, N, H, W, C, ips, ops, (hipStream_t)stream);
}

int miseg_upsample_nearest_fwd(const void* in, const void* addend, void* out, int32_t N, int32_t H,
                               int32_t W, int32_t C, int32_t shift, int32_t ips, int32_t aps,
                               int32_t ops, void* stream) {
  return launch_upsample_nearest(in, addend, out, N, H, W, C, shift, ips, aps, ops,
                                 (hipStream_t)stream);
}

int miseg_upsample_bilinear2x_fwd(const void* in, void* out, int32_t N, int32_t H, int32_t W,
                                  int32_t C, int32_t ips, int32_t ops, void* stream) {
  return launch_upsample_bilinear2x(in, out, N, H, W, C, ips, ops, (hipStream_t)stream);
}

int miseg_scale_shift_relu_fwd(const void* in, void* out, const float* scale, const float* shift,
                               int64_t pixels, int32_t C, int32_t ips, int32_t ops, void* stream) {
  return launch_scale_shift_relu(in, out, scale, shift, pixels, C, ips, ops, (hipStream_t)stream);
}

int miseg_collapse_u8(const float* probs, uint8_t* out, int64_t pixels, int32_t C, int32_t value,
                      void* stream) {
  return launch_collapse_u8(probs, out, pixels, C, value, (hipStream_t)stream);
}

int miseg_loss_sums(const float* gt, const float* pr, double* sums, int64_t pixels, int32_t C,
                    void* stream) {
  return launch_loss_sums(gt, pr, sums, pixels, C, (hipStream_t)stream);
}

// ---- training ----
int miseg_conv2d_wgrad(const miseg_wgrad_desc* d, void* stream) {
  return launch_wgrad(d, (hipStream_t)stream);
}
int miseg_bn_finalize(const double* stats, int32_t replicas, int32_t stride, int64_t pixels, int32_t C,
                      const float* gamma, const float* beta, float* mm, float* mv, float momentum,
                      float eps, float* mean, float* var, float* scale, float* shift, void* stream) {
  return launch_bn_finalize(stats, replicas, stride, pixels, C, gamma, beta, mm, mv, momentum, eps, mean,
                            var, scale, shift, (hipStream_t)stream);
}

int miseg_bn_stats_fwd(const void* x, int64_t pixels, int32_t C, int32_t ps, const float* gamma,
                       const float* beta, float* mm, float* mv, float momentum, float eps,
                       float* mean, float* var, float* scale, float* shift, double* ws, void* stream) {
  return launch_bn_stats(x, pixels, C, ps, gamma, beta, mm, mv, momentum, eps, mean, var, scale,
                         shift, ws, (hipStream_t)stream);
}
int miseg_bn_stats_fwd_noalloc(const void* x, int64_t pixels, int32_t C, int32_t ps, const float* gamma,
                               const float* beta, float* mm, float* mv, float momentum, float eps,
                               float* mean, float* var, float* scale, float* shift, void* stream) {
  return launch_bn_stats_noalloc(x, pixels, C, ps, gamma, beta, mm, mv, momentum, eps, mean, var, scale,
                                 shift, (hipStream_t)stream);
}
int miseg_input_stats_fwd(const float* x, int64_t pixels, const float* beta, float* mm, float* mv,
                          float momentum, float eps, float* mean, float* var, float* scale,
                          float* shift, double* ws, void* stream) {
  return launch_input_stats(x, pixels, beta, mm, mv, momentum, eps, mean, var, scale, shift, ws,
                            (hipStream_t)stream);
}
int miseg_bn_relu_bwd_reduce(const void* x, const void* dy, int64_t pixels, int32_t C, int32_t xps,
                             int32_t dps, const float* scale, const float* shift, const float* mean,
                             const float* var, float eps, int32_t relu, double* sums, void* stream) {
  return launch_bn_relu_bwd_reduce(x, dy, pixels, C, xps, dps, scale, shift, mean, var, eps, relu,
                                   sums, (hipStream_t)stream);
}
int miseg_bn_relu_bwd_apply(const void* x, const void* dy, void* dx, int64_t pixels, int32_t C,
                            int32_t xps, int32_t dps, int32_t ops, const float* scale,
                            const float* shift, const float* mean, const float* var, float eps,
                            int32_t relu, const double* sums, const void* addend,
                            int32_t addend_pix_stride, void* stream) {
  return launch_bn_relu_bwd_apply(x, dy, dx, pixels, C, xps, dps, ops, scale, shift, mean, var, eps,
                                  relu, sums, addend, addend_pix_stride, (hipStream_t)stream);
}
int miseg_bn_bwd_sums(const double* stats, int32_t replicas, int32_t stride, int32_t C, const float* mean,
                      const float* var, float eps, double* sums, void* stream) {
  return launch_bn_bwd_sums(stats, replicas, stride, C, mean, var, eps, sums, (hipStream_t)stream);
}
int miseg_image_minmax(const void* img, int32_t dtype, int32_t N, int64_t elems, float* minmax,
                       void* stream) {
  return launch_image_minmax(img, dtype, N, elems, minmax, (hipStream_t)stream);
}
int miseg_image_normalize(const void* img, int32_t dtype, int32_t N, int64_t elems, const float* minmax,
                          float* out, void* stream) {
  return launch_image_normalize(img, dtype, N, elems, minmax, out, (hipStream_t)stream);
}
int miseg_resize_bilinear(const void* in, int32_t dtype, float* out, int32_t N, int32_t H, int32_t W,
                          int32_t C, int32_t Ho, int32_t Wo, void* stream) {
  return launch_resize_bilinear(in, dtype, out, N, H, W, C, Ho, Wo, (hipStream_t)stream);
}
int miseg_resize_nearest_u8(const uint8_t* in, uint8_t* out, int32_t N, int32_t H, int32_t W, int32_t C,
                            int32_t Ho, int32_t Wo, int32_t mode, void* stream) {
  return launch_resize_nearest_u8(in, out, N, H, W, C, Ho, Wo, mode, (hipStream_t)stream);
}
int miseg_one_hot(const uint8_t* ids, float* out, int64_t pixels, int32_t C, void* stream) {
  return launch_one_hot(ids, out, pixels, C, (hipStream_t)stream);
}
int miseg_augment(const float* img, const float* mask, float* img_out, float* mask_out, int32_t N,
                  int32_t H, int32_t W, int32_t MC, const float* params, double* workspace,
                  void* stream) {
  return launch_augment(img, mask, img_out, mask_out, N, H, W, MC, params, workspace,
                        (hipStream_t)stream);
}
int miseg_maxpool3x3s2_bwd(const void* x, const void* dy, void* dx, int32_t N, int32_t H, int32_t W,
                           int32_t C, int32_t xps, int32_t dps, int32_t ops, const void* addend, int32_t aps,
                           void* stream) {
  return launch_maxpool3x3s2_bwd(x, dy, dx, N, H, W, C, xps, dps, ops, addend, aps, (hipStream_t)stream);
}
int miseg_maxpool2x2_fwd(const void* in, void* out, int32_t N, int32_t H, int32_t W, int32_t C,
                         int32_t ips, int32_t ops, int32_t fmt, void* stream) {
  MISEG_CHECK_FMT(fmt);
  if (fmt == MISEG_FMT_F16) return miseg_f16::launch_maxpool2x2(in, out, N, H, W, C, ips, ops, (hipStream_t)stream);
  return launch_maxpool2x2(in, out, N, H, W, C, ips, ops, (hipStream_t)stream);
}
int miseg_im2col3x3_fwd(const float* in, void* out, int32_t N, int32_t H, int32_t W, int32_t taps,
                        int32_t fmt, void* stream) {
  MISEG_CHECK_FMT(fmt);
  if (fmt == MISEG_FMT_F16) return miseg_f16::launch_im2col3x3(in, out, N, H, W, taps, (hipStream_t)stream);
  return launch_im2col3x3(in, out, N, H, W, taps, (hipStream_t)stream);
}
int miseg_maxpool2x2_bwd(const void* x, const void* dy, void* dx, int32_t N, int32_t H, int32_t W,
                         int32_t C, int32_t xps, int32_t dps, int32_t ops, void* stream) {
  return launch_maxpool2x2_bwd(x, dy, dx, N, H, W, C, xps, dps, ops, (hipStream_t)stream);
}
int miseg_relu_bwd(const void* y, const void* dy, const void* addend, void* dx, int64_t pixels, int32_t C,
                   int32_t yps, int32_t dps, int32_t aps, int32_t ops, void* stream) {
  return launch_relu_bwd(y, dy, addend, dx, pixels, C, yps, dps, aps, ops, (hipStream_t)stream);
}
int miseg_avgpool2x2_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                         int32_t dps, int32_t ops, void* stream) {
  return launch_avgpool2x2_bwd(dy, dx, N, H, W, C, dps, ops, (hipStream_t)stream);
}
int miseg_upsample_nearest_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                               int32_t shift, int32_t dps, int32_t ops, void* stream) {
  return launch_upsample_nearest_bwd(dy, dx, N, H, W, C, shift, dps, ops, (hipStream_t)stream);
}
int miseg_upsample_bilinear2x_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W,
                                  int32_t C, int32_t dps, int32_t ops, void* stream) {
  return launch_upsample_bilinear2x_bwd(dy, dx, N, H, W, C, dps, ops, (hipStream_t)stream);
}
int miseg_copy_channels(const void* in, const void* addend, void* out, int64_t pixels, int32_t C,
                        int32_t ips, int32_t aps, int32_t ops, void* stream) {
  return launch_copy_channels(in, addend, out, pixels, C, ips, aps, ops, (hipStream_t)stream);
}
int miseg_channel_sums(const void* x, int64_t pixels, int32_t C, int32_t ps, double* sums,
                       void* stream) {
  return launch_channel_sums(x, pixels, C, ps, sums, (hipStream_t)stream);
}
int miseg_loss_grad(const float* gt, const float* pr, const double* sums, void* dlogits,
                    int64_t pixels, int32_t C, int32_t out_stride, float loss_scale,
                    float temperature, int32_t use_cce, int32_t use_dice, void* stream) {
  return launch_loss_grad(gt, pr, sums, dlogits, pixels, C, out_stride, loss_scale, temperature,
                          use_cce, use_dice, (hipStream_t)stream);
}
int miseg_loss_grad_ex(const float* gt, const float* pr, const double* sums, void* dlogits,
                       int64_t pixels, int32_t C, int32_t out_stride, float loss_scale,
                       float temperature, int32_t flags, float alpha, float gamma, void* stream) {
  return launch_loss_grad_ex(gt, pr, sums, dlogits, pixels, C, out_stride, loss_scale, temperature, flags,
                             alpha, gamma, (hipStream_t)stream);
}
int miseg_loss_sums_focal(const float* gt, const float* pr, double* sums, int64_t pixels, int32_t C,
                          float alpha, float gamma, void* stream) {
  return launch_loss_sums_focal(gt, pr, sums, pixels, C, alpha, gamma, (hipStream_t)stream);
}
int miseg_adam_step(float* p, float* m, float* v, const float* g, void* p_bf16, int64_t count,
                    float lr_t, float beta1, float beta2, float eps, float grad_scale, void* stream) {
  return launch_adam(p, m, v, g, p_bf16, count, lr_t, beta1, beta2, eps, grad_scale,
                     (hipStream_t)stream);
}

// ---- plan ----
miseg_plan* miseg_plan_create(void) { return new miseg_plan(); }
void miseg_plan_destroy(miseg_plan* p) { delete p; }
int miseg_plan_num_ops(const miseg_plan* p) { return p ? (int)p->ops.size() : 0; }

int miseg_plan_set_format(miseg_plan* p, int32_t fmt) {
  MISEG_CHECK_ARG(p, "plan: null plan");
  MISEG_CHECK_FMT(fmt);
  MISEG_CHECK_ARG(p->ops.empty(), "plan: the storage format must be chosen before the first op is added");
  p->fmt = fmt;
  return MISEG_OK;
}

int miseg_plan_add_conv(miseg_plan* p, const miseg_conv_desc* d) {
  MISEG_CHECK_ARG(p && d, "plan: null argument");
  MISEG_CHECK_ARG(d->fmt == p->fmt, "plan: the convolution's storage format (%d) differs from the plan's (%d)",
                  (int)d->fmt, p->fmt);
  int rc = p->fmt == MISEG_FMT_F16 ? miseg_f16::validate_conv(d) : validate_conv(d);
  if (rc != MISEG_OK) return rc;
  Op o{};
  o.kind = OP_CONV;
  o.conv = *d;
  p->ops.push_back(o);
  return MISEG_OK;
}

int miseg_plan_add_conv_c32_head(miseg_plan* p, const miseg_conv_desc* body, const miseg_conv_desc* head) {
  MISEG_CHECK_ARG(p && body && head, "plan: null argument");
  MISEG_CHECK_ARG(body->fmt == p->fmt && head->fmt == p->fmt,
                  "plan: the convolutions' storage format differs from the plan's (%d)", p->fmt);
  int rc = p->fmt == MISEG_FMT_F16 ? miseg_f16::validate_conv_c32_head(body, head)
                                   : validate_conv_c32_head(body, head);
  if (rc != MISEG_OK) return rc;
  Op o{};
  o.kind = OP_CONV_HEAD;
  o.conv = *body;
  o.conv2 = *head;
  p->ops.push_back(o);
  return MISEG_OK;
}

int miseg_plan_add_dense_layer(miseg_plan* p, const miseg_conv_desc* c1, const miseg_conv_desc* c2) {
  MISEG_CHECK_ARG(p && c1 && c2, "plan: null argument");
  MISEG_CHECK_ARG(c1->fmt == p->fmt && c2->fmt == p->fmt,
                  "plan: the convolutions' storage format differs from the plan's (%d)", p->fmt);
  // the kernel's coefficient area holds twice as many input channels with fp16 storage
  int rc = p->fmt == MISEG_FMT_F16 ? miseg_f16::validate_dense_layer(c1, c2) : validate_dense_layer(c1, c2);
  if (rc != MISEG_OK) return rc;
  Op o{};
  o.kind = OP_DENSE;
  o.conv = *c1;
  o.conv2 = *c2;
  p->ops.push_back(o);
  return MISEG_OK;
}

int miseg_plan_add_linknet_tail(miseg_plan* p, const miseg_linknet_tail_desc* d) {
  MISEG_CHECK_ARG(p && d, "plan: null argument");
  MISEG_CHECK_ARG(d->fmt == p->fmt, "plan: the tail's storage format differs from the plan's (%d)", p->fmt);
  const int rc = p->fmt == MISEG_FMT_F16 ? miseg_f16::validate_linknet_tail(d) : validate_linknet_tail(d);
  if (rc != MISEG_OK) return rc;
  Op o{};
  o.kind = OP_LTAIL;
  o.tail = *d;
  p->ops.push_back(o);
  return MISEG_OK;
}

int miseg_plan_add_linknet_block(miseg_plan* p, const miseg_linknet_block_desc* d) {
  MISEG_CHECK_ARG(p && d, "plan: null argument");
  MISEG_CHECK_ARG(d->fmt == p->fmt, "plan: the block's storage format differs from the plan's (%d)", p->fmt);
  const int rc = p->fmt == MISEG_FMT_F16 ? miseg_f16::validate_linknet_block(d) : validate_linknet_block(d);
  if (rc != MISEG_OK) return rc;
  Op o{};
  o.kind = OP_LBLOCK;
  o.block = *d;
  p->ops.push_back(o);
  return MISEG_OK;
}

int miseg_plan_add_fpn_head(miseg_plan* p, const miseg_fpn_head_desc* d) {
  MISEG_CHECK_ARG(p && d, "plan: null argument");
  MISEG_CHECK_ARG(d->fmt == p->fmt, "plan: the head's storage format differs from the plan's (%d)", p->fmt);
  const int rc = p->fmt == MISEG_FMT_F16 ? miseg_f16::validate_fpn_head(d) : validate_fpn_head(d);
  if (rc != MISEG_OK) return rc;
  Op o{};
  o.kind = OP_FPNHEAD;
  o.fhead = *d;
  p->ops.push_back(o);
  return MISEG_OK;
}

int miseg_plan_add_prep_input(miseg_plan* p, const float* in, void* out, const float* scale,
                              const float* shift, int32_t N, int32_t H, int32_t W) {
  MISEG_CHECK_ARG(p, "plan: null plan");
  Op o{};
  o.kind = OP_PREP; o.in = in; o.out = out; o.scale = scale; o.shift = shift;
  o.N = N; o.H = H; o.W = W;
  p->ops.push_back(o);
  return MISEG_OK;
}

int miseg_plan_add_maxpool3x3s2(miseg_plan* p, const void* in, void* out, const float* scale,
                                const float* shift, int32_t N, int32_t H, int32_t W, int32_t C,
                                int32_t ips, int32_t ops) {
  MISEG_CHECK_ARG(p, "plan: null plan");
  Op o{};
  o.kind = OP_MAXPOOL; o.in = in; o.out = out; o.scale = scale; o.shift = shift;
  o.N = N; o.H = H; o.W = W; o.C = C; o.ips = ips; o.ops = ops;
  p->ops.push_back(o);
  return MISEG_OK;
}

int miseg_plan_add_avgpool2x2(miseg_plan* p, const void* in, void* out, int32_t N, int32_t H,
                              int32_t W, int32_t C, int32_t ips, int32_t ops) {
  MISEG_CHECK_ARG(p, "plan: null plan");
  Op o{};
  o.kind = OP_AVGPOOL; o.in = in; o.out = out;
  o.N = N; o.H = H; o.W = W; o.C = C; o.ips = ips; o.ops = ops;
  p->ops.push_back(o);
  return MISEG_OK;
}

int miseg_plan_add_maxpool2x2(miseg_plan* p, const void* in, void* out, int32_t N, int32_t H,
                              int32_t W, int32_t C, int32_t ips, int32_t ops) {
  MISEG_CHECK_ARG(p, "plan: null plan");
  Op o{};
  o.kind = OP_MAXPOOL2; o.in = in; o.out = out;
  o.N = N; o.H = H; o.W = W; o.C = C; o.ips = ips; o.ops = ops;
  p->ops.push_back(o);
  return MISEG_OK;
}

int miseg_head_gather_fwd(const float* y, const float* bias, float* probs, void* collapse_out, int32_t N,
                          int32_t h, int32_t w, int32_t classes, int32_t y_pix_stride, int32_t act,
                          float temperature, int32_t collapse_value, int32_t class_major, void* stream) {
  return launch_head_gather(y, bias, probs, (uint8_t*)collapse_out, N, h, w, classes, y_pix_stride, act, temperature,
                            collapse_value, class_major, (hipStream_t)stream);
}

int miseg_head_gather_bwd(const void* dz, int32_t dz_pix_stride, void* dy, int32_t N, int32_t h, int32_t w,
                          int32_t classes, int32_t dy_pix_stride, void* stream) {
  return launch_head_gather_bwd(dz, dz_pix_stride, dy, N, h, w, classes, dy_pix_stride, (hipStream_t)stream);
}

int miseg_plan_add_head_gather(miseg_plan* p, const float* y, const float* bias, float* probs, void* collapse_out,
                               int32_t N, int32_t h, int32_t w, int32_t classes, int32_t y_pix_stride,
                               int32_t act, float temperature, int32_t collapse_value) {
  MISEG_CHECK_ARG(p, "plan: null plan");
  Op o{};
  o.kind = OP_HEADGATHER; o.in = y; o.scale = bias; o.out = probs; o.in2 = collapse_out;
  o.N = N; o.H = h; o.W = w; o.C = classes; o.ips = y_pix_stride;
  o.act = act; o.temperature = temperature; o.cval = collapse_value;
  p->ops.push_back(o);
  return MISEG_OK;
}

int miseg_plan_add_im2col3x3(miseg_plan* p, const float* in, void* out, int32_t N, int32_t H, int32_t W) {
  MISEG_CHECK_ARG(p, "plan: null plan");
  Op o{};
  o.kind = OP_IM2COL; o.in = in; o.out = out;
  o.N = N; o.H = H; o.W = W;
  p->ops.push_back(o);
  return MISEG_OK;
}

int miseg_plan_add_upsample_nearest(miseg_plan* p, const void* in, const void* addend, void* out,
                                    int32_t N, int32_t H, int32_t W, int32_t C, int32_t shift,
                                    int32_t ips, int32_t aps, int32_t ops) {
  MISEG_CHECK_ARG(p, "plan: null plan");
  Op o{};
  o.kind = OP_UPNEAR; o.in = in; o.in2 = addend; o.out = out;
  o.N = N; o.H = H; o.W = W; o.C = C; o.shift_bits = shift; o.ips = ips; o.aps = aps; o.ops = ops;
  p->ops.push_back(o);
  return MISEG_OK;
}

int miseg_plan_add_upsample_bilinear2x(miseg_plan* p, const void* in, void* out, int32_t N,
                                       int32_t H, int32_t W, int32_t C, int32_t ips, int32_t ops) {
  MISEG_CHECK_ARG(p, "plan: null plan");
  Op o{};
  o.kind = OP_UPBIL; o.in = in; o.out = out;
  o.N = N; o.H = H; o.W = W; o.C = C; o.ips = ips; o.ops = ops;
  p->ops.push_back(o);
  return MISEG_OK;
}

int miseg_plan_add_scale_shift_relu(miseg_plan* p, const void* in, void* out, const float* scale,
                                    const float* shift, int64_t pixels, int32_t C, int32_t ips,
                                    int32_t ops) {
  MISEG_CHECK_ARG(p, "plan: null plan");
  Op o{};
  o.kind = OP_SSR; o.in = in; o.out = out; o.scale = scale; o.shift = shift;
  o.pixels = pixels; o.C = C; o.ips = ips; o.ops = ops;
  p->ops.push_back(o);
  return MISEG_OK;
}

int miseg_plan_run_range(const miseg_plan* p, int first, int last, void* stream, float* ms) {
  MISEG_CHECK_ARG(p, "plan: null plan");
  MISEG_CHECK_ARG(first >= 0 && last <= (int)p->ops.size() && first <= last, "plan: bad op range");
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ms) {
    MISEG_CHECK_HIP(hipEventCreate(&e0));
    MISEG_CHECK_HIP(hipEventCreate(&e1));
    MISEG_CHECK_HIP(hipEventRecord(e0, s));
  }
  int rc = MISEG_OK;
  for (int i = first; i < last && rc == MISEG_OK; ++i) rc = run_op(p->ops[i], s, p->fmt);
  if (ms) {
    hipError_t e = hipEventRecord(e1, s);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc == MISEG_OK && e != hipSuccess) {
      set_error("plan timing failed: %s", hipGetErrorString(e));
      rc = MISEG_E_HIP;
    }
  }
  return rc;
}

int miseg_plan_run(const miseg_plan* p, void* stream) {
  MISEG_CHECK_ARG(p, "plan: null plan");
  return miseg_plan_run_range(p, 0, (int)p->ops.size(), stream, nullptr);
}

int64_t miseg_plan_op_macs(const miseg_plan* p, int index) {
  if (!p || index < 0 || index >= (int)p->ops.size()) return 0;
  const Op& o = p->ops[index];
  if (o.kind == OP_LTAIL) {   // per input pixel: 64*16 (a) + 4 phases * 4 taps * 16*16 (b); per output pixel: 16*16 + 9*16*classes
    const miseg_linknet_tail_desc& t = o.tail;
    return (int64_t)t.N * t.H * t.W * (64 * 16 + 4 * 4 * 256 + 4 * (256 + 144 * t.classes));
  }
  if (o.kind == OP_LBLOCK) {   // per input pixel: cin*m (a) + 4 phases * 4 taps * m*m (b); per output pixel: m*64 (c)
    const int64_t cin = o.block.cin, m = cin / 4;
    return (int64_t)o.block.N * o.block.H * o.block.W * (cin * m + 4 * 4 * m * m + 4 * m * 64);
  }
  if (o.kind == OP_FPNHEAD) return (int64_t)o.fhead.N * o.fhead.H * o.fhead.W * 4 * 9 * 128 * o.fhead.classes;
  if (o.kind != OP_CONV && o.kind != OP_CONV_HEAD && o.kind != OP_DENSE) return 0;
  const miseg_conv_desc& d = o.conv;
  if (o.kind == OP_DENSE)   // the layer's own arithmetic (the kernel recomputes the 1x1 on the 3x3's halo)
    return (int64_t)d.N * d.H * d.W * ((int64_t)d.Cin * 128 + 9 * 128 * 32);
  int64_t per = (int64_t)d.N * d.Ho * d.Wo * d.Cout * d.KH * d.KW * d.Cin;
  // the 7x7x3 stem runs as 7x1 taps over 8 pixels x 4 channels (K = 224): count the layer's own 147
  if (d.KH == 7 && d.KW == 1 && d.Cin == 32 && d.stride_y == 2) per = (int64_t)d.N * d.Ho * d.Wo * d.Cout * 147;
  if (o.kind == OP_CONV_HEAD) return per + (int64_t)d.N * d.Ho * d.Wo * o.conv2.Cout * 9 * o.conv2.Cin;
  return d.transposed ? 4 * per : per;
}

int miseg_plan_op_kernel(const miseg_plan* p, int index, char* buf, int32_t buflen) {
  MISEG_CHECK_ARG(p && buf && buflen > 0, "plan_op_kernel: null argument");
  MISEG_CHECK_ARG(index >= 0 && index < (int)p->ops.size(), "plan_op_kernel: bad index");
  const Op& o = p->ops[index];
  static const char* names[] = {"", "prep_input_kernel", "maxpool3x3s2_kernel", "avgpool2x2_kernel",
                                "upsample_nearest_kernel", "upsample_bilinear2x_kernel",
                                "scale_shift_relu_kernel", "", "", "", "", "maxpool2x2_kernel",
                                "im2col3x3_kernel", "head_gather_kernel", "linknet_block_kernel"};
  if (o.kind == OP_CONV && p->fmt == MISEG_FMT_F16)
    miseg_f16::conv_kernel_name(&o.conv, buf, buflen);
  else if (o.kind == OP_CONV)
    conv_kernel_name(&o.conv, buf, buflen);
  else if (o.kind == OP_CONV_HEAD)
    snprintf(buf, buflen, "conv3x3_c32_head_kernel");
  else if (o.kind == OP_DENSE)
    snprintf(buf, buflen, "dense_layer_kernel");
  else if (o.kind == OP_LTAIL)
    snprintf(buf, buflen, "linknet_tail_kernel");
  else if (o.kind == OP_FPNHEAD)
    snprintf(buf, buflen, "fpn_head_kernel");
  else
    snprintf(buf, buflen, "%s", names[o.kind]);
  return MISEG_OK;
}

}  // extern "C"
