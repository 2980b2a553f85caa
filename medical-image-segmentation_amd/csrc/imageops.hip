// The steps either side of the network, on the device (gfx950): the input contract of
// /root/reference utils/data.py (bilinear resize :133-138, per-image min-max normalisation :64-80,
// one-hot masks :50-54, the augmentations of :240-295) and the way back of utils/predict.py:164-166
// (nearest resize of the collapsed mask to the original size).  All tensors are NHWC, interleaved
// channels; every kernel is a single pass bounded by HBM bandwidth.
#include "common.h"

namespace miseg {

static inline int io_grid(long long work, int block = 256, int cap = 8192) {
  long long g = (work + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// order-preserving map float -> uint so that unsigned atomicMin/Max order floats
__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

template <typename T>
__global__ void minmax_kernel(const T* __restrict__ img, long long per_image, unsigned* __restrict__ ord) {
  const int n = blockIdx.y;
  const T* p = img + (long long)n * per_image;
  float lo = 3.0e38f, hi = -3.0e38f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_image;
       i += (long long)gridDim.x * blockDim.x) {
    const float v = (float)p[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_down(lo, o, 64));
    hi = fmaxf(hi, __shfl_down(hi, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&ord[2 * n], f2ord(lo));
    atomicMax(&ord[2 * n + 1], f2ord(hi));
  }
}

__global__ void minmax_init_kernel(unsigned* ord, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    ord[2 * i] = 0xffffffffu;
    ord[2 * i + 1] = 0u;
  }
}
__global__ void minmax_decode_kernel(unsigned* ord, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 2 * n) reinterpret_cast<float*>(ord)[i] = ord2f(ord[i]);
}

int launch_image_minmax(const void* img, int dtype, int N, long long per_image, float* minmax,
                        hipStream_t s) {
  MISEG_CHECK_ARG(img && minmax && N > 0 && per_image > 0 && (dtype == 0 || dtype == 1),
                  "image_minmax: bad argument (dtype 0 = u8, 1 = f32)");
  unsigned* ord = reinterpret_cast<unsigned*>(minmax);
  hipLaunchKernelGGL(minmax_init_kernel, dim3((N + 63) / 64), dim3(64), 0, s, ord, N);
  const dim3 grid(io_grid(per_image, 256, 1024), N);
  if (dtype == 0)
    hipLaunchKernelGGL(minmax_kernel<unsigned char>, grid, dim3(256), 0, s, (const unsigned char*)img,
                       per_image, ord);
  else
    hipLaunchKernelGGL(minmax_kernel<float>, grid, dim3(256), 0, s, (const float*)img, per_image, ord);
  hipLaunchKernelGGL(minmax_decode_kernel, dim3((2 * N + 63) / 64), dim3(64), 0, s, ord, N);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// (x - min) / (max - min), the division as the reference does it (no reciprocal)
template <typename T>
__global__ void normalize_kernel(const T* __restrict__ img, long long per_image,
                                 const float* __restrict__ minmax, float* __restrict__ out) {
  const int n = blockIdx.y;
  const float lo = minmax[2 * n], range = minmax[2 * n + 1] - lo;
  const T* p = img + (long long)n * per_image;
  float* o = out + (long long)n * per_image;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_image;
       i += (long long)gridDim.x * blockDim.x)
    o[i] = ((float)p[i] - lo) / range;
}

int launch_image_normalize(const void* img, int dtype, int N, long long per_image, const float* minmax,
                           float* out, hipStream_t s) {
  MISEG_CHECK_ARG(img && minmax && out && N > 0 && per_image > 0 && (dtype == 0 || dtype == 1),
                  "image_normalize: bad argument");
  const dim3 grid(io_grid(per_image, 256, 2048), N);
  if (dtype == 0)
    hipLaunchKernelGGL(normalize_kernel<unsigned char>, grid, dim3(256), 0, s, (const unsigned char*)img,
                       per_image, minmax, out);
  else
    hipLaunchKernelGGL(normalize_kernel<float>, grid, dim3(256), 0, s, (const float*)img, per_image,
                       minmax, out);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// tf.image.resize(method="bilinear"): half-pixel centres, no antialiasing
template <typename T>
__global__ void resize_bilinear_kernel(const T* __restrict__ in, float* __restrict__ out, int N, int H,
                                       int W, int C, int Ho, int Wo, float sy, float sx) {
  const long long total = (long long)N * Ho * Wo * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
    const float fly = floorf(fy), flx = floorf(fx);
    const int y0 = max((int)fly, 0), y1 = min((int)ceilf(fy), H - 1);
    const int x0 = max((int)flx, 0), x1 = min((int)ceilf(fx), W - 1);
    const float ly = fy - fly, lx = fx - flx;
    const T* b = in + (long long)n * H * W * C + c;
    const float v00 = (float)b[((long long)y0 * W + x0) * C], v01 = (float)b[((long long)y0 * W + x1) * C];
    const float v10 = (float)b[((long long)y1 * W + x0) * C], v11 = (float)b[((long long)y1 * W + x1) * C];
    const float top = v00 + (v01 - v00) * lx, bot = v10 + (v11 - v10) * lx;
    out[i] = top + (bot - top) * ly;
  }
}

int launch_resize_bilinear(const void* in, int dtype, float* out, int N, int H, int W, int C, int Ho,
                           int Wo, hipStream_t s) {
  MISEG_CHECK_ARG(in && out && N > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 &&
                      (dtype == 0 || dtype == 1),
                  "resize_bilinear: bad argument");
  const long long total = (long long)N * Ho * Wo * C;
  const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
  if (dtype == 0)
    hipLaunchKernelGGL(resize_bilinear_kernel<unsigned char>, dim3(io_grid(total)), dim3(256), 0, s,
                       (const unsigned char*)in, out, N, H, W, C, Ho, Wo, sy, sx);
  else
    hipLaunchKernelGGL(resize_bilinear_kernel<float>, dim3(io_grid(total)), dim3(256), 0, s,
                       (const float*)in, out, N, H, W, C, Ho, Wo, sy, sx);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// mode 0: tf.image.resize(method="nearest"): src = min(floor((dst + 0.5) * scale), size - 1)
// mode 1: cv2.resize(..., INTER_NEAREST):    src = min(floor(dst * scale), size - 1)
__global__ void resize_nearest_u8_kernel(const unsigned char* __restrict__ in,
                                         unsigned char* __restrict__ out, int N, int H, int W, int C,
                                         int Ho, int Wo, float sy, float sx, float off) {
  const long long total = (long long)N * Ho * Wo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    long long r = i / Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const int iy = min((int)floorf(((float)oy + off) * sy), H - 1);
    const int ix = min((int)floorf(((float)ox + off) * sx), W - 1);
    const unsigned char* src = in + (((long long)n * H + iy) * W + ix) * C;
    unsigned char* dst = out + i * C;
    for (int c = 0; c < C; ++c) dst[c] = src[c];
  }
}

int launch_resize_nearest_u8(const unsigned char* in, unsigned char* out, int N, int H, int W, int C,
                             int Ho, int Wo, int mode, hipStream_t s) {
  MISEG_CHECK_ARG(in && out && N > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 &&
                      (mode == 0 || mode == 1),
                  "resize_nearest_u8: bad argument (mode 0 = tf.image, 1 = cv2)");
  const long long total = (long long)N * Ho * Wo;
  // cv2 computes the scale in double and the product in float; for integer ratios both agree
  const float sy = (float)((double)H / (double)Ho), sx = (float)((double)W / (double)Wo);
  hipLaunchKernelGGL(resize_nearest_u8_kernel, dim3(io_grid(total)), dim3(256), 0, s, in, out, N, H, W, C,
                     Ho, Wo, sy, sx, mode == 0 ? 0.5f : 0.0f);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// tf.one_hot(ids, depth=C, axis=2) as float32: ids outside [0, C) give an all-zero pixel
__global__ void one_hot_kernel(const unsigned char* __restrict__ ids, float* __restrict__ out,
                               long long pixels, int C) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pixels;
       i += (long long)gridDim.x * blockDim.x) {
    const int id = ids[i];
    float* o = out + i * C;
    for (int c = 0; c < C; ++c) o[c] = c == id ? 1.f : 0.f;
  }
}

int launch_one_hot(const unsigned char* ids, float* out, long long pixels, int C, hipStream_t s) {
  MISEG_CHECK_ARG(ids && out && pixels > 0 && C > 0 && C <= 256, "one_hot: bad argument");
  hipLaunchKernelGGL(one_hot_kernel, dim3(io_grid(pixels)), dim3(256), 0, s, ids, out, pixels, C);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ---------------------------------------------------------------------------------- augmentation
// Per-image parameters (float[8] each): flip_lr, flip_ud (0/1), brightness delta, contrast factor,
// hue delta, saturation factor, contrast_on, colour_on.  The image is float32 RGB; the mask (any
// channel count, float32) only takes the flips.  Order as utils/data.py:267-293: flips,
// brightness, contrast, hue, saturation.  Contrast needs the per-channel mean of the image after
// brightness: adding a constant shifts the mean by the same constant, so the means are taken from
// the input (`chan_mean`, float[3] per image) and shifted.
__device__ __forceinline__ void rgb_to_hsv(float r, float g, float b, float& h, float& s, float& v) {
  v = fmaxf(r, fmaxf(g, b));
  const float mn = fminf(r, fminf(g, b));
  const float range = v - mn;
  s = v > 0.f ? range / v : 0.f;
  const float norm = 1.0f / (6.0f * range);
  if (range <= 0.f) h = 0.f;
  else if (r == v) h = norm * (g - b);
  else if (g == v) h = norm * (b - r) + 2.0f / 6.0f;
  else h = norm * (r - g) + 4.0f / 6.0f;
  if (h < 0.f) h += 1.f;
}
__device__ __forceinline__ void hsv_to_rgb(float h, float s, float v, float& r, float& g, float& b) {
  const float c = s * v, m = v - c;
  const float dh = h * 6.f;
  float fmodu = dh;
  while (fmodu <= 0.f) fmodu += 2.f;
  while (fmodu >= 2.f) fmodu -= 2.f;
  const float x = c * (1.f - fabsf(fmodu - 1.f));
  const int hc = (int)dh;
  float rr = 0, gg = 0, bb = 0;
  switch (hc) {
    case 0: rr = c; gg = x; break;
    case 1: rr = x; gg = c; break;
    case 2: gg = c; bb = x; break;
    case 3: gg = x; bb = c; break;
    case 4: rr = x; bb = c; break;
    default: rr = c; bb = x; break;
  }
  r = rr + m; g = gg + m; b = bb + m;
}

__global__ void channel_mean3_kernel(const float* __restrict__ img, long long pixels, double* sums) {
  const int n = blockIdx.y;
  const float* p = img + (long long)n * pixels * 3;
  float a0 = 0, a1 = 0, a2 = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pixels;
       i += (long long)gridDim.x * blockDim.x) {
    a0 += p[3 * i];
    a1 += p[3 * i + 1];
    a2 += p[3 * i + 2];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a0 += __shfl_down(a0, o, 64);
    a1 += __shfl_down(a1, o, 64);
    a2 += __shfl_down(a2, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&sums[3 * n], (double)a0);
    atomicAdd(&sums[3 * n + 1], (double)a1);
    atomicAdd(&sums[3 * n + 2], (double)a2);
  }
}

__global__ void augment_kernel(const float* __restrict__ img, const float* __restrict__ mask,
                               float* __restrict__ img_out, float* __restrict__ mask_out, int H, int W,
                               int MC, const float* __restrict__ params,
                               const double* __restrict__ sums) {
  const int n = blockIdx.y;
  const float* pr = params + 8 * n;
  const bool flr = pr[0] > 0.5f, fud = pr[1] > 0.5f;
  const float bright = pr[2], contrast = pr[3], hue = pr[4], sat = pr[5];
  const bool contrast_on = pr[6] > 0.5f, colour_on = pr[7] > 0.5f;
  const long long pixels = (long long)H * W;
  float mean[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) mean[c] = (float)(sums[3 * n + c] / (double)pixels) + bright;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pixels;
       i += (long long)gridDim.x * blockDim.x) {
    const int oy = (int)(i / W), ox = (int)(i - (long long)oy * W);
    const int iy = fud ? H - 1 - oy : oy, ix = flr ? W - 1 - ox : ox;
    const long long src = (long long)n * pixels + (long long)iy * W + ix;
    const long long dst = (long long)n * pixels + i;
    float r = img[3 * src] + bright, g = img[3 * src + 1] + bright, b = img[3 * src + 2] + bright;
    if (contrast_on) {
      r = (r - mean[0]) * contrast + mean[0];
      g = (g - mean[1]) * contrast + mean[1];
      b = (b - mean[2]) * contrast + mean[2];
    }
    if (colour_on) {
      float h, s, v;
      rgb_to_hsv(r, g, b, h, s, v);
      h += hue;
      h -= floorf(h);
      s = fminf(fmaxf(s * sat, 0.f), 1.f);
      hsv_to_rgb(h, s, v, r, g, b);
    }
    img_out[3 * dst] = r;
    img_out[3 * dst + 1] = g;
    img_out[3 * dst + 2] = b;
    if (mask)
      for (int c = 0; c < MC; ++c) mask_out[dst * MC + c] = mask[src * MC + c];
  }
}

int launch_augment(const float* img, const float* mask, float* img_out, float* mask_out, int N, int H,
                   int W, int MC, const float* params, double* workspace, hipStream_t s) {
  MISEG_CHECK_ARG(img && img_out && params && workspace && N > 0 && H > 0 && W > 0 &&
                      ((mask == nullptr) == (mask_out == nullptr)) && (!mask || MC > 0) &&
                      img != img_out && (!mask || mask != mask_out),
                  "augment: bad argument (outputs must not alias inputs)");
  MISEG_CHECK_HIP(hipMemsetAsync(workspace, 0, sizeof(double) * 3 * N, s));
  const long long pixels = (long long)H * W;
  hipLaunchKernelGGL(channel_mean3_kernel, dim3(io_grid(pixels, 256, 512), N), dim3(256), 0, s, img,
                     pixels, workspace);
  hipLaunchKernelGGL(augment_kernel, dim3(io_grid(pixels, 256, 2048), N), dim3(256), 0, s, img, mask,
                     img_out, mask_out, H, W, MC, params, workspace);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

}  // namespace miseg
