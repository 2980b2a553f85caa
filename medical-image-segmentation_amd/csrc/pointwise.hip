// Bandwidth-bound kernels around the convolutions (gfx950): input preparation, pooling,
// upsampling, BatchNorm+ReLU application, probability collapse and the loss/metric sums.
// All bf16 tensors are NHWC with a pixel stride (elements) so they can be channel slices
// of wider buffers; every thread moves 16 bytes (8 bf16 channels) per access.
#include "common.h"

namespace MISEG_KNS {

static inline int grid_for(long long work, int block = 256) {
  long long g = (work + block - 1) / block;
  const long long cap = 256 * 16;  // grid-stride beyond ~16 blocks per CU
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------------------------
// fp32 [N][H][W][3] -> bf16 [N][H+6][W+8][4], image at (3,3), zero elsewhere
// ---------------------------------------------------------------------------------
// IND: channel 3 = 1 inside the image (the indicator the training path uses to obtain the input
// BatchNorm's beta gradient from the stem's weight gradient), else 0 everywhere
template <bool IND>
__global__ void prep_input_kernel(const float* __restrict__ in, h16_t* __restrict__ out,
                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                  int N, int H, int W) {
  const int Hp = H + 6, Wp = W + 8;
  const long long total = (long long)N * Hp * Wp;
  const float s0 = scale[0], s1 = scale[1], s2 = scale[2];
  const float t0 = shift[0], t1 = shift[1], t2 = shift[2];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int xp = (int)(i % Wp);
    const long long r = i / Wp;
    const int yp = (int)(r % Hp);
    const int n = (int)(r / Hp);
    const int y = yp - 3, x = xp - 3;
    uint2 v = make_uint2(0, 0);
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
      const float* px = in + (((long long)n * H + y) * W + x) * 3;
      v.x = pack_h16x2(px[0] * s0 + t0, px[1] * s1 + t1);
      v.y = pack_h16x2(px[2] * s2 + t2, IND ? 1.f : 0.f);
    }
    *reinterpret_cast<uint2*>(out + i * 4) = v;
  }
}

int launch_prep_input(const float* in, void* out, const float* scale, const float* shift, int N,
                      int H, int W, hipStream_t s, bool indicator) {
  MISEG_CHECK_ARG(in && out && scale && shift, "prep_input: null pointer");
  MISEG_CHECK_ARG(N > 0 && H > 0 && W > 0 && (W & 1) == 0 && (H & 1) == 0,
                  "prep_input: H and W must be positive and even");
  const long long total = (long long)N * (H + 6) * (W + 8);
  if (indicator)
    hipLaunchKernelGGL(prep_input_kernel<true>, dim3(grid_for(total)), dim3(256), 0, s, in, (h16_t*)out,
                       scale, shift, N, H, W);
  else
    hipLaunchKernelGGL(prep_input_kernel<false>, dim3(grid_for(total)), dim3(256), 0, s, in, (h16_t*)out,
                       scale, shift, N, H, W);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ---------------------------------------------------------------------------------
// max pool 3x3 stride 2 after zero pad 1 (inputs are post-ReLU, so 0 == -inf padding)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void max8(float (&m)[8], const uint4 v) {
  m[0] = fmaxf(m[0], h16lo_to_f32(v.x)); m[1] = fmaxf(m[1], h16hi_to_f32(v.x));
  m[2] = fmaxf(m[2], h16lo_to_f32(v.y)); m[3] = fmaxf(m[3], h16hi_to_f32(v.y));
  m[4] = fmaxf(m[4], h16lo_to_f32(v.z)); m[5] = fmaxf(m[5], h16hi_to_f32(v.z));
  m[6] = fmaxf(m[6], h16lo_to_f32(v.w)); m[7] = fmaxf(m[7], h16hi_to_f32(v.w));
}
__device__ __forceinline__ void unpack8(float (&m)[8], const uint4 v) {
  m[0] = h16lo_to_f32(v.x); m[1] = h16hi_to_f32(v.x);
  m[2] = h16lo_to_f32(v.y); m[3] = h16hi_to_f32(v.y);
  m[4] = h16lo_to_f32(v.z); m[5] = h16hi_to_f32(v.z);
  m[6] = h16lo_to_f32(v.w); m[7] = h16hi_to_f32(v.w);
}
__device__ __forceinline__ uint4 pack8(const float (&m)[8]) {
  return make_uint4(pack_h16x2(m[0], m[1]), pack_h16x2(m[2], m[3]), pack_h16x2(m[4], m[5]),
                    pack_h16x2(m[6], m[7]));
}

// One thread: 8 channels of MP_ROWS vertically adjacent output pixels.  An input row between two
// output rows belongs to both windows; walking down the column keeps it in registers (the running
// maximum of each input row over its three columns), so a thread fetches 2 * MP_ROWS + 1 input rows
// instead of 3 * MP_ROWS, and neighbouring threads of a wave cover neighbouring columns.
constexpr int MP_ROWS = 4;
__global__ void maxpool3x3s2_kernel(const h16_t* __restrict__ in, h16_t* __restrict__ out,
                                    const float* __restrict__ scale,
                                    const float* __restrict__ shift, int N, int H, int W, int C,
                                    int ips, int ops) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int C8 = C >> 3;
  const int Hg = (Ho + MP_ROWS - 1) / MP_ROWS;
  const long long total = (long long)N * Hg * Wo * C8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    long long r = i / C8;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int og = (int)(r % Hg);
    const int n = (int)(r / Hg);
    const int oy0 = og * MP_ROWS;
    float sc[8], sh[8];
    if (scale) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sc[j] = scale[c8 * 8 + j];
        sh[j] = shift[c8 * 8 + j];
      }
    }
    // maximum over the three columns of input row y (zeros outside the image: inputs are >= 0)
    auto row_max = [&](int y, float (&m)[8]) {
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = 0.f;
      if ((unsigned)y >= (unsigned)H) return;
      const h16_t* rowp = in + (((long long)n * H + y) * W) * ips + c8 * 8;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int x = 2 * ox - 1 + dx;
        if ((unsigned)x >= (unsigned)W) continue;
        max8(m, *reinterpret_cast<const uint4*>(rowp + (long long)x * ips));
      }
    };
    float carry[8];                       // input row 2*oy - 1 of the next output row
    row_max(2 * oy0 - 1, carry);
#pragma unroll
    for (int k = 0; k < MP_ROWS; ++k) {
      const int oy = oy0 + k;
      if (oy >= Ho) break;
      float a[8], b[8], m[8];
      row_max(2 * oy, a);
      row_max(2 * oy + 1, b);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        m[j] = fmaxf(fmaxf(carry[j], a[j]), b[j]);
        carry[j] = b[j];
      }
      if (scale) {
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j] * sc[j] + sh[j], 0.f);
      }
      *reinterpret_cast<uint4*>(out + (((long long)n * Ho + oy) * Wo + ox) * ops + c8 * 8) = pack8(m);
    }
  }
}

int launch_maxpool3x3s2(const void* in, void* out, const float* scale, const float* shift, int N,
                        int H, int W, int C, int ips, int ops, hipStream_t s) {
  MISEG_CHECK_ARG(in && out, "maxpool: null pointer");
  MISEG_CHECK_ARG((scale == nullptr) == (shift == nullptr), "maxpool: scale and shift go together");
  MISEG_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0 && (ips & 7) == 0 &&
                      (ops & 7) == 0 && ips >= C && ops >= C,
                  "maxpool: C and strides must be multiples of 8");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = (long long)N * ((Ho + MP_ROWS - 1) / MP_ROWS) * Wo * (C >> 3);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total)), dim3(256), 0, s, (const h16_t*)in,
                     (h16_t*)out, scale, shift, N, H, W, C, ips, ops);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ---------------------------------------------------------------------------------
// average pool 2x2 stride 2
// ---------------------------------------------------------------------------------
__global__ void avgpool2x2_kernel(const h16_t* __restrict__ in, h16_t* __restrict__ out, int N,
                                  int H, int W, int C, int ips, int ops) {
  const int Ho = H >> 1, Wo = W >> 1, C8 = C >> 3;
  const long long total = (long long)N * Ho * Wo * C8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    long long r = i / C8;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float t[8];
        unpack8(t, *reinterpret_cast<const uint4*>(
                       in + (((long long)n * H + 2 * oy + dy) * W + 2 * ox + dx) * ips + c8 * 8));
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += t[j];
      }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= 0.25f;
    *reinterpret_cast<uint4*>(out + (((long long)n * Ho + oy) * Wo + ox) * ops + c8 * 8) =
        pack8(acc);
  }
}

int launch_avgpool2x2(const void* in, void* out, int N, int H, int W, int C, int ips, int ops,
                      hipStream_t s) {
  MISEG_CHECK_ARG(in && out, "avgpool: null pointer");
  MISEG_CHECK_ARG(N > 0 && H > 1 && W > 1 && C > 0 && (C & 7) == 0 && (ips & 7) == 0 &&
                      (ops & 7) == 0 && ips >= C && ops >= C,
                  "avgpool: C and strides must be multiples of 8");
  const long long total = (long long)N * (H >> 1) * (W >> 1) * (C >> 3);
  hipLaunchKernelGGL(avgpool2x2_kernel, dim3(grid_for(total)), dim3(256), 0, s, (const h16_t*)in,
                     (h16_t*)out, N, H, W, C, ips, ops);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ---------------------------------------------------------------------------------
// max pool 2x2 stride 2 (keras_applications VGG block pools)
// ---------------------------------------------------------------------------------
__global__ void maxpool2x2_kernel(const h16_t* __restrict__ in, h16_t* __restrict__ out, int N,
                                  int H, int W, int C, int ips, int ops) {
  const int Ho = H >> 1, Wo = W >> 1, C8 = C >> 3;
  const long long total = (long long)N * Ho * Wo * C8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    long long r = i / C8;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float best[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) best[j] = -3.0e38f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float t[8];
        unpack8(t, *reinterpret_cast<const uint4*>(
                       in + (((long long)n * H + 2 * oy + dy) * W + 2 * ox + dx) * ips + c8 * 8));
#pragma unroll
        for (int j = 0; j < 8; ++j) best[j] = fmaxf(best[j], t[j]);
      }
    *reinterpret_cast<uint4*>(out + (((long long)n * Ho + oy) * Wo + ox) * ops + c8 * 8) =
        pack8(best);      // exact: the maximum is one of the stored values
  }
}

int launch_maxpool2x2(const void* in, void* out, int N, int H, int W, int C, int ips, int ops,
                      hipStream_t s) {
  MISEG_CHECK_ARG(in && out, "maxpool2x2: null pointer");
  MISEG_CHECK_ARG(N > 0 && H > 1 && W > 1 && (H & 1) == 0 && (W & 1) == 0 && C > 0 && (C & 7) == 0 &&
                      (ips & 7) == 0 && (ops & 7) == 0 && ips >= C && ops >= C,
                  "maxpool2x2: H and W must be even, C and strides multiples of 8");
  const long long total = (long long)N * (H >> 1) * (W >> 1) * (C >> 3);
  hipLaunchKernelGGL(maxpool2x2_kernel, dim3(grid_for(total)), dim3(256), 0, s, (const h16_t*)in,
                     (h16_t*)out, N, H, W, C, ips, ops);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ---------------------------------------------------------------------------------
// fp32 [N][H][W][3] -> 16-bit [N][H][W][32]: the 3x3 neighbourhood of every pixel (zero outside the
// image), element (ky*3 + kx)*3 + c = in[y + ky - 1][x + kx - 1][c], elements 27..31 zero.  A 3x3
// 'same' convolution of the 3-channel image is then a 1x1 convolution with 32 input channels.
// taps == 1: the centre tap only, i.e. the image itself padded to 32 channels.
// ---------------------------------------------------------------------------------
__global__ void im2col3x3_kernel(const float* __restrict__ in, h16_t* __restrict__ out, int N, int H,
                                 int W, int taps) {
  const long long total = (long long)N * H * W * 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i & 3);
    long long r = i >> 2;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = q * 8 + j;
      const int t = k / 3, c = k - t * 3;
      const int tap = taps == 9 ? t : 4;
      const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
      v[j] = (k < 3 * taps && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                 ? in[(((long long)n * H + yy) * W + xx) * 3 + c]
                 : 0.f;
    }
    *reinterpret_cast<uint4*>(out + i * 8) = pack8(v);
  }
}

int launch_im2col3x3(const float* in, void* out, int N, int H, int W, int taps, hipStream_t s) {
  MISEG_CHECK_ARG(in && out && N > 0 && H > 0 && W > 0 && (taps == 9 || taps == 1), "im2col3x3: bad argument");
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for((long long)N * H * W * 4)), dim3(256), 0, s, in,
                     (h16_t*)out, N, H, W, taps);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ---------------------------------------------------------------------------------
// nearest upsample by 2^shift, optional add
// ---------------------------------------------------------------------------------
__global__ void upsample_nearest_kernel(const h16_t* __restrict__ in,
                                        const h16_t* __restrict__ addend,
                                        h16_t* __restrict__ out, int N, int H, int W, int C,
                                        int shift, int ips, int aps, int ops) {
  const int Ho = H << shift, Wo = W << shift, C8 = C >> 3;
  const long long total = (long long)N * Ho * Wo * C8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    long long r = i / C8;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    uint4 v = *reinterpret_cast<const uint4*>(
        in + (((long long)n * H + (oy >> shift)) * W + (ox >> shift)) * ips + c8 * 8);
    const long long opix = ((long long)n * Ho + oy) * Wo + ox;
    if (addend) {
      float a[8], b[8];
      unpack8(a, v);
      unpack8(b, *reinterpret_cast<const uint4*>(addend + opix * aps + c8 * 8));
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += b[j];
      v = pack8(a);
    }
    *reinterpret_cast<uint4*>(out + opix * ops + c8 * 8) = v;
  }
}

int launch_upsample_nearest(const void* in, const void* addend, void* out, int N, int H, int W,
                            int C, int shift, int ips, int aps, int ops, hipStream_t s) {
  MISEG_CHECK_ARG(in && out, "upsample_nearest: null pointer");
  MISEG_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0 && (ips & 7) == 0 &&
                      (ops & 7) == 0 && (!addend || (aps & 7) == 0) && shift >= 0 && shift <= 4,
                  "upsample_nearest: C and strides must be multiples of 8, shift in 0..4");
  const long long total = ((long long)N * (H << shift) * (W << shift)) * (C >> 3);
  hipLaunchKernelGGL(upsample_nearest_kernel, dim3(grid_for(total)), dim3(256), 0, s,
                     (const h16_t*)in, (const h16_t*)addend, (h16_t*)out, N, H, W, C, shift, ips,
                     aps, ops);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ---------------------------------------------------------------------------------
// bilinear x2, half-pixel centres: src = (dst + 0.5)/2 - 0.5 clamped to [0, n-1]
// ---------------------------------------------------------------------------------
__global__ void upsample_bilinear2x_kernel(const h16_t* __restrict__ in, h16_t* __restrict__ out,
                                           int N, int H, int W, int C, int ips, int ops) {
  const int Ho = 2 * H, Wo = 2 * W, C8 = C >> 3;
  const long long total = (long long)N * Ho * Wo * C8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    long long r = i / C8;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float sy = fminf(fmaxf((oy + 0.5f) * 0.5f - 0.5f, 0.f), (float)(H - 1));
    float sx = fminf(fmaxf((ox + 0.5f) * 0.5f - 0.5f, 0.f), (float)(W - 1));
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float fy = sy - y0, fx = sx - x0;
    const h16_t* base = in + (long long)n * H * W * ips + c8 * 8;
    float a[8], b[8], c[8], d[8], o[8];
    unpack8(a, *reinterpret_cast<const uint4*>(base + ((long long)y0 * W + x0) * ips));
    unpack8(b, *reinterpret_cast<const uint4*>(base + ((long long)y0 * W + x1) * ips));
    unpack8(c, *reinterpret_cast<const uint4*>(base + ((long long)y1 * W + x0) * ips));
    unpack8(d, *reinterpret_cast<const uint4*>(base + ((long long)y1 * W + x1) * ips));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float top = a[j] * (1.f - fx) + b[j] * fx;
      const float bot = c[j] * (1.f - fx) + d[j] * fx;
      o[j] = top * (1.f - fy) + bot * fy;
    }
    *reinterpret_cast<uint4*>(out + (((long long)n * Ho + oy) * Wo + ox) * ops + c8 * 8) = pack8(o);
  }
}

int launch_upsample_bilinear2x(const void* in, void* out, int N, int H, int W, int C, int ips,
                               int ops, hipStream_t s) {
  MISEG_CHECK_ARG(in && out, "upsample_bilinear2x: null pointer");
  MISEG_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0 && (ips & 7) == 0 &&
                      (ops & 7) == 0,
                  "upsample_bilinear2x: C and strides must be multiples of 8");
  const long long total = (long long)N * 4 * H * W * (C >> 3);
  hipLaunchKernelGGL(upsample_bilinear2x_kernel, dim3(grid_for(total)), dim3(256), 0, s,
                     (const h16_t*)in, (h16_t*)out, N, H, W, C, ips, ops);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ---------------------------------------------------------------------------------
// y = relu(x*scale + shift)
// ---------------------------------------------------------------------------------
__global__ void scale_shift_relu_kernel(const h16_t* __restrict__ in, h16_t* __restrict__ out,
                                        const float* __restrict__ scale,
                                        const float* __restrict__ shift, long long pixels, int C,
                                        int ips, int ops) {
  const int C8 = C >> 3;
  const long long total = pixels * C8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    const long long pix = i / C8;
    float a[8];
    unpack8(a, *reinterpret_cast<const uint4*>(in + pix * ips + c8 * 8));
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = fmaxf(a[j] * scale[c8 * 8 + j] + shift[c8 * 8 + j], 0.f);
    *reinterpret_cast<uint4*>(out + pix * ops + c8 * 8) = pack8(a);
  }
}

int launch_scale_shift_relu(const void* in, void* out, const float* scale, const float* shift,
                            long long pixels, int C, int ips, int ops, hipStream_t s) {
  MISEG_CHECK_ARG(in && out && scale && shift, "scale_shift_relu: null pointer");
  MISEG_CHECK_ARG(pixels > 0 && C > 0 && (C & 7) == 0 && (ips & 7) == 0 && (ops & 7) == 0,
                  "scale_shift_relu: C and strides must be multiples of 8");
  hipLaunchKernelGGL(scale_shift_relu_kernel, dim3(grid_for(pixels * (C >> 3))), dim3(256), 0, s,
                     (const h16_t*)in, (h16_t*)out, scale, shift, pixels, C, ips, ops);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ---------------------------------------------------------------------------------
// collapse_probabilities: channels decided in order against already-collapsed values
// ---------------------------------------------------------------------------------
__global__ void collapse_u8_kernel(const float* __restrict__ probs, uint8_t* __restrict__ out,
                                   long long pixels, int C, float value) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pixels;
       i += (long long)gridDim.x * blockDim.x) {
    float p[8], cv[8];
    for (int j = 0; j < C; ++j) p[j] = probs[i * C + j];
    for (int a = 0; a < C; ++a) {
      bool win = true;
      for (int j = 0; j < C; ++j) {
        if (j == a) continue;
        const float other = j < a ? cv[j] : p[j];
        win = win && (p[a] > other);
      }
      cv[a] = win ? value : 0.f;
    }
    for (int j = 0; j < C; ++j) out[i * C + j] = (uint8_t)cv[j];
  }
}

int launch_collapse_u8(const float* probs, uint8_t* out, long long pixels, int C, int value,
                       hipStream_t s) {
  MISEG_CHECK_ARG(probs && out, "collapse: null pointer");
  MISEG_CHECK_ARG(pixels > 0 && C >= 1 && C <= 8 && value >= 0 && value <= 255,
                  "collapse: need 1 <= C <= 8 and 0 <= value <= 255");
  hipLaunchKernelGGL(collapse_u8_kernel, dim3(grid_for(pixels)), dim3(256), 0, s, probs, out,
                     pixels, C, (float)value);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ---------------------------------------------------------------------------------
// loss / metric sums: per class sum(gt*pr), sum(pr), sum(gt), sum(gt*log(clip(pr/sum pr)))
// ---------------------------------------------------------------------------------
// FOCAL: a fifth row, sum(gt * alpha * (1 - q)^gamma * log q) with q = clip(pr) (no renormalisation:
// segmentation_models categorical_focal_loss)
template <int C, bool FOCAL>
__global__ void loss_sums_kernel(const float* __restrict__ gt, const float* __restrict__ pr,
                                 double* __restrict__ sums, long long pixels, float alpha, float gamma) {
  constexpr int R = FOCAL ? 5 : 4;
  float acc[R * C];
#pragma unroll
  for (int j = 0; j < R * C; ++j) acc[j] = 0.f;
  // per-thread fp32 partials over at most a few thousand pixels, then fp64 atomics
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pixels;
       i += (long long)gridDim.x * blockDim.x) {
    float g[C], p[C], s = 0.f;
#pragma unroll
    for (int j = 0; j < C; ++j) {
      g[j] = gt[i * C + j];
      p[j] = pr[i * C + j];
      s += p[j];
    }
    const float inv = 1.f / s;
#pragma unroll
    for (int j = 0; j < C; ++j) {
      const float q = fminf(fmaxf(p[j] * inv, 1e-7f), 1.f - 1e-7f);
      acc[j] += g[j] * p[j];
      acc[C + j] += p[j];
      acc[2 * C + j] += g[j];
      acc[3 * C + j] += g[j] * __logf(q);
      if (FOCAL) {
        const float qf = fminf(fmaxf(p[j], 1e-7f), 1.f - 1e-7f);
        acc[4 * C + j] += g[j] * alpha * __powf(1.f - qf, gamma) * __logf(qf);
      }
    }
  }
  __shared__ double red[R * C];
  if (threadIdx.x < R * C) red[threadIdx.x] = 0.0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < R * C; ++j) {
    float v = acc[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&red[j], (double)v);
  }
  __syncthreads();
  if (threadIdx.x < R * C) atomicAdd(&sums[threadIdx.x], red[threadIdx.x]);
}

template <bool FOCAL>
static int launch_loss_sums_t(const float* gt, const float* pr, double* sums, long long pixels, int C,
                              float alpha, float gamma, hipStream_t s) {
  MISEG_CHECK_ARG(gt && pr && sums, "loss_sums: null pointer");
  MISEG_CHECK_ARG(pixels > 0 && C >= 1 && C <= 8, "loss_sums: need 1 <= C <= 8");
  MISEG_CHECK_HIP(hipMemsetAsync(sums, 0, sizeof(double) * (FOCAL ? 5 : 4) * C, s));
  const dim3 grid(grid_for(pixels)), block(256);
#define MISEG_LS(CC)                                                                                 \
  case CC:                                                                                           \
    hipLaunchKernelGGL((loss_sums_kernel<CC, FOCAL>), grid, block, 0, s, gt, pr, sums, pixels, alpha, \
                       gamma);                                                                       \
    break;
  switch (C) {
    MISEG_LS(1) MISEG_LS(2) MISEG_LS(3) MISEG_LS(4) MISEG_LS(5) MISEG_LS(6) MISEG_LS(7)
    default:
      hipLaunchKernelGGL((loss_sums_kernel<8, FOCAL>), grid, block, 0, s, gt, pr, sums, pixels, alpha, gamma);
  }
#undef MISEG_LS
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

int launch_loss_sums(const float* gt, const float* pr, double* sums, long long pixels, int C,
                     hipStream_t s) {
  return launch_loss_sums_t<false>(gt, pr, sums, pixels, C, 0.f, 0.f, s);
}

int launch_loss_sums_focal(const float* gt, const float* pr, double* sums, long long pixels, int C,
                           float alpha, float gamma, hipStream_t s) {
  MISEG_CHECK_ARG(alpha > 0.f && gamma >= 0.f, "loss_sums_focal: need alpha > 0, gamma >= 0");
  return launch_loss_sums_t<true>(gt, pr, sums, pixels, C, alpha, gamma, s);
}

}  // namespace MISEG_KNS
