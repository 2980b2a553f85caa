// Bandwidth-bound kernels of the training pass on gfx950: BatchNormalization batch
// statistics and its backward, pooling / upsampling backward, channel-slice copies,
// the loss gradient at the logits and the Keras-form Adam update.  bf16 tensors are NHWC
// views with a pixel stride in elements; every thread moves 16 bytes (8 channels) per access.
#include "common.h"

namespace miseg {

// Grid of a reduction kernel: every block ends with 2*C global atomics, so blocks must be few and
// each thread should make at least ~16 trips; 1024 blocks (4 per CU) saturate HBM.
static inline int reduce_grid(long long pixels, int C8, int block = 256) {
  long long g = pixels * C8 / (block * 16);
  if (g > 1024) g = 1024;
  const int need = (C8 + block - 1) / block;     // at least one thread per channel group
  if (g < need) g = need;
  return (int)g;
}

static inline int grid_cap(long long work, int block = 256, int cap = 2048) {
  long long g = (work + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ void unpack8f(float (&m)[8], const uint4 v) {
  m[0] = bf16lo_to_f32(v.x); m[1] = bf16hi_to_f32(v.x);
  m[2] = bf16lo_to_f32(v.y); m[3] = bf16hi_to_f32(v.y);
  m[4] = bf16lo_to_f32(v.z); m[5] = bf16hi_to_f32(v.z);
  m[6] = bf16lo_to_f32(v.w); m[7] = bf16hi_to_f32(v.w);
}
__device__ __forceinline__ uint4 pack8f(const float (&m)[8]) {
  return make_uint4(pack_bf16x2(m[0], m[1]), pack_bf16x2(m[2], m[3]), pack_bf16x2(m[4], m[5]),
                    pack_bf16x2(m[6], m[7]));
}

// Work item i = pixel * C8 + c8.  The grid stride is rounded down to a multiple of C8 so a
// thread keeps one channel group for its whole trip and can accumulate in registers.
struct ChanWalk {
  long long pix0, pstep;      // first pixel of this thread and pixels between its visits
  int c8;
  bool active;
  __device__ ChanWalk(long long pixels, int C8) {
    const long long threads = (long long)gridDim.x * blockDim.x;
    pstep = threads / C8;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    active = gid < pstep * C8;
    pix0 = gid / C8;
    c8 = (int)(gid - pix0 * C8);
  }
};

// Block-level reduction of per-thread channel partials through LDS double atomics, then
// one global double atomic per channel and block.
template <int NACC>
__device__ __forceinline__ void flush_channel_sums(const float (&acc)[NACC][8], int c8, bool active,
                                                   int C, double* gsums, double* lsums) {
  for (int i = threadIdx.x; i < NACC * C; i += blockDim.x) lsums[i] = 0.0;
  __syncthreads();
  if (active) {
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int j = 0; j < 8; ++j) atomicAdd(&lsums[a * C + c8 * 8 + j], (double)acc[a][j]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NACC * C; i += blockDim.x)
    if (lsums[i] != 0.0) atomicAdd(&gsums[i], lsums[i]);
}

// ------------------------------------------------------------------ BN forward statistics
__global__ void bn_stats_kernel(const bf16_t* __restrict__ x, long long pixels, int C, int ps,
                                double* __restrict__ sums) {
  extern __shared__ double lsums[];
  const int C8 = C >> 3;
  ChanWalk w(pixels, C8);
  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = 0.f;
  if (w.active) {
    const bf16_t* xp = x + w.c8 * 8;
    long long pix = w.pix0;
    for (; pix + 3 * w.pstep < pixels; pix += 4 * w.pstep) {
      uint4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const uint4*>(xp + (pix + u * w.pstep) * ps);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[8];
        unpack8f(v, r[u]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc[0][j] += v[j];
          acc[1][j] += v[j] * v[j];
        }
      }
    }
    for (; pix < pixels; pix += w.pstep) {
      float v[8];
      unpack8f(v, *reinterpret_cast<const uint4*>(xp + pix * ps));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc[0][j] += v[j];
        acc[1][j] += v[j] * v[j];
      }
    }
  }
  flush_channel_sums<2>(acc, w.c8, w.active, C, sums, lsums);
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, long long count, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* moving_mean, float* moving_var, float momentum, float eps,
                                   float* mean, float* var, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double m = sums[c] / (double)count;
  double v = sums[C + c] / (double)count - m * m;
  if (v < 0) v = 0;
  mean[c] = (float)m;
  var[c] = (float)v;
  const float inv = 1.0f / sqrtf((float)v + eps);
  const float sc = gamma ? gamma[c] * inv : inv;
  scale[c] = sc;
  shift[c] = beta[c] - (float)m * sc;
  if (moving_mean) {
    const double unbiased = count > 1 ? v * (double)count / (double)(count - 1) : v;
    moving_mean[c] = momentum * moving_mean[c] + (1.f - momentum) * (float)m;
    moving_var[c] = momentum * moving_var[c] + (1.f - momentum) * (float)unbiased;
  }
}

// The same from the replicated sums a convolution's epilogue gathered: stats[r][2][stride], r < replicas.
// Block = 64 channels x FIN_RG replica groups: a thread adds every FIN_RG-th replica of its channel (its
// loads are independent and in flight together; one thread walking all replicas waited for 2 x replicas
// loads one after the other, 17 us for 32 replicas), the groups meet in LDS.
constexpr int FIN_RG = 8;
__global__ __launch_bounds__(64 * FIN_RG) void bn_finalize_replicas_kernel(
    const double* __restrict__ stats, int replicas, int stride, long long count, int C,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* moving_mean, float* moving_var,
    float momentum, float eps, float* mean, float* var, float* scale, float* shift) {
  __shared__ double part[FIN_RG][2][64];
  const int lc = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lc;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int r = rg; r < replicas; r += FIN_RG) {
      s1 += stats[((long long)r * 2 + 0) * stride + c];
      s2 += stats[((long long)r * 2 + 1) * stride + c];
    }
  }
  part[rg][0][lc] = s1;
  part[rg][1][lc] = s2;
  __syncthreads();
  if (rg != 0 || c >= C) return;
#pragma unroll
  for (int g = 1; g < FIN_RG; ++g) {
    s1 += part[g][0][lc];
    s2 += part[g][1][lc];
  }
  const double m = s1 / (double)count;
  double v = s2 / (double)count - m * m;
  if (v < 0) v = 0;
  mean[c] = (float)m;
  var[c] = (float)v;
  const float inv = 1.0f / sqrtf((float)v + eps);
  const float sc = gamma ? gamma[c] * inv : inv;
  scale[c] = sc;
  shift[c] = beta[c] - (float)m * sc;
  if (moving_mean) {
    const double unbiased = count > 1 ? v * (double)count / (double)(count - 1) : v;
    moving_mean[c] = momentum * moving_mean[c] + (1.f - momentum) * (float)m;
    moving_var[c] = momentum * moving_var[c] + (1.f - momentum) * (float)unbiased;
  }
}

int launch_bn_finalize(const double* stats, int replicas, int stride, long long pixels, int C,
                       const float* gamma, const float* beta, float* mm, float* mv, float momentum,
                       float eps, float* mean, float* var, float* scale, float* shift, hipStream_t s) {
  MISEG_CHECK_ARG(stats && beta && mean && var && scale && shift, "bn_finalize: null pointer");
  MISEG_CHECK_ARG(pixels > 0 && C > 0 && replicas > 1 && stride >= C, "bn_finalize: bad extents");
  MISEG_CHECK_ARG((mm == nullptr) == (mv == nullptr), "bn_finalize: moving stats go together");
  hipLaunchKernelGGL(bn_finalize_replicas_kernel, dim3((C + 63) / 64), dim3(64 * FIN_RG), 0, s, stats, replicas, stride,
                     pixels, C, gamma, beta, mm, mv, momentum, eps, mean, var, scale, shift);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

int launch_bn_stats(const void* x, long long pixels, int C, int ps, const float* gamma,
                    const float* beta, float* mm, float* mv, float momentum, float eps, float* mean,
                    float* var, float* scale, float* shift, double* ws, hipStream_t s) {
  MISEG_CHECK_ARG(x && beta && mean && var && scale && shift && ws, "bn_stats: null pointer");
  MISEG_CHECK_ARG(pixels > 0 && C > 0 && (C & 7) == 0 && (ps & 7) == 0 && ps >= C && C <= 4096,
                  "bn_stats: C must be a multiple of 8 (<= 4096), stride a multiple of 8");
  MISEG_CHECK_ARG((mm == nullptr) == (mv == nullptr), "bn_stats: moving stats go together");
  MISEG_CHECK_HIP(hipMemsetAsync(ws, 0, sizeof(double) * 2 * C, s));
  const int grid = reduce_grid(pixels, C >> 3);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(grid), dim3(256), sizeof(double) * 2 * C, s,
                     (const bf16_t*)x, pixels, C, ps, ws);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, ws, pixels, C, gamma,
                     beta, mm, mv, momentum, eps, mean, var, scale, shift);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ---- the same without a workspace argument: the double[2*C] partial sums live in a file-scope
// device array, for callers that cannot allocate (HIP graph capture, serving loops).  Each GPU
// holds its own copy of the array at its own address; g_bn_scratch_ptr[dev] is that address,
// looked up once per device.  In host code `g_bn_scratch` itself is only a handle for
// hipGetSymbolAddress, never a device address.
constexpr int kNoAllocMaxC = 4096;
constexpr int kNoAllocMaxDevices = 64;
static __device__ __align__(256) double g_bn_scratch[2 * kNoAllocMaxC];   // 64 KB per GPU
static double* g_bn_scratch_ptr[kNoAllocMaxDevices] = {};

// Address of this device's scratch array (looked up on the first call per device; idempotent, and a
// race between two first calls stores the same value twice).
int noalloc_setup(double** out) {
  int dev = -1;
  MISEG_CHECK_HIP(hipGetDevice(&dev));
  MISEG_CHECK_ARG(dev >= 0 && dev < kNoAllocMaxDevices, "no-alloc scratch: device index %d out of range", dev);
  double* ptr = __atomic_load_n(&g_bn_scratch_ptr[dev], __ATOMIC_ACQUIRE);
  if (!ptr) {
    MISEG_CHECK_HIP(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(g_bn_scratch)));
    MISEG_CHECK_ARG(ptr != nullptr, "no-alloc scratch: hipGetSymbolAddress returned a null address");
    __atomic_store_n(&g_bn_scratch_ptr[dev], ptr, __ATOMIC_RELEASE);
  }
  if (out) *out = ptr;
  return MISEG_OK;
}

int launch_bn_stats_noalloc(const void* x, long long pixels, int C, int ps, const float* gamma,
                            const float* beta, float* mm, float* mv, float momentum, float eps,
                            float* mean, float* var, float* scale, float* shift, hipStream_t s) {
  MISEG_CHECK_ARG(C > 0 && C <= kNoAllocMaxC,
                  "bn_stats_noalloc: C = %d exceeds the built-in scratch (at most %d channels)", C, kNoAllocMaxC);
  double* ws = nullptr;
  const int rc = noalloc_setup(&ws);
  if (rc != MISEG_OK) return rc;
  // launch_bn_stats clears ws[0, 2*C) on the stream before the reduction, so whatever the previous
  // call left behind (possibly for more channels) does not show
  return launch_bn_stats(x, pixels, C, ps, gamma, beta, mm, mv, momentum, eps, mean, var, scale, shift, ws, s);
}

// fp32 3-channel network input
// Per-channel sum and sum of squares of the fp32 (N,H,W,3) input.  A thread takes groups of four pixels =
// three 16-byte loads (consecutive threads on consecutive 48-byte groups); a block adds its six sums in LDS
// and issues six atomics (one atomic per wave and value on the same six addresses serialised the kernel:
// 0.79 ms for 0.94 GB).
__global__ __launch_bounds__(256) void input_stats_kernel(const float* __restrict__ x, long long pixels,
                                                          double* __restrict__ sums) {
  __shared__ double part[4][6];
  float acc[6] = {0, 0, 0, 0, 0, 0};
  const long long groups = pixels >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < groups;
       g += (long long)gridDim.x * blockDim.x) {
    const float4 a = x4[3 * g], b = x4[3 * g + 1], c = x4[3 * g + 2];
    const float v[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      acc[j % 3] += v[j];
      acc[3 + j % 3] += v[j] * v[j];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(pixels & 3)) {      // the last pixels & 3 pixels
    const long long i = (groups << 2) + threadIdx.x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = x[i * 3 + c];
      acc[c] += v;
      acc[3 + c] += v * v;
    }
  }
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float v = acc[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) part[wave][j] = (double)v;
  }
  __syncthreads();
  if (threadIdx.x < 6) atomicAdd(&sums[threadIdx.x], part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

int launch_input_stats(const float* x, long long pixels, const float* beta, float* mm, float* mv,
                       float momentum, float eps, float* mean, float* var, float* scale,
                       float* shift, double* ws, hipStream_t s) {
  MISEG_CHECK_ARG(x && beta && mean && var && scale && shift && ws, "input_stats: null pointer");
  MISEG_CHECK_ARG(pixels > 0, "input_stats: no pixels");
  MISEG_CHECK_HIP(hipMemsetAsync(ws, 0, sizeof(double) * 6, s));
  MISEG_CHECK_ARG(((uintptr_t)x & 15) == 0, "input_stats: the input must be 16-byte aligned");
  const long long want = ((pixels >> 2) + 255) / 256;
  hipLaunchKernelGGL(input_stats_kernel, dim3((unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want))), dim3(256), 0, s, x,
                     pixels, ws);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(64), 0, s, ws, pixels, 3, (const float*)nullptr,
                     beta, mm, mv, momentum, eps, mean, var, scale, shift);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ------------------------------------------------------------------ BN + ReLU backward
__global__ void bn_relu_bwd_reduce_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                          long long pixels, int C, int xps, int dps,
                                          const float* __restrict__ scale,
                                          const float* __restrict__ shift,
                                          const float* __restrict__ mean,
                                          const float* __restrict__ var, float eps, int relu,
                                          double* __restrict__ sums) {
  extern __shared__ double lsums[];
  const int C8 = C >> 3;
  ChanWalk w(pixels, C8);
  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = 0.f;
  if (w.active) {
    float sc[8], sh[8], mu[8], inv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = w.c8 * 8 + j;
      sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c];
      inv[j] = 1.0f / sqrtf(var[c] + eps);
    }
    const bf16_t* xp = x + w.c8 * 8;
    const bf16_t* dp = dy + w.c8 * 8;
    long long pix = w.pix0;
    for (; pix + w.pstep < pixels; pix += 2 * w.pstep) {
      uint4 rx[2], rg[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        rx[u] = *reinterpret_cast<const uint4*>(xp + (pix + u * w.pstep) * xps);
        rg[u] = *reinterpret_cast<const uint4*>(dp + (pix + u * w.pstep) * dps);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float xv[8], g[8];
        unpack8f(xv, rx[u]);
        unpack8f(g, rg[u]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float gg = (relu && xv[j] * sc[j] + sh[j] <= 0.f) ? 0.f : g[j];
          acc[0][j] += gg;
          acc[1][j] += gg * (xv[j] - mu[j]) * inv[j];
        }
      }
    }
    for (; pix < pixels; pix += w.pstep) {
      float xv[8], g[8];
      unpack8f(xv, *reinterpret_cast<const uint4*>(xp + pix * xps));
      unpack8f(g, *reinterpret_cast<const uint4*>(dp + pix * dps));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gg = (relu && xv[j] * sc[j] + sh[j] <= 0.f) ? 0.f : g[j];
        acc[0][j] += gg;
        acc[1][j] += gg * (xv[j] - mu[j]) * inv[j];
      }
    }
  }
  flush_channel_sums<2>(acc, w.c8, w.active, C, sums, lsums);
}

__global__ void bn_relu_bwd_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                         bf16_t* __restrict__ dx, long long pixels, int C, int xps,
                                         int dps, int ops, const float* __restrict__ scale,
                                         const float* __restrict__ shift,
                                         const float* __restrict__ mean,
                                         const float* __restrict__ var, float eps, int relu,
                                         const double* __restrict__ sums,
                                         const bf16_t* __restrict__ addend, int aps) {
  const int C8 = C >> 3;
  ChanWalk w(pixels, C8);
  if (!w.active) return;
  float sc[8], sh[8], mu[8], inv[8], k1[8], k2[8];
  const float rcount = 1.0f / (float)pixels;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = w.c8 * 8 + j;
    sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c];
    inv[j] = 1.0f / sqrtf(var[c] + eps);
    k1[j] = (float)sums[c] * rcount;
    k2[j] = (float)sums[C + c] * rcount;
  }
  for (long long pix = w.pix0; pix < pixels; pix += w.pstep) {
    float xv[8], g[8], o[8];
    unpack8f(xv, *reinterpret_cast<const uint4*>(x + pix * xps + w.c8 * 8));
    unpack8f(g, *reinterpret_cast<const uint4*>(dy + pix * dps + w.c8 * 8));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gg = (relu && xv[j] * sc[j] + sh[j] <= 0.f) ? 0.f : g[j];
      const float xhat = (xv[j] - mu[j]) * inv[j];
      o[j] = sc[j] * (gg - k1[j] - xhat * k2[j]);
    }
    if (addend) {
      float a[8];
      unpack8f(a, *reinterpret_cast<const uint4*>(addend + pix * aps + w.c8 * 8));
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += a[j];
    }
    *reinterpret_cast<uint4*>(dx + pix * ops + w.c8 * 8) = pack8f(o);
  }
}

// The two reductions of the backward pass from what a data-gradient epilogue gathered
// (MISEG_RES_BN_RELU_MASK): stats[r][0][c] = sum(g), stats[r][1][c] = sum(g * x) over the block's pixels, g
// the gradient behind the ReLU.  sums[c] = sum(g) (the beta gradient), sums[C + c] = sum(g * xhat) =
// (sum(g * x) - mean * sum(g)) / sqrt(var + eps) (the gamma gradient), in fp64.
__global__ __launch_bounds__(64 * FIN_RG) void bn_bwd_sums_kernel(const double* __restrict__ stats, int replicas,
                                                                  int stride, int C, const float* __restrict__ mean,
                                                                  const float* __restrict__ var, float eps,
                                                                  double* __restrict__ sums) {
  __shared__ double part[FIN_RG][2][64];
  const int lc = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lc;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int r = rg; r < replicas; r += FIN_RG) {
      s1 += stats[((long long)r * 2 + 0) * stride + c];
      s2 += stats[((long long)r * 2 + 1) * stride + c];
    }
  }
  part[rg][0][lc] = s1;
  part[rg][1][lc] = s2;
  __syncthreads();
  if (rg != 0 || c >= C) return;
#pragma unroll
  for (int i = 1; i < FIN_RG; ++i) {
    s1 += part[i][0][lc];
    s2 += part[i][1][lc];
  }
  sums[c] = s1;
  sums[C + c] = (s2 - (double)mean[c] * s1) / sqrt((double)var[c] + (double)eps);
}

int launch_bn_bwd_sums(const double* stats, int replicas, int stride, int C, const float* mean, const float* var,
                       float eps, double* sums, hipStream_t s) {
  MISEG_CHECK_ARG(stats && mean && var && sums, "bn_bwd_sums: null pointer");
  MISEG_CHECK_ARG(replicas > 0 && C > 0 && stride >= C, "bn_bwd_sums: needs replicas > 0 and 0 < C <= stride");
  hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3((C + 63) / 64), dim3(64 * FIN_RG), 0, s, stats, replicas, stride, C,
                     mean, var, eps, sums);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

static int check_bn_bwd(const void* x, const void* dy, long long pixels, int C, int xps, int dps) {
  MISEG_CHECK_ARG(x && dy, "bn_relu_bwd: null pointer");
  MISEG_CHECK_ARG(pixels > 0 && C > 0 && (C & 7) == 0 && (xps & 7) == 0 && (dps & 7) == 0 &&
                      C <= 4096,
                  "bn_relu_bwd: C must be a multiple of 8 (<= 4096), strides multiples of 8");
  return MISEG_OK;
}

int launch_bn_relu_bwd_reduce(const void* x, const void* dy, long long pixels, int C, int xps,
                              int dps, const float* scale, const float* shift, const float* mean,
                              const float* var, float eps, int relu, double* sums, hipStream_t s) {
  int rc = check_bn_bwd(x, dy, pixels, C, xps, dps);
  if (rc) return rc;
  MISEG_CHECK_ARG(scale && shift && mean && var && sums, "bn_relu_bwd_reduce: null pointer");
  MISEG_CHECK_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, s));
  hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3(reduce_grid(pixels, C >> 3)), dim3(256),
                     sizeof(double) * 2 * C, s, (const bf16_t*)x, (const bf16_t*)dy, pixels, C, xps,
                     dps, scale, shift, mean, var, eps, relu, sums);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

int launch_bn_relu_bwd_apply(const void* x, const void* dy, void* dx, long long pixels, int C,
                             int xps, int dps, int ops, const float* scale, const float* shift,
                             const float* mean, const float* var, float eps, int relu,
                             const double* sums, const void* addend, int aps, hipStream_t s) {
  int rc = check_bn_bwd(x, dy, pixels, C, xps, dps);
  if (rc) return rc;
  MISEG_CHECK_ARG(dx && scale && shift && mean && var && sums && (ops & 7) == 0 &&
                      (!addend || ((aps & 7) == 0 && aps >= C)),
                  "bn_relu_bwd_apply: null pointer or bad stride");
  hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(grid_cap(pixels * (C >> 3), 256, 4096)), dim3(256),
                     0, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, pixels, C, xps, dps, ops,
                     scale, shift, mean, var, eps, relu, sums, (const bf16_t*)addend, aps);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ------------------------------------------------------------------ channel sums (bias grads)
__global__ void channel_sums_kernel(const bf16_t* __restrict__ x, long long pixels, int C, int ps,
                                    double* __restrict__ sums) {
  extern __shared__ double lsums[];
  const int C8 = C >> 3;
  ChanWalk w(pixels, C8);
  float acc[1][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = 0.f;
  if (w.active) {
    for (long long pix = w.pix0; pix < pixels; pix += w.pstep) {
      float v[8];
      unpack8f(v, *reinterpret_cast<const uint4*>(x + pix * ps + w.c8 * 8));
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[0][j] += v[j];
    }
  }
  flush_channel_sums<1>(acc, w.c8, w.active, C, sums, lsums);
}

int launch_channel_sums(const void* x, long long pixels, int C, int ps, double* sums, hipStream_t s) {
  MISEG_CHECK_ARG(x && sums && pixels > 0 && C > 0 && (C & 7) == 0 && (ps & 7) == 0 && C <= 4096,
                  "channel_sums: bad argument");
  MISEG_CHECK_HIP(hipMemsetAsync(sums, 0, sizeof(double) * C, s));
  hipLaunchKernelGGL(channel_sums_kernel, dim3(reduce_grid(pixels, C >> 3)), dim3(256),
                     sizeof(double) * C, s, (const bf16_t*)x, pixels, C, ps, sums);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ------------------------------------------------------------------ pooling / upsampling backward
// One thread owns a 2x2 block of input pixels (rows 2a, 2a+1; columns 2b, 2b+1) and 8 channels.
// The windows that can route a gradient into the block are (a..a+1) x (b..b+1); together they
// cover the 5x5 patch of rows 2a-1..2a+3, which is loaded once.
// addend (may be null): a gradient the pool's input already holds, added to the result.
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                        bf16_t* __restrict__ dx, int N, int H, int W, int C, int xps,
                                        int dps, int ops, const bf16_t* __restrict__ addend, int aps) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, C8 = C >> 3;
  const int Hb = (H + 1) / 2, Wb = (W + 1) / 2;
  const long long total = (long long)N * Hb * Wb * C8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    long long r = i / C8;
    const int bx = (int)(r % Wb);
    r /= Wb;
    const int by = (int)(r % Hb);
    const int n = (int)(r / Hb);
    const bf16_t* xb = x + (long long)n * H * W * xps + c8 * 8;
    uint4 patch[5][5];                                            // bf16x8 per pixel
#pragma unroll
    for (int py = 0; py < 5; ++py) {
      const int y = 2 * by - 1 + py;
#pragma unroll
      for (int px = 0; px < 5; ++px) {
        const int xx = 2 * bx - 1 + px;
        patch[py][px] = ((unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W)
                            ? *reinterpret_cast<const uint4*>(xb + ((long long)y * W + xx) * xps)
                            : make_uint4(0, 0, 0, 0);             // ZeroPadding2D
      }
    }
    float acc[2][2][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[q >> 1][q & 1][j] = 0.f;
#pragma unroll
    for (int wy = 0; wy < 2; ++wy) {
      const int oy = by + wy;
#pragma unroll
      for (int wx = 0; wx < 2; ++wx) {
        const int ox = bx + wx;
        const bool live = oy < Ho && ox < Wo;
        float g[8];
        unpack8f(g, live ? *reinterpret_cast<const uint4*>(
                               dy + (((long long)n * Ho + oy) * Wo + ox) * dps + c8 * 8)
                         : make_uint4(0, 0, 0, 0));
        // first maximum of the window in row-major order
        float best[8];
        int bidx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { best[j] = -3.0e38f; bidx[j] = -1; }
#pragma unroll
        for (int dy_ = 0; dy_ < 3; ++dy_)
#pragma unroll
          for (int dx_ = 0; dx_ < 3; ++dx_) {
            float v[8];
            unpack8f(v, patch[2 * wy + dy_][2 * wx + dx_]);
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (v[j] > best[j]) { best[j] = v[j]; bidx[j] = dy_ * 3 + dx_; }
          }
        // block pixel (qy, qx) sits at patch (1+qy, 1+qx) = window position (1+qy-2wy, 1+qx-2wx)
#pragma unroll
        for (int qy = 0; qy < 2; ++qy) {
          const int ry = 1 + qy - 2 * wy;
#pragma unroll
          for (int qx = 0; qx < 2; ++qx) {
            const int rx = 1 + qx - 2 * wx;
            if (ry >= 0 && ry <= 2 && rx >= 0 && rx <= 2) {
#pragma unroll
              for (int j = 0; j < 8; ++j)
                if (bidx[j] == ry * 3 + rx) acc[qy][qx][j] += g[j];
            }
          }
        }
      }
    }
#pragma unroll
    for (int qy = 0; qy < 2; ++qy)
#pragma unroll
      for (int qx = 0; qx < 2; ++qx) {
        const int iy = 2 * by + qy, ix = 2 * bx + qx;
        if (iy < H && ix < W) {
          if (addend) {
            float a[8];
            unpack8f(a, *reinterpret_cast<const uint4*>(addend + (((long long)n * H + iy) * W + ix) * aps + c8 * 8));
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[qy][qx][j] += a[j];
          }
          *reinterpret_cast<uint4*>(dx + (((long long)n * H + iy) * W + ix) * ops + c8 * 8) = pack8f(acc[qy][qx]);
        }
      }
  }
}

int launch_maxpool3x3s2_bwd(const void* x, const void* dy, void* dx, int N, int H, int W, int C,
                            int xps, int dps, int ops, const void* addend, int aps, hipStream_t s) {
  MISEG_CHECK_ARG(x && dy && dx, "maxpool_bwd: null pointer");
  MISEG_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0 && (xps & 7) == 0 &&
                      (dps & 7) == 0 && (ops & 7) == 0 && (!addend || (aps & 7) == 0),
                  "maxpool_bwd: C and strides must be multiples of 8");
  const long long total = (long long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C >> 3);
  hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(grid_cap(total, 256, 16384)), dim3(256), 0, s,
                     (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, xps, dps, ops,
                     (const bf16_t*)addend, aps);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ------------------------------------------------------------------ max pool 2x2 backward
// One thread per output window and 8 channels: the gradient goes to the first maximum of the window in
// row-major order, the other three inputs get zero.
__global__ void maxpool2x2_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                      bf16_t* __restrict__ dx, int N, int H, int W, int C, int xps,
                                      int dps, int ops) {
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
    float g[8], best[8];
    int bidx[8];
    unpack8f(g, *reinterpret_cast<const uint4*>(dy + (((long long)n * Ho + oy) * Wo + ox) * dps + c8 * 8));
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -3.0e38f; bidx[j] = -1; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[8];
      unpack8f(v, *reinterpret_cast<const uint4*>(
                      x + (((long long)n * H + 2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * xps + c8 * 8));
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (v[j] > best[j]) { best[j] = v[j]; bidx[j] = q; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = bidx[j] == q ? g[j] : 0.f;
      *reinterpret_cast<uint4*>(dx + (((long long)n * H + 2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * ops +
                                c8 * 8) = pack8f(o);
    }
  }
}

int launch_maxpool2x2_bwd(const void* x, const void* dy, void* dx, int N, int H, int W, int C, int xps,
                          int dps, int ops, hipStream_t s) {
  MISEG_CHECK_ARG(x && dy && dx && N > 0 && H > 1 && W > 1 && (H & 1) == 0 && (W & 1) == 0 && C > 0 &&
                      (C & 7) == 0 && (xps & 7) == 0 && (dps & 7) == 0 && (ops & 7) == 0,
                  "maxpool2x2_bwd: H and W must be even, C and strides multiples of 8");
  const long long total = (long long)N * (H >> 1) * (W >> 1) * (C >> 3);
  hipLaunchKernelGGL(maxpool2x2_bwd_kernel, dim3(grid_cap(total, 256, 8192)), dim3(256), 0, s,
                     (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, xps, dps, ops);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ------------------------------------------------------------------ ReLU backward (no BatchNorm in between)
// dx = (y > 0 ? dy : 0) [+ addend]; y is the ReLU's output.
__global__ void relu_bwd_kernel(const bf16_t* __restrict__ y, const bf16_t* __restrict__ dy,
                                const bf16_t* __restrict__ add, bf16_t* __restrict__ dx, long long pixels,
                                int C, int yps, int dps, int aps, int ops) {
  const int C8 = C >> 3;
  ChanWalk w(pixels, C8);
  if (!w.active) return;
  const int c8 = w.c8;
  for (long long pix = w.pix0; pix < pixels; pix += w.pstep) {
    float a[8], g[8];
    unpack8f(a, *reinterpret_cast<const uint4*>(y + pix * yps + c8 * 8));
    unpack8f(g, *reinterpret_cast<const uint4*>(dy + pix * dps + c8 * 8));
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = a[j] > 0.f ? g[j] : 0.f;
    if (add) {
      float b[8];
      unpack8f(b, *reinterpret_cast<const uint4*>(add + pix * aps + c8 * 8));
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] += b[j];
    }
    *reinterpret_cast<uint4*>(dx + pix * ops + c8 * 8) = pack8f(g);
  }
}

int launch_relu_bwd(const void* y, const void* dy, const void* add, void* dx, long long pixels, int C,
                    int yps, int dps, int aps, int ops, hipStream_t s) {
  MISEG_CHECK_ARG(y && dy && dx && pixels > 0 && C > 0 && (C & 7) == 0 && (yps & 7) == 0 &&
                      (dps & 7) == 0 && (ops & 7) == 0 && (!add || (aps & 7) == 0),
                  "relu_bwd: bad argument");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_cap(pixels * (C >> 3), 256, 8192)), dim3(256), 0, s,
                     (const bf16_t*)y, (const bf16_t*)dy, (const bf16_t*)add, (bf16_t*)dx, pixels, C, yps,
                     dps, aps, ops);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

__global__ void avgpool2x2_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int N,
                                      int H, int W, int C, int dps, int ops) {
  const int Ho = H >> 1, Wo = W >> 1, C8 = C >> 3;
  const long long total = (long long)N * H * W * C8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    long long r = i / C8;
    const int ix = (int)(r % W);
    r /= W;
    const int iy = (int)(r % H);
    const int n = (int)(r / H);
    float g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((iy >> 1) < Ho && (ix >> 1) < Wo) {
      unpack8f(g, *reinterpret_cast<const uint4*>(
                      dy + (((long long)n * Ho + (iy >> 1)) * Wo + (ix >> 1)) * dps + c8 * 8));
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] *= 0.25f;
    }
    *reinterpret_cast<uint4*>(dx + (((long long)n * H + iy) * W + ix) * ops + c8 * 8) = pack8f(g);
  }
}

int launch_avgpool2x2_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dps, int ops,
                          hipStream_t s) {
  MISEG_CHECK_ARG(dy && dx && N > 0 && H > 1 && W > 1 && C > 0 && (C & 7) == 0 && (dps & 7) == 0 &&
                      (ops & 7) == 0,
                  "avgpool_bwd: bad argument");
  const long long total = (long long)N * H * W * (C >> 3);
  hipLaunchKernelGGL(avgpool2x2_bwd_kernel, dim3(grid_cap(total, 256, 8192)), dim3(256), 0, s,
                     (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, dps, ops);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

__global__ void upsample_nearest_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx,
                                            int N, int H, int W, int C, int shift, int dps, int ops) {
  const int s = 1 << shift, Wo = W << shift, Ho = H << shift, C8 = C >> 3;
  const long long total = (long long)N * H * W * C8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    long long r = i / C8;
    const int ix = (int)(r % W);
    r /= W;
    const int iy = (int)(r % H);
    const int n = (int)(r / H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = 0; a < s; ++a)
      for (int b = 0; b < s; ++b) {
        float g[8];
        unpack8f(g, *reinterpret_cast<const uint4*>(
                        dy + (((long long)n * Ho + (iy * s + a)) * Wo + (ix * s + b)) * dps + c8 * 8));
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += g[j];
      }
    *reinterpret_cast<uint4*>(dx + (((long long)n * H + iy) * W + ix) * ops + c8 * 8) = pack8f(acc);
  }
}

int launch_upsample_nearest_bwd(const void* dy, void* dx, int N, int H, int W, int C, int shift,
                                int dps, int ops, hipStream_t s) {
  MISEG_CHECK_ARG(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0 && (dps & 7) == 0 &&
                      (ops & 7) == 0 && shift >= 0 && shift <= 4,
                  "upsample_nearest_bwd: bad argument");
  const long long total = (long long)N * H * W * (C >> 3);
  hipLaunchKernelGGL(upsample_nearest_bwd_kernel, dim3(grid_cap(total, 256, 8192)), dim3(256), 0, s,
                     (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, shift, dps, ops);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// forward coefficients of the x2 half-pixel bilinear resize along one axis
__device__ __forceinline__ void bil_coef(int o, int n, int& i0, int& i1, float& f) {
  float s = fminf(fmaxf((o + 0.5f) * 0.5f - 0.5f, 0.f), (float)(n - 1));
  i0 = (int)s;
  i1 = min(i0 + 1, n - 1);
  f = s - i0;
}

__global__ void upsample_bilinear2x_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx,
                                               int N, int H, int W, int C, int dps, int ops) {
  const int Ho = 2 * H, Wo = 2 * W, C8 = C >> 3;
  const long long total = (long long)N * H * W * C8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    long long r = i / C8;
    const int ix = (int)(r % W);
    r /= W;
    const int iy = (int)(r % H);
    const int n = (int)(r / H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int oy = 2 * iy - 1; oy <= 2 * iy + 2; ++oy) {
      if ((unsigned)oy >= (unsigned)Ho) continue;
      int y0, y1;
      float fy;
      bil_coef(oy, H, y0, y1, fy);
      const float wy = (y0 == iy ? 1.f - fy : 0.f) + (y1 == iy ? fy : 0.f);
      if (wy == 0.f) continue;
      for (int ox = 2 * ix - 1; ox <= 2 * ix + 2; ++ox) {
        if ((unsigned)ox >= (unsigned)Wo) continue;
        int x0, x1;
        float fx;
        bil_coef(ox, W, x0, x1, fx);
        const float wx = (x0 == ix ? 1.f - fx : 0.f) + (x1 == ix ? fx : 0.f);
        if (wx == 0.f) continue;
        float g[8];
        unpack8f(g, *reinterpret_cast<const uint4*>(
                        dy + (((long long)n * Ho + oy) * Wo + ox) * dps + c8 * 8));
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += wy * wx * g[j];
      }
    }
    *reinterpret_cast<uint4*>(dx + (((long long)n * H + iy) * W + ix) * ops + c8 * 8) = pack8f(acc);
  }
}

int launch_upsample_bilinear2x_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dps,
                                   int ops, hipStream_t s) {
  MISEG_CHECK_ARG(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0 && (dps & 7) == 0 &&
                      (ops & 7) == 0,
                  "upsample_bilinear2x_bwd: bad argument");
  const long long total = (long long)N * H * W * (C >> 3);
  hipLaunchKernelGGL(upsample_bilinear2x_bwd_kernel, dim3(grid_cap(total, 256, 8192)), dim3(256), 0, s,
                     (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, dps, ops);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ------------------------------------------------------------------ channel-slice copy / add
__global__ void copy_channels_kernel(const bf16_t* __restrict__ in, const bf16_t* __restrict__ add,
                                     bf16_t* __restrict__ out, long long pixels, int C, int ips,
                                     int aps, int ops) {
  const int C8 = C >> 3;
  ChanWalk w(pixels, C8);
  if (!w.active) return;
  const int c8 = w.c8;
  for (long long pix = w.pix0; pix < pixels; pix += w.pstep) {
    uint4 v = *reinterpret_cast<const uint4*>(in + pix * ips + c8 * 8);
    if (add) {
      float a[8], b[8];
      unpack8f(a, v);
      unpack8f(b, *reinterpret_cast<const uint4*>(add + pix * aps + c8 * 8));
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += b[j];
      v = pack8f(a);
    }
    *reinterpret_cast<uint4*>(out + pix * ops + c8 * 8) = v;
  }
}

int launch_copy_channels(const void* in, const void* add, void* out, long long pixels, int C, int ips,
                         int aps, int ops, hipStream_t s) {
  MISEG_CHECK_ARG(in && out && pixels > 0 && C > 0 && (C & 7) == 0 && (ips & 7) == 0 &&
                      (ops & 7) == 0 && (!add || (aps & 7) == 0),
                  "copy_channels: bad argument");
  hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_cap(pixels * (C >> 3), 256, 8192)), dim3(256), 0,
                     s, (const bf16_t*)in, (const bf16_t*)add, (bf16_t*)out, pixels, C, ips, aps, ops);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// ------------------------------------------------------------------ loss gradient at the logits
template <int C>
__global__ void loss_grad_kernel(const float* __restrict__ gt, const float* __restrict__ pr,
                                 const double* __restrict__ sums, bf16_t* __restrict__ dz,
                                 long long pixels, int out_stride, float loss_scale,
                                 float inv_temperature, int use_cce, int use_dice, int use_focal,
                                 float alpha, float gamma) {
  // Dice: score_c = (2 tp_c + s) / (sum gt_c + sum pr_c + s);  d(1 - mean score)/d pr_{p,c}
  //       = -(2 gt_{p,c} den_c - num_c) / den_c^2 / C
  float dnum[C], dden[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const double num = 2.0 * sums[c] + 1e-5;
    const double den = sums[C + c] + sums[2 * C + c] + 1e-5;
    dnum[c] = (float)(num / (den * den) / C);
    dden[c] = (float)(2.0 / den / C);
  }
  const float cce_w = 1.0f / ((float)pixels * C);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pixels;
       i += (long long)gridDim.x * blockDim.x) {
    float g[C], p[C], s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      p[c] = pr[i * C + c];
      s += p[c];
    }
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float t = gt[i * C + c];
      float gc = 0.f;
      if (use_cce) {
        const float q = p[c] / s;
        if (q > 1e-7f && q < 1.f - 1e-7f) gc += -t * cce_w / p[c];
      }
      if (use_dice) gc += dnum[c] - t * dden[c];
      if (use_focal && t != 0.f && p[c] > 1e-7f && p[c] < 1.f - 1e-7f) {
        // f = -t alpha (1-p)^gamma log p / (pixels C)
        const float om = 1.f - p[c];
        gc += -t * alpha * cce_w * (__powf(om, gamma) / p[c] - gamma * __powf(om, gamma - 1.f) * __logf(p[c]));
      }
      g[c] = gc * loss_scale;
      dot += g[c] * p[c];
    }
    // (C == 1, the sigmoid head: dz = g p (1 - p) = p (g - dot) with dot = g p, the same expression)
    if ((out_stride & 7) == 0) {
      // whole line in 16-byte stores: the C gradients, then zeros up to out_stride
      float o[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = c < C ? p[c < C ? c : 0] * (g[c < C ? c : 0] - dot) * inv_temperature : 0.f;
      uint4* line = reinterpret_cast<uint4*>(dz + i * out_stride);
      line[0] = pack8f(o);
      for (int k = 1; k < (out_stride >> 3); ++k) line[k] = make_uint4(0, 0, 0, 0);
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c)
        dz[i * out_stride + c] = f32_to_bf16(p[c] * (g[c] - dot) * inv_temperature);
    }
  }
}

int launch_loss_grad_ex(const float* gt, const float* pr, const double* sums, void* dz, long long pixels,
                        int C, int out_stride, float loss_scale, float temperature, int flags, float alpha,
                        float gamma, hipStream_t s) {
  MISEG_CHECK_ARG(gt && pr && sums && dz, "loss_grad: null pointer");
  MISEG_CHECK_ARG(pixels > 0 && C >= 1 && C <= 8 && out_stride >= C && temperature > 0.f,
                  "loss_grad: need 1 <= C <= 8, out_stride >= C, temperature > 0");
  MISEG_CHECK_ARG((flags & 7) != 0 && (flags & ~7) == 0, "loss_grad: flags select cce (1), dice (2), focal (4)");
  MISEG_CHECK_ARG(!(flags & 4) || (alpha > 0.f && gamma >= 1.f), "loss_grad: focal needs alpha > 0, gamma >= 1");
  const dim3 grid(grid_cap(pixels, 256, 4096)), block(256);
  const float it = C == 1 ? 1.f : 1.f / temperature;
  bf16_t* o = (bf16_t*)dz;
  const int cce = flags & 1, dice = (flags >> 1) & 1, focal = (flags >> 2) & 1;
#define MISEG_LG(CC)                                                                             \
  case CC:                                                                                       \
    hipLaunchKernelGGL(loss_grad_kernel<CC>, grid, block, 0, s, gt, pr, sums, o, pixels,         \
                       out_stride, loss_scale, it, cce, dice, focal, alpha, gamma);              \
    break;
  switch (C) {
    MISEG_LG(1) MISEG_LG(2) MISEG_LG(3) MISEG_LG(4) MISEG_LG(5) MISEG_LG(6) MISEG_LG(7)
    default:
      hipLaunchKernelGGL(loss_grad_kernel<8>, grid, block, 0, s, gt, pr, sums, o, pixels, out_stride,
                         loss_scale, it, cce, dice, focal, alpha, gamma);
  }
#undef MISEG_LG
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

int launch_loss_grad(const float* gt, const float* pr, const double* sums, void* dz, long long pixels,
                     int C, int out_stride, float loss_scale, float temperature, int use_cce,
                     int use_dice, hipStream_t s) {
  MISEG_CHECK_ARG(C >= 2, "loss_grad: need 2 <= C <= 8 (miseg_loss_grad_ex serves the sigmoid head)");
  return launch_loss_grad_ex(gt, pr, sums, dz, pixels, C, out_stride, loss_scale, temperature,
                             (use_cce ? 1 : 0) | (use_dice ? 2 : 0), 0.25f, 2.f, s);
}

// ------------------------------------------------------------------ Adam (Keras form)
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ g, bf16_t* __restrict__ pb, long long count,
                            float lr_t, float b1, float b2, float eps, float gscale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float pi = p[i] - lr_t * mi / (sqrtf(vi) + eps);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
    if (pb) pb[i] = f32_to_bf16(pi);
  }
}

int launch_adam(float* p, float* m, float* v, const float* g, void* pb, long long count, float lr_t,
                float b1, float b2, float eps, float gscale, hipStream_t s) {
  MISEG_CHECK_ARG(p && m && v && g && count > 0, "adam: bad argument");
  hipLaunchKernelGGL(adam_kernel, dim3(grid_cap(count, 256, 4096)), dim3(256), 0, s, p, m, v, g,
                     (bf16_t*)pb, count, lr_t, b1, b2, eps, gscale);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

}  // namespace miseg
