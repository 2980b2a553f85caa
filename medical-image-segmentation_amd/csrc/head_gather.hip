// The segmentation head of the FPN decoder after its channel contraction has been done at half resolution
// (gfx950).  final_upsampling (bilinear x2) followed by head_conv (3x3, 128 -> classes) is linear in the
// 128-channel tensor, so the nine taps' channel contractions y[tap][class] = W[tap][:, class] . x can run
// where x lives, at half resolution, as one 1x1 convolution with 9 * classes output channels; what is left
// per output pixel is, for every tap whose position lies inside the image (the 3x3 convolution's zero
// padding), the bilinear sample of that tap's `classes` values at the tap's position, their sum plus
// bias, the activation and the collapse.  This kernel does that last step: fp32 in, fp32 out.
#include "common.h"

namespace miseg {

// y: fp32 [N][h][w][ypix], channel tap * C + c (tap = ky * 3 + kx), or c * 9 + tap when CM (class-major:
// the rows of a [class][3][3][cin] weight tensor taken as they lie).  Output extents 2h x 2w.
// Bilinear source of destination index q (tf.image.resize, half-pixel centres): s = (q + 0.5) / 2 - 0.5
// clamped to [0, n - 1]; rows floor(s) and min(floor(s) + 1, n - 1) with weights 1 - f, f.
//
// One thread finishes the 2x2 output pixels of one half-resolution cell (Y, X).  Their nine taps sit on
// output rows 2Y-1 .. 2Y+2, whose bilinear sources are the cell rows Y-1 .. Y+1 (clamped): a 3x3 cell
// neighbourhood.  The weight of neighbourhood row ni in output row 2Y-1+qi is zero for a row outside the
// image (the convolution's zero padding); the thread walks the nine cells once and adds each cell's
// 9 * C values to the four pixels' logits with weight wy[a + ty][ni] * wx[b + tx][nj].
//
// A block owns a tile of 32 x 8 cells.  Read straight from memory, a wave's 16-byte loads would touch 64
// cache lines each (neighbouring cells are ypix floats apart) and every line would be fetched by seven or
// eight separate instructions, which is what bounds the kernel; instead the block copies its tile plus a
// one-cell halo into LDS with consecutive lanes on consecutive 16-byte pieces, and the nine-cell walk
// reads LDS.  The LDS cell stride NV * 4 floats is 4 mod 8, so sixteen lanes' 16-byte reads fall on
// different banks.
constexpr int HG_TW = 32, HG_TH = 8;

template <int C, bool CM>
__global__ __launch_bounds__(256) void head_gather_kernel(const float* __restrict__ y, const float* __restrict__ bias,
                                                          float* __restrict__ probs, uint8_t* __restrict__ collapse,
                                                          int N, int h, int w, int ypix, int act, float inv_temperature,
                                                          int collapse_value, int tiles_x, int tiles_y) {
  constexpr int NV = (9 * C + 3) / 4;        // 16-byte vectors per cell
  constexpr int HW_ = HG_TW + 2, HH_ = HG_TH + 2;
  __shared__ float4 tile[HH_ * HW_ * NV];
  const int H = 2 * h, W = 2 * w;
  float b[C];
#pragma unroll
  for (int c = 0; c < C; ++c) b[c] = bias ? bias[c] : 0.f;
  const int bt = blockIdx.x;
  const int n = bt / (tiles_x * tiles_y);
  const int tr = bt - n * tiles_x * tiles_y;
  const int Y0 = (tr / tiles_x) * HG_TH, X0 = (tr % tiles_x) * HG_TW;
  const float* yn = y + (long long)n * h * w * ypix;
  for (int e = threadIdx.x; e < HH_ * HW_ * NV; e += 256) {
    const int cell = e / NV, k = e - cell * NV;
    const int cy = cell / HW_, cx = cell - cy * HW_;
    // cells outside the image only ever meet zero weights; clamped so that what they hold is finite
    const int yy = min(max(Y0 - 1 + cy, 0), h - 1), xx = min(max(X0 - 1 + cx, 0), w - 1);
    tile[e] = reinterpret_cast<const float4*>(yn + ((long long)yy * w + xx) * ypix)[k];
  }
  __syncthreads();
  const int lx = threadIdx.x % HG_TW, ly = threadIdx.x / HG_TW;
  const int X = X0 + lx, Y = Y0 + ly;
  if (X < w && Y < h) {
    // bilinear sources of the four output rows / columns 2Y-1+qi, 2X-1+qi as neighbourhood indices 0..2
    // (row Y-1+index; -1 = no contribution) and weights; an output row outside the image contributes nothing
    int iy0[4], iy1[4], ix0[4], ix1[4];
    float fy[4], fx[4];
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) {
      const int qy = 2 * Y - 1 + qi, qx = 2 * X - 1 + qi;
      const float sy = fminf(fmaxf(qy * 0.5f - 0.25f, 0.f), (float)(h - 1));
      const float sx = fminf(fmaxf(qx * 0.5f - 0.25f, 0.f), (float)(w - 1));
      const int y0 = (int)sy, x0 = (int)sx;
      fy[qi] = sy - (float)y0;
      fx[qi] = sx - (float)x0;
      const bool vy = (unsigned)qy < (unsigned)H, vx = (unsigned)qx < (unsigned)W;
      iy0[qi] = vy ? y0 - (Y - 1) : -1;
      iy1[qi] = vy ? min(y0 + 1, h - 1) - (Y - 1) : -1;
      ix0[qi] = vx ? x0 - (X - 1) : -1;
      ix1[qi] = vx ? min(x0 + 1, w - 1) - (X - 1) : -1;
    }
    float v[2][2][C];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int c = 0; c < C; ++c) v[q >> 1][q & 1][c] = b[c];
    // a real loop: unrolled, the compiler hoists all nine cells' loads (252 registers, one wave per SIMD)
#pragma unroll 1
    for (int cell = 0; cell < 9; ++cell) {
      const int ni = cell / 3, nj = cell - 3 * ni;
      float wyc[4], wxc[4];
#pragma unroll
      for (int qi = 0; qi < 4; ++qi) {
        wyc[qi] = (iy0[qi] == ni ? 1.f - fy[qi] : 0.f) + (iy1[qi] == ni ? fy[qi] : 0.f);
        wxc[qi] = (ix0[qi] == nj ? 1.f - fx[qi] : 0.f) + (ix1[qi] == nj ? fx[qi] : 0.f);
      }
      const float4* src = tile + ((ly + ni) * HW_ + lx + nj) * NV;
      float t[NV * 4];
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const float4 q = src[k];
        t[4 * k] = q.x; t[4 * k + 1] = q.y; t[4 * k + 2] = q.z; t[4 * k + 3] = q.w;
      }
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
              const float wgt = wyc[a + ty] * wxc[bb + tx];
#pragma unroll
              for (int c = 0; c < C; ++c) v[a][bb][c] += wgt * t[CM ? c * 9 + ty * 3 + tx : (ty * 3 + tx) * C + c];
            }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      // the two pixels of an output row are adjacent in memory: 2 * C floats and 2 * C bytes, written as
      // 8-byte and 2-byte pieces
      float pr2[2 * C], cv2[2 * C];
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        float* pr = pr2 + bb * C;
        if (act == MISEG_ACT_SIGMOID) {
#pragma unroll
          for (int c = 0; c < C; ++c) pr[c] = 1.f / (1.f + __expf(-v[a][bb][c]));
        } else {
          float mx = -3.0e38f;
#pragma unroll
          for (int c = 0; c < C; ++c) mx = fmaxf(mx, v[a][bb][c] * inv_temperature);
          float sum = 0.f;
#pragma unroll
          for (int c = 0; c < C; ++c) {
            pr[c] = __expf(v[a][bb][c] * inv_temperature - mx);
            sum += pr[c];
          }
          const float inv = 1.f / sum;
#pragma unroll
          for (int c = 0; c < C; ++c) pr[c] *= inv;
        }
        // collapse_probabilities (utils/utils.py:50-68): channel c keeps `collapse_value` where it is strictly
        // greater than every other channel; the comparison of channel c sees the already collapsed values of
        // the channels before it
        float* cv = cv2 + bb * C;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          bool win = true;
#pragma unroll
          for (int j = 0; j < C; ++j) {
            if (j == c) continue;
            const float other = j < c ? cv[j] : pr[j];
            win = win && (pr[c] > other);
          }
          cv[c] = win ? (float)collapse_value : 0.f;
        }
      }
      const long long o = (((long long)n * H + 2 * Y + a) * W + 2 * X) * C;
      float2* po = reinterpret_cast<float2*>(probs + o);
#pragma unroll
      for (int k = 0; k < C; ++k) po[k] = make_float2(pr2[2 * k], pr2[2 * k + 1]);
      if (collapse) {
        unsigned short* co = reinterpret_cast<unsigned short*>(collapse + o);
#pragma unroll
        for (int k = 0; k < C; ++k)
          co[k] = (unsigned short)((unsigned)cv2[2 * k] | ((unsigned)cv2[2 * k + 1] << 8));
      }
    }
  }
}

int launch_head_gather(const float* y, const float* bias, float* probs, uint8_t* collapse, int N, int h, int w,
                       int classes, int ypix, int act, float temperature, int collapse_value, int class_major,
                       hipStream_t s) {
  MISEG_CHECK_ARG(y && probs && N > 0 && h > 0 && w > 0, "head_gather: bad argument");
  MISEG_CHECK_ARG(classes >= 1 && classes <= 4 && ypix >= (9 * classes + 3) / 4 * 4 && (ypix & 3) == 0 &&
                      ((uintptr_t)y & 15) == 0,
                  "head_gather: 1..4 classes, 9 * classes channels in 16-byte aligned pixels");
  MISEG_CHECK_ARG(act == MISEG_ACT_SOFTMAX || act == MISEG_ACT_SIGMOID, "head_gather: softmax or sigmoid");
  MISEG_CHECK_ARG(temperature > 0.f, "head_gather: temperature must be positive");
  const int tiles_x = (w + HG_TW - 1) / HG_TW, tiles_y = (h + HG_TH - 1) / HG_TH;
  const long long g = (long long)N * tiles_x * tiles_y;
  MISEG_CHECK_ARG(g < (1ll << 31), "head_gather: too many tiles for one launch");
  const float it = 1.f / temperature;
#define MISEG_HG(Cn)                                                                                                \
  if (class_major)                                                                                                  \
    hipLaunchKernelGGL((head_gather_kernel<Cn, true>), dim3((int)g), dim3(256), 0, s, y, bias, probs, collapse, N, \
                       h, w, ypix, act, it, collapse_value, tiles_x, tiles_y);                                     \
  else                                                                                                              \
    hipLaunchKernelGGL((head_gather_kernel<Cn, false>), dim3((int)g), dim3(256), 0, s, y, bias, probs, collapse, N, \
                       h, w, ypix, act, it, collapse_value, tiles_x, tiles_y)
  switch (classes) {
    case 1: MISEG_HG(1); break;
    case 2: MISEG_HG(2); break;
    case 3: MISEG_HG(3); break;
    default: MISEG_HG(4); break;
  }
#undef MISEG_HG
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

// The adjoint of the gather above, for training: dz = d loss / d logits (bf16 [N][2h][2w][dzs], the first C
// channels) -> dy = d loss / d y (bf16 [N][h][w][ypix], channel c * 9 + tap, zeros from 9 * C on).
//   dy[Y][X][c][ky][kx] = sum over the tap positions q = (qy, qx) inside the image whose bilinear sample
//   uses cell (Y, X), of its weight there times dz at the output pixel (qy - ky + 1, qx - kx + 1), when that
//   pixel exists.
// The rows that sample cell row Y are qy = 2Y-1 .. 2Y+2, so the output pixels involved are the 6 x 6 patch
// from (2Y-2, 2X-2); one thread per cell walks the patch row by row: R[ky][col] += wy[r + ky - 2] * P[r][col],
// then dy[ky][kx] = sum_qj wx[qj] * R[ky][qj - kx + 2].
template <int C>
__global__ __launch_bounds__(256) void head_gather_bwd_kernel(const bf16_t* __restrict__ dz, bf16_t* __restrict__ dy,
                                                              int N, int h, int w, int dzs, int ypix) {
  const int H = 2 * h, W = 2 * w;
  const long long total = (long long)N * h * w;
  constexpr int NO = (9 * C + 7) / 8 * 8;
  const bool vec = (dzs & 3) == 0 && ((uintptr_t)dz & 7) == 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int X = (int)(i % w);
    const long long r_ = i / w;
    const int Y = (int)(r_ % h);
    const int n = (int)(r_ / h);
    // weight of cell row Y (column X) in the bilinear sample at row 2Y-1+qi (column 2X-1+qi); zero for a
    // position outside the image
    float wy[4], wx[4];
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) {
      const int qy = 2 * Y - 1 + qi, qx = 2 * X - 1 + qi;
      const float sy = fminf(fmaxf(qy * 0.5f - 0.25f, 0.f), (float)(h - 1));
      const float sx = fminf(fmaxf(qx * 0.5f - 0.25f, 0.f), (float)(w - 1));
      const int y0 = (int)sy, x0 = (int)sx;
      const float fy = sy - (float)y0, fx = sx - (float)x0;
      const float cy = (y0 == Y ? 1.f - fy : 0.f) + (min(y0 + 1, h - 1) == Y ? fy : 0.f);
      const float cx = (x0 == X ? 1.f - fx : 0.f) + (min(x0 + 1, w - 1) == X ? fx : 0.f);
      wy[qi] = (unsigned)qy < (unsigned)H ? cy : 0.f;
      wx[qi] = (unsigned)qx < (unsigned)W ? cx : 0.f;
    }
    float R[3][6][C];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int col = 0; col < 6; ++col)
#pragma unroll
        for (int c = 0; c < C; ++c) R[ky][col][c] = 0.f;
    const bf16_t* dzn = dz + (long long)n * H * W * dzs;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int oy = 2 * Y - 2 + r;
      if ((unsigned)oy >= (unsigned)H) continue;
      float P[6][C];
#pragma unroll
      for (int col = 0; col < 6; ++col) {
        const int ox = 2 * X - 2 + col;
        const bool ok = (unsigned)ox < (unsigned)W;
        const bf16_t* src = dzn + ((long long)oy * W + (ok ? ox : 0)) * dzs;
        if (vec) {
          // the pixel's first four channels in one 8-byte load
          const uint2 v = *reinterpret_cast<const uint2*>(src);
          const float f[4] = {bf16lo_to_f32(v.x), bf16hi_to_f32(v.x), bf16lo_to_f32(v.y), bf16hi_to_f32(v.y)};
#pragma unroll
          for (int c = 0; c < C; ++c) P[col][c] = ok ? f[c] : 0.f;
        } else {
#pragma unroll
          for (int c = 0; c < C; ++c) P[col][c] = ok ? bf16_to_f32(src[c]) : 0.f;
        }
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int qi = r + ky - 2;            // the tap position's row is 2Y-1+qi
        if (qi < 0 || qi > 3) continue;
#pragma unroll
        for (int col = 0; col < 6; ++col)
#pragma unroll
          for (int c = 0; c < C; ++c) R[ky][col][c] += wy[qi] * P[col][c];
      }
    }
    float o[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) o[j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int qj = 0; qj < 4; ++qj)
#pragma unroll
          for (int c = 0; c < C; ++c) o[c * 9 + ky * 3 + kx] += wx[qj] * R[ky][qj - kx + 2][c];
    uint4* dst = reinterpret_cast<uint4*>(dy + i * ypix);
#pragma unroll
    for (int k = 0; k < NO / 8; ++k) {
      float o8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o8[j] = o[8 * k + j];
      dst[k] = make_uint4(pack_bf16x2(o8[0], o8[1]), pack_bf16x2(o8[2], o8[3]), pack_bf16x2(o8[4], o8[5]),
                          pack_bf16x2(o8[6], o8[7]));
    }
    for (int k = NO / 8; k < (ypix >> 3); ++k) dst[k] = make_uint4(0, 0, 0, 0);
  }
}

int launch_head_gather_bwd(const void* dz, int dzs, void* dy, int N, int h, int w, int classes, int ypix,
                           hipStream_t s) {
  MISEG_CHECK_ARG(dz && dy && N > 0 && h > 0 && w > 0, "head_gather_bwd: bad argument");
  MISEG_CHECK_ARG(classes >= 1 && classes <= 4 && dzs >= classes && (ypix & 7) == 0 &&
                      ypix >= (9 * classes + 7) / 8 * 8 && ((uintptr_t)dy & 15) == 0,
                  "head_gather_bwd: 1..4 classes, dy pixels of a multiple of 8 channels >= 9 * classes, 16-byte aligned");
  const long long total = (long long)N * h * w;
  long long g = (total + 255) / 256;
  if (g > 256 * 64) g = 256 * 64;
#define MISEG_HGB(Cn)                                                                                         \
  hipLaunchKernelGGL(head_gather_bwd_kernel<Cn>, dim3((int)g), dim3(256), 0, s, (const bf16_t*)dz, (bf16_t*)dy, N, \
                     h, w, dzs, ypix)
  switch (classes) {
    case 1: MISEG_HGB(1); break;
    case 2: MISEG_HGB(2); break;
    case 3: MISEG_HGB(3); break;
    default: MISEG_HGB(4); break;
  }
#undef MISEG_HGB
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

}  // namespace miseg
