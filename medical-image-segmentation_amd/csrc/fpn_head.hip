// The end of the FPN decoder in one kernel: `final_upsampling` (UpSampling2D(2, 'bilinear'), half-pixel
// centres with clamped edges) and `head_conv` (3x3, 128 -> classes, bias) with softmax / sigmoid and the
// collapse_probabilities mask (/root/reference utils/model.py:200-211, utils/utils.py:50-68).  The
// up-sampled tensor is 256 bytes per output pixel; written and read back it is 10 GB per 4 images at
// 1920x2560, against 1.3 GB for the half-resolution tensor it is interpolated from.
//
// A block of 4 waves owns 8 x 32 output pixels (three blocks per CU); per chunk of 32 channels:
//   1. interpolation of the (8+2) x (32+2) up-sampled pixels the 3x3 needs -> LDS, zero outside the
//      image (the head's padding).  Lane = column, wave = group of 8 channels: a lane reads its two
//      half-resolution neighbour columns x six rows straight from global memory into registers (the
//      next chunk's twelve loads are issued before this chunk's matrix instructions, so nothing
//      waits for them), interpolates the six rows along x and feeds all ten output rows from them.
//      bf16 storage: fp32 arithmetic in the order of the stand-alone kernel (t = a (1 - fx) + b fx,
//      then o = t0 (1 - fy) + t1 fy), rounded once.  fp16 storage: packed fp16 arithmetic.  The
//      clamped edges are the interior formula with the weight of the missing neighbour forced to
//      zero, which is bit-identical;
//   2. head: 9 taps x 32 channels = 9 steps of v_mfma_f32_16x16x32 per 16 pixels, accumulated over
//      the four chunks (class rows >= classes are zero);
//   3. bias, activation, collapse; rows leave as 16-byte stores through LDS staging.
#include "conv_common.h"

namespace MISEG_KNS {

struct FpnHeadParams {
  const h16_t* in;
  const h16_t* wd;      // [16][9][128]
  const float* bias;
  float* out;
  uint8_t* collapse_out;
  int N, H, W;          // half-resolution extent; the output is 2H x 2W
  int in_pix_stride, in_row_stride;
  long long in_img_stride;
  int out_pix_stride, out_row_stride;
  long long out_img_stride;
  int tiles_y, tiles_x, mtiles;
};

namespace {
constexpr int FH_TH = 8, FH_TW = 32;
constexpr int FH_UW = FH_TW + 2, FH_UH = FH_TH + 2;       // up-sampled tile with the 3x3's halo
constexpr int FH_UPX = FH_UH * FH_UW;
constexpr int FH_LH = FH_TH / 2 + 2;                        // half-resolution rows behind them
constexpr int FH_UPITCH = 80;
constexpr int FH_SCR_OFF = (FH_UPX * FH_UPITCH + 15) / 16 * 16;
constexpr int FH_SCR_WAVE = 1024 + 256;
constexpr int FH_W_OFF = FH_SCR_OFF + 4 * FH_SCR_WAVE;     // head weights of the (at most 4) classes: [4][9][128]
constexpr int FH_LDS = FH_W_OFF + 4 * 9 * 128 * 2;
static_assert(3 * FH_LDS <= 160 * 1024, "three blocks per CU");
typedef f32x4_t f32x4;

__device__ __forceinline__ void fh_unpack8(float* f, uint4 v) {
  f[0] = h16lo_to_f32(v.x); f[1] = h16hi_to_f32(v.x); f[2] = h16lo_to_f32(v.y); f[3] = h16hi_to_f32(v.y);
  f[4] = h16lo_to_f32(v.z); f[5] = h16hi_to_f32(v.z); f[6] = h16lo_to_f32(v.w); f[7] = h16hi_to_f32(v.w);
}
}  // namespace

__global__ __launch_bounds__(256, 3) void fpn_head_kernel(const FpnHeadParams t, const ConvParams hd) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[FH_LDS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = lane & 15, kg = lane >> 4;

  int tile;
  {
    const int ntot = t.mtiles, bid = blockIdx.x;
    const int qq = ntot >> 3, r = ntot & 7, x = bid & 7, loc = bid >> 3;
    tile = (x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq) + loc;
  }
  const int per_img = t.tiles_y * t.tiles_x;
  const int n = tile / per_img;
  const int trem = tile - n * per_img;
  const int tyi = trem / t.tiles_x;
  const int txi = trem - tyi * t.tiles_x;
  const int Y0 = tyi * FH_TH, X0 = txi * FH_TW;          // output origin
  const int y0 = Y0 >> 1;                                // first half-resolution row
  const int Ho = 2 * t.H, Wo = 2 * t.W;
  unsigned char* UP = lds;

  // head weights to LDS
  for (int i = tid; i < 4 * 9 * 16; i += 256)
    *reinterpret_cast<uint4*>(lds + FH_W_OFF + i * 16) = *reinterpret_cast<const uint4*>(t.wd + i * 8);
  // interpolation geometry of this lane's column (lane < 34) and of the ten rows
  const int col = lane < FH_UW ? lane : FH_UW - 1;
  const int ox = X0 - 1 + col;
  const bool col_in = (unsigned)ox < (unsigned)Wo;
  // the two half-resolution neighbour columns (clamped into the image: where the clamp bites, the
  // neighbour's weight is zero or the up-sampled pixel lies outside the image) and the six rows
  const h16_t* img = t.in + n * t.in_img_stride + wave * 8;
  const int ixa = min(max((ox - 1) >> 1, 0), t.W - 1), ixb = min(max(((ox - 1) >> 1) + 1, 0), t.W - 1);
  const h16_t* pa = img + (long long)ixa * t.in_pix_stride;
  const h16_t* pb = img + (long long)ixb * t.in_pix_stride;
  long long rowoff[FH_LH];
#pragma unroll
  for (int r = 0; r < FH_LH; ++r) rowoff[r] = (long long)min(max(y0 - 1 + r, 0), t.H - 1) * t.in_row_stride;
  uint4 cur[2 * FH_LH];
  auto load_chunk = [&](int ch) {
    if (lane < FH_UW) {
#pragma unroll
      for (int r = 0; r < FH_LH; ++r) {
        cur[2 * r] = *reinterpret_cast<const uint4*>(pa + rowoff[r] + ch * 32);
        cur[2 * r + 1] = *reinterpret_cast<const uint4*>(pb + rowoff[r] + ch * 32);
      }
    }
  };
  load_chunk(0);
  float fx = (ox & 1) ? 0.25f : 0.75f;
  if (ox == 0) fx = 1.f;                                  // clamped edges: the missing neighbour gets weight 0
  if (ox == Wo - 1) fx = 0.f;
  __syncthreads();

  f32x4 hacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) hacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int ch = 0; ch < 4; ++ch) {
    // ---- 1: interpolate channels [32 ch + 8 wave, + 8) of column `col` ----
#if MISEG_BUILD_F16 && !defined(MISEG_FH_F32)
    // fp16 storage: the interpolation in packed fp16 arithmetic (one v_pk_mul + one v_pk_fma per four
    // values and step; each step rounds to 11 bits, so a value can differ from the fp32-evaluated one
    // by about one ulp of the storage format)
    if (lane < FH_UW) {
      h16x8 T[FH_LH];
      const _Float16 wx1 = (_Float16)fx, wx0 = (_Float16)(1.f - fx);
#pragma unroll
      for (int r = 0; r < FH_LH; ++r)
        T[r] = __builtin_bit_cast(h16x8, cur[2 * r]) * wx0 + __builtin_bit_cast(h16x8, cur[2 * r + 1]) * wx1;
#pragma unroll
      for (int j = 0; j < FH_UH; ++j) {
        const int oy = Y0 - 1 + j;
        float fy = (oy & 1) ? 0.25f : 0.75f;
        if (oy == 0) fy = 1.f;
        if (oy == Ho - 1) fy = 0.f;
        const bool in = col_in && (unsigned)oy < (unsigned)Ho;
        const _Float16 wy1 = (_Float16)fy, wy0 = (_Float16)(1.f - fy);
        h16x8 o = T[j >> 1] * wy0 + T[(j >> 1) + 1] * wy1;
        uint4 pk = __builtin_bit_cast(uint4, o);
        if (!in) pk = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(UP + (j * FH_UW + col) * FH_UPITCH + wave * 16) = pk;
      }
    }
#else
    if (lane < FH_UW) {
      float T[FH_LH][8];
#pragma unroll
      for (int r = 0; r < FH_LH; ++r) {
        float a[8], b[8];
        fh_unpack8(a, cur[2 * r]);
        fh_unpack8(b, cur[2 * r + 1]);
#pragma unroll
        for (int j = 0; j < 8; ++j) T[r][j] = a[j] * (1.f - fx) + b[j] * fx;
      }
#pragma unroll
      for (int j = 0; j < FH_UH; ++j) {
        const int oy = Y0 - 1 + j;
        // rows (oy - 1) >> 1 and the next one: tile rows j >> 1 and (j >> 1) + 1 (Y0 is even)
        float fy = (oy & 1) ? 0.25f : 0.75f;
        if (oy == 0) fy = 1.f;
        if (oy == Ho - 1) fy = 0.f;
        const bool in = col_in && (unsigned)oy < (unsigned)Ho;
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = T[j >> 1][k] * (1.f - fy) + T[(j >> 1) + 1][k] * fy;
        uint4 pk = make_uint4(pack_h16x2(o[0], o[1]), pack_h16x2(o[2], o[3]), pack_h16x2(o[4], o[5]), pack_h16x2(o[6], o[7]));
        if (!in) pk = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(UP + (j * FH_UW + col) * FH_UPITCH + wave * 16) = pk;
      }
    }
#endif
    if (ch + 1 < 4) load_chunk(ch + 1);     // in flight during the matrix instructions below
    __syncthreads();

    // ---- 2: head, wave -> output rows 2 wave, 2 wave + 1 ----
    const unsigned char* xbase = UP + (2 * wave * FH_UW + row) * FH_UPITCH + kg * 16;
    const unsigned char* wbase = lds + FH_W_OFF + ((row & 3) * 9 * 128 + ch * 32 + kg * 8) * 2;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // class rows >= 4 do not exist: zero operand
      h16x8 fw = *reinterpret_cast<const h16x8*>(wbase + tap * 256);
      if (row >= 4) fw = __builtin_bit_cast(h16x8, make_uint4(0, 0, 0, 0));
#pragma unroll
      for (int tl = 0; tl < 4; ++tl)
        hacc[tl] = mfma_16x16x32(fw, *reinterpret_cast<const h16x8*>(
                                         xbase + (((tl >> 1) + tap / 3) * FH_UW + (tl & 1) * 16 + tap % 3) * FH_UPITCH),
                                 hacc[tl]);
    }
    __syncthreads();
  }

  // ---- 3: bias, activation, collapse, stores ----
  const int C = hd.Cout;
  float bias[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bias[j] = (t.bias && j < C) ? t.bias[j] : 0.f;
  float* sw = reinterpret_cast<float*>(lds + FH_SCR_OFF + wave * FH_SCR_WAVE);
  uint8_t* sm = reinterpret_cast<uint8_t*>(sw) + 1024;
  if (lane < 16) {
#pragma unroll
    for (int tl = 0; tl < 4; ++tl)
      *reinterpret_cast<float4*>(sw + (tl * 16 + row) * 4) = make_float4(hacc[tl][0], hacc[tl][1], hacc[tl][2], hacc[tl][3]);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const float4 lg = *reinterpret_cast<const float4*>(sw + lane * 4);
  float v[4] = {lg.x + bias[0], lg.y + bias[1], lg.z + bias[2], lg.w + bias[3]};
  float pr[4], cv[4];
  head_eval(hd, v, pr, cv);
  const long long HoWo = (long long)Ho * Wo;
  const int orow = Y0 + 2 * wave + (lane >> 5);
  const int ocol = X0 + (lane & 31);
  const long long o_row0 = n * t.out_img_stride + (long long)(Y0 + 2 * wave) * t.out_row_stride + (long long)X0 * t.out_pix_stride;
  const long long m_row0 = ((long long)n * HoWo + (long long)(Y0 + 2 * wave) * Wo + X0) * C;
  float* out = t.out;
  const bool whole = Y0 + FH_TH <= Ho && X0 + FH_TW <= Wo && t.out_pix_stride == C &&
                     (((uintptr_t)(out + o_row0) | ((uintptr_t)t.out_row_stride * 4)) & 15) == 0 &&
                     (!t.collapse_out || (((uintptr_t)(t.collapse_out + m_row0) | ((uintptr_t)Wo * C)) & 15) == 0);
  if (whole) {   // block-uniform
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < C) {
        sw[lane * C + j] = pr[j];
        sm[lane * C + j] = (uint8_t)cv[j];
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int cpr = 8 * C;             // 16-byte chunks per 32-pixel row of probabilities
    if (lane < 2 * cpr) {
      const int r = lane / cpr, c = lane - r * cpr;
      *reinterpret_cast<float4*>(out + o_row0 + (long long)r * t.out_row_stride + c * 4) =
          *reinterpret_cast<const float4*>(sw + r * 32 * C + c * 4);
    }
    const int cpm = 2 * C;             // 16-byte chunks per row of the mask
    if (t.collapse_out && lane < 2 * cpm) {
      const int r = lane / cpm, c = lane - r * cpm;
      *reinterpret_cast<uint4*>(t.collapse_out + m_row0 + (long long)r * Wo * C + c * 16) =
          *reinterpret_cast<const uint4*>(sm + r * 32 * C + c * 16);
    }
  } else if (orow < Ho && ocol < Wo) {
    float* o = out + n * t.out_img_stride + (long long)orow * t.out_row_stride + (long long)ocol * t.out_pix_stride;
    uint8_t* c8 = t.collapse_out ? t.collapse_out + ((long long)n * HoWo + (long long)orow * Wo + ocol) * C : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < C) {
        o[j] = pr[j];
        if (c8) c8[j] = (uint8_t)cv[j];
      }
  }
}

// Everything the kernel and its grid assume about a descriptor; called when a plan records the op and again at launch.
int validate_fpn_head(const miseg_fpn_head_desc* d) {
  MISEG_CHECK_ARG(d != nullptr, "fpn_head: null descriptor");
  MISEG_CHECK_ARG(d->in && d->wd && d->out, "fpn_head: null pointer");
  MISEG_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0, "fpn_head: non-positive dimension");
  if (d->classes < 1 || d->classes > 4) {
    set_error("fpn_head: classes=%d must be 1..4", d->classes);
    return MISEG_E_UNSUPPORTED_SHAPE;
  }
  MISEG_CHECK_ARG((d->act == MISEG_ACT_SOFTMAX || d->act == MISEG_ACT_SIGMOID) && d->temperature > 0.f,
                  "fpn_head: needs softmax/sigmoid and a positive temperature");
  MISEG_CHECK_ARG((d->in_pix_stride & 7) == 0 && (d->in_row_stride & 7) == 0 && (d->in_img_stride & 7) == 0 &&
                      d->in_pix_stride >= 128 && (((uintptr_t)d->in | (uintptr_t)d->wd) & 15) == 0,
                  "fpn_head: input needs 128 channels per pixel, strides in multiples of 8 elements, 16-byte alignment");
  MISEG_CHECK_ARG(((uintptr_t)d->out & 3) == 0, "fpn_head: output misaligned");
  // the output is (N, 2H, 2W, classes) fp32: a pixel must hold the classes, rows and images must not overlap
  MISEG_CHECK_ARG(d->out_pix_stride >= d->classes && d->out_row_stride >= 2ll * d->W * d->out_pix_stride &&
                      d->out_img_stride >= 2ll * d->H * d->out_row_stride,
                  "fpn_head: output strides do not hold a (2H, 2W, classes) image");
  return MISEG_OK;
}

int launch_fpn_head(const miseg_fpn_head_desc* d, hipStream_t stream) {
  const int rc = validate_fpn_head(d);
  if (rc != MISEG_OK) return rc;
  FpnHeadParams t;
  t.in = (const h16_t*)d->in;
  t.wd = (const h16_t*)d->wd;
  t.bias = d->bias;
  t.out = d->out;
  t.collapse_out = (uint8_t*)d->collapse_out;
  t.N = d->N; t.H = d->H; t.W = d->W;
  t.in_pix_stride = d->in_pix_stride; t.in_row_stride = d->in_row_stride; t.in_img_stride = d->in_img_stride;
  t.out_pix_stride = d->out_pix_stride; t.out_row_stride = d->out_row_stride; t.out_img_stride = d->out_img_stride;
  t.tiles_y = (2 * d->H + FH_TH - 1) / FH_TH;
  t.tiles_x = (2 * d->W + FH_TW - 1) / FH_TW;
  const long long tiles = (long long)d->N * t.tiles_y * t.tiles_x;
  if (tiles >= (1ll << 31)) {
    set_error("fpn_head: too many tiles");
    return MISEG_E_UNSUPPORTED_SHAPE;
  }
  t.mtiles = (int)tiles;
  ConvParams hd{};
  hd.Cout = d->classes;
  hd.act = d->act;
  hd.inv_temperature = d->act == MISEG_ACT_SOFTMAX ? 1.f / d->temperature : 1.f;
  hd.collapse_value = d->collapse_value;
  hipLaunchKernelGGL(fpn_head_kernel, dim3(t.mtiles), dim3(256), 0, stream, t, hd);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("fpn_head launch failed: %s", hipGetErrorString(e));
    return MISEG_E_HIP;
  }
  return MISEG_OK;
}

}  // namespace MISEG_KNS
