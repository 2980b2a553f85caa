// The tail of the LinkNet decoder in one kernel: segmentation_models DecoderBlock `decoder_stage4`
// (1x1 64 -> 16, Conv2DTranspose(4, 2, 'same') 16 -> 16, 1x1 16 -> 16, each followed by
// BatchNorm + ReLU) and the head (`final_conv` 3x3 16 -> classes with bias, softmax / sigmoid)
// plus the collapse_probabilities mask (/root/reference utils/model.py:188-198,
// utils/utils.py:50-68).  At 1920x2560 the three 16-channel full-resolution tensors between
// these layers are the widest tensors of the network per unit of arithmetic; run layer by layer
// they cost 4 x 64 B per output pixel of HBM traffic against 32 B in (per output pixel) and 15 B
// out for the fused form, and the whole tail is 2 kMAC per pixel.
//
// A block of 4 waves owns 8 x 64 output pixels = 4 x 32 input pixels (four blocks per CU):
//   0. the (4+2) x (32+2) input pixels x 64 channels arrive by LDS-DMA (144-byte pixel pitch);
//   1. 1x1 64 -> 16 on those 204 pixels (v_mfma_f32_16x16x32: 16 couts x 16 pixels x 32 k),
//      scale/shift + ReLU, zero outside the image (the transpose's padding) -> LDS;
//   2. the transpose as four 2x2 phase convolutions, one phase per wave (its [16][64] weights in
//      registers; K = 4 taps x 16 channels), on the (8+2) x (64+2) output pixels the head's 3x3
//      needs;
//   3. 1x1 16 -> 16 on the transpose's accumulators where they lie: a lane's 4 channels of its pixel
//      are 4 of its 8 k slots of the 32-deep instruction (the others zero, the weights dealt to
//      match), zero outside the image (the head's padding) -> LDS, over the dead input tile;
//   4. head: 9 taps x 16 channels = five 32-deep steps per 16 pixels, bias, activation, collapse;
//      rows leave as contiguous 16-byte stores through a wave-private LDS staging area.
// After step 0 no LDS-DMA is pending, so all LDS traffic is ordinary compiler-scheduled code.
#include "conv_common.h"

namespace MISEG_KNS {

struct TailParams {
  const h16_t* in;
  const h16_t *wa, *wb, *wc, *wd;
  const float *sa, *ta, *sb, *tb, *sc, *tc, *bias;
  float* out;
  uint8_t* collapse_out;
  int N, H, W;            // input extent; the output is 2H x 2W
  int in_pix_stride, in_row_stride;
  long long in_img_stride;
  int out_pix_stride, out_row_stride;
  long long out_img_stride;
  int tiles_y, tiles_x, mtiles;
};

namespace {
constexpr int LT_IH = 4, LT_IW = 32;               // input pixels per block
constexpr int LT_HW = LT_IW + 2, LT_HH = LT_IH + 2;
constexpr int LT_HPX = LT_HH * LT_HW;              // 204 input pixels with halo
constexpr int LT_XPITCH = 144;
constexpr int LT_NHI = (LT_HPX * LT_XPITCH + 1023) / 1024;   // 29 wave-instructions
constexpr int LT_MPITCH = 48;                      // 16 channels + 16 bytes of padding
constexpr int LT_APITCH = 32;                      // the first 1x1's tile: 16 channels, no padding (its 16-pixel
                                                   // fragments and quad writes cover contiguous bytes)
constexpr int LT_MA = (LT_HPX * LT_APITCH + 1023) / 1024 * 1024;   // 1x1 output tile
constexpr int LT_OW = 2 * LT_IW + 2, LT_OH = 2 * LT_IH + 2;
constexpr int LT_OPX = LT_OH * LT_OW;              // 660 output pixels with halo
constexpr int LT_PW = LT_IW + 1, LT_PH = LT_IH + 1;          // pixels of one phase: 5 x 33
constexpr int LT_PPX = LT_PH * LT_PW;
constexpr int LT_BC_OFF = LT_MA;                   // input tile first, then the output-resolution tile
constexpr int LT_BC = LT_OPX * LT_MPITCH;
static_assert(LT_NHI * 1024 <= LT_BC + 1024, "the input tile fits under the output-resolution tile");
constexpr int LT_SCR_OFF = 0;                      // the head's staging area overlays the first 1x1's dead tile
constexpr int LT_SCR_WAVE = 1024 + 256;
static_assert(4 * LT_SCR_WAVE <= LT_MA, "the staging area fits under the first tile");
constexpr int LT_LDS = LT_BC_OFF + (LT_BC > LT_NHI * 1024 ? LT_BC : LT_NHI * 1024);
static_assert(4 * LT_LDS <= 160 * 1024, "four blocks per CU");
typedef f32x4_t f32x4;

__device__ __forceinline__ uint2 lt_pack4(float a, float b, float c, float d) {
  return make_uint2(pack_h16x2_raw(a, b), pack_h16x2_raw(c, d));
}
}  // namespace

__global__ __launch_bounds__(256, 4) void linknet_tail_kernel(const TailParams t, const ConvParams hd) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LT_LDS];
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
  const int y0 = tyi * LT_IH, x0 = txi * LT_IW;          // input origin
  const int Y0 = 2 * y0, X0 = 2 * x0;                     // output origin
  const int Ho = 2 * t.H, Wo = 2 * t.W;

  unsigned char* XH = lds + LT_BC_OFF;
  unsigned char* MA = lds;
  unsigned char* BC = lds + LT_BC_OFF;

  // ---- 0: input halo by LDS-DMA ----
  {
    const h16_t* img = t.in + n * t.in_img_stride;
    const h16_t* zero_src = reinterpret_cast<const h16_t*>(g_zero_line);
#pragma unroll
    for (int j = 0; j < (LT_NHI + 3) / 4; ++j) {
      const int i = 4 * j + wave;
      if (i < LT_NHI) {
        const int o = i * 1024 + lane * 16;
        const int hr = o / LT_XPITCH;
        const int sl = (o - hr * LT_XPITCH) >> 4;
        const int hy = hr / LT_HW, hx = hr - hy * LT_HW;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = hr < LT_HPX && sl < 8 && (unsigned)iy < (unsigned)t.H && (unsigned)ix < (unsigned)t.W;
        const h16_t* src = ok ? img + (long long)iy * t.in_row_stride + (long long)ix * t.in_pix_stride + sl * 8 : zero_src;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(XH + i * 1024), 16, 0, 0);
      }
    }
  }
  // weights and coefficients to registers while the halo is in flight
  h16x8 fa[2], fb[2], fc, fd[5];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    fa[ks] = *reinterpret_cast<const h16x8*>(t.wa + row * 64 + ks * 32 + kg * 8);
    fb[ks] = *reinterpret_cast<const h16x8*>(t.wb + (wave * 16 + row) * 64 + ks * 32 + kg * 8);
  }
  // the second 1x1 consumes the transpose's accumulators as they lie: lane (pixel, kg) holds channels 4 kg .. 4 kg + 3
  // of its pixel, which become k slots 8 kg .. 8 kg + 3 of the 32-deep instruction (slots 8 kg + 4 .. 8 kg + 7 zero)
  {
    const uint2 w4 = *reinterpret_cast<const uint2*>(t.wc + row * 32 + 4 * kg);
    fc = __builtin_bit_cast(h16x8, make_uint4(w4.x, w4.y, 0u, 0u));
  }
#pragma unroll
  for (int s = 0; s < 5; ++s) fd[s] = *reinterpret_cast<const h16x8*>(t.wd + row * 160 + s * 32 + kg * 8);
  const float4 sa4 = *reinterpret_cast<const float4*>(t.sa + 4 * kg), ta4 = *reinterpret_cast<const float4*>(t.ta + 4 * kg);
  const float4 sb4 = *reinterpret_cast<const float4*>(t.sb + 4 * kg), tb4 = *reinterpret_cast<const float4*>(t.tb + 4 * kg);
  const float4 sc4 = *reinterpret_cast<const float4*>(t.sc + 4 * kg), tc4 = *reinterpret_cast<const float4*>(t.tc + 4 * kg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // Every phase below reads all of a wave's fragments first, then runs its matrix instructions, then
  // writes: the loads of different tiles are independent, but the compiler cannot move an LDS read
  // above an earlier LDS write of the same array, so a tile-by-tile loop would expose the LDS
  // latency once per tile.

  // ---- 1: 1x1 64 -> 16 on the input tile (13 tiles of 16 pixels: wave, wave + 4, ...) ----
  {
    constexpr int NT1 = ((LT_HPX + 15) / 16 + 3) / 4;
    h16x8 xb[NT1][2];
#pragma unroll
    for (int k = 0; k < NT1; ++k) {
      const int px = (wave + 4 * k) * 16 + row;
      const int pc = px < LT_HPX ? px : LT_HPX - 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        xb[k][ks] = *reinterpret_cast<const h16x8*>(XH + pc * LT_XPITCH + ks * 64 + kg * 16);
    }
    f32x4 acc[NT1];
#pragma unroll
    for (int k = 0; k < NT1; ++k) {
      acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) acc[k] = mfma_16x16x32(fa[ks], xb[k][ks], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < NT1; ++k) {
      const int px = (wave + 4 * k) * 16 + row;
      const int pc = px < LT_HPX ? px : LT_HPX - 1;
      const int hy = pc / LT_HW, hx = pc - hy * LT_HW;
      const bool inside = (unsigned)(y0 - 1 + hy) < (unsigned)t.H && (unsigned)(x0 - 1 + hx) < (unsigned)t.W;
      uint2 o = lt_pack4(h16_relu(acc[k][0] * sa4.x + ta4.x), h16_relu(acc[k][1] * sa4.y + ta4.y),
                         h16_relu(acc[k][2] * sa4.z + ta4.z), h16_relu(acc[k][3] * sa4.w + ta4.w));
      if (!inside) o = make_uint2(0, 0);
      if (px < LT_HPX) *reinterpret_cast<uint2*>(MA + px * LT_APITCH + kg * 8) = o;
    }
  }
  __syncthreads();

  // ---- 2 + 3: the transpose, phase (a, b) = wave (11 tiles in two batches), and the 1x1 16 -> 16 behind it on
  // the tile's accumulators in registers; zero outside the image (the head's padding) ----
  {
    const int pa = wave >> 1, pb = wave & 1;
    constexpr int NT2 = (LT_PPX + 15) / 16, HB = (NT2 + 1) / 2;
#pragma unroll
    for (int h0 = 0; h0 < NT2; h0 += HB) {
      h16x8 xb[HB][2];
#pragma unroll
      for (int k = 0; k < HB; ++k) {
        const int p = (h0 + k) * 16 + row;
        const int pc = p < LT_PPX ? p : LT_PPX - 1;
        const int r = pc / LT_PW, c = pc - r * LT_PW;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)      // taps (dy, dx) = (ks, kg >> 1), channels 8 (kg & 1) ..
          xb[k][ks] = *reinterpret_cast<const h16x8*>(MA + ((r + ks) * LT_HW + c + (kg >> 1)) * LT_APITCH + (kg & 1) * 16);
      }
#pragma unroll
      for (int k = 0; k < HB; ++k) {
        if (h0 + k >= NT2) continue;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) acc = mfma_16x16x32(fb[ks], xb[k][ks], acc);
        const int p = (h0 + k) * 16 + row;
        const int pc = p < LT_PPX ? p : LT_PPX - 1;
        const int r = pc / LT_PW, c = pc - r * LT_PW;
        const uint2 ob = lt_pack4(h16_relu(acc[0] * sb4.x + tb4.x), h16_relu(acc[1] * sb4.y + tb4.y),
                                  h16_relu(acc[2] * sb4.z + tb4.z), h16_relu(acc[3] * sb4.w + tb4.w));
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
        acc2 = mfma_16x16x32(fc, __builtin_bit_cast(h16x8, make_uint4(ob.x, ob.y, 0u, 0u)), acc2);
        const int ry = 2 * r - pa + 1, rx = 2 * c - pb + 1;
        const bool inside = (unsigned)(Y0 - 1 + ry) < (unsigned)Ho && (unsigned)(X0 - 1 + rx) < (unsigned)Wo;
        uint2 o = lt_pack4(h16_relu(acc2[0] * sc4.x + tc4.x), h16_relu(acc2[1] * sc4.y + tc4.y),
                           h16_relu(acc2[2] * sc4.z + tc4.z), h16_relu(acc2[3] * sc4.w + tc4.w));
        if (!inside) o = make_uint2(0, 0);
        if (p < LT_PPX) *reinterpret_cast<uint2*>(BC + (ry * LT_OW + rx) * LT_MPITCH + kg * 8) = o;
      }
    }
  }
  __syncthreads();

  // ---- 4: head ----
  int hoff[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    int tap = 2 * s + (kg >> 1);
    if (tap > 8) tap = 8;                // K = 144 .. 159: zero weights
    hoff[s] = ((tap / 3) * LT_OW + tap % 3) * LT_MPITCH + (kg & 1) * 16;
  }
  const int C = hd.Cout;
  float bias[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bias[j] = (t.bias && j < C) ? t.bias[j] : 0.f;
  float* sw = reinterpret_cast<float*>(lds + LT_SCR_OFF + wave * LT_SCR_WAVE);
  uint8_t* sm = reinterpret_cast<uint8_t*>(sw) + 1024;
  const long long HoWo = (long long)Ho * Wo;
#pragma unroll 1
  for (int rr = 0; rr < 2; ++rr) {
    const int orow = 2 * wave + rr;
    f32x4 hacc[4];
    {
      h16x8 xb[4][5];
#pragma unroll
      for (int tl = 0; tl < 4; ++tl) {
        const unsigned char* base = BC + (orow * LT_OW + tl * 16 + row) * LT_MPITCH;
#pragma unroll
        for (int s = 0; s < 5; ++s) xb[tl][s] = *reinterpret_cast<const h16x8*>(base + hoff[s]);
      }
#pragma unroll
      for (int tl = 0; tl < 4; ++tl) {
        hacc[tl] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 5; ++s) hacc[tl] = mfma_16x16x32(fd[s], xb[tl][s], hacc[tl]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
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
    const int oy = Y0 + orow, ox = X0 + lane;
    const long long o_row0 = n * t.out_img_stride + (long long)oy * t.out_row_stride + (long long)X0 * t.out_pix_stride;
    const long long m_row0 = ((long long)n * HoWo + (long long)oy * Wo + X0) * C;
    const bool whole = oy < Ho && X0 + 2 * LT_IW <= Wo && t.out_pix_stride == C &&
                       (((uintptr_t)(t.out + o_row0)) & 15) == 0 &&
                       (!t.collapse_out || (((uintptr_t)(t.collapse_out + m_row0)) & 15) == 0);
    if (whole) {   // wave-uniform
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
      if (lane < 16 * C)      // 64 pixels x C floats = 16 C chunks of 16 bytes
        *reinterpret_cast<float4*>(t.out + o_row0 + lane * 4) = *reinterpret_cast<const float4*>(sw + lane * 4);
      if (t.collapse_out && lane < 4 * C)
        *reinterpret_cast<uint4*>(t.collapse_out + m_row0 + lane * 16) = *reinterpret_cast<const uint4*>(sm + lane * 16);
    } else if (oy < Ho && ox < Wo) {
      float* o = t.out + n * t.out_img_stride + (long long)oy * t.out_row_stride + (long long)ox * t.out_pix_stride;
      uint8_t* c8 = t.collapse_out ? t.collapse_out + ((long long)n * HoWo + (long long)oy * Wo + ox) * C : nullptr;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < C) {
          o[j] = pr[j];
          if (c8) c8[j] = (uint8_t)cv[j];
        }
    }
  }
}

// Everything the kernel and its grid assume about a descriptor; called when a plan records the op and again at launch.
int validate_linknet_tail(const miseg_linknet_tail_desc* d) {
  MISEG_CHECK_ARG(d != nullptr, "linknet_tail: null descriptor");
  MISEG_CHECK_ARG(d->in && d->wa && d->wb && d->wc && d->wd && d->out, "linknet_tail: null pointer");
  MISEG_CHECK_ARG(d->sa && d->ta && d->sb && d->tb && d->sc && d->tc, "linknet_tail: null scale/shift vector");
  MISEG_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0, "linknet_tail: non-positive dimension");
  if (d->classes < 1 || d->classes > 4) {
    set_error("linknet_tail: classes=%d must be 1..4", d->classes);
    return MISEG_E_UNSUPPORTED_SHAPE;
  }
  MISEG_CHECK_ARG((d->act == MISEG_ACT_SOFTMAX || d->act == MISEG_ACT_SIGMOID) && d->temperature > 0.f,
                  "linknet_tail: needs softmax/sigmoid and a positive temperature");
  MISEG_CHECK_ARG((d->in_pix_stride & 7) == 0 && (d->in_row_stride & 7) == 0 && (d->in_img_stride & 7) == 0 &&
                      d->in_pix_stride >= 64,
                  "linknet_tail: input strides must be multiples of 8 elements, 64 channels per pixel");
  const uintptr_t al = (uintptr_t)d->in | (uintptr_t)d->wa | (uintptr_t)d->wb | (uintptr_t)d->wc | (uintptr_t)d->wd |
                       (uintptr_t)d->sa | (uintptr_t)d->ta | (uintptr_t)d->sb | (uintptr_t)d->tb | (uintptr_t)d->sc |
                       (uintptr_t)d->tc;
  MISEG_CHECK_ARG((al & 15) == 0, "linknet_tail: input, weights and coefficient vectors must be 16-byte aligned");
  MISEG_CHECK_ARG(((uintptr_t)d->out & 3) == 0, "linknet_tail: output misaligned");
  // the output is (N, 2H, 2W, classes) fp32: a pixel must hold the classes, rows and images must not overlap
  MISEG_CHECK_ARG(d->out_pix_stride >= d->classes && d->out_row_stride >= 2ll * d->W * d->out_pix_stride &&
                      d->out_img_stride >= 2ll * d->H * d->out_row_stride,
                  "linknet_tail: output strides do not hold a (2H, 2W, classes) image");
  return MISEG_OK;
}

int launch_linknet_tail(const miseg_linknet_tail_desc* d, hipStream_t stream) {
  const int rc = validate_linknet_tail(d);
  if (rc != MISEG_OK) return rc;
  TailParams t;
  t.in = (const h16_t*)d->in;
  t.wa = (const h16_t*)d->wa; t.wb = (const h16_t*)d->wb; t.wc = (const h16_t*)d->wc; t.wd = (const h16_t*)d->wd;
  t.sa = d->sa; t.ta = d->ta; t.sb = d->sb; t.tb = d->tb; t.sc = d->sc; t.tc = d->tc; t.bias = d->bias;
  t.out = d->out;
  t.collapse_out = (uint8_t*)d->collapse_out;
  t.N = d->N; t.H = d->H; t.W = d->W;
  t.in_pix_stride = d->in_pix_stride; t.in_row_stride = d->in_row_stride; t.in_img_stride = d->in_img_stride;
  t.out_pix_stride = d->out_pix_stride; t.out_row_stride = d->out_row_stride; t.out_img_stride = d->out_img_stride;
  t.tiles_y = (d->H + LT_IH - 1) / LT_IH;
  t.tiles_x = (d->W + LT_IW - 1) / LT_IW;
  const long long tiles = (long long)d->N * t.tiles_y * t.tiles_x;
  if (tiles >= (1ll << 31)) {
    set_error("linknet_tail: too many tiles");
    return MISEG_E_UNSUPPORTED_SHAPE;
  }
  t.mtiles = (int)tiles;
  ConvParams hd{};
  hd.Cout = d->classes;
  hd.act = d->act;
  hd.inv_temperature = d->act == MISEG_ACT_SOFTMAX ? 1.f / d->temperature : 1.f;
  hd.collapse_value = d->collapse_value;
  hipLaunchKernelGGL(linknet_tail_kernel, dim3(t.mtiles), dim3(256), 0, stream, t, hd);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("linknet_tail launch failed: %s", hipGetErrorString(e));
    return MISEG_E_HIP;
  }
  return MISEG_OK;
}

// ---------------------------------------------------------------------------------------------
// A whole LinkNet decoder block whose skip tensor has 64 channels (segmentation_models DecoderBlock
// `decoder_stage3`: 1x1 CIN -> CIN/4, Conv2DTranspose(4, 2, 'same') CIN/4 -> CIN/4, 1x1 CIN/4 -> 64, each
// followed by BatchNorm + ReLU, then Add with the skip tensor; /root/reference utils/model.py:188-198), for
// CIN = 64 (the ResNets: the previous block ends on the 64-channel stage-1 skip) and CIN = 128 (the DenseNets:
// it ends on pool2_conv's 128 channels).  Layer by layer the two CIN/4-channel tensors cost as much traffic as
// 64-channel ones (they are stored padded for the matrix kernels): 384 B per output pixel against CIN/2 B in,
// 128 B skip and 128 B out here.  Steps 0-2 are those of the tail kernel above without the head's halo (a
// phase owns 4 x 32 of the block's 8 x 64 output pixels); step 3 is the 1x1 to 64 channels (four cout groups
// of v_mfma_f32_16x16x32, their weight rows dealt so that a lane holds 8 consecutive couts of one pixel over a
// pair of them) with scale/shift, ReLU and the skip addition in registers: 16-byte skip loads and stores, the
// four kg lanes of a pixel cover 64 contiguous bytes and the two pairs complete the pixel's 128-byte line.
// ---------------------------------------------------------------------------------------------
struct BlockParams {
  const h16_t* in;
  const h16_t *wa, *wb, *wc;
  const float *sa, *ta, *sb, *tb, *sc, *tc;
  const h16_t* skip;
  h16_t* out;
  int N, H, W;            // input extent; skip and output are 2H x 2W
  int in_pix_stride, in_row_stride;
  long long in_img_stride;
  int skip_pix_stride, skip_row_stride;
  long long skip_img_stride;
  int out_pix_stride, out_row_stride;
  long long out_img_stride;
  int tiles_y, tiles_x, mtiles;
};

namespace {
constexpr int LB_OW = 2 * LT_IW, LB_OH = 2 * LT_IH;                  // 8 x 64 output pixels
template <int CIN>
struct LbGeom {
  static constexpr int MID = CIN / 4;
  static constexpr int XPITCH = CIN * 2 + 16;                       // input halo: bytes per pixel
  static constexpr int NHI = (LT_HPX * XPITCH + 1023) / 1024;       // wave-instructions that fetch it
  static constexpr int MPITCH = MID * 2 + 16;                       // CIN/4-channel tiles
  static constexpr int MA = (LT_HPX * MPITCH + 1023) / 1024 * 1024;
  static constexpr int BC = LB_OH * LB_OW * MPITCH;
  static constexpr int LDS = MA + (BC > NHI * 1024 ? BC : NHI * 1024);
  static constexpr int OCC = 3 * LDS <= 160 * 1024 ? 3 : 2;
  static_assert(2 * LDS <= 160 * 1024, "two blocks per CU");
};
}  // namespace

template <int CIN>
__global__ __launch_bounds__(256, LbGeom<CIN>::OCC) void linknet_block_kernel(const BlockParams t) {
  typedef LbGeom<CIN> G;
  constexpr int MID = G::MID, MG = MID / 16, XPITCH = G::XPITCH, MPITCH = G::MPITCH;
  constexpr int KS1 = CIN / 32;               // K steps of the first 1x1
  constexpr int KS2 = 4 * MID / 32;           // ... of a transpose phase: k = tap * MID + channel
  __shared__ __attribute__((aligned(1024))) unsigned char lds[G::LDS];
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
  const int y0 = tyi * LT_IH, x0 = txi * LT_IW;
  const int Y0 = 2 * y0, X0 = 2 * x0;
  const int Ho = 2 * t.H, Wo = 2 * t.W;

  unsigned char* MA = lds;
  unsigned char* XH = lds + G::MA;            // the output-resolution tile overlays the dead input halo
  unsigned char* BC = lds + G::MA;

  // ---- 0: input halo by LDS-DMA ----
  {
    const h16_t* img = t.in + n * t.in_img_stride;
    const h16_t* zero_src = reinterpret_cast<const h16_t*>(g_zero_line);
#pragma unroll
    for (int j = 0; j < (G::NHI + 3) / 4; ++j) {
      const int i = 4 * j + wave;
      if (i < G::NHI) {
        const int o = i * 1024 + lane * 16;
        const int hr = o / XPITCH;
        const int sl = (o - hr * XPITCH) >> 4;
        const int hy = hr / LT_HW, hx = hr - hy * LT_HW;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = hr < LT_HPX && sl < CIN / 8 && (unsigned)iy < (unsigned)t.H && (unsigned)ix < (unsigned)t.W;
        const h16_t* src = ok ? img + (long long)iy * t.in_row_stride + (long long)ix * t.in_pix_stride + sl * 8 : zero_src;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(XH + i * 1024), 16, 0, 0);
      }
    }
  }
  h16x8 fa[MG][KS1], fb[MG][KS2], fc[4];
#pragma unroll
  for (int mg = 0; mg < MG; ++mg) {
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
      fa[mg][ks] = *reinterpret_cast<const h16x8*>(t.wa + (mg * 16 + row) * CIN + ks * 32 + kg * 8);
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks)
      fb[mg][ks] = *reinterpret_cast<const h16x8*>(t.wb + (wave * MID + mg * 16 + row) * (4 * MID) + ks * 32 + kg * 8);
  }
  // the last 1x1's weight rows are dealt so that a lane ends up with 8 CONSECUTIVE couts over a pair of matrix
  // instructions: instruction (pair, half) row 4 q + j holds cout 32 pair + 8 q + 4 half + j, hence lane
  // (pixel, kg) accumulator j of it is cout 32 pair + 8 kg + 4 half + j: 16-byte skip loads and stores
#pragma unroll
  for (int g = 0; g < 4; ++g)
    fc[g] = *reinterpret_cast<const h16x8*>(t.wc + ((g >> 1) * 32 + 8 * (row >> 2) + 4 * (g & 1) + (row & 3)) * 32 + kg * 8);
  float4 sa4[MG], ta4[MG], sb4[MG], tb4[MG];
#pragma unroll
  for (int mg = 0; mg < MG; ++mg) {
    sa4[mg] = *reinterpret_cast<const float4*>(t.sa + mg * 16 + 4 * kg);
    ta4[mg] = *reinterpret_cast<const float4*>(t.ta + mg * 16 + 4 * kg);
    sb4[mg] = *reinterpret_cast<const float4*>(t.sb + mg * 16 + 4 * kg);
    tb4[mg] = *reinterpret_cast<const float4*>(t.tb + mg * 16 + 4 * kg);
  }
  float4 sc4[4], tc4[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    sc4[g] = *reinterpret_cast<const float4*>(t.sc + (g >> 1) * 32 + 8 * kg + 4 * (g & 1));
    tc4[g] = *reinterpret_cast<const float4*>(t.tc + (g >> 1) * 32 + 8 * kg + 4 * (g & 1));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- 1: 1x1 CIN -> MID on the input tile (13 tiles of 16 pixels: wave, wave + 4, ...) ----
  {
    constexpr int NT1 = ((LT_HPX + 15) / 16 + 3) / 4;
    h16x8 xb[NT1][KS1];
#pragma unroll
    for (int k = 0; k < NT1; ++k) {
      const int px = (wave + 4 * k) * 16 + row;
      const int pc = px < LT_HPX ? px : LT_HPX - 1;
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks)
        xb[k][ks] = *reinterpret_cast<const h16x8*>(XH + pc * XPITCH + ks * 64 + kg * 16);
    }
    f32x4 acc[NT1][MG];
#pragma unroll
    for (int k = 0; k < NT1; ++k)
#pragma unroll
      for (int mg = 0; mg < MG; ++mg) {
        acc[k][mg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) acc[k][mg] = mfma_16x16x32(fa[mg][ks], xb[k][ks], acc[k][mg]);
      }
#pragma unroll
    for (int k = 0; k < NT1; ++k) {
      const int px = (wave + 4 * k) * 16 + row;
      const int pc = px < LT_HPX ? px : LT_HPX - 1;
      const int hy = pc / LT_HW, hx = pc - hy * LT_HW;
      const bool inside = (unsigned)(y0 - 1 + hy) < (unsigned)t.H && (unsigned)(x0 - 1 + hx) < (unsigned)t.W;
#pragma unroll
      for (int mg = 0; mg < MG; ++mg) {
        uint2 o = lt_pack4(h16_relu(acc[k][mg][0] * sa4[mg].x + ta4[mg].x), h16_relu(acc[k][mg][1] * sa4[mg].y + ta4[mg].y),
                           h16_relu(acc[k][mg][2] * sa4[mg].z + ta4[mg].z), h16_relu(acc[k][mg][3] * sa4[mg].w + ta4[mg].w));
        if (!inside) o = make_uint2(0, 0);
        if (px < LT_HPX) *reinterpret_cast<uint2*>(MA + px * MPITCH + mg * 32 + kg * 8) = o;
      }
    }
  }
  __syncthreads();

  // ---- 2: the transpose, phase (a, b) = wave: output pixels (2 r' + a, 2 c' + b), r' < 4, c' < 32 ----
  {
    const int pa = wave >> 1, pb = wave & 1;
    constexpr int NT2 = LT_IH * LT_IW / 16;      // 8 tiles, four at a time
    int koff[KS2];                               // k = ks * 32 + kg * 8 = tap * MID + channel, tap = dy * 2 + dx
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
      const int kidx = ks * 32 + kg * 8;
      const int tap = kidx / MID, ch = kidx - tap * MID;
      koff[ks] = ((tap >> 1) * LT_HW + (tap & 1)) * MPITCH + ch * 2;
    }
#pragma unroll
    for (int h0 = 0; h0 < NT2; h0 += 4) {
      h16x8 xb[4][KS2];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int p = (h0 + k) * 16 + row;
        const int r = (p >> 5) + pa, c = (p & 31) + pb;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks)
          xb[k][ks] = *reinterpret_cast<const h16x8*>(MA + (r * LT_HW + c) * MPITCH + koff[ks]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int p = (h0 + k) * 16 + row;
        const int rr = p >> 5, cc = p & 31;
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS2; ++ks) acc = mfma_16x16x32(fb[mg][ks], xb[k][ks], acc);
          const uint2 o = lt_pack4(h16_relu(acc[0] * sb4[mg].x + tb4[mg].x), h16_relu(acc[1] * sb4[mg].y + tb4[mg].y),
                                   h16_relu(acc[2] * sb4[mg].z + tb4[mg].z), h16_relu(acc[3] * sb4[mg].w + tb4[mg].w));
          *reinterpret_cast<uint2*>(BC + ((2 * rr + pa) * LB_OW + 2 * cc + pb) * MPITCH + mg * 32 + kg * 8) = o;
        }
      }
    }
  }
  __syncthreads();

  // ---- 3: 1x1 MID -> 64 (K padded to 32), scale/shift, ReLU, + skip: wave = 16 columns of every output row ----
  {
    const int ox = X0 + wave * 16 + row;
    const h16_t* skp = t.skip + n * t.skip_img_stride + (long long)ox * t.skip_pix_stride + 8 * kg;
    h16_t* outp = t.out + n * t.out_img_stride + (long long)ox * t.out_pix_stride + 8 * kg;
#pragma unroll
    for (int h0 = 0; h0 < LB_OH; h0 += 4) {
      h16x8 xb[4];
      uint4 sk[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int oy = Y0 + h0 + i;
        const unsigned char* px = BC + ((h0 + i) * LB_OW + wave * 16 + row) * MPITCH;
        if constexpr (MID == 16) {
          xb[i] = *reinterpret_cast<const h16x8*>(px + (kg & 1) * 16);
          if (kg >= 2) xb[i] = __builtin_bit_cast(h16x8, make_uint4(0, 0, 0, 0));
        } else {
          xb[i] = *reinterpret_cast<const h16x8*>(px + kg * 16);
        }
        const bool okp = oy < Ho && ox < Wo;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
          sk[i][pr] = okp ? *reinterpret_cast<const uint4*>(skp + (long long)oy * t.skip_row_stride + pr * 32) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int oy = Y0 + h0 + i;
        const bool okp = oy < Ho && ox < Wo;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
          lo = mfma_16x16x32(fc[2 * pr], xb[i], lo);
          hi = mfma_16x16x32(fc[2 * pr + 1], xb[i], hi);
          const float4 s0 = sc4[2 * pr], t0 = tc4[2 * pr], s1 = sc4[2 * pr + 1], t1 = tc4[2 * pr + 1];
          const uint4 r = sk[i][pr];
          const float v0 = h16_relu(lo[0] * s0.x + t0.x) + h16lo_to_f32(r.x);
          const float v1 = h16_relu(lo[1] * s0.y + t0.y) + h16hi_to_f32(r.x);
          const float v2 = h16_relu(lo[2] * s0.z + t0.z) + h16lo_to_f32(r.y);
          const float v3 = h16_relu(lo[3] * s0.w + t0.w) + h16hi_to_f32(r.y);
          const float v4 = h16_relu(hi[0] * s1.x + t1.x) + h16lo_to_f32(r.z);
          const float v5 = h16_relu(hi[1] * s1.y + t1.y) + h16hi_to_f32(r.z);
          const float v6 = h16_relu(hi[2] * s1.z + t1.z) + h16lo_to_f32(r.w);
          const float v7 = h16_relu(hi[3] * s1.w + t1.w) + h16hi_to_f32(r.w);
          if (okp) {
            const uint2 a = lt_pack4(h16_act(v0, false), h16_act(v1, false), h16_act(v2, false), h16_act(v3, false));
            const uint2 b = lt_pack4(h16_act(v4, false), h16_act(v5, false), h16_act(v6, false), h16_act(v7, false));
            *reinterpret_cast<uint4*>(outp + (long long)oy * t.out_row_stride + pr * 32) = make_uint4(a.x, a.y, b.x, b.y);
          }
        }
      }
    }
  }
}

int validate_linknet_block(const miseg_linknet_block_desc* d) {
  MISEG_CHECK_ARG(d != nullptr, "linknet_block: null descriptor");
  MISEG_CHECK_ARG(d->in && d->wa && d->wb && d->wc && d->skip && d->out, "linknet_block: null pointer");
  MISEG_CHECK_ARG(d->sa && d->ta && d->sb && d->tb && d->sc && d->tc, "linknet_block: null scale/shift vector");
  MISEG_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0, "linknet_block: non-positive dimension");
  MISEG_CHECK_ARG((d->in_pix_stride & 7) == 0 && (d->in_row_stride & 7) == 0 && (d->in_img_stride & 7) == 0 &&
                      d->in_pix_stride >= d->cin,
                  "linknet_block: input strides must be multiples of 8 elements, cin channels per pixel");
  if (d->cin != 64 && d->cin != 128) {
    set_error("linknet_block: cin=%d must be 64 or 128", d->cin);
    return MISEG_E_UNSUPPORTED_SHAPE;
  }
  MISEG_CHECK_ARG((d->skip_pix_stride & 7) == 0 && (d->skip_row_stride & 7) == 0 && (d->skip_img_stride & 7) == 0 &&
                      d->skip_pix_stride >= 64 && (d->out_pix_stride & 7) == 0 && (d->out_row_stride & 7) == 0 &&
                      (d->out_img_stride & 7) == 0 && d->out_pix_stride >= 64,
                  "linknet_block: skip and output strides must be multiples of 8 elements, 64 channels per pixel");
  const uintptr_t al = (uintptr_t)d->in | (uintptr_t)d->wa | (uintptr_t)d->wb | (uintptr_t)d->wc | (uintptr_t)d->sa |
                       (uintptr_t)d->ta | (uintptr_t)d->sb | (uintptr_t)d->tb | (uintptr_t)d->sc | (uintptr_t)d->tc;
  MISEG_CHECK_ARG((al & 15) == 0, "linknet_block: input, weights and coefficient vectors must be 16-byte aligned");
  MISEG_CHECK_ARG((((uintptr_t)d->skip | (uintptr_t)d->out) & 15) == 0, "linknet_block: skip / output misaligned");
  // skip and output are (N, 2H, 2W, 64): rows and images must not overlap
  MISEG_CHECK_ARG(d->out_row_stride >= 2ll * d->W * d->out_pix_stride && d->out_img_stride >= 2ll * d->H * d->out_row_stride &&
                      d->skip_row_stride >= 2ll * d->W * d->skip_pix_stride &&
                      d->skip_img_stride >= 2ll * d->H * d->skip_row_stride,
                  "linknet_block: skip / output strides do not hold a (2H, 2W, 64) image");
  return MISEG_OK;
}

int launch_linknet_block(const miseg_linknet_block_desc* d, hipStream_t stream) {
  const int rc = validate_linknet_block(d);
  if (rc != MISEG_OK) return rc;
  BlockParams t;
  t.in = (const h16_t*)d->in;
  t.wa = (const h16_t*)d->wa; t.wb = (const h16_t*)d->wb; t.wc = (const h16_t*)d->wc;
  t.sa = d->sa; t.ta = d->ta; t.sb = d->sb; t.tb = d->tb; t.sc = d->sc; t.tc = d->tc;
  t.skip = (const h16_t*)d->skip;
  t.out = (h16_t*)d->out;
  t.N = d->N; t.H = d->H; t.W = d->W;
  t.in_pix_stride = d->in_pix_stride; t.in_row_stride = d->in_row_stride; t.in_img_stride = d->in_img_stride;
  t.skip_pix_stride = d->skip_pix_stride; t.skip_row_stride = d->skip_row_stride; t.skip_img_stride = d->skip_img_stride;
  t.out_pix_stride = d->out_pix_stride; t.out_row_stride = d->out_row_stride; t.out_img_stride = d->out_img_stride;
  t.tiles_y = (d->H + LT_IH - 1) / LT_IH;
  t.tiles_x = (d->W + LT_IW - 1) / LT_IW;
  const long long tiles = (long long)d->N * t.tiles_y * t.tiles_x;
  if (tiles >= (1ll << 31)) {
    set_error("linknet_block: too many tiles");
    return MISEG_E_UNSUPPORTED_SHAPE;
  }
  t.mtiles = (int)tiles;
  if (d->cin == 64)
    hipLaunchKernelGGL(linknet_block_kernel<64>, dim3(t.mtiles), dim3(256), 0, stream, t);
  else
    hipLaunchKernelGGL(linknet_block_kernel<128>, dim3(t.mtiles), dim3(256), 0, stream, t);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("linknet_block launch failed: %s", hipGetErrorString(e));
    return MISEG_E_HIP;
  }
  return MISEG_OK;
}

}  // namespace MISEG_KNS
