// Weight gradient of a convolution on gfx950 (the backward-filter pass that tf.keras runs
// inside Model.fit, /root/reference model_train.py:284-329, for every Conv2D /
// Conv2DTranspose the networks of utils/model.py:176-211 contain).
//
//   dW[co][tap][ci] = sum over output pixels p of  dY[p][co] * X[p shifted by tap][ci]
//
// GEMM view per tap: D[co][ci] = sum_k A[co][k] B[k][ci] with k = output pixel.  Both
// operands live in memory pixel-major (NHWC), i.e. k is the slow index, so the MFMA
// fragments (8 consecutive k per lane) are columns of the staged [pixel][channel] LDS
// tiles: they are fetched with ds_read_b64_tr_b16, the transposing LDS read.
//
// Block = 4 waves computing a [BC couts] x [BC cins] tile of one tap over a slice of the
// pixels (split-K); K-step = 64 pixels; tiles are staged by LDS-DMA, double buffered, with
// the 16-byte chunk index XOR-swizzled on the source side.  Partial tiles are added to the
// fp32 gradient with float atomics (two 128-byte row segments per wave instruction).
#include "common.h"
#include <cmath>

#include <type_traits>

namespace miseg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// 16-byte source of LDS-DMA loads for rows that fall outside the tensors
__device__ __attribute__((aligned(128))) unsigned char g_zero_line_wg[128];

struct WgradParams {
  const bf16_t* x;
  const bf16_t* dy;
  float* dw;
  int N, H, W, Cin, Ho, Wo, Cout;
  int KH, KW, stride_y, stride_x, pad_y, pad_x;
  int x_pix, x_row;
  long long x_img;
  int dy_pix, dy_row;
  long long dy_img;
  int M, ksteps_total, ksplit, ksteps_per_split;
  int co_tiles, ci_tiles, taps;
  const float* x_scale;     // X operand = relu(x * x_scale[ci] + x_shift[ci]) when set (halo kernel, 3x3, 32 x 32)
  const float* x_shift;
};

template <int BC>
__device__ __forceinline__ int wg_swz(int row, int chunk) {
  // BC = 128: 16 chunks per 256-B row; BC = 64: 8 chunks per 128-B row (two rows per bank row)
  return BC == 128 ? (chunk ^ ((row & 3) << 2)) : (chunk ^ (((row >> 1) & 1) << 2));
}

template <int BC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
  constexpr int RB = BC * 2;              // bytes per LDS row (one pixel)
  constexpr int RPI = 1024 / RB;          // rows per wave-instruction
  constexpr int LPR = 64 / RPI;           // lanes per row = 16-byte chunks per row
  constexpr int NI = 64 / RPI / 4;        // wave-instructions per wave per operand tile
  constexpr int WT = BC / 64;             // 32x32 MFMA tiles per wave per dimension
  constexpr int TILE = 64 * RB;           // bytes of one operand tile
  constexpr int STAGE = 2 * TILE;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  int item = blockIdx.x;
  const int ci_t = item % p.ci_tiles;
  item /= p.ci_tiles;
  const int co_t = item % p.co_tiles;
  item /= p.co_tiles;
  const int tap = item % p.taps;
  const int kz = item / p.taps;
  const int ky = tap / p.KW, kx = tap - ky * p.KW;
  const int co0 = co_t * BC, ci0 = ci_t * BC;
  const int ks_begin = kz * p.ksteps_per_split;
  int ks_end = ks_begin + p.ksteps_per_split;
  if (ks_end > p.ksteps_total) ks_end = p.ksteps_total;

  const int HoWo = p.Ho * p.Wo;
  const int lrow = lane / LPR;            // row within the wave-instruction
  const int slot = lane % LPR;            // LDS chunk position within the row
  const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(g_zero_line_wg);

  // per staged row: output pixel coordinates (advanced by 64 pixels per K-step)
  int r_n[NI], r_oy[NI], r_ox[NI], r_m[NI], r_ch[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = (j * 4 + wave) * RPI + lrow;
    const int m = ks_begin * 64 + row;
    r_m[j] = m;
    const int mm = m < p.M ? m : 0;
    r_n[j] = mm / HoWo;
    const int rem = mm - r_n[j] * HoWo;
    r_oy[j] = rem / p.Wo;
    r_ox[j] = rem - r_oy[j] * p.Wo;
    r_ch[j] = wg_swz<BC>(row, slot) * 8;   // source channel offset that lands in this slot
  }

  auto issue = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const bool in_range = r_m[j] < p.M;
      // dY tile: [pixel][co0 + ...]
      {
        const int c = co0 + r_ch[j];
        const bf16_t* src = (in_range && c < p.Cout)
                                ? p.dy + r_n[j] * p.dy_img + (long long)r_oy[j] * p.dy_row +
                                      (long long)r_ox[j] * p.dy_pix + c
                                : zero_src;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)src,
            (__attribute__((address_space(3))) void*)(lds + buf * STAGE + (j * 4 + wave) * 1024),
            16, 0, 0);
      }
      // X tile for this tap: [pixel shifted by the tap][ci0 + ...]
      {
        const int iy = r_oy[j] * p.stride_y - p.pad_y + ky;
        const int ix = r_ox[j] * p.stride_x - p.pad_x + kx;
        const int c = ci0 + r_ch[j];
        const bool ok = in_range && c < p.Cin && (unsigned)iy < (unsigned)p.H &&
                        (unsigned)ix < (unsigned)p.W;
        const bf16_t* src = ok ? p.x + r_n[j] * p.x_img + (long long)iy * p.x_row +
                                     (long long)ix * p.x_pix + c
                               : zero_src;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)src,
            (__attribute__((address_space(3))) void*)(lds + buf * STAGE + TILE +
                                                      (j * 4 + wave) * 1024),
            16, 0, 0);
      }
      // advance this row by 64 pixels
      r_m[j] += 64;
      r_ox[j] += 64;
      while (r_ox[j] >= p.Wo) {
        r_ox[j] -= p.Wo;
        if (++r_oy[j] == p.Ho) {
          r_oy[j] = 0;
          ++r_n[j];
        }
      }
    }
  };

  f32x16 acc[WT][WT];
#pragma unroll
  for (int a = 0; a < WT; ++a)
#pragma unroll
    for (int b = 0; b < WT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // transposing-read geometry: 16-lane group g reads a 4(k) x 16(channel) block; lane 4q+p of
  // the group addresses row q, channels 4p..4p+3; lane i of the group receives channel i.
  const int g = lane >> 4;
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  const int kgrp = 8 * (g >> 1) + tq;      // k row inside a 16-deep sub-step (first read)
  const int cgrp = 16 * (g & 1) + 4 * tp;  // channel inside the 32-wide MFMA tile

  auto frag = [&](const unsigned char* tile, int krow0, int col0) __attribute__((always_inline)) {
    // 8 consecutive k (pixels) of one channel column: two transposed 4x16 reads
    const int col = col0 + cgrp;
    const int chunk = col >> 3, within = (col & 7) * 2;
    const int r0 = krow0 + kgrp, r1 = r0 + 4;
    const unsigned char* a0 = tile + r0 * RB + wg_swz<BC>(r0, chunk) * 16 + within;
    const unsigned char* a1 = tile + r1 * RB + wg_swz<BC>(r1, chunk) * 16 + within;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(a1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  auto compute = [&](int buf) __attribute__((always_inline)) {
    const unsigned char* dyt = lds + buf * STAGE;
    const unsigned char* xt = dyt + TILE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[WT], bfr[WT];
#pragma unroll
      for (int a = 0; a < WT; ++a) af[a] = frag(dyt, ks * 16, wm * (WT * 32) + a * 32);
#pragma unroll
      for (int b = 0; b < WT; ++b) bfr[b] = frag(xt, ks * 16, wn * (WT * 32) + b * 32);
#pragma unroll
      for (int a = 0; a < WT; ++a)
#pragma unroll
        for (int b = 0; b < WT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    }
  };

  if (ks_begin < ks_end) {
    issue(0);
    for (int ks = ks_begin; ks < ks_end; ++ks) {
      const int it = ks - ks_begin;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (ks + 1 < ks_end) issue((it + 1) & 1);
      compute(it & 1);
    }
  }

  // accumulate the tile: lane column = ci, register row = co
  const int lcol = lane & 31, lhalf = lane >> 5;
#pragma unroll
  for (int a = 0; a < WT; ++a)
#pragma unroll
    for (int b = 0; b < WT; ++b) {
      const int ci = ci0 + wn * (WT * 32) + b * 32 + lcol;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * (WT * 32) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
        if (co < p.Cout && ci < p.Cin)
          atomicAdd(p.dw + ((long long)co * p.taps + tap) * p.Cin + ci, acc[a][b][r]);
      }
    }
}

// ---------------------------------------------------------------------------------------------
// All taps per block, for thin layers with any stride: Cin <= 32, Cout <= 64, 2..8 taps (the 7x7
// stride-2 stem in its 7x1 form over 32-"channel" pixel windows).  The per-tap kernel above reads the
// dY tile once per tap (seven times 2.5 GB for the stem at 16 x 1920 x 2560); here a K-step's dY tile
// [64 pixels][64 couts] is staged once, next to the X tiles [64 pixels][32 channels] of every tap, and
// wave (wm, wn) multiplies cout group wm with the taps [4 wn, 4 wn + 4).  Split-K over pixels, fp32
// atomics at the end, as above.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void wgrad_alltaps_kernel(const WgradParams p) {
  constexpr int YT = 64 * 128;            // dY tile bytes: 64 pixel rows of 128 B
  constexpr int XT = 64 * 64;             // X tile bytes per tap: 64 pixel rows of 64 B
  constexpr int STAGE = YT + 8 * XT;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int kz = blockIdx.x;
  const int ks_begin = kz * p.ksteps_per_split;
  int ks_end = ks_begin + p.ksteps_per_split;
  if (ks_end > p.ksteps_total) ks_end = p.ksteps_total;
  const int HoWo = p.Ho * p.Wo;
  const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(g_zero_line_wg);

  // staged rows: dY instruction j covers rows (2 wave + j) * 8 + lane / 8 (8 chunks of 16 B per row), an X
  // instruction covers rows wave * 16 + lane / 4 (4 chunks per row); both advance by 64 pixels per K-step
  struct Row { int n, oy, ox, m; };
  auto locate = [&](int row) {
    Row r;
    r.m = ks_begin * 64 + row;
    const int mm = r.m < p.M ? r.m : 0;
    r.n = mm / HoWo;
    const int rem = mm - r.n * HoWo;
    r.oy = rem / p.Wo;
    r.ox = rem - r.oy * p.Wo;
    return r;
  };
  auto advance = [&](Row& r) {
    r.m += 64;
    r.ox += 64;
    while (r.ox >= p.Wo) {
      r.ox -= p.Wo;
      if (++r.oy == p.Ho) {
        r.oy = 0;
        ++r.n;
      }
    }
  };
  Row ry[2] = {locate((2 * wave) * 8 + (lane >> 3)), locate((2 * wave + 1) * 8 + (lane >> 3))};
  Row rx = locate(wave * 16 + (lane >> 2));
  int ych[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) ych[j] = wg_swz<64>((2 * wave + j) * 8 + (lane >> 3), lane & 7) * 8;
  const int xch = (lane & 3) * 8;

  auto issue = [&](int buf) __attribute__((always_inline)) {
    unsigned char* st = lds + buf * STAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const Row& r = ry[j];
      const bf16_t* src = (r.m < p.M && ych[j] < p.Cout)
                              ? p.dy + r.n * p.dy_img + (long long)r.oy * p.dy_row + (long long)r.ox * p.dy_pix + ych[j]
                              : zero_src;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + (2 * wave + j) * 1024), 16, 0, 0);
    }
    for (int tap = 0; tap < p.taps; ++tap) {
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
      const int iy = rx.oy * p.stride_y - p.pad_y + ky, ix = rx.ox * p.stride_x - p.pad_x + kx;
      const bool ok = rx.m < p.M && xch < p.Cin && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const bf16_t* src = ok ? p.x + rx.n * p.x_img + (long long)iy * p.x_row + (long long)ix * p.x_pix + xch : zero_src;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + YT + tap * XT + wave * 1024), 16,
                                       0, 0);
    }
    advance(ry[0]);
    advance(ry[1]);
    advance(rx);
  };

  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // transposing reads: 16-lane group g takes a 4 (pixels) x 16 (channels) block, see the per-tap kernel
  const int g = lane >> 4;
  const int kgrp = 8 * (g >> 1) + ((lane & 15) >> 2);
  const int cgrp = 16 * (g & 1) + 4 * (lane & 3);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto read2 = [&](const unsigned char* a0, const unsigned char* a1) __attribute__((always_inline)) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a1));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  const int tap0 = wn * 4;
  const int ntap = p.taps - tap0 < 4 ? p.taps - tap0 : 4;      // may be <= 0: that wave only stages

  auto compute = [&](int buf) __attribute__((always_inline)) {
    const unsigned char* yt = lds + buf * STAGE;
    const unsigned char* xt = yt + YT + tap0 * XT;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int r0 = ks * 16 + kgrp, r1 = r0 + 4;
      const int ycol = wm * 32 + cgrp;
      const bf16x8 yf = read2(yt + r0 * 128 + wg_swz<64>(r0, ycol >> 3) * 16 + (ycol & 7) * 2,
                              yt + r1 * 128 + wg_swz<64>(r1, ycol >> 3) * 16 + (ycol & 7) * 2);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t < ntap) {
          const bf16x8 xf = read2(xt + t * XT + r0 * 64 + cgrp * 2, xt + t * XT + r1 * 64 + cgrp * 2);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf, xf, acc[t], 0, 0, 0);
        }
      }
    }
  };

  if (ks_begin < ks_end) {
    issue(0);
    for (int ks = ks_begin; ks < ks_end; ++ks) {
      const int it = ks - ks_begin;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (ks + 1 < ks_end) issue((it + 1) & 1);
      compute(it & 1);
    }
  }

  const int lcol = lane & 31, lhalf = lane >> 5;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t >= ntap) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
      if (co < p.Cout && lcol < p.Cin)
        atomicAdd(p.dw + ((long long)co * p.taps + tap0 + t) * p.Cin + lcol, acc[t][r]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Halo-resident weight gradient for stride-1 convolutions with small filters (3x3, the 2x2 phases
// of Conv2DTranspose, 1x1).  The per-tap kernel above re-reads both operand tiles for every tap;
// here a block keeps the X halo tile and the dY tile of an 8x32 pixel block in LDS and a wave
// computes ALL taps of its [32 co] x [32 ci] sub-tile from them: the dY fragment of a pixel run is
// read once and multiplied with the X fragments of the KH x KW shifted runs, and those stay in
// registers while the wave walks down the rows (a row of X is used by KH rows of dY).  Blocks are
// persistent over pixel tiles and flush their accumulators with float atomics once at the end.
//
// Waves: 4.  CO_T x CI_T (32 or 64 each) is cut into 32x32 sub-tiles; when there are fewer than 4,
// the remaining factor splits the pixels (the two 16-pixel column windows, then row halves).
//
// XPRO: the layer's input is relu(x * x_scale + x_shift) of a tensor that was never stored activated (the
// forward convolution applied the BatchNorm + ReLU to its halo tile, conv_c32p.hip); the same happens here to
// the X halo tile once it has landed, every lane on the 16-byte pieces it requested, pixels outside the image
// left at zero (they are padding, not bn(0)).
template <int N, typename F>
__device__ __forceinline__ void wg_static_for(F&& f) {
  if constexpr (N > 0) {
    wg_static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int KH, int KW, int CO_T, int CI_T, bool XPRO = false>
__global__ __launch_bounds__(256, 2) void wgrad_halo_kernel(const WgradParams p) {
  constexpr int TH = 8, TW = 32;
  constexpr int HR = TH + KH - 1, HWD = TW + KW - 1, HP = HR * HWD;
  constexpr int XRB = CI_T * 2, YRB = CO_T * 2;          // LDS bytes per pixel row
  constexpr int XPPI = 1024 / XRB, YPPI = 1024 / YRB;    // pixels per wave-instruction
  constexpr int XLPR = 64 / XPPI, YLPR = 64 / YPPI;      // lanes (16-byte chunks) per pixel
  constexpr int XNI = (HP + XPPI - 1) / XPPI;            // wave-instructions for the halo
  constexpr int YNI = TH * TW / YPPI;                    // ... for the dY tile
  constexpr int XBYTES = XNI * 1024, YBYTES = YNI * 1024;
  constexpr int WCO = CO_T / 32, WCI = CI_T / 32, WK = 4 / (WCO * WCI);
  constexpr int TAPS = KH * KW;
  __shared__ __attribute__((aligned(16))) unsigned char lds[XBYTES + YBYTES];
  unsigned char* const xt = lds;
  unsigned char* const yt = lds + XBYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = wave % (WCO * WCI);
  const int wm = q / WCI, wn = q % WCI, kpart = wave / (WCO * WCI);

  const int pair = blockIdx.y;
  const int co0 = (pair / p.ci_tiles) * CO_T, ci0 = (pair % p.ci_tiles) * CI_T;
  const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
  const int ntiles = p.N * tiles_y * tiles_x;
  const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(g_zero_line_wg);

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // transposing-read geometry (see the per-tap kernel)
  const int g = lane >> 4;
  const int kgrp = 8 * (g >> 1) + ((lane & 15) >> 2);
  const int cgrp = 16 * (g & 1) + 4 * (lane & 3);

  auto xswz = [](int row, int chunk) __attribute__((always_inline)) {
    return CI_T == 64 ? (chunk ^ (((row >> 1) & 1) << 2)) : chunk;
  };
  auto yswz = [](int row, int chunk) __attribute__((always_inline)) {
    return CO_T == 64 ? (chunk ^ (((row >> 1) & 1) << 2)) : chunk;
  };
  auto read2 = [&](const unsigned char* a0, const unsigned char* a1) __attribute__((always_inline)) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  // Fragment addresses = a per-lane base + a compile-time offset.  A fragment's first LDS row is
  // r = row0 + kgrp; the XOR swizzle only looks at bit 1 of r, and 4 divides every run-time part of
  // row0 (16 * w, ya * HWD with ya in {0, 4} and HWD even, TW = 32), so the swizzled offset depends
  // on the lane and on (compile-time part of row0) mod 4 only: four bases for X, one for dY, and
  // the rest of the address is an immediate (the second half of a fragment is 4 rows further,
  // which leaves bit 1 of the row unchanged).
  const int run_w = WK == 1 ? 0 : (WK == 2 ? kpart : (kpart & 1));         // column window, if run-time
  const int run_ya = WK == 4 ? (kpart >> 1) * (TH / 2) : 0;
  const unsigned char* xl[4];
  {
    const int col = wn * 32 + cgrp, chunk = col >> 3, within = (col & 7) * 2;
#pragma unroll
    for (int m = 0; m < 4; ++m)
      xl[m] = xt + (kgrp + run_ya * HWD + run_w * 16) * XRB + xswz(m + kgrp, chunk) * 16 + within;
  }
  const unsigned char* yl;
  {
    const int col = wm * 32 + cgrp, chunk = col >> 3, within = (col & 7) * 2;
    yl = yt + (kgrp + run_ya * TW + run_w * 16) * YRB + yswz(kgrp, chunk) * 16 + within;
  }
  // row0c: the compile-time part of the fragment's first pixel row
  auto xfrag = [&](auto row0c) __attribute__((always_inline)) {
    constexpr int row0 = decltype(row0c)::value;
    const unsigned char* a = xl[row0 & 3] + row0 * XRB;
    return read2(a, a + 4 * XRB);
  };
  auto yfrag = [&](auto row0c) __attribute__((always_inline)) {
    constexpr int row0 = decltype(row0c)::value;
    static_assert((row0 & 3) == 0, "dY fragments start at multiples of 16 pixels");
    const unsigned char* a = yl + row0 * YRB;
    return read2(a, a + 4 * YRB);
  };

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t = tile;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int n = t / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    __syncthreads();                         // the previous tile's fragments have been read
    // X halo: LDS pixel hp = hr * HWD + hc  <-  x[n][oy0 + hr - pad_y][ox0 + hc - pad_x]
#pragma unroll
    for (int j = 0; j < (XNI + 3) / 4; ++j) {
      const int inst = j * 4 + wave;
      if (inst < XNI) {
        const int hp = inst * XPPI + lane / XLPR, slot = lane % XLPR;
        const int hr = hp / HWD, hc = hp - hr * HWD;
        const int iy = oy0 + hr - p.pad_y, ix = ox0 + hc - p.pad_x;
        const int c = ci0 + xswz(hp, slot) * 8;
        const bool ok = hp < HP && c < p.Cin && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const bf16_t* src = ok ? p.x + n * p.x_img + (long long)iy * p.x_row + (long long)ix * p.x_pix + c
                               : zero_src;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(xt + inst * 1024), 16, 0, 0);
      }
    }
    // dY tile: LDS pixel dp = y * TW + x
#pragma unroll
    for (int j = 0; j < YNI / 4; ++j) {
      const int inst = j * 4 + wave;
      const int dp = inst * YPPI + lane / YLPR, slot = lane % YLPR;
      const int oy = oy0 + dp / TW, ox = ox0 + dp % TW;
      const int c = co0 + yswz(dp, slot) * 8;
      const bool ok = c < p.Cout && oy < p.Ho && ox < p.Wo;
      const bf16_t* src = ok ? p.dy + n * p.dy_img + (long long)oy * p.dy_row + (long long)ox * p.dy_pix + c
                             : zero_src;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(yt + inst * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (XPRO) {
#pragma unroll
      for (int j = 0; j < (XNI + 3) / 4; ++j) {
        const int inst = j * 4 + wave;
        if (inst < XNI) {
          const int hp = inst * XPPI + lane / XLPR, slot = lane % XLPR;
          const int hr = hp / HWD, hc = hp - hr * HWD;
          const int iy = oy0 + hr - p.pad_y, ix = ox0 + hc - p.pad_x;
          const int c = ci0 + xswz(hp, slot) * 8;
          const bool ok = hp < HP && c < p.Cin && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
          if (ok) {
            uint4* piece = reinterpret_cast<uint4*>(xt + inst * 1024 + lane * 16);
            const uint4 v = *piece;
            const float4 s0 = *reinterpret_cast<const float4*>(p.x_scale + c), s1 = *reinterpret_cast<const float4*>(p.x_scale + c + 4);
            const float4 t0 = *reinterpret_cast<const float4*>(p.x_shift + c), t1 = *reinterpret_cast<const float4*>(p.x_shift + c + 4);
            uint4 o;
            o.x = pack_bf16x2(fmaxf(bf16lo_to_f32(v.x) * s0.x + t0.x, 0.f), fmaxf(bf16hi_to_f32(v.x) * s0.y + t0.y, 0.f));
            o.y = pack_bf16x2(fmaxf(bf16lo_to_f32(v.y) * s0.z + t0.z, 0.f), fmaxf(bf16hi_to_f32(v.y) * s0.w + t0.w, 0.f));
            o.z = pack_bf16x2(fmaxf(bf16lo_to_f32(v.z) * s1.x + t1.x, 0.f), fmaxf(bf16hi_to_f32(v.z) * s1.y + t1.y, 0.f));
            o.w = pack_bf16x2(fmaxf(bf16lo_to_f32(v.w) * s1.z + t1.z, 0.f), fmaxf(bf16hi_to_f32(v.w) * s1.w + t1.w, 0.f));
            *piece = o;
          }
        }
      }
      __syncthreads();
    }

    // w_c: the column window when it is a compile-time choice (WK == 1), else 0 (folded into the bases)
    auto roll = [&](auto rows_c, auto w_c) __attribute__((always_inline)) {
      constexpr int ROWS = decltype(rows_c)::value;
      constexpr int w = decltype(w_c)::value;
      bf16x8 xf[KH][KW];
      wg_static_for<ROWS + KH - 1>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;           // halo row ya + i
        wg_static_for<KW>([&](auto kc) __attribute__((always_inline)) {
          constexpr int kx = decltype(kc)::value;
          xf[i % KH][kx] = xfrag(std::integral_constant<int, i * HWD + w * 16 + kx>{});
        });
        if constexpr (i >= KH - 1) {
          constexpr int y = i - (KH - 1);
          const bf16x8 yf = yfrag(std::integral_constant<int, y * TW + w * 16>{});
          wg_static_for<KH>([&](auto kyc) __attribute__((always_inline)) {
            constexpr int ky = decltype(kyc)::value;
            wg_static_for<KW>([&](auto kc) __attribute__((always_inline)) {
              constexpr int kx = decltype(kc)::value;
              acc[ky * KW + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                  yf, xf[(y + ky) % KH][kx], acc[ky * KW + kx], 0, 0, 0);
            });
          });
        }
      });
    };
    if constexpr (WK == 1) {
      roll(std::integral_constant<int, TH>{}, std::integral_constant<int, 0>{});
      roll(std::integral_constant<int, TH>{}, std::integral_constant<int, 1>{});
    } else if constexpr (WK == 2) {
      roll(std::integral_constant<int, TH>{}, std::integral_constant<int, 0>{});
    } else {
      roll(std::integral_constant<int, TH / 2>{}, std::integral_constant<int, 0>{});
    }
  }

  const int lcol = lane & 31, lhalf = lane >> 5;
  const int ci = ci0 + wn * 32 + lcol;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
      if (co < p.Cout && ci < p.Cin)
        atomicAdd(p.dw + ((long long)co * TAPS + t) * p.Cin + ci, acc[t][r]);
    }
}

template <int KH, int KW>
static void launch_halo_wgrad(const WgradParams& p, hipStream_t stream) {
  const bool co64 = p.Cout > 32, ci64 = p.Cin > 32;
  WgradParams q = p;
  q.co_tiles = co64 ? (p.Cout + 63) / 64 : 1;
  q.ci_tiles = ci64 ? (p.Cin + 63) / 64 : 1;
  const int pairs = q.co_tiles * q.ci_tiles;
  const int ntiles = p.N * ((p.Ho + 7) / 8) * ((p.Wo + 31) / 32);
  int nb = (1024 + pairs - 1) / pairs;                  // ~4 resident blocks per CU over all pairs
  if (nb > ntiles) nb = ntiles;
  {
    // Every block ends by adding its whole weight tile to memory with atomics, whose total count is what a
    // launch with few pixel tiles per block spends its time on (DenseNet's 128 -> 32 layers at 240x320 x 4:
    // 1024 blocks x 36 864 atomics for 2.3 tiles of arithmetic each, 0.21 ms; 0.09 ms with a quarter of the
    // blocks).  Blocks per pair that balance the two: flush ~ nb * pairs * A, arithmetic ~ ntiles * t / nb.
    const int cot = co64 ? 64 : 32, cit = ci64 ? 64 : 32;
    const int wk = 4 / ((cot / 32) * (cit / 32));
    const double A = (double)cot * cit * KH * KW * wk;              // atomics per block
    const double tt = 0.25 + 0.25 * KH * KW * (wk == 1 ? 2 : 1);    // relative time of one pixel tile
    int best = (int)(sqrt((double)ntiles * 6.0e5 * tt / (pairs * A)) + 0.5);
    if (best < 1) best = 1;
    if (best < nb) nb = best;
  }
  const dim3 grid((unsigned)nb, (unsigned)pairs);
  if (co64 && ci64)
    hipLaunchKernelGGL((wgrad_halo_kernel<KH, KW, 64, 64>), grid, dim3(256), 0, stream, q);
  else if (co64)
    hipLaunchKernelGGL((wgrad_halo_kernel<KH, KW, 64, 32>), grid, dim3(256), 0, stream, q);
  else if (ci64)
    hipLaunchKernelGGL((wgrad_halo_kernel<KH, KW, 32, 64>), grid, dim3(256), 0, stream, q);
  else if (q.x_scale && KH == 3)      // (launch_wgrad admits a prologue only for this shape)
    hipLaunchKernelGGL((wgrad_halo_kernel<KH, KW, 32, 32, KH == 3>), grid, dim3(256), 0, stream, q);
  else
    hipLaunchKernelGGL((wgrad_halo_kernel<KH, KW, 32, 32>), grid, dim3(256), 0, stream, q);
}

int launch_wgrad(const miseg_wgrad_desc* d, hipStream_t stream) {
  MISEG_CHECK_ARG(d && d->x && d->dy && d->dw, "wgrad: null pointer");
  MISEG_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->Cin > 0 &&
                      d->Cout > 0 && d->KH > 0 && d->KW > 0 && d->stride_y > 0 && d->stride_x > 0,
                  "wgrad: non-positive dimension");
  MISEG_CHECK_ARG((d->Cin & 7) == 0 && (d->Cout & 7) == 0,
                  "wgrad: Cin and Cout must be multiples of 8 (pad the tensors)");
  MISEG_CHECK_ARG((d->x_pix & 7) == 0 && (d->x_row & 7) == 0 && (d->x_img & 7) == 0 &&
                      (d->dy_pix & 7) == 0 && (d->dy_row & 7) == 0 && (d->dy_img & 7) == 0 &&
                      ((uintptr_t)d->x & 15) == 0 && ((uintptr_t)d->dy & 15) == 0,
                  "wgrad: strides must be multiples of 8 elements, pointers 16-byte aligned");
  const long long M = (long long)d->N * d->Ho * d->Wo;
  if (M >= (1ll << 31)) {
    set_error("wgrad: N*Ho*Wo=%lld exceeds 2^31", M);
    return MISEG_E_UNSUPPORTED_SHAPE;
  }
  WgradParams p;
  p.x = (const bf16_t*)d->x; p.dy = (const bf16_t*)d->dy; p.dw = d->dw;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin;
  p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.KH = d->KH; p.KW = d->KW; p.stride_y = d->stride_y; p.stride_x = d->stride_x;
  p.pad_y = d->pad_y; p.pad_x = d->pad_x;
  p.x_pix = d->x_pix; p.x_row = d->x_row; p.x_img = d->x_img;
  p.dy_pix = d->dy_pix; p.dy_row = d->dy_row; p.dy_img = d->dy_img;
  p.M = (int)M;
  p.ksteps_total = (p.M + 63) / 64;
  p.taps = d->KH * d->KW;
  p.co_tiles = p.ci_tiles = 1;
  p.ksplit = p.ksteps_per_split = 1;
  p.x_scale = d->x_scale;
  p.x_shift = d->x_shift;
  const bool halo_shape = d->stride_y == 1 && d->stride_x == 1 && d->KH == d->KW && d->KH <= 3 &&
                          d->pad_y >= 0 && d->pad_x >= 0 && d->pad_y < d->KH && d->pad_x < d->KW &&
                          (d->Cin & 31) == 0;
  const bool want_halo = d->tile_hint == 2 || d->tile_hint == 0;
  MISEG_CHECK_ARG(d->tile_hint >= 0 && d->tile_hint <= 2, "wgrad: tile_hint must be 0, 1 or 2");
  if (d->tile_hint == 2 && !halo_shape) {
    set_error("wgrad: the halo kernel needs stride 1, a square filter up to 3x3 and Cin %% 32 == 0");
    return MISEG_E_UNSUPPORTED_SHAPE;
  }
  if (d->x_scale || d->x_shift) {
    MISEG_CHECK_ARG(d->x_scale && d->x_shift && ((uintptr_t)d->x_scale & 15) == 0 && ((uintptr_t)d->x_shift & 15) == 0,
                    "wgrad: the X prologue needs both x_scale and x_shift, 16-byte aligned");
    if (!(halo_shape && want_halo && d->KH == 3 && d->Cin <= 32 && d->Cout <= 32)) {
      set_error("wgrad: the X prologue is built for the halo kernel on 3x3 layers with Cin, Cout <= 32");
      return MISEG_E_UNSUPPORTED_SHAPE;
    }
  }
  if (halo_shape && want_halo) {
    if (d->KH == 3) launch_halo_wgrad<3, 3>(p, stream);
    else if (d->KH == 2) launch_halo_wgrad<2, 2>(p, stream);
    else launch_halo_wgrad<1, 1>(p, stream);
    MISEG_CHECK_HIP(hipGetLastError());
    return MISEG_OK;
  }
  if (d->tile_hint == 0 && d->Cin <= 32 && d->Cout <= 64 && p.taps >= 2 && p.taps <= 8) {
    // ~2 blocks per CU, each with at least a few K-steps between its staging and its flush
    long long want = 512;
    if (want > p.ksteps_total / 4) want = p.ksteps_total / 4;
    if (want < 1) want = 1;
    p.ksteps_per_split = (int)((p.ksteps_total + want - 1) / want);
    p.ksplit = (p.ksteps_total + p.ksteps_per_split - 1) / p.ksteps_per_split;
    hipLaunchKernelGGL(wgrad_alltaps_kernel, dim3((unsigned)p.ksplit), dim3(256), 0, stream, p);
    MISEG_CHECK_HIP(hipGetLastError());
    return MISEG_OK;
  }
  const bool big = d->Cin > 64 && d->Cout > 64;
  const int bc = big ? 128 : 64;
  p.co_tiles = (d->Cout + bc - 1) / bc;
  p.ci_tiles = (d->Cin + bc - 1) / bc;
  const long long base = (long long)p.co_tiles * p.ci_tiles * p.taps;
  long long want = (2048 + base - 1) / base;       // aim at ~8 blocks per CU in total
  if (want > p.ksteps_total) want = p.ksteps_total;
  if (want < 1) want = 1;
  p.ksteps_per_split = (int)((p.ksteps_total + want - 1) / want);
  p.ksplit = (p.ksteps_total + p.ksteps_per_split - 1) / p.ksteps_per_split;
  const dim3 grid((unsigned)(base * p.ksplit));
  if (big)
    hipLaunchKernelGGL(conv_wgrad_kernel<128>, grid, dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL(conv_wgrad_kernel<64>, grid, dim3(256), 0, stream, p);
  MISEG_CHECK_HIP(hipGetLastError());
  return MISEG_OK;
}

}  // namespace miseg
