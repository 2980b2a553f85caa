// Implicit-GEMM convolution for gfx950 (MI355X): bf16 operands, fp32 accumulate on
// v_mfma_f32_32x32x16_bf16.  One kernel family serves Conv2D 1x1/3x3 (stride 1 or 2,
// explicit symmetric padding), the 7x7 stride-2 stem (as a 7x1 convolution over a
// channel-padded, spatially padded image) and Conv2DTranspose(4, stride 2, 'same')
// as four 2x2 phase convolutions.  Replaces the tf.keras layers instantiated through
// /root/reference utils/model.py:176-211.
//
// GEMM view:  D[cout][pixel] = sum_k Wt[cout][k] * X[k][pixel],  k = (tap, cin).
// The weight tile is the MFMA "A" operand and the pixel tile the "B" operand, so that
// each lane ends up with runs of 4 consecutive output channels of one pixel
// (NHWC-contiguous) in its accumulator registers.
//
// Block = 256 threads = 4 waves; a K-step is one filter tap x BK input channels of a
// [BM pixels] x [BN couts] tile, double-buffered in LDS with one barrier per K-step.
// Two staging paths: LDS-DMA (global_load_lds, BK = 64) for plain bf16 operands, and a
// register-staged path (BK = 32) that can apply BatchNorm+ReLU to the operand on the way.
#include "conv_common.h"

namespace MISEG_KNS {

// ---------------------------------------------------------------------------------
// Kernel A: register-staged, BK = 32.  Serves operands that need the BatchNorm+ReLU
// prologue and layers whose Cin is not a multiple of 64.
// ---------------------------------------------------------------------------------
template <int WARPS_M, int WARPS_N, int MT, int NT, bool PRO, bool BNM = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvParams p) {
  constexpr int BM = WARPS_M * MT * 32;
  constexpr int BN = WARPS_N * NT * 32;
  constexpr int PA = BM / 64;
  constexpr int PB = (BN + 63) / 64;
  static_assert(WARPS_M * WARPS_N == 4, "4 waves per block");
  static_assert(BM % 64 == 0, "BM multiple of 64");
  constexpr int STAGE = (BM + BN) * 64;  // bytes per pipeline stage: A tile then B tile
  constexpr int BOFF = BM * 64;
  constexpr int EPI = 4 * 32 * (NT * 32 + 4) * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[cmax(2 * STAGE, EPI)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WARPS_N;
  const int wn = wave % WARPS_N;
  const TileCoord tc = tile_coord<BM, BN>(p);

  const int KK = p.KH * p.KW;
  const int nchunk = p.Cin >> 5;
  const int ksteps = KK * nchunk;
  const int HoWo = p.Ho * p.Wo;

  // ---- staging assignment: thread -> (row, 16-byte chunk) ----
  const int srow = tid >> 2;
  const int chunk = tid & 3;

  const h16_t* a_ptr[PA];
  int a_iy0[PA], a_ix0[PA];
  bool a_ok[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = tc.m0 + i * 64 + srow;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    const int n = mm / HoWo;
    const int rem = mm - n * HoWo;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    a_iy0[i] = oy * p.stride_y - tc.pad_y;
    a_ix0[i] = ox * p.stride_x - tc.pad_x;
    a_ptr[i] = p.in + n * p.in_img_stride + (long long)a_iy0[i] * p.in_row_stride +
               (long long)a_ix0[i] * p.in_pix_stride + chunk * 8;
  }
  const h16_t* w_ptr[PB];
  bool w_ok[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int co = tc.cout0 + i * 64 + srow;
    w_ok[i] = (BN >= 64 || srow < BN) && co < p.Cout;
    w_ptr[i] = tc.wbase + (long long)(w_ok[i] ? co : 0) * KK * p.Cin + chunk * 8;
  }

  uint4 ra[PA], rb[PB];
  int dy = 0, dx = 0, cc = 0;  // K-step cursor of the NEXT tile to load

  auto load_tiles = [&](int ks) {
    const int aoff = dy * p.in_row_stride + dx * p.in_pix_stride + cc * 32;
    float4 s0, s1, t0, t1;
    if (PRO) {
      const int c = cc * 32 + chunk * 8;
      s0 = *reinterpret_cast<const float4*>(p.in_scale + c);
      s1 = *reinterpret_cast<const float4*>(p.in_scale + c + 4);
      t0 = *reinterpret_cast<const float4*>(p.in_shift + c);
      t1 = *reinterpret_cast<const float4*>(p.in_shift + c + 4);
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const bool ok = a_ok[i] && (unsigned)(a_iy0[i] + dy) < (unsigned)p.H &&
                      (unsigned)(a_ix0[i] + dx) < (unsigned)p.W;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) {
        v = *reinterpret_cast<const uint4*>(a_ptr[i] + aoff);
        if (PRO) {
          v.x = pack_h16x2_raw(h16_relu(h16lo_to_f32(v.x) * s0.x + t0.x), h16_relu(h16hi_to_f32(v.x) * s0.y + t0.y));
          v.y = pack_h16x2_raw(h16_relu(h16lo_to_f32(v.y) * s0.z + t0.z), h16_relu(h16hi_to_f32(v.y) * s0.w + t0.w));
          v.z = pack_h16x2_raw(h16_relu(h16lo_to_f32(v.z) * s1.x + t1.x), h16_relu(h16hi_to_f32(v.z) * s1.y + t1.y));
          v.w = pack_h16x2_raw(h16_relu(h16lo_to_f32(v.w) * s1.z + t1.z), h16_relu(h16hi_to_f32(v.w) * s1.w + t1.w));
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (w_ok[i]) v = *reinterpret_cast<const uint4*>(w_ptr[i] + ks * 32);
      rb[i] = v;
    }
    if (++cc == nchunk) {
      cc = 0;
      if (++dx == p.KW) {
        dx = 0;
        ++dy;
      }
    }
  };

  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PA; ++i)
      *reinterpret_cast<uint4*>(lds + buf * STAGE + swz64(i * 64 + srow, chunk)) = ra[i];
#pragma unroll
    for (int i = 0; i < PB; ++i)
      if (BN >= 64 || srow < BN)
        *reinterpret_cast<uint4*>(lds + buf * STAGE + BOFF + swz64(i * 64 + srow, chunk)) = rb[i];
  };

  f32x16 acc[NT][MT];
  zero_acc<MT, NT>(acc);
  const int lrow = lane & 31;
  const int lhalf = lane >> 5;

  auto compute = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = kk * 2 + lhalf;
      h16x8 xf[MT], wf[NT];
#pragma unroll
      for (int b = 0; b < MT; ++b)
        xf[b] = *reinterpret_cast<const h16x8*>(
            lds + buf * STAGE + swz64(wm * (MT * 32) + b * 32 + lrow, ch));
#pragma unroll
      for (int a = 0; a < NT; ++a)
        wf[a] = *reinterpret_cast<const h16x8*>(
            lds + buf * STAGE + BOFF + swz64(wn * (NT * 32) + a * 32 + lrow, ch));
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
          acc[a][b] = mfma_32x32x16(wf[a], xf[b], acc[a][b]);
    }
  };

  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int ks = 0; ks < ksteps; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < ksteps) load_tiles(ks + 1);
    compute(buf);
    if (ks + 1 < ksteps) store_tiles(buf ^ 1);
    __syncthreads();
  }
  const LinearDecode dec{tc.m0, p.M, HoWo, p.Wo};
  if constexpr (BNM) {
    // data gradient behind relu(bn(x)): the masking form of the LDS epilogue
    __syncthreads();
    conv_epilogue<WARPS_N, MT, NT, LinearDecode, true>(p, tc, dec, acc, lds, wm, wn, lane);
    return;
  }
  if (p.direct && direct_epilogue_ok(p)) {
    // large, bandwidth-bound launches (full-resolution 1x1 layers): no LDS round trip, no waits
    // between the rows' stores; short launches are faster through the LDS form
    if (p.stats) __syncthreads();   // the statistics' block sums reuse the staging buffers
    conv_epilogue_direct_groups<MT, NT, WARPS_N * NT * 32>(p, tc, dec, acc, wm, tc.cout0 + wn * (NT * 32), lane, lds);
    return;
  }
  __syncthreads();   // the epilogue regions alias the staging buffers other waves may still read
  conv_epilogue<WARPS_N, MT, NT>(p, tc, dec, acc, lds, wm, wn, lane);
}

// ---------------------------------------------------------------------------------
// Kernel B: LDS-DMA staged (global_load_lds, 16 B per lane), BK = 64, for plain bf16
// operands with Cin % 64 == 0.  Each wave-instruction fills 8 rows of 128 B; the LDS
// image is lane-linear, so the chunk swizzle is applied to the per-lane SOURCE address.
// Taps outside the image read a line of zeros instead.  Tile k+1 is in flight while the
// MFMAs of tile k run; one vmcnt(0) + barrier per K-step.
// ---------------------------------------------------------------------------------
template <int WARPS_M, int WARPS_N, int MT, int NT, bool BNM = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_dma_kernel(const ConvParams p) {
  constexpr int BM = WARPS_M * MT * 32;
  constexpr int BN = WARPS_N * NT * 32;
  constexpr int NA = BM / 32;  // A wave-instructions per wave per stage
  constexpr int NB = BN / 32;
  static_assert(WARPS_M * WARPS_N == 4, "4 waves per block");
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int BOFF = BM * 128;
  constexpr int EPI = 4 * 32 * (NT * 32 + 4) * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[cmax(2 * STAGE, EPI)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WARPS_N;
  const int wn = wave % WARPS_N;
  const TileCoord tc = tile_coord<BM, BN>(p);

  const int KK = p.KH * p.KW;
  const int nchunk = p.Cin >> 6;
  const int ksteps = KK * nchunk;
  const int HoWo = p.Ho * p.Wo;

  // lane -> (row within the 8-row group, 16-byte slot); instruction j of this wave fills
  // rows [(4j + wave)*8, +8) of the A (or B) tile.
  const int r8 = lane >> 3;
  const int slot = lane & 7;

  const h16_t* a_ptr[NA];
  int a_iy0[NA], a_ix0[NA];
  bool a_ok[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int row = (4 * j + wave) * 8 + r8;
    const int m = tc.m0 + row;
    a_ok[j] = m < p.M;
    const int mm = a_ok[j] ? m : 0;
    const int n = mm / HoWo;
    const int rem = mm - n * HoWo;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    a_iy0[j] = oy * p.stride_y - tc.pad_y;
    a_ix0[j] = ox * p.stride_x - tc.pad_x;
    const int ch = slot ^ ((row >> 1) & 7);
    a_ptr[j] = p.in + n * p.in_img_stride + (long long)a_iy0[j] * p.in_row_stride +
               (long long)a_ix0[j] * p.in_pix_stride + ch * 8;
  }
  const h16_t* w_ptr[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int row = (4 * j + wave) * 8 + r8;
    const int co = tc.cout0 + row;
    const int ch = slot ^ ((row >> 1) & 7);
    w_ptr[j] = co < p.Cout ? tc.wbase + (long long)co * KK * p.Cin + ch * 8 : nullptr;
  }
  const h16_t* zero_src = reinterpret_cast<const h16_t*>(g_zero_line) + slot * 8;

  int dy = 0, dx = 0, cc = 0;
  auto issue = [&](int ks, int buf) {
    const int aoff = dy * p.in_row_stride + dx * p.in_pix_stride + cc * 64;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const bool ok = a_ok[j] && (unsigned)(a_iy0[j] + dy) < (unsigned)p.H &&
                      (unsigned)(a_ix0[j] + dx) < (unsigned)p.W;
      const h16_t* src = ok ? a_ptr[j] + aoff : zero_src;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(lds + buf * STAGE + (4 * j + wave) * 1024),
          16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const h16_t* src = w_ptr[j] ? w_ptr[j] + ks * 64 : zero_src;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(lds + buf * STAGE + BOFF +
                                                    (4 * j + wave) * 1024),
          16, 0, 0);
    }
    if (++cc == nchunk) {
      cc = 0;
      if (++dx == p.KW) {
        dx = 0;
        ++dy;
      }
    }
  };

  f32x16 acc[NT][MT];
  zero_acc<MT, NT>(acc);
  const int lrow = lane & 31;
  const int lhalf = lane >> 5;

  auto compute = [&](int buf) __attribute__((always_inline)) {
    // fragment reads run two 16-deep sub-steps ahead of the MFMAs that consume them
    h16x8 xf[2][MT], wf[2][NT];
#define MISEG_RD(KK)                                                                        \
  {                                                                                         \
    const int ch = (KK) * 2 + lhalf;                                                        \
    _Pragma("unroll") for (int b = 0; b < MT; ++b) xf[(KK) & 1][b] =                        \
        *reinterpret_cast<const h16x8*>(lds + buf * STAGE + swz128(wm * (MT * 32) + b * 32 + lrow, ch));                                               \
    _Pragma("unroll") for (int a = 0; a < NT; ++a) wf[(KK) & 1][a] =                        \
        *reinterpret_cast<const h16x8*>(lds + buf * STAGE + BOFF + swz128(wn * (NT * 32) + a * 32 + lrow, ch));                                               \
  }
    MISEG_RD(0)
    MISEG_RD(1)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
          acc[a][b] = mfma_32x32x16(wf[kk & 1][a], xf[kk & 1][b], acc[a][b]);
      if (kk + 2 < 4) MISEG_RD(kk + 2)
    }
#undef MISEG_RD
  };

  issue(0, 0);
  for (int ks = 0; ks < ksteps; ++ks) {
    // tile ks has landed (own DMA retired, then everyone's after the barrier); the barrier
    // also means every wave is done reading the other buffer, which tile ks+1 now overwrites
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ks + 1 < ksteps) issue(ks + 1, (ks + 1) & 1);
    compute(ks & 1);
  }
  const LinearDecode dec{tc.m0, p.M, HoWo, p.Wo};
  if constexpr (BNM) {
    // data gradient behind relu(bn(x)): the masking form of the LDS epilogue
    __syncthreads();
    conv_epilogue<WARPS_N, MT, NT, LinearDecode, true>(p, tc, dec, acc, lds, wm, wn, lane);
    return;
  }
  if (p.direct && direct_epilogue_ok(p)) {
    // large, bandwidth-bound launches (full-resolution 1x1 layers): no LDS round trip, no waits
    // between the rows' stores; short launches are faster through the LDS form
    if (p.stats) __syncthreads();   // the statistics' block sums reuse the staging buffers
    conv_epilogue_direct_groups<MT, NT, WARPS_N * NT * 32>(p, tc, dec, acc, wm, tc.cout0 + wn * (NT * 32), lane, lds);
    return;
  }
  __syncthreads();   // the epilogue regions alias the staging buffers other waves may still read
  conv_epilogue<WARPS_N, MT, NT>(p, tc, dec, acc, lds, wm, wn, lane);
}

// ---------------------------------------------------------------------------------
// Kernel C: 3x3, stride 1, pad 1 with the input halo tile resident in LDS.  A block of 8
// waves computes a TH x TW patch of output pixels for BN couts.  Per 64-channel chunk the
// (TH+2) x (TW+2) input halo is brought into LDS once by LDS-DMA and serves all nine
// taps (the pixel operand of tap (dy,dx) is the same image read at a shifted row), so
// the global->LDS traffic per K-step is one [BN][64] weight tile plus 1/9 of a halo
// instead of a [BM][64] activation tile.  Weight tiles are double buffered per tap, the
// next chunk's halo is fetched one wave-instruction per tap step.
// ---------------------------------------------------------------------------------

template <int TH, int TW, int WARPS_M, int WARPS_N, int MT, int NT, int HB, bool CONVT>
__global__ __launch_bounds__(WARPS_M * WARPS_N * 64) void conv3x3_halo_kernel(const ConvParams p) {
  // CONVT: one phase (a,b) of Conv2DTranspose(4, 2, 'same'): a 2x2 convolution whose taps
  // (dy,dx) read halo pixel (y + a + dy, x + b + dx); same halo, four taps per chunk.
  constexpr int TAPS = CONVT ? 4 : 9;
  constexpr int TAPW = CONVT ? 2 : 3;
  // HB = 2: halo double buffered (next chunk fetched during the current one's taps);
  // HB = 1: one halo buffer, refilled between chunks -- half the LDS, so two blocks share a
  //         CU and cover each other's refill.
  constexpr int NW = WARPS_M * WARPS_N;
  static_assert(NW == 8 || NW == 4, "4 or 8 waves per block");
  constexpr int BM = TH * TW;
  constexpr int BN = WARPS_N * NT * 32;
  static_assert(BM == WARPS_M * MT * 32, "pixel tile must match the wave layout");
  constexpr int HWD = TW + 2;
  constexpr int HROWS = (TH + 2) * HWD;          // halo pixels, one 128-B LDS row each
  constexpr int NHI = (HROWS + 7) / 8;           // halo wave-instructions per chunk
  constexpr int NHW = (NHI + NW - 1) / NW;       // ... per wave
  static_assert(HB == 1 || NHW <= TAPS, "next halo must fit in the tap steps of a chunk");
  constexpr int HALO_BYTES = NHI * 1024;
  constexpr int NBI = BN / 8;                    // weight wave-instructions per K-step
  constexpr int NBW = (NBI + NW - 1) / NW;
  constexpr int W_BYTES = BN * 128;
  constexpr int WOFF = HB * HALO_BYTES;
  constexpr int EPI = NW * 32 * (NT * 32 + 4) * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[cmax(HB * HALO_BYTES + 2 * W_BYTES, EPI)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WARPS_N;
  const int wn = wave % WARPS_N;
  const TileCoord tc = tile_coord<BM, BN>(p);
  const int mtile = tc.m0 / BM;
  const int per_img = p.tiles_y * p.tiles_x;
  const int n = mtile / per_img;
  const int trem = mtile - n * per_img;
  const int tyi = trem / p.tiles_x;
  const int y0 = tyi * TH;
  const int x0 = (trem - tyi * p.tiles_x) * TW;

  const int nchunk = p.Cin >> 6;
  const int steps = nchunk * TAPS;
  const int r8 = lane >> 3;
  const int slot = lane & 7;
  const h16_t* zero_src = reinterpret_cast<const h16_t*>(g_zero_line) + slot * 8;

  // halo sources: instruction j of this wave fills halo rows [(j*NW + wave)*8, +8)
  const h16_t* h_src[NHW];
#pragma unroll
  for (int j = 0; j < NHW; ++j) {
    const int hr = (j * NW + wave) * 8 + r8;
    const int hy = hr / HWD;
    const int hx = hr - hy * HWD;
    const int iy = y0 - 1 + hy;
    const int ix = x0 - 1 + hx;
    const bool ok = hr < HROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const int ch = slot ^ ((hr >> 1) & 7);
    h_src[j] = ok ? p.in + n * p.in_img_stride + (long long)iy * p.in_row_stride +
                        (long long)ix * p.in_pix_stride + ch * 8
                  : nullptr;
  }
  const h16_t* w_src[NBW];
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int row = (j * NW + wave) * 8 + r8;
    const int co = tc.cout0 + row;
    const int ch = slot ^ ((row >> 1) & 7);
    w_src[j] = (row < BN && co < p.Cout) ? tc.wbase + (long long)co * TAPS * p.Cin + ch * 8 : nullptr;
  }

  auto issue_halo = [&](int j, int chunk, int hb) {
    if ((j * NW + wave) < NHI) {
      const h16_t* src = h_src[j] ? h_src[j] + chunk * 64 : zero_src;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(lds + hb * HALO_BYTES + (j * NW + wave) * 1024),
          16, 0, 0);
    }
  };
  auto issue_w = [&](int tap, int chunk, int wb) {
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      if ((j * NW + wave) < NBI) {
        const h16_t* src = w_src[j] ? w_src[j] + tap * p.Cin + chunk * 64 : zero_src;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)src,
            (__attribute__((address_space(3))) void*)(lds + WOFF + wb * W_BYTES +
                                                      (j * NW + wave) * 1024),
            16, 0, 0);
      }
    }
  };

  f32x16 acc[NT][MT];
  zero_acc<MT, NT>(acc);
  const int lrow = lane & 31;
  const int lhalf = lane >> 5;
  int hr0[MT];   // halo row of this lane's pixel for tap (0,0)
#pragma unroll
  for (int b = 0; b < MT; ++b) {
    const int pidx = (wm * MT + b) * 32 + lrow;
    hr0[b] = (pidx / TW + (CONVT ? tc.ph_a : 0)) * HWD + (pidx % TW) + (CONVT ? tc.ph_b : 0);
  }

  auto compute = [&](int hb, int wb, int dy, int dx) __attribute__((always_inline)) {
    const unsigned char* hbase = lds + hb * HALO_BYTES;
    const unsigned char* wbase = lds + WOFF + wb * W_BYTES;
    int hrow[MT], hsw[MT];
#pragma unroll
    for (int b = 0; b < MT; ++b) {
      hrow[b] = hr0[b] + dy * HWD + dx;
      hsw[b] = (hrow[b] >> 1) & 7;
    }
    // fragment reads run two 16-deep sub-steps ahead of the MFMAs that consume them
    h16x8 xf[2][MT], wf[2][NT];
#define MISEG_RD(KK)                                                                        \
  {                                                                                         \
    const int ch = (KK) * 2 + lhalf;                                                        \
    _Pragma("unroll") for (int b = 0; b < MT; ++b) xf[(KK) & 1][b] =                        \
        *reinterpret_cast<const h16x8*>(hbase + hrow[b] * 128 + ((ch ^ hsw[b]) << 4));                                               \
    _Pragma("unroll") for (int a = 0; a < NT; ++a) wf[(KK) & 1][a] =                        \
        *reinterpret_cast<const h16x8*>(wbase + swz128(wn * (NT * 32) + a * 32 + lrow, ch));                                               \
  }
    MISEG_RD(0)
    MISEG_RD(1)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
          acc[a][b] = mfma_32x32x16(wf[kk & 1][a], xf[kk & 1][b], acc[a][b]);
      if (kk + 2 < 4) MISEG_RD(kk + 2)
    }
#undef MISEG_RD
  };

#pragma unroll
  for (int j = 0; j < NHW; ++j) issue_halo(j, 0, 0);
  issue_w(0, 0, 0);
  int chunk = 0, tap = 0, dy = 0, dx = 0;
  for (int s = 0; s < steps; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // next step's coordinates
    int ntap = tap + 1, nchk = chunk;
    if (ntap == TAPS) {
      ntap = 0;
      nchk = chunk + 1;
    }
    if (s + 1 < steps) issue_w(ntap, nchk, (s + 1) & 1);
    if (HB == 2 && chunk + 1 < nchunk && tap < NHW) {
      // one halo instruction of the next chunk per tap step
#pragma unroll
      for (int j = 0; j < NHW; ++j)
        if (j == tap) issue_halo(j, chunk + 1, (chunk + 1) & 1);
    }
    compute(HB == 2 ? (chunk & 1) : 0, s & 1, dy, dx);
    if (HB == 1 && tap == TAPS - 1 && chunk + 1 < nchunk) {
      __syncthreads();   // every wave is done with this chunk's halo
#pragma unroll
      for (int j = 0; j < NHW; ++j) issue_halo(j, chunk + 1, 0);
    }
    tap = ntap;
    chunk = nchk;
    if (++dx == TAPW) {
      dx = 0;
      if (++dy == TAPW) dy = 0;
    }
  }
  const PatchDecode<TH, TW> dec{n, y0, x0, p.Ho, p.Wo};
  // Straight from the accumulators (no LDS, no waits between the rows' stores) where it pays: wave
  // tiles of two cout groups.  The short launches of the one-group tiles (DenseNet's 128 -> 32
  // layers) measured 5 % slower with it.
  if (NT >= 2 && direct_epilogue_ok(p)) {
    conv_epilogue_direct_groups<MT, NT, WARPS_N * NT * 32>(p, tc, dec, acc, wm, tc.cout0 + wn * (NT * 32), lane, lds);
    return;
  }
  __syncthreads();   // the epilogue regions alias the staging buffers other waves may still read
  conv_epilogue<WARPS_N, MT, NT>(p, tc, dec, acc, lds, wm, wn, lane);
}

// ---------------------------------------------------------------------------------
// Kernel D: 3x3, stride 1, pad 1, Cin = 32, Cout <= 32 (the full-resolution decoder conv
// and the segmentation head).  These layers are bandwidth-bound: a block brings the
// (TH+2) x (TW+2) x 32 halo and all nine 32x32 weight taps into LDS with one LDS-DMA burst,
// synchronises once, and runs the 9 x 2 MFMA sub-steps without further barriers; four
// blocks per CU overlap each other's load, compute and store phases.  (A persistent
// variant with the weights resident and the next halo prefetched measured no faster: two
// 80 KB blocks per CU keep fewer bytes in flight than four 40 KB ones.)
// ---------------------------------------------------------------------------------
template <int TH, int TW>
__global__ __launch_bounds__(256) void conv3x3_c32_kernel(const ConvParams p) {
  constexpr int BM = TH * TW;
  constexpr int MT = BM / 128;
  static_assert(BM % 128 == 0, "patch must split over 4 waves of 32-pixel tiles");
  constexpr int HWD = TW + 2;
  constexpr int HROWS = (TH + 2) * HWD;
  constexpr int NHI = (HROWS + 15) / 16;   // wave-instructions of 16 rows x 64 B
  constexpr int NHW = (NHI + 3) / 4;
  constexpr int HALO_BYTES = NHI * 1024;
  constexpr int NWI = 9 * 32 / 16;
  constexpr int NWW = (NWI + 3) / 4;
  constexpr int W_BYTES = NWI * 1024;
  constexpr int EPI = 4 * 32 * (32 + 4) * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[cmax(HALO_BYTES + W_BYTES, EPI)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const TileCoord tc = tile_coord<BM, 32>(p);
  const int mtile = tc.m0 / BM;
  const int per_img = p.tiles_y * p.tiles_x;
  const int n = mtile / per_img;
  const int trem = mtile - n * per_img;
  const int tyi = trem / p.tiles_x;
  const int y0 = tyi * TH;
  const int x0 = (trem - tyi * p.tiles_x) * TW;
  const int r16 = lane >> 2;
  const int slot = lane & 3;
  const h16_t* zero_src = reinterpret_cast<const h16_t*>(g_zero_line) + slot * 8;

#pragma unroll
  for (int j = 0; j < NHW; ++j) {
    const int i = j * 4 + wave;
    if (i < NHI) {
      const int hr = i * 16 + r16;
      const int hy = hr / HWD;
      const int hx = hr - hy * HWD;
      const int iy = y0 - 1 + hy;
      const int ix = x0 - 1 + hx;
      const bool ok = hr < HROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const int ch = slot ^ ((hr >> 2) & 3);
      const h16_t* src = ok ? p.in + n * p.in_img_stride + (long long)iy * p.in_row_stride +
                                   (long long)ix * p.in_pix_stride + ch * 8
                             : zero_src;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lds + i * 1024),
                                       16, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < NWW; ++j) {
    const int i = j * 4 + wave;
    if (i < NWI) {
      const int wr = i * 16 + r16;      // LDS row = tap * 32 + cout
      const int tap = wr >> 5;
      const int co = wr & 31;
      const int ch = slot ^ ((wr >> 2) & 3);
      const h16_t* src = co < p.Cout ? tc.wbase + ((long long)co * 9 + tap) * 32 + ch * 8 : zero_src;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(lds + HALO_BYTES + i * 1024), 16, 0, 0);
    }
  }

  f32x16 acc[1][MT];
  zero_acc<MT, 1>(acc);
  const int lrow = lane & 31;
  const int lhalf = lane >> 5;
  int hr0[MT];
#pragma unroll
  for (int b = 0; b < MT; ++b) {
    const int pidx = (wave * MT + b) * 32 + lrow;
    hr0[b] = (pidx / TW) * HWD + (pidx % TW);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3, dx = tap % 3;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = kk * 2 + lhalf;
      const h16x8 wf = *reinterpret_cast<const h16x8*>(lds + HALO_BYTES + swz64(tap * 32 + lrow, ch));
#pragma unroll
      for (int b = 0; b < MT; ++b) {
        const int hrow = hr0[b] + dy * HWD + dx;
        const h16x8 xf = *reinterpret_cast<const h16x8*>(lds + swz64(hrow, ch));
        acc[0][b] = mfma_32x32x16(wf, xf, acc[0][b]);
      }
    }
  }
  const PatchDecode<TH, TW> dec{n, y0, x0, p.Ho, p.Wo};
  __syncthreads();   // the epilogue regions alias the staging buffers other waves may still read
  conv_epilogue<1, MT, 1>(p, tc, dec, acc, lds, wave, 0, lane);
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
struct TileCfg {
  int bm, bn;
};
static const TileCfg kCfgs[] = {
    {128, 128},  // 1
    {256, 64},   // 2
    {256, 32},   // 3
    {64, 64},    // 4
    {128, 64},   // 5
    {128, 32},   // 6
    {64, 128},   // 7
};

template <int WM, int WN, int MT, int NT>
static hipError_t launch_cfg(const ConvParams& p, dim3 grid, bool pro, bool dma, hipStream_t s) {
#if !MISEG_BUILD_F16
  // data gradients behind relu(bn(x)) (training runs on bf16 storage): instantiations of their own
  if (p.res_mask_c && !pro) {
    if (dma)
      hipLaunchKernelGGL((conv_igemm_dma_kernel<WM, WN, MT, NT, true>), grid, dim3(256), 0, s, p);
    else
      hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, MT, NT, false, true>), grid, dim3(256), 0, s, p);
    return hipGetLastError();
  }
#endif
  if (pro)
    hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, MT, NT, true>), grid, dim3(256), 0, s, p);
  else if (dma)
    hipLaunchKernelGGL((conv_igemm_dma_kernel<WM, WN, MT, NT>), grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, MT, NT, false>), grid, dim3(256), 0, s, p);
  return hipGetLastError();
}

static int choose_cfg(int M, int Cout) {
  // Prefer the widest cout tile the layer fills, then the tallest pixel tile that still
  // leaves at least ~2 blocks per CU.  (With two blocks resident per CU the 128x128 tile is within a few
  // per cent of the best of all seven on 1x1, strided and Cin = 32 layers of 19 K .. 20 M pixels.)
  auto blocks = [&](int bm, int bn) {
    return (long long)((M + bm - 1) / bm) * ((Cout + bn - 1) / bn);
  };
  const long long want = 512;
  if (Cout > 64) {
    if (blocks(128, 128) >= want) return 1;
    if (blocks(64, 128) >= want) return 7;
    return 4;
  }
  if (Cout > 32) {
    if (blocks(256, 64) >= want) return 2;
    if (blocks(128, 64) >= want) return 5;
    return 4;
  }
  if (blocks(256, 32) >= want) return 3;
  return 6;
}

int validate_conv(const miseg_conv_desc* d) {
  MISEG_CHECK_ARG(d != nullptr, "conv: null descriptor");
  MISEG_CHECK_ARG(d->in && d->weight, "conv: null input or weight pointer");
  MISEG_CHECK_ARG(d->out || d->out2, "conv: no output pointer");
  MISEG_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0,
                  "conv: non-positive dimension");
  if (d->Cin <= 0 || (d->Cin & 31)) {
    set_error("conv: Cin=%d must be a positive multiple of 32", d->Cin);
    return MISEG_E_UNSUPPORTED_SHAPE;
  }
  MISEG_CHECK_ARG(d->KH > 0 && d->KW > 0 && d->stride_y > 0 && d->stride_x > 0,
                  "conv: bad kernel/stride");
  MISEG_CHECK_ARG(d->in_valid_c >= 0 && (d->in_valid_c & 7) == 0 && d->in_valid_c < d->Cin,
                  "conv: in_valid_c must be 0 or a multiple of 8 below Cin");
  MISEG_CHECK_ARG((d->in_pix_stride & 7) == 0 && (d->in_row_stride & 7) == 0 &&
                      (d->in_img_stride & 7) == 0,
                  "conv: input strides must be multiples of 8 elements (16-byte loads)");
  MISEG_CHECK_ARG(((uintptr_t)d->in & 15) == 0 && ((uintptr_t)d->weight & 15) == 0,
                  "conv: input/weight must be 16-byte aligned");
  if (d->transposed) {
    MISEG_CHECK_ARG(d->KH == 2 && d->KW == 2 && d->stride_y == 1 && d->stride_x == 1 &&
                        d->Ho == d->H && d->Wo == d->W,
                    "conv: transposed mode needs KH=KW=2, stride 1, Ho=H, Wo=W");
    if (d->transposed == MISEG_CONV_DGRAD_T) {
      MISEG_CHECK_ARG((d->Cin & 63) == 0 && !d->in_scale && !d->collapse_out,
                      "conv: transpose-conv data gradient needs Cin %% 64 == 0 and no operand prologue");
      MISEG_CHECK_ARG((long long)d->H * 2 * d->in_row_stride * 2 < (1ll << 31) &&
                          (long long)d->Cout * 16 * d->Cin * 2 < (1ll << 31),
                      "conv: transpose-conv data gradient: image or weight set too large for 32-bit offsets");
    } else {
      MISEG_CHECK_ARG(d->transposed == 1, "conv: transposed must be 0, 1 or MISEG_CONV_DGRAD_T");
    }
  } else {
    // every output pixel's window must start inside the padded image
    MISEG_CHECK_ARG((d->Ho - 1) * d->stride_y - d->pad_y < d->H &&
                        (d->Wo - 1) * d->stride_x - d->pad_x < d->W,
                    "conv: output extent exceeds input extent");
  }
  if (d->act == MISEG_ACT_SOFTMAX || d->act == MISEG_ACT_SIGMOID) {
    MISEG_CHECK_ARG(d->Cout <= 4 && d->out && d->out_dtype == MISEG_DT_F32 && !d->out2,
                    "conv: softmax/sigmoid epilogue needs Cout<=4, fp32 out, no out2");
    MISEG_CHECK_ARG(d->temperature > 0.f, "conv: temperature must be positive");
  } else if (d->Cout % 4 == 0) {
    if (d->out) {
      const int al = d->out_dtype == MISEG_DT_F32 ? 3 : 3;
      MISEG_CHECK_ARG((d->out_pix_stride & al) == 0 && (d->out_row_stride & al) == 0 &&
                          (d->out_img_stride & al) == 0,
                      "conv: output strides must be multiples of 4 elements");
      MISEG_CHECK_ARG(((uintptr_t)d->out & (d->out_dtype == MISEG_DT_F32 ? 15 : 7)) == 0,
                      "conv: output pointer misaligned");
    }
    if (d->out2) {
      MISEG_CHECK_ARG(d->scale2 && d->shift2, "conv: out2 needs scale2/shift2");
      MISEG_CHECK_ARG((d->out2_pix_stride & 3) == 0 && (d->out2_row_stride & 3) == 0 &&
                          (d->out2_img_stride & 3) == 0 && ((uintptr_t)d->out2 & 7) == 0,
                      "conv: out2 strides/pointer misaligned");
    }
    if (d->residual) {
      MISEG_CHECK_ARG((d->res_pix_stride & 3) == 0 && (d->res_row_stride & 3) == 0 &&
                          (d->res_img_stride & 3) == 0 &&
                          ((uintptr_t)d->residual & (d->res_dtype == MISEG_DT_F32 ? 15 : 7)) == 0,
                      "conv: residual strides/pointer misaligned");
    }
  } else if (d->out2) {
    MISEG_CHECK_ARG(d->scale2 && d->shift2, "conv: out2 needs scale2/shift2");
  }
  MISEG_CHECK_ARG(d->res_upsample_shift >= 0 && d->res_upsample_shift <= 4 && (!d->res_upsample_shift || d->residual),
                  "conv: res_upsample_shift must be 0..4 and needs a residual");
  if (d->res_after_act)
    MISEG_CHECK_ARG(d->residual && !d->out2 && d->act != MISEG_ACT_SOFTMAX && d->act != MISEG_ACT_SIGMOID,
                    "conv: res_after_act needs a residual, no second output and no head activation");
  if (d->in_scale || d->in_shift) {
    MISEG_CHECK_ARG(d->in_scale && d->in_shift, "conv: prologue needs both in_scale and in_shift");
    MISEG_CHECK_ARG(((uintptr_t)d->in_scale & 15) == 0 && ((uintptr_t)d->in_shift & 15) == 0,
                    "conv: prologue vectors must be 16-byte aligned");
  }
  MISEG_CHECK_ARG(d->res_mode == MISEG_RES_ADD || d->res_mode == MISEG_RES_BN_RELU_MASK, "conv: unknown res_mode %d",
                  (int)d->res_mode);
  if (d->res_mode == MISEG_RES_BN_RELU_MASK) {
    MISEG_CHECK_ARG(d->residual && d->res_dtype == MISEG_DT_BF16 && d->scale && d->shift && d->out &&
                        d->out_dtype == MISEG_DT_BF16 && !d->out2 && d->act == MISEG_ACT_NONE && !d->res_after_act &&
                        !d->res_upsample_shift,
                    "conv: the BatchNorm-ReLU mask needs x as a 16-bit residual, scale and shift, a 16-bit output, "
                    "no activation and no second output");
    MISEG_CHECK_ARG(d->transposed != 1, "conv: the BatchNorm-ReLU mask is for data gradients, not for transpose convolutions");
    MISEG_CHECK_ARG(d->res_mask_c > 0 && (d->res_mask_c & 7) == 0 && d->res_mask_c <= d->Cout && (d->Cout & 7) == 0,
                    "conv: res_mask_c must be a positive multiple of 8, at most Cout (a multiple of 8)");
    MISEG_CHECK_ARG((d->out_pix_stride & 7) == 0 && (d->out_row_stride & 7) == 0 && (d->out_img_stride & 7) == 0 &&
                        ((uintptr_t)d->out & 15) == 0 && (d->res_pix_stride & 7) == 0 && (d->res_row_stride & 7) == 0 &&
                        (d->res_img_stride & 7) == 0 && ((uintptr_t)d->residual & 15) == 0,
                    "conv: the BatchNorm-ReLU mask needs 16-byte aligned output and residual strides");
  }
  if (d->stats) {
    const int stat_c = d->res_mode == MISEG_RES_BN_RELU_MASK ? d->res_mask_c : d->Cout;
    MISEG_CHECK_ARG(d->out && d->out_dtype == MISEG_DT_BF16 && (d->Cout & 7) == 0 && d->stats_replicas > 0 &&
                        d->stats_stride >= stat_c && ((uintptr_t)d->stats & 7) == 0,
                    "conv: stats need a bf16 output, Cout % 8 == 0, replicas > 0, stride >= Cout");
    MISEG_CHECK_ARG((d->out_pix_stride & 7) == 0 && (d->out_row_stride & 7) == 0 && (d->out_img_stride & 7) == 0 &&
                        ((uintptr_t)d->out & 15) == 0 && !d->out2 &&
                        (!d->residual || ((d->res_pix_stride & 7) == 0 && (d->res_row_stride & 7) == 0 &&
                                          (d->res_img_stride & 7) == 0 && ((uintptr_t)d->residual & 15) == 0)),
                    "conv: stats need 16-byte aligned output (and residual) strides and no second output");
  }
  if (d->scale) MISEG_CHECK_ARG(((uintptr_t)d->scale & 15) == 0, "conv: scale misaligned");
  if (d->shift) MISEG_CHECK_ARG(((uintptr_t)d->shift & 15) == 0, "conv: shift misaligned");
  const long long M = (long long)d->N * d->Ho * d->Wo;
  if (M >= (1ll << 31)) {
    set_error("conv: N*Ho*Wo=%lld exceeds 2^31", M);
    return MISEG_E_UNSUPPORTED_SHAPE;
  }
  MISEG_CHECK_ARG(d->tile_hint >= 0 && (d->tile_hint & 15) <= 15, "conv: bad tile_hint");
  return MISEG_OK;
}

enum KernelKind { K_REG = 0, K_REG_PRO = 1, K_DMA = 2, K_HALO = 3, K_C32 = 4, K_HALO_PW = 5 };

hipError_t launch_halo_pw(const ConvParams& p, dim3 grid, hipStream_t s);
void halo_pw_kernel_name(const ConvParams& p, char* buf, int len);
bool conv_c32p_ok(const ConvParams& p);
hipError_t launch_conv_c32p(const ConvParams& p, hipStream_t s);
bool conv_c64p_ok(const ConvParams& p);
hipError_t launch_conv_c64p(ConvParams p, hipStream_t s);
bool conv_stem7_ok(const ConvParams& p);
hipError_t launch_conv_stem7(ConvParams p, hipStream_t s);
bool conv_pw_pro_ok(const ConvParams& p);
hipError_t launch_conv_pw_pro(ConvParams p, hipStream_t s);
bool conv_pw_thin_ok(const ConvParams& p);
hipError_t launch_conv_pw_thin(ConvParams p, hipStream_t s);
// short-contraction 1x1 layers go to the persistent point-wise kernel (MISEG_NO_PW_THIN keeps them on the tiles)
static bool pw_thin_wanted(const miseg_conv_desc* d, const ConvParams& p, KernelKind kind) {
  static const bool off = getenv("MISEG_NO_PW_THIN") != nullptr;
  return !off && kind == K_DMA && (d->tile_hint & 15) == 0 && !p.res_mask_c && conv_pw_thin_ok(p);
}
struct KernelChoice {
  KernelKind kind;
  int cfg;  // 1..7 index into kCfgs; 8..10 halo variants
};

static KernelChoice choose_kernel_base(const miseg_conv_desc* d) {
  const int hint = d->tile_hint & 15;
  const bool pro = d->in_scale != nullptr;
  const bool halo_ok = !pro && !d->transposed && d->KH == 3 && d->KW == 3 && d->stride_y == 1 &&
                       d->stride_x == 1 && d->pad_y == 1 && d->pad_x == 1 && d->Ho == d->H &&
                       d->Wo == d->W && (d->Cin & 63) == 0 && !(d->tile_hint & MISEG_TILE_NO_HALO) &&
                       !(d->tile_hint & MISEG_TILE_NO_DMA);
  const bool c32_ok = !pro && !d->transposed && d->KH == 3 && d->KW == 3 && d->stride_y == 1 &&
                      d->stride_x == 1 && d->pad_y == 1 && d->pad_x == 1 && d->Ho == d->H &&
                      d->Wo == d->W && d->Cin == 32 && d->Cout <= 32 &&
                      !(d->tile_hint & MISEG_TILE_NO_HALO);
  if (c32_ok && (hint == 0 || hint == 14)) return {K_C32, 14};
  // wave-private weight rings: per-lane source offsets are 32-bit byte offsets from the image base
  const bool pw_ok = hint == 0 && (d->Cout > 32 || d->transposed) && !(d->tile_hint & MISEG_TILE_SHARED_W) &&
                     (long long)d->H * d->in_row_stride * 2 < (1ll << 31) &&
                     (long long)d->Cout * 16 * d->Cin * 2 < (1ll << 31);
  const bool convt_ok = !pro && d->transposed && (d->Cin & 63) == 0 &&
                        !(d->tile_hint & MISEG_TILE_NO_HALO) && !(d->tile_hint & MISEG_TILE_NO_DMA);
  if ((convt_ok || halo_ok) && pw_ok) return {K_HALO_PW, 0};
  if (convt_ok && (hint == 0 || hint == 11 || hint == 12 || hint == 15)) {
    if (hint) return {K_HALO, hint};
    return {K_HALO, d->Cout > 64 ? 11 : (d->Cout > 32 ? 12 : 15)};
  }
  if (halo_ok && (hint == 0 || (hint >= 8 && hint <= 13) || hint == 15)) {
    if (hint >= 8) return {K_HALO, hint};
    if (d->Cout > 64) {
      const long long blocks = (long long)d->N * ((d->Ho + 7) / 8) * ((d->Wo + 31) / 32) *
                               ((d->Cout + 127) / 128);
      // few, short blocks: the 8-wave 8x16 patch keeps more CUs busy; otherwise the 4-wave
      // 8x32 patch with 128x64 wave tiles (two blocks per CU) has the best MFMA density
      return {K_HALO, (blocks >= 512 || d->Cin >= 512) ? 11 : 10};
    }
    if (d->Cout > 32) return {K_HALO, 12};
    return {K_HALO, 15};
  }
  const int M = d->N * d->Ho * d->Wo;
  const int cfg = (hint >= 1 && hint <= 7) ? hint : choose_cfg(M, d->Cout);
  if (pro) return {K_REG_PRO, cfg};
  if ((d->Cin & 63) == 0 && !(d->tile_hint & MISEG_TILE_NO_DMA)) return {K_DMA, cfg};
  return {K_REG, cfg};
}

// A prologue (in_scale: the input is relu(in * in_scale + in_shift)) lives in the register-staged generic kernel
// and, for the 32-channel 3x3 layers it can take, in the persistent 32-channel kernel (conv_c32p_ok decides, on
// the filled-in parameters): there the layer is chosen as if it had no prologue.
static KernelChoice choose_kernel(const miseg_conv_desc* d, const ConvParams& p) {
  if (d->in_scale && !(d->tile_hint & MISEG_TILE_NO_PERSIST)) {
    miseg_conv_desc e = *d;
    e.in_scale = e.in_shift = nullptr;
    const KernelChoice k = choose_kernel_base(&e);
    if (k.kind == K_C32 && conv_c32p_ok(p)) return k;
  }
  return choose_kernel_base(d);
}

static void fill_params(const miseg_conv_desc* d, ConvParams& p);

// Name of the kernel instantiation launch_conv() picks for `d`, as rocprofv3 prints it.
void conv_kernel_name(const miseg_conv_desc* d, char* buf, int len) {
  ConvParams pp;
  fill_params(d, pp);
  if (d->transposed == MISEG_CONV_DGRAD_T) {
    pp.transposed = 0;
    pp.dgrad_t = 1;
    halo_pw_kernel_name(pp, buf, len);
    return;
  }
  const KernelChoice kc = choose_kernel(d, pp);
  const bool persist = !(d->tile_hint & MISEG_TILE_NO_PERSIST);
  if ((d->tile_hint & 15) == 0 && persist && conv_stem7_ok(pp)) {
    snprintf(buf, len, "conv_stem7_kernel");
    return;
  }
  if (kc.kind == K_C32 && persist && conv_c32p_ok(pp)) {
    if (pp.in_scale)
      snprintf(buf, len, "conv3x3_c32p_kernel<false, true>");
    else
      snprintf(buf, len, "conv3x3_c32p_kernel<%s, false>", pp.res_mask_c && !MISEG_BUILD_F16 ? "true" : "false");
    return;
  }
  if (kc.kind == K_REG_PRO && (d->tile_hint & 15) == 0 && conv_pw_pro_ok(pp)) {
    snprintf(buf, len, "conv_pw_pro_kernel");
    return;
  }
  if (pw_thin_wanted(d, pp, kc.kind)) {
    snprintf(buf, len, "conv_pw_thin_kernel<%d, 2>", d->Cin / 16);
    return;
  }
  static const char* shapes[] = {"2, 2, 2, 2", "4, 1, 2, 2", "4, 1, 2, 1", "2, 2, 1, 1",
                                 "4, 1, 1, 2", "4, 1, 1, 1", "2, 2, 1, 2"};
  static const char* halos[] = {"8, 32, 4, 2, 2, 2, 2", "8, 32, 4, 2, 2, 1, 2", "8, 16, 4, 2, 1, 2, 2",
                                "8, 32, 2, 2, 4, 2, 1", "8, 32, 4, 1, 2, 2, 1", "8, 16, 2, 2, 2, 2, 2",
                                "", "8, 32, 4, 1, 2, 1, 1"};
  if (kc.kind == K_C32)
    snprintf(buf, len, "conv3x3_c32_kernel<8, 32>");
  else if (kc.kind == K_HALO_PW && persist && conv_c64p_ok(pp))
    snprintf(buf, len, "conv3x3_c64p_kernel<%s, %s>", pp.stats ? "true" : "false", pp.res_mask_c ? "true" : "false");
  else if (kc.kind == K_HALO_PW)
    halo_pw_kernel_name(pp, buf, len);
  else if (kc.kind == K_HALO)
    snprintf(buf, len, "conv3x3_halo_kernel<%s, %s>", halos[kc.cfg - 8], d->transposed ? "true" : "false");
  else if (kc.kind == K_DMA)
    snprintf(buf, len, "conv_igemm_dma_kernel<%s, %s>", shapes[kc.cfg - 1],
             pp.res_mask_c && !MISEG_BUILD_F16 ? "true" : "false");
  else
    snprintf(buf, len, "conv_igemm_kernel<%s, %s, %s>", shapes[kc.cfg - 1], kc.kind == K_REG_PRO ? "true" : "false",
             pp.res_mask_c && !MISEG_BUILD_F16 && kc.kind != K_REG_PRO ? "true" : "false");
}

static void fill_params(const miseg_conv_desc* d, ConvParams& p) {
  p.in = (const h16_t*)d->in;
  p.weight = (const h16_t*)d->weight;
  p.out = d->out;
  p.out2 = (h16_t*)d->out2;
  p.residual = d->residual;
  p.scale = d->scale;
  p.shift = d->shift;
  p.scale2 = d->scale2;
  p.shift2 = d->shift2;
  p.in_scale = d->in_scale;
  p.in_shift = d->in_shift;
  p.collapse_out = (uint8_t*)d->collapse_out;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin;
  p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.KH = d->KH; p.KW = d->KW;
  p.stride_y = d->stride_y; p.stride_x = d->stride_x;
  p.pad_y = d->pad_y; p.pad_x = d->pad_x;
  p.in_pix_stride = d->in_pix_stride; p.in_row_stride = d->in_row_stride;
  p.in_img_stride = d->in_img_stride;
  p.out_pix_stride = d->out_pix_stride; p.out_row_stride = d->out_row_stride;
  p.out_img_stride = d->out_img_stride;
  p.out2_pix_stride = d->out2_pix_stride; p.out2_row_stride = d->out2_row_stride;
  p.out2_img_stride = d->out2_img_stride;
  p.res_pix_stride = d->res_pix_stride; p.res_row_stride = d->res_row_stride;
  p.res_img_stride = d->res_img_stride;
  p.out_dtype = d->out_dtype; p.res_dtype = d->res_dtype;
  p.act = d->act; p.transposed = d->transposed;
  p.inv_temperature = d->act == MISEG_ACT_SOFTMAX ? 1.f / d->temperature : 1.f;
  p.collapse_value = d->collapse_value;
  p.stats = (double*)d->stats;
  p.stats_rep = d->stats_replicas;
  p.stats_stride = d->stats_stride;
  p.M = d->N * d->Ho * d->Wo;
  p.dgrad_t = 0;
  p.res_after = d->res_after_act ? 1 : 0;
  p.res_shift = d->res_upsample_shift;
  p.out2_linear = d->out2_linear ? 1 : 0;
  p.in_valid_c = d->in_valid_c > 0 ? d->in_valid_c : d->Cin;
  p.res_mask_c = d->res_mode == MISEG_RES_BN_RELU_MASK ? d->res_mask_c : 0;
  // one million output pixels, or 64 million output values: from there on the epilogue's HBM traffic, not its
  // latency, is what a generic launch waits for (measured on the full-resolution 1x1 layers of LinkNet and FPN:
  // 1.1-2.2x faster; LinkNet's 416 -> 640 layer at 120x160x8)
  p.direct = (long long)d->N * d->Ho * d->Wo >= (1 << 20) || (long long)d->N * d->Ho * d->Wo * d->Cout >= (64ll << 20);
  auto al = [](const void* ptr, long long a, long long b, long long c, int mask) {
    return (((uintptr_t)ptr & 15) == 0) && ((a | b | c) & mask) == 0;
  };
  p.vec_ok = 1;
  if (d->out)
    p.vec_ok &= al(d->out, d->out_pix_stride, d->out_row_stride, d->out_img_stride,
                   d->out_dtype == MISEG_DT_F32 ? 3 : 7);
  if (d->out2)
    p.vec_ok &= al(d->out2, d->out2_pix_stride, d->out2_row_stride, d->out2_img_stride, 7);
  if (d->residual)
    p.vec_ok &= al(d->residual, d->res_pix_stride, d->res_row_stride, d->res_img_stride,
                   d->res_dtype == MISEG_DT_F32 ? 3 : 7);
  if (d->scale2) p.vec_ok &= ((uintptr_t)d->scale2 & 15) == 0 && ((uintptr_t)d->shift2 & 15) == 0;
  if (d->act == MISEG_ACT_SOFTMAX || d->act == MISEG_ACT_SIGMOID) p.vec_ok = 0;
}

int launch_conv(const miseg_conv_desc* d, hipStream_t stream) {
  int rc = validate_conv(d);
  if (rc != MISEG_OK) return rc;
  ConvParams p;
  fill_params(d, p);
  if (d->in_valid_c > 0 && (d->transposed || choose_kernel(d, p).kind != K_C32 || !conv_c32p_ok(p) ||
                            (d->tile_hint & MISEG_TILE_NO_PERSIST))) {
    set_error("conv: in_valid_c needs a 3x3 stride-1 layer with Cin = 32, Cout <= 32, 16-bit output, no residual");
    return MISEG_E_UNSUPPORTED_SHAPE;
  }
  if (p.res_mask_c && (MISEG_BUILD_F16 || !p.vec_ok)) {
    set_error("conv: the BatchNorm-ReLU mask needs bf16 storage and 16-byte aligned tensors");
    return MISEG_E_UNSUPPORTED_SHAPE;
  }
  if (d->transposed == MISEG_CONV_DGRAD_T) {
    // dX = sum over the four phase lattices of dY: one launch of the ring kernel in its data-gradient mode
    p.transposed = 0;
    p.dgrad_t = 1;
    p.tiles_y = (d->Ho + 7) / 8;
    p.tiles_x = (d->Wo + 31) / 32;
    p.mtiles = d->N * p.tiles_y * p.tiles_x;
    p.ntiles = (d->Cout + 127) / 128;
    const hipError_t e2 = launch_halo_pw(p, dim3(p.mtiles * p.ntiles), stream);
    if (e2 != hipSuccess) {
      set_error("conv launch failed: %s", hipGetErrorString(e2));
      return MISEG_E_HIP;
    }
    return MISEG_OK;
  }
  const KernelChoice kc = choose_kernel(d, p);
  const int cfg = kc.cfg;
  const bool pro = kc.kind == K_REG_PRO;
  const bool dma = kc.kind == K_DMA;
  if (p.res_mask_c) {
    // the BatchNorm-ReLU mask exists in the ring kernels, the persistent 32-channel kernel and the generic tiles
    // (bf16 storage); a layer that would run elsewhere is declined, and the caller runs the two passes instead
    const bool tiles = (kc.kind == K_REG || kc.kind == K_DMA) && cfg >= 1 && cfg <= 7 && !MISEG_BUILD_F16;
    const bool ring = kc.kind == K_HALO_PW && !d->transposed && !MISEG_BUILD_F16;     // (or the persistent 64-channel kernel)
    const bool c32p = kc.kind == K_C32 && conv_c32p_ok(p) && !(d->tile_hint & MISEG_TILE_NO_PERSIST) && !MISEG_BUILD_F16;
    if (!(tiles || ring || c32p) || !p.vec_ok) {
      set_error("conv: the BatchNorm-ReLU mask is not built for the kernel this layer runs on");
      return MISEG_E_UNSUPPORTED_SHAPE;
    }
  }
  p.tiles_y = p.tiles_x = 0;
  if (kc.kind == K_C32) {
    p.tiles_y = (d->Ho + 7) / 8;
    p.tiles_x = (d->Wo + 31) / 32;
    p.mtiles = d->N * p.tiles_y * p.tiles_x;
    p.ntiles = 1;
  } else if (kc.kind == K_HALO_PW) {
    p.tiles_y = (d->Ho + 7) / 8;
    p.tiles_x = (d->Wo + 31) / 32;
    p.mtiles = d->N * p.tiles_y * p.tiles_x;
    p.ntiles = d->transposed ? (d->Cout + 31) / 32 : (d->Cout <= 64 ? 1 : (d->Cout + 127) / 128);
  } else if (kc.kind == K_HALO) {
    const int tw = (cfg == 10 || cfg == 13) ? 16 : 32;
    const int bn = cfg == 15 ? 32 : ((cfg == 9 || cfg == 12) ? 64 : 128);
    p.tiles_y = (d->Ho + 7) / 8;
    p.tiles_x = (d->Wo + tw - 1) / tw;
    p.mtiles = d->N * p.tiles_y * p.tiles_x;
    p.ntiles = (d->Cout + bn - 1) / bn;
  } else {
    const TileCfg tc = kCfgs[cfg - 1];
    p.mtiles = (p.M + tc.bm - 1) / tc.bm;
    p.ntiles = (d->Cout + tc.bn - 1) / tc.bn;
  }
  // the wave-private-ring kernel computes the four transpose-conv phases in one block
  const dim3 grid(p.mtiles * p.ntiles * ((d->transposed && kc.kind != K_HALO_PW) ? 4 : 1), 1, 1);
  hipError_t e;
  if ((d->tile_hint & 15) == 0 && !(d->tile_hint & MISEG_TILE_NO_PERSIST) && conv_stem7_ok(p)) {
    e = launch_conv_stem7(p, stream);
    if (e != hipSuccess) {
      set_error("conv launch failed: %s", hipGetErrorString(e));
      return MISEG_E_HIP;
    }
    return MISEG_OK;
  }
  if (kc.kind == K_REG_PRO && (d->tile_hint & 15) == 0 && conv_pw_pro_ok(p)) {
    e = launch_conv_pw_pro(p, stream);
    if (e != hipSuccess) {
      set_error("conv launch failed: %s", hipGetErrorString(e));
      return MISEG_E_HIP;
    }
    return MISEG_OK;
  }
  if (pw_thin_wanted(d, p, kc.kind)) {
    e = launch_conv_pw_thin(p, stream);
    if (e != hipSuccess) {
      set_error("conv launch failed: %s", hipGetErrorString(e));
      return MISEG_E_HIP;
    }
    return MISEG_OK;
  }
  if (kc.kind == K_C32 && conv_c32p_ok(p) && !(d->tile_hint & MISEG_TILE_NO_PERSIST)) {
    e = launch_conv_c32p(p, stream);
    if (e != hipSuccess) {
      set_error("conv launch failed: %s", hipGetErrorString(e));
      return MISEG_E_HIP;
    }
    return MISEG_OK;
  }
  if (kc.kind == K_HALO_PW && conv_c64p_ok(p) && !(d->tile_hint & MISEG_TILE_NO_PERSIST)) {
    e = launch_conv_c64p(p, stream);
    if (e != hipSuccess) {
      set_error("conv launch failed: %s", hipGetErrorString(e));
      return MISEG_E_HIP;
    }
    return MISEG_OK;
  }
  if (kc.kind == K_HALO_PW) {
    e = launch_halo_pw(p, grid, stream);
    if (e != hipSuccess) {
      set_error("conv launch failed: %s", hipGetErrorString(e));
      return MISEG_E_HIP;
    }
    return MISEG_OK;
  }
  switch (cfg) {
    case 1: e = launch_cfg<2, 2, 2, 2>(p, grid, pro, dma, stream); break;
    case 2: e = launch_cfg<4, 1, 2, 2>(p, grid, pro, dma, stream); break;
    case 3: e = launch_cfg<4, 1, 2, 1>(p, grid, pro, dma, stream); break;
    case 4: e = launch_cfg<2, 2, 1, 1>(p, grid, pro, dma, stream); break;
    case 5: e = launch_cfg<4, 1, 1, 2>(p, grid, pro, dma, stream); break;
    case 6: e = launch_cfg<4, 1, 1, 1>(p, grid, pro, dma, stream); break;
    case 7: e = launch_cfg<2, 2, 1, 2>(p, grid, pro, dma, stream); break;
    case 8:
      hipLaunchKernelGGL((conv3x3_halo_kernel<8, 32, 4, 2, 2, 2, 2, false>), grid, dim3(512), 0, stream, p);
      e = hipGetLastError();
      break;
    case 9:
      hipLaunchKernelGGL((conv3x3_halo_kernel<8, 32, 4, 2, 2, 1, 2, false>), grid, dim3(512), 0, stream, p);
      e = hipGetLastError();
      break;
    case 10:
      hipLaunchKernelGGL((conv3x3_halo_kernel<8, 16, 4, 2, 1, 2, 2, false>), grid, dim3(512), 0, stream, p);
      e = hipGetLastError();
      break;
    case 11:
      if (d->transposed)
        hipLaunchKernelGGL((conv3x3_halo_kernel<8, 32, 2, 2, 4, 2, 1, true>), grid, dim3(256), 0, stream, p);
      else
        hipLaunchKernelGGL((conv3x3_halo_kernel<8, 32, 2, 2, 4, 2, 1, false>), grid, dim3(256), 0, stream, p);
      e = hipGetLastError();
      break;
    case 12:
      if (d->transposed)
        hipLaunchKernelGGL((conv3x3_halo_kernel<8, 32, 4, 1, 2, 2, 1, true>), grid, dim3(256), 0, stream, p);
      else
        hipLaunchKernelGGL((conv3x3_halo_kernel<8, 32, 4, 1, 2, 2, 1, false>), grid, dim3(256), 0, stream, p);
      e = hipGetLastError();
      break;
    case 15:
      if (d->transposed)
        hipLaunchKernelGGL((conv3x3_halo_kernel<8, 32, 4, 1, 2, 1, 1, true>), grid, dim3(256), 0, stream, p);
      else
        hipLaunchKernelGGL((conv3x3_halo_kernel<8, 32, 4, 1, 2, 1, 1, false>), grid, dim3(256), 0, stream, p);
      e = hipGetLastError();
      break;
    case 13:
      hipLaunchKernelGGL((conv3x3_halo_kernel<8, 16, 2, 2, 2, 2, 2, false>), grid, dim3(256), 0, stream, p);
      e = hipGetLastError();
      break;
    default:
      hipLaunchKernelGGL((conv3x3_c32_kernel<8, 32>), grid, dim3(256), 0, stream, p);
      e = hipGetLastError();
      break;
  }
  if (e != hipSuccess) {
    set_error("conv launch failed: %s", hipGetErrorString(e));
    return MISEG_E_HIP;
  }
  return MISEG_OK;
}

hipError_t launch_c32_head(const ConvParams& body, const ConvParams& head, dim3 grid, hipStream_t s);

// The last U-Net decoder convolution (3x3, 32 -> 32, scale/shift, ReLU) and the segmentation
// head (3x3, 32 -> classes <= 4, softmax / sigmoid, optional collapse mask) in one kernel.
int validate_conv_c32_head(const miseg_conv_desc* b, const miseg_conv_desc* h) {
  MISEG_CHECK_ARG(b && h, "conv_c32_head: null descriptor");
  MISEG_CHECK_ARG(b->in && b->weight && h->weight && h->out, "conv_c32_head: null pointer");
  MISEG_CHECK_ARG(b->N > 0 && b->H > 0 && b->W > 0, "conv_c32_head: non-positive dimension");
  auto shape3x3 = [](const miseg_conv_desc* d) {
    return d->KH == 3 && d->KW == 3 && d->stride_y == 1 && d->stride_x == 1 && d->pad_y == 1 &&
           d->pad_x == 1 && d->Ho == d->H && d->Wo == d->W && !d->transposed && d->Cin == 32;
  };
  if (!shape3x3(b) || !shape3x3(h) || b->Cout != 32 || h->Cout < 1 || h->Cout > 4) {
    set_error("conv_c32_head: needs 3x3 stride-1 pad-1 convolutions 32 -> 32 -> (1..4) channels");
    return MISEG_E_UNSUPPORTED_SHAPE;
  }
  MISEG_CHECK_ARG(h->N == b->N && h->H == b->H && h->W == b->W, "conv_c32_head: body and head extents differ");
  MISEG_CHECK_ARG(!b->residual && !b->out2 && !b->in_scale && !h->residual && !h->out2 && !h->in_scale,
                  "conv_c32_head: residual, second output and input prologue are not supported");
  MISEG_CHECK_ARG(b->act == MISEG_ACT_NONE || b->act == MISEG_ACT_RELU, "conv_c32_head: body activation");
  MISEG_CHECK_ARG((h->act == MISEG_ACT_SOFTMAX || h->act == MISEG_ACT_SIGMOID) &&
                      h->out_dtype == MISEG_DT_F32 && h->temperature > 0.f,
                  "conv_c32_head: head needs softmax/sigmoid, fp32 output, positive temperature");
  MISEG_CHECK_ARG((b->in_pix_stride & 7) == 0 && (b->in_row_stride & 7) == 0 && (b->in_img_stride & 7) == 0 &&
                      ((uintptr_t)b->in & 15) == 0 && ((uintptr_t)b->weight & 15) == 0 &&
                      ((uintptr_t)h->weight & 15) == 0,
                  "conv_c32_head: input/weights need 16-byte alignment and strides in multiples of 8");
  if (b->scale) MISEG_CHECK_ARG(((uintptr_t)b->scale & 15) == 0, "conv_c32_head: scale misaligned");
  if (b->shift) MISEG_CHECK_ARG(((uintptr_t)b->shift & 15) == 0, "conv_c32_head: shift misaligned");
  return MISEG_OK;
}

int launch_conv_c32_head(const miseg_conv_desc* b, const miseg_conv_desc* h, hipStream_t stream) {
  int rc = validate_conv_c32_head(b, h);
  if (rc != MISEG_OK) return rc;
  ConvParams pb, ph;
  fill_params(b, pb);
  fill_params(h, ph);
  pb.tiles_y = (b->H + 7) / 8;
  pb.tiles_x = (b->W + 31) / 32;
  pb.mtiles = b->N * pb.tiles_y * pb.tiles_x;
  pb.ntiles = 1;
  pb.vec_ok = ph.vec_ok = 0;
  ph.tiles_y = pb.tiles_y; ph.tiles_x = pb.tiles_x; ph.mtiles = pb.mtiles; ph.ntiles = 1;
  // persistent blocks, two per CU
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int blocks = pb.mtiles < 2 * cus ? pb.mtiles : 2 * cus;
  hipError_t e = launch_c32_head(pb, ph, dim3(blocks), stream);
  if (e != hipSuccess) {
    set_error("conv_c32_head launch failed: %s", hipGetErrorString(e));
    return MISEG_E_HIP;
  }
  return MISEG_OK;
}

hipError_t launch_dense_layer_kernel(const ConvParams& c1, const ConvParams& c2, dim3 grid, hipStream_t s);
int dense_layer_max_cin();

// One DenseNet layer (BatchNorm + ReLU -> 1x1, C -> 128 -> BatchNorm + ReLU -> 3x3, 128 -> 32) in one
// kernel; c1 describes the 1x1 convolution with its operand prologue, c2 the 3x3 convolution.
int validate_dense_layer(const miseg_conv_desc* a, const miseg_conv_desc* b) {
  MISEG_CHECK_ARG(a && b, "dense_layer: null descriptor");
  MISEG_CHECK_ARG(a->in && a->weight && b->weight && b->out, "dense_layer: null pointer");
  MISEG_CHECK_ARG(a->N > 0 && a->H > 0 && a->W > 0, "dense_layer: non-positive dimension");
  const bool s1 = a->KH == 1 && a->KW == 1 && a->stride_y == 1 && a->stride_x == 1 && a->pad_y == 0 &&
                  a->pad_x == 0 && a->Ho == a->H && a->Wo == a->W && !a->transposed && a->Cout == 128 &&
                  a->Cin > 0 && (a->Cin & 31) == 0 && a->Cin <= dense_layer_max_cin();
  const bool s2 = b->KH == 3 && b->KW == 3 && b->stride_y == 1 && b->stride_x == 1 && b->pad_y == 1 &&
                  b->pad_x == 1 && b->Ho == b->H && b->Wo == b->W && !b->transposed && b->Cin == 128 &&
                  b->Cout == 32;
  if (!s1 || !s2) {
    set_error("dense_layer: needs a 1x1 convolution C -> 128 (C %% 32 == 0, C <= %d) and a 3x3 stride-1 pad-1 "
              "convolution 128 -> 32", dense_layer_max_cin());
    return MISEG_E_UNSUPPORTED_SHAPE;
  }
  MISEG_CHECK_ARG(b->N == a->N && b->H == a->H && b->W == a->W, "dense_layer: the two convolutions' extents differ");
  MISEG_CHECK_ARG(a->in_scale && a->in_shift && ((uintptr_t)a->in_scale & 15) == 0 && ((uintptr_t)a->in_shift & 15) == 0,
                  "dense_layer: needs 16-byte aligned in_scale / in_shift");
  MISEG_CHECK_ARG(a->act == MISEG_ACT_RELU && !a->residual && !a->out2 && !a->stats,
                  "dense_layer: the 1x1 convolution takes scale/shift + ReLU only");
  MISEG_CHECK_ARG(b->act == MISEG_ACT_NONE && !b->residual && !b->out2 && !b->in_scale && !b->stats &&
                      b->out_dtype == MISEG_DT_BF16,
                  "dense_layer: the 3x3 convolution takes a plain 16-bit output (optional scale/shift)");
  MISEG_CHECK_ARG((a->in_pix_stride & 7) == 0 && (a->in_row_stride & 7) == 0 && (a->in_img_stride & 7) == 0 &&
                      ((uintptr_t)a->in & 15) == 0 && ((uintptr_t)a->weight & 15) == 0 && ((uintptr_t)b->weight & 15) == 0,
                  "dense_layer: input/weights need 16-byte alignment and strides in multiples of 8");
  MISEG_CHECK_ARG((b->out_pix_stride & 7) == 0 && (b->out_row_stride & 7) == 0 && (b->out_img_stride & 7) == 0 &&
                      ((uintptr_t)b->out & 15) == 0,
                  "dense_layer: output needs 16-byte alignment and strides in multiples of 8");
  MISEG_CHECK_ARG((long long)a->H * a->in_row_stride * 2 < (1ll << 31) && (long long)128 * a->Cin * 2 < (1ll << 31),
                  "dense_layer: image too large for 32-bit offsets");
  if (a->scale) MISEG_CHECK_ARG(a->shift != nullptr, "dense_layer: scale without shift");
  return MISEG_OK;
}

int launch_dense_layer(const miseg_conv_desc* a, const miseg_conv_desc* b, hipStream_t stream) {
  int rc = validate_dense_layer(a, b);
  if (rc != MISEG_OK) return rc;
  ConvParams pa, pb;
  fill_params(a, pa);
  fill_params(b, pb);
  pa.tiles_y = (a->H + 7) / 8;
  pa.tiles_x = (a->W + 15) / 16;
  pa.mtiles = a->N * pa.tiles_y * pa.tiles_x;
  pa.ntiles = 1;
  pb.tiles_y = pa.tiles_y; pb.tiles_x = pa.tiles_x; pb.mtiles = pa.mtiles; pb.ntiles = 1;
  hipError_t e = launch_dense_layer_kernel(pa, pb, dim3(pa.mtiles), stream);
  if (e != hipSuccess) {
    set_error("dense_layer launch failed: %s", hipGetErrorString(e));
    return MISEG_E_HIP;
  }
  return MISEG_OK;
}

}  // namespace MISEG_KNS
