// 3x3 stride-1 convolution (and the 2x2 phases of Conv2DTranspose(4, 2, 'same')) with the
// input halo tile resident in LDS and WAVE-PRIVATE weight rings: no workgroup barrier inside
// a 64-channel chunk.  Replaces the tf.keras Conv2D / Conv2DTranspose layers instantiated
// through /root/reference utils/model.py:176-211 for Cin % 64 == 0.
//
// A block of 4 waves computes an 8 x 32 patch of output pixels for 32 * WARPS_N couts; a
// wave owns MT pixel rows of the patch and ONE 32-cout group, so its accumulators are
// MT 32x32 tiles and every K sub-step is MT MFMAs fed by MT pixel fragments + 1 weight
// fragment.
//
//  * Halo: the (8+2) x (32+2) pixels x 64 channels of the current chunk, one 144-byte LDS row
//    per pixel (128 B of data + 16 B of padding).  With that pitch the 16 lanes a ds_read_b128
//    serves per cycle fall on 16 distinct 16-byte slots with plain linear addressing, so the
//    address of fragment (row b, tap dy/dx, sub-step kk) is one per-lane base plus a
//    compile-time immediate: no address arithmetic in the main loop.  Brought in by LDS-DMA
//    (lane-linear LDS image; lanes that land on the padding fetch the zero line).
//  * Weights: each wave streams the [32 couts][64 k] slice of every (tap, chunk) step it needs
//    into its own two-slot ring by LDS-DMA, one step ahead, and waits only on its own vmcnt.
//    Waves that share a cout group fetch the slice twice (from L2); in exchange the waves of
//    a block drift freely between the two barriers that bracket a halo swap.
//  * Fragment reads roll 8 MFMAs ahead of the MFMAs that consume them, across tap boundaries
//    too.
//
// LDS per block: 48 KB halo + 32 KB weight rings = 80 KB, so two blocks share a CU and cover
// each other's halo swaps, prologues and epilogues.
#include "conv_common.h"

namespace miseg {

namespace {

constexpr int PW_TH = 8, PW_TW = 32;
constexpr int PW_HWD = PW_TW + 2;
constexpr int PW_HROWS = (PW_TH + 2) * PW_HWD;   // 340 halo pixels
constexpr int PW_PITCH = 144;                    // bytes per halo pixel row in LDS
constexpr int PW_NHI = (PW_HROWS * PW_PITCH + 1023) / 1024;   // 48 halo wave-instructions
constexpr int PW_HALO_BYTES = PW_NHI * 1024;
constexpr int PW_WST = 32 * 128;                 // one ring slot: [32 couts][64 k] bf16
constexpr int PW_LDS = PW_HALO_BYTES + 4 * 2 * PW_WST;
static_assert(2 * PW_LDS <= 160 * 1024, "two blocks per CU");

}  // namespace

template <int WARPS_M, int MT, bool CONVT>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_pw_kernel(const ConvParams p) {
  constexpr int TH = PW_TH, TW = PW_TW, HWD = PW_HWD, PITCH = PW_PITCH;
  constexpr int WARPS_N = 4 / WARPS_M;
  static_assert(WARPS_M == 1 && MT == TH, "a wave owns all 8 rows of the patch");
  constexpr int TAPS = CONVT ? 4 : 9;
  constexpr int TAPW = CONVT ? 2 : 3;
  constexpr int BM = TH * TW;
  constexpr int BN = CONVT ? 32 : WARPS_N * 32;   // transpose-conv: the four waves are the four phases
  constexpr int NHW = PW_NHI / 4;                 // halo wave-instructions per wave
  static_assert(PW_NHI % 4 == 0, "halo instructions split evenly over the waves");
  constexpr int WOFF = PW_HALO_BYTES;
  constexpr int EPI = 4 * 32 * (32 + 4) * 4;
  static_assert(EPI <= PW_LDS, "epilogue regions alias the staging buffers");
  __shared__ __attribute__((aligned(16))) unsigned char lds[PW_LDS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WARPS_N;
  const int wn = wave % WARPS_N;
  TileCoord tc = tile_coord<BM, BN, false>(p);
  const int cgrp = CONVT ? 0 : wn;               // 32-cout group of this wave inside the block
  if (CONVT) {
    tc.ph_a = wn >> 1;
    tc.ph_b = wn & 1;
    tc.wbase = p.weight + (long long)wn * p.Cout * 4 * p.Cin;
  }
  const int mtile = tc.m0 / BM;
  const int per_img = p.tiles_y * p.tiles_x;
  const int n = mtile / per_img;
  const int trem = mtile - n * per_img;
  const int tyi = trem / p.tiles_x;
  const int y0 = tyi * TH;
  const int x0 = (trem - tyi * p.tiles_x) * TW;
  const int nchunk = p.Cin >> 6;

  const unsigned char* zero_src = g_zero_line;
  const unsigned char* img = reinterpret_cast<const unsigned char*>(p.in + n * p.in_img_stride);

  // halo sources: instruction j of this wave fills LDS bytes [(4j + wave) * 1024, +1024); a
  // lane's 16 bytes belong to pixel row o / 144, channel slot (o % 144) / 16 (slot 8 = padding).
  // Byte offsets from the image base (checked on the host to fit 31 bits); -1 = zero line.
  int h_off[NHW];
#pragma unroll
  for (int j = 0; j < NHW; ++j) {
    const int o = (4 * j + wave) * 1024 + lane * 16;
    const int hr = o / PITCH;
    const int sl = (o - hr * PITCH) >> 4;
    const int hy = hr / HWD;
    const int hx = hr - hy * HWD;
    const int iy = y0 - 1 + hy;
    const int ix = x0 - 1 + hx;
    const bool ok = hr < PW_HROWS && sl < 8 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    h_off[j] = ok ? (iy * p.in_row_stride + ix * p.in_pix_stride + sl * 8) * 2 : -1;
  }
  // weight sources: instruction j fills ring rows [8j, 8j + 8) (row = cout of this wave's group);
  // the 16-byte chunk index is XOR-swizzled with the row on the SOURCE side (the LDS image of
  // an LDS-DMA is lane-linear).  Byte offsets from the phase's weight base.
  int w_off[4];
  {
    const int r8 = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = j * 8 + r8;
      const int co = tc.cout0 + cgrp * 32 + row;
      const int ch = slot ^ ((row >> 1) & 7);
      w_off[j] = co < p.Cout ? (co * TAPS * p.Cin + ch * 8) * 2 : -1;
    }
  }
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(tc.wbase);
  unsigned char* wring = lds + WOFF + wave * (2 * PW_WST);

  auto issue_halo = [&](int chunk) {
#pragma unroll
    for (int j = 0; j < NHW; ++j) {
      const unsigned char* src = h_off[j] >= 0 ? img + (unsigned)h_off[j] + chunk * 128 : zero_src;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(lds + (4 * j + wave) * 1024), 16, 0, 0);
    }
  };
  auto issue_w = [&](int tap, int chunk, int slot) {
    const int koff = (tap * p.Cin + chunk * 64) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned char* src = w_off[j] >= 0 ? wsrc + (unsigned)w_off[j] + koff : zero_src;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(wring + slot * PW_WST + j * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[1][MT];
  zero_acc<MT, 1>(acc);
  const int lrow = lane & 31;
  const int lhalf = lane >> 5;
  // pixel fragment (b, dy, dx, kk) = xlane + ((b + dy) * HWD + dx) * PITCH + kk * 32
  const unsigned char* xlane =
      lds + ((wm * MT + (CONVT ? tc.ph_a : 0)) * HWD + lrow + (CONVT ? tc.ph_b : 0)) * PITCH + lhalf * 16;
  // weight fragment kk of ring slot s = wring + s * PW_WST + wlane[kk]
  int wlane[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) wlane[kk] = swz128(lrow, kk * 2 + lhalf);

  // Fragment reads are issued by hand (the compiler's own wait placement drains lgkmcnt to 0 at
  // every sub-step, exposing the latency of the two reads issued last): a pixel fragment is
  // read one sub-step (8 MFMAs) ahead of the MFMA that consumes it, a weight fragment two
  // sub-steps ahead.  Per sub-step the issue order is x0 .. x7, w; LDS reads return in
  // order, so before MFMA b it is enough that all but the reads issued after x_b have landed.
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const unsigned xl = (unsigned)(unsigned long long)(lds_byte*)xlane;
  const unsigned wr = (unsigned)(unsigned long long)(lds_byte*)wring;
  bf16x8 xf[MT], wf[2];
#define MISEG_DS_READ(dst, addr, off) ds_read_b128_at<off>(dst, addr)
#define MISEG_LGKM_WAIT(frag, cnt) lgkm_wait<cnt>(frag)

  issue_halo(0);
  issue_w(0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int slot0 = 0;   // ring slot of tap 0 of the current chunk
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const bool more = chunk + 1 < nchunk;
    // fragments of the chunk's first sub-steps
    static_for<MT>([&](auto bc) __attribute__((always_inline)) {
      constexpr int b = decltype(bc)::value;
      MISEG_DS_READ(xf[b], xl, b * HWD * PITCH);
    });
    {
      const unsigned w0 = wr + slot0 * PW_WST;
      MISEG_DS_READ(wf[0], w0 + wlane[0], 0);
      MISEG_DS_READ(wf[1], w0 + wlane[1], 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]), "+v"(xf[4]), "+v"(xf[5]),
                   "+v"(xf[6]), "+v"(xf[7]), "+v"(wf[0]), "+v"(wf[1]));

    static_for<TAPS>([&](auto tc_) __attribute__((always_inline)) {
      constexpr int tap = decltype(tc_)::value;
      constexpr bool last = tap == TAPS - 1;
      const int slot = slot0 ^ (tap & 1);
      const unsigned wcur = wr + slot * PW_WST;
      const unsigned wnxt = wr + (slot ^ 1) * PW_WST;
      // next step's weights into the other slot (its last reads were waited for a step ago)
      if (!last)
        issue_w(tap + 1, chunk, slot ^ 1);
      else if (more)
        issue_w(0, chunk + 1, slot ^ 1);
      static_for<4>([&](auto kc) __attribute__((always_inline)) {
        constexpr int kk = decltype(kc)::value;
        if (kk == 2 && !last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next slot landed
        // what this sub-step reads ahead, and what the previous one did
        constexpr int ntap = tap + (kk + 1) / 4, nkk = (kk + 1) % 4;
        constexpr bool has_x = ntap < TAPS;
        constexpr bool has_w = kk < 2 || !last;
        constexpr bool first = tap == 0 && kk == 0;          // fragments came from the chunk top
        constexpr bool prev_w = kk == 0 ? true : (kk - 1 < 2 || !last);
        static_for<MT>([&](auto bc) __attribute__((always_inline)) {
          constexpr int b = decltype(bc)::value;
          if constexpr (!first) {
            // reads issued after x_b: the rest of the previous sub-step, then this one's so far
            constexpr int younger = (MT - 1 - b) + (prev_w ? 1 : 0) + (has_x ? b : 0);
            MISEG_LGKM_WAIT(xf[b], younger);
          }
          acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk & 1], xf[b], acc[0][b], 0, 0, 0);
          if constexpr (has_x)
            MISEG_DS_READ(xf[b], xl, ((b + ntap / TAPW) * HWD + ntap % TAPW) * PITCH + nkk * 32);
          if constexpr (b == MT - 1 && has_w) {
            if constexpr (kk < 2)
              MISEG_DS_READ(wf[kk & 1], wcur + wlane[kk + 2], 0);
            else
              MISEG_DS_READ(wf[kk & 1], wnxt + wlane[kk - 2], 0);
          }
        });
      });
    });
    slot0 ^= (TAPS & 1);
    if (more) {
      __syncthreads();   // every wave is done with this chunk's halo
      issue_halo(chunk + 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // halo and the next step's weights
      __syncthreads();
    }
  }
#undef MISEG_DS_READ
#undef MISEG_LGKM_WAIT
  const PatchDecode<TH, TW> dec{n, y0, x0, p.Ho, p.Wo};
  __syncthreads();   // the epilogue regions alias the staging buffers other waves may still read
  conv_epilogue<WARPS_N, MT, 1>(p, tc, dec, acc, lds, wm, cgrp, lane, wave);
}

hipError_t launch_halo_pw(const ConvParams& p, dim3 grid, hipStream_t s) {
  if (p.transposed)
    hipLaunchKernelGGL((conv3x3_halo_pw_kernel<1, 8, true>), grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((conv3x3_halo_pw_kernel<1, 8, false>), grid, dim3(256), 0, s, p);
  return hipGetLastError();
}

}  // namespace miseg
