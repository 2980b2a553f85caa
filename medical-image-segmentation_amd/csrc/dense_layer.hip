// One DenseNet layer in one kernel: BatchNorm + ReLU on the concatenated input, 1x1 convolution
// to 128 channels, BatchNorm + ReLU, 3x3 convolution to 32 channels appended to the same
// concatenation buffer (keras_applications `conv_block`: <name>_0_bn, _0_relu, _1_conv, _1_bn,
// _1_relu, _2_conv, _concat; reached from /root/reference utils/model.py:188-211).  Run as two
// launches the 128-channel tensor between the two convolutions costs 512 bytes per pixel of HBM
// traffic for a layer whose own input and output are 2 C + 64 bytes per pixel; here it lives in LDS.
//
// A block of 4 waves owns an 8 x 16 patch of output pixels (two blocks per CU):
//   A. 1x1 convolution on the (8+2) x (16+2) = 180 pixels the 3x3 needs (flattened to six
//      32-pixel MFMA tiles), K = C in chunks of 32 channels.  A chunk's pixels [192][32] and
//      weights [128][32] arrive by LDS-DMA in a three-slot ring, two chunks ahead, one barrier
//      per chunk.  The input BatchNorm + ReLU is applied to the pixel fragments in registers
//      after the LDS read (packed fp16 arithmetic with fp16 storage: 8 VALU operations per
//      fragment, which feeds two MFMAs), so the operand needs no pass of its own and no staging
//      registers.  Wave (cg, ph): couts [64 cg, 64 cg + 64) x pixel tiles 3 ph .. 3 ph + 2.
//   B. scale/shift + ReLU of the 128-channel tile into LDS (272-byte pixel pitch, rows padded to
//      a multiple of 256 bytes: conflict-free linear addressing, every tap an immediate offset),
//      pixels outside the image forced to zero (they are the 3x3's zero padding).  The tile
//      overlays the ring.
//   C. 3x3 convolution out of LDS: the [32][128] weight slice of a tap arrives by LDS-DMA in a
//      three-slot ring two taps ahead, one barrier per tap.  Wave (rp, kh): output rows
//      4 rp .. 4 rp + 3 (two 2 x 16 pixel tiles) from channels [64 kh, 64 kh + 64) of the tile;
//      the two K halves exchange one accumulator tile each through LDS and each finishes one.
// LDS: max(3 x 20 KB ring + input coefficients, 50 KB tile + 24 KB tap ring) + 1 KB = 75 KB.
#include "conv_common.h"

namespace MISEG_KNS {

namespace {
constexpr int DL_TH = 8, DL_TW = 16;
constexpr int DL_MW = DL_TW + 2, DL_MH = DL_TH + 2;
constexpr int DL_MPX = DL_MH * DL_MW;            // 180 pixels of the intermediate tile
constexpr int DL_XT = 6;                         // 32-pixel tiles covering them
static_assert(DL_XT * 32 >= DL_MPX, "pixel tiles cover the halo");
constexpr int DL_XST = DL_XT * 32 * 64;          // pixel part of a ring stage: [192][32 ch]
constexpr int DL_W1ST = 128 * 64;                // weight part: [128 couts][32 ch]
constexpr int DL_STAGE = DL_XST + DL_W1ST;
constexpr int DL_NST = 3;
constexpr int DL_RING = DL_NST * DL_STAGE;
constexpr int DL_MPITCH = 272;                   // bytes per pixel of the intermediate tile
constexpr int DL_MROW = 5120;                    // bytes per row of it (18 x 272 = 4896, padded)
static_assert(DL_MW * DL_MPITCH <= DL_MROW && DL_MROW % 256 == 0, "row pitch");
constexpr int DL_MID = DL_MH * DL_MROW;
constexpr int DL_W2_OFF = DL_MID;
constexpr int DL_W2_SLOT = 32 * 256;             // one tap: [32 couts][128 ch]
constexpr int DL_PB_END = DL_W2_OFF + 3 * DL_W2_SLOT;
constexpr int DL_ST0_OFF = DL_RING;              // input scale/shift: needed while the ring lives
static_assert(DL_RING <= DL_PB_END, "the coefficients sit behind the ring, inside the tap ring's space");
constexpr int DL_ST0_MAX = DL_PB_END - DL_ST0_OFF;
constexpr int DL_BN1_OFF = DL_PB_END;            // scale[128], shift[128] of the 1x1's BatchNorm (fp32)
constexpr int DL_LDS = DL_BN1_OFF + 1024;
static_assert(2 * DL_LDS <= 160 * 1024, "two blocks per CU");

#if MISEG_BUILD_F16
typedef _Float16 dl_coef_t;
#else
typedef float dl_coef_t;
#endif

__device__ __forceinline__ void dl_barrier() {
  // LDS hand-off between waves without draining vmcnt (pending LDS-DMA of later stages)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <int N>
__device__ __forceinline__ void dl_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void dl_dma16(const void* src, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
}  // namespace

#ifdef MISEG_DL_PROFILE
// phase timers of wave w of block b (100 MHz ticks): [0] coefficients + first stages issued, [1] the 1x1's
// chunk loop, [2] barrier + tile write, [3] the 3x3's tap loop, [4] exchange + stores
__device__ unsigned long long g_dl_prof[1 << 16][8];
#define DL_TICK(i) do { const unsigned long long t__ = wall_clock64(); dl_t[i] += t__ - dl_last; dl_last = t__; } while (0)
#else
#define DL_TICK(i) do { } while (0)
#endif

// c1: the 1x1 convolution (in = concatenation buffer, Cin = C, in_scale/in_shift = the first
// BatchNorm, weight [128][C], scale/shift = the second BatchNorm, ReLU); c2: the 3x3 convolution
// (weight [32][3][3][128], out = the 32 new channels).
__global__ __launch_bounds__(256, 2) void dense_layer_kernel(const ConvParams p, const ConvParams q) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[DL_LDS];
#ifdef MISEG_DL_PROFILE
  unsigned long long dl_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dl_last = wall_clock64();
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 31, lhalf = lane >> 5;
  const int C = p.Cin;
  const int nch = C >> 5;

  // XCD-aware patch order: blocks b and b + 8 share an L2; each XCD gets a contiguous run
  int tile;
  {
    const int ntot = p.mtiles, bid = blockIdx.x;
    const int qq = ntot >> 3, r = ntot & 7, x = bid & 7, loc = bid >> 3;
    tile = (x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq) + loc;
  }
  const int per_img = p.tiles_y * p.tiles_x;
  const int n = tile / per_img;
  const int trem = tile - n * per_img;
  const int tyi = trem / p.tiles_x;
  const int y0 = tyi * DL_TH;
  const int x0 = (trem - tyi * p.tiles_x) * DL_TW;

  // ---- coefficients into LDS (plain stores: no LDS-DMA is pending yet) ----
  {
    dl_coef_t* st0 = reinterpret_cast<dl_coef_t*>(lds + DL_ST0_OFF);
    for (int i = tid * 4; i < C; i += 1024) {
      const float4 s = *reinterpret_cast<const float4*>(p.in_scale + i);
      const float4 t = *reinterpret_cast<const float4*>(p.in_shift + i);
      st0[i] = (dl_coef_t)s.x; st0[i + 1] = (dl_coef_t)s.y; st0[i + 2] = (dl_coef_t)s.z; st0[i + 3] = (dl_coef_t)s.w;
      st0[C + i] = (dl_coef_t)t.x; st0[C + i + 1] = (dl_coef_t)t.y; st0[C + i + 2] = (dl_coef_t)t.z;
      st0[C + i + 3] = (dl_coef_t)t.w;
    }
    float* bn1 = reinterpret_cast<float*>(lds + DL_BN1_OFF);
    bn1[tid] = tid < 128 ? (p.scale ? p.scale[tid] : 1.f) : (p.shift ? p.shift[tid - 128] : 0.f);
  }

  // ---- LDS-DMA sources ----
  const unsigned char* zero_src = g_zero_line + (lane & 3) * 16;
  const unsigned char* img = reinterpret_cast<const unsigned char*>(p.in + n * p.in_img_stride);
  int x_off[3];     // pixel rows 16 i + lane / 4 of instruction i = wave, wave + 4, wave + 8
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int row = (wave + 4 * j) * 16 + (lane >> 2);
    const int hy = row / DL_MW, hx = row - hy * DL_MW;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool ok = row < DL_MPX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const int ch = (lane & 3) ^ ((row >> 2) & 3);
    x_off[j] = ok ? (iy * p.in_row_stride + ix * p.in_pix_stride + ch * 8) * 2 : -1;
  }
  int w1_off[2];    // cout rows of instruction i = wave, wave + 4
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave + 4 * j) * 16 + (lane >> 2);
    const int ch = (lane & 3) ^ ((row >> 2) & 3);
    w1_off[j] = (row * C + ch * 8) * 2;
  }
  const unsigned char* w1src = reinterpret_cast<const unsigned char*>(p.weight);
  auto issue_stage = [&](int chunk, int slot) {
    unsigned char* st = lds + slot * DL_STAGE;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      dl_dma16(x_off[j] >= 0 ? img + (unsigned)x_off[j] + chunk * 64 : zero_src, st + (wave + 4 * j) * 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      dl_dma16(w1src + (unsigned)w1_off[j] + chunk * 64, st + DL_XST + (wave + 4 * j) * 1024);
  };
  int w2_off[2];    // rows 4 i + lane / 16 of instruction i = wave, wave + 4
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave + 4 * j) * 4 + (lane >> 4);
    const int ch = (lane & 15) ^ (row & 15);
    w2_off[j] = (row * 9 * 128 + ch * 8) * 2;
  }
  const unsigned char* w2src = reinterpret_cast<const unsigned char*>(q.weight);
  auto issue_w2 = [&](int tap, int slot) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      dl_dma16(w2src + (unsigned)w2_off[j] + tap * 256, lds + DL_W2_OFF + slot * DL_W2_SLOT + (wave + 4 * j) * 1024);
  };

  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const unsigned lds_a = (unsigned)(unsigned long long)(lds_byte*)lds;

  // =============================== A: the 1x1 convolution ===============================
  const int cg = wave & 1, ph = wave >> 1;
  f32x16 acc[2][3];
  zero_acc<3, 2>(acc);
  {
    // fragment addresses inside a stage (k sub-step 0; sub-step 1 = chunk index + 2)
    unsigned xa[2][3], wa[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int b = 0; b < 3; ++b) xa[ks][b] = swz64((ph * 3 + b) * 32 + lrow, ks * 2 + lhalf);
#pragma unroll
      for (int a = 0; a < 2; ++a) wa[ks][a] = DL_XST + swz64(cg * 64 + a * 32 + lrow, ks * 2 + lhalf);
    }
    const unsigned st_a = lds_a + DL_ST0_OFF + lhalf * 8 * (unsigned)sizeof(dl_coef_t);
    const unsigned tofs = C * (unsigned)sizeof(dl_coef_t);

    issue_stage(0, 0);
    if (nch > 1) issue_stage(1, 1);
    DL_TICK(0);
    int slot = 0;
    for (int ch = 0; ch < nch; ++ch) {
      if (ch + 1 < nch) dl_vmcnt<5>(); else dl_vmcnt<0>();
      dl_barrier();
      if (ch + 2 < nch) issue_stage(ch + 2, slot >= 1 ? slot - 1 : 2);
      const unsigned sb = lds_a + slot * DL_STAGE;
      h16x8 xf[2][3], wf[2][2];
#if MISEG_BUILD_F16
      h16x8 cs[2], ct[2];
#else
      h16x8 cs[2][2], ct[2][2];     // 8 fp32 each, as raw 16-byte pairs
#endif
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const unsigned ca = st_a + (ch * 32 + ks * 16) * (unsigned)sizeof(dl_coef_t);
#if MISEG_BUILD_F16
        ds_read_b128_at<0>(cs[ks], ca);
        ds_read_b128_at<0>(ct[ks], ca + tofs);
#else
        ds_read_b128_at<0>(cs[ks][0], ca);
        ds_read_b128_at<16>(cs[ks][1], ca);
        ds_read_b128_at<0>(ct[ks][0], ca + tofs);
        ds_read_b128_at<16>(ct[ks][1], ca + tofs);
#endif
#pragma unroll
        for (int a = 0; a < 2; ++a) ds_read_b128_at<0>(wf[ks][a], sb + wa[ks][a]);
#pragma unroll
        for (int b = 0; b < 3; ++b) ds_read_b128_at<0>(xf[ks][b], sb + xa[ks][b]);
      }
      static_for<2>([&](auto kc) __attribute__((always_inline)) {
        constexpr int ks = decltype(kc)::value;
#if MISEG_BUILD_F16
        constexpr int later = ks == 0 ? 7 : 0;
        asm volatile("s_waitcnt lgkmcnt(%7)"
                     : "+v"(cs[ks]), "+v"(ct[ks]), "+v"(wf[ks][0]), "+v"(wf[ks][1]), "+v"(xf[ks][0]),
                       "+v"(xf[ks][1]), "+v"(xf[ks][2])
                     : "n"(later));
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          h16x8 v = xf[ks][b] * cs[ks] + ct[ks];
          const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
          xf[ks][b] = __builtin_elementwise_max(v, z);
        }
#else
        constexpr int later = ks == 0 ? 9 : 0;
        asm volatile("s_waitcnt lgkmcnt(%9)"
                     : "+v"(cs[ks][0]), "+v"(cs[ks][1]), "+v"(ct[ks][0]), "+v"(ct[ks][1]), "+v"(wf[ks][0]),
                       "+v"(wf[ks][1]), "+v"(xf[ks][0]), "+v"(xf[ks][1]), "+v"(xf[ks][2])
                     : "n"(later));
        {
          const float4 s0 = __builtin_bit_cast(float4, cs[ks][0]), s1 = __builtin_bit_cast(float4, cs[ks][1]);
          const float4 t0 = __builtin_bit_cast(float4, ct[ks][0]), t1 = __builtin_bit_cast(float4, ct[ks][1]);
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            uint4 v = __builtin_bit_cast(uint4, xf[ks][b]);
            v.x = pack_h16x2_raw(h16_relu(h16lo_to_f32(v.x) * s0.x + t0.x), h16_relu(h16hi_to_f32(v.x) * s0.y + t0.y));
            v.y = pack_h16x2_raw(h16_relu(h16lo_to_f32(v.y) * s0.z + t0.z), h16_relu(h16hi_to_f32(v.y) * s0.w + t0.w));
            v.z = pack_h16x2_raw(h16_relu(h16lo_to_f32(v.z) * s1.x + t1.x), h16_relu(h16hi_to_f32(v.z) * s1.y + t1.y));
            v.w = pack_h16x2_raw(h16_relu(h16lo_to_f32(v.w) * s1.z + t1.z), h16_relu(h16hi_to_f32(v.w) * s1.w + t1.w));
            xf[ks][b] = __builtin_bit_cast(h16x8, v);
          }
        }
#endif
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) acc[a][b] = mfma_32x32x16(wf[ks][a], xf[ks][b], acc[a][b]);
      });
      slot = slot == 2 ? 0 : slot + 1;
    }
  }
  DL_TICK(1);
  dl_barrier();          // every wave is done with the ring: the tile and the tap ring overlay it
  issue_w2(0, 0);
  issue_w2(1, 1);
  issue_w2(2, 2);

  // =============================== B: the intermediate tile ===============================
  {
    const unsigned bn_a = lds_a + DL_BN1_OFF + (cg * 64 + 4 * lhalf) * 4;
    unsigned mid_w[3];
    bool inside[3], valid[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int idx = (ph * 3 + b) * 32 + lrow;
      valid[b] = idx < DL_MPX;
      const int hy = idx / DL_MW, hx = idx - hy * DL_MW;
      inside[b] = (unsigned)(y0 - 1 + hy) < (unsigned)p.H && (unsigned)(x0 - 1 + hx) < (unsigned)p.W;
      mid_w[b] = lds_a + hy * DL_MROW + hx * DL_MPITCH + (cg * 64 + 4 * lhalf) * 2;
    }
    static_for<2>([&](auto ac) __attribute__((always_inline)) {
      constexpr int a = decltype(ac)::value;
      h16x8 raw[8];
      static_for<4>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        ds_read_b128_at<(a * 32 + 8 * g) * 4>(raw[g], bn_a);
        ds_read_b128_at<512 + (a * 32 + 8 * g) * 4>(raw[4 + g], bn_a);
      });
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]),
                     "+v"(raw[6]), "+v"(raw[7]));
      static_for<3>([&](auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
        if (valid[b]) {
          static_for<4>([&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            const float4 sc = __builtin_bit_cast(float4, raw[g]);
            const float4 sh = __builtin_bit_cast(float4, raw[4 + g]);
            const float v0 = h16_relu(acc[a][b][4 * g] * sc.x + sh.x);
            const float v1 = h16_relu(acc[a][b][4 * g + 1] * sc.y + sh.y);
            const float v2 = h16_relu(acc[a][b][4 * g + 2] * sc.z + sh.z);
            const float v3 = h16_relu(acc[a][b][4 * g + 3] * sc.w + sh.w);
            uint2 o = make_uint2(pack_h16x2_raw(v0, v1), pack_h16x2_raw(v2, v3));
            if (!inside[b]) o = make_uint2(0, 0);
            const unsigned wa_ = mid_w[b];     // (asm operands do not capture by themselves)
            asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(wa_), "v"(o), "n"((a * 32 + 8 * g) * 2) : "memory");
          });
        }
      });
    });
  }

  DL_TICK(2);
  // =============================== C: the 3x3 convolution ===============================
  const int rp = wave >> 1, kh = wave & 1;
  f32x16 oacc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[j][r] = 0.f;
  {
    const unsigned mid_a = lds_a + (4 * rp + (lrow >> 4)) * DL_MROW + (lrow & 15) * DL_MPITCH + kh * 128 + lhalf * 16;
    unsigned w2a[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w2a[ks] = lrow * 256 + (((kh * 8 + ks * 2 + lhalf) ^ (lrow & 15)) << 4);
    static_for<9>([&](auto tc_) __attribute__((always_inline)) {
      constexpr int tap = decltype(tc_)::value;
      constexpr int dy = tap / 3, dx = tap % 3;
      if constexpr (tap == 0) dl_vmcnt<4>();
      else if constexpr (tap < 8) dl_vmcnt<2>();
      else dl_vmcnt<0>();
      dl_barrier();      // tap 0: the tile is complete; later: every wave has left tap - 1
      if constexpr (tap >= 1 && tap + 2 <= 8) issue_w2(tap + 2, (tap + 2) % 3);
      const unsigned wb = lds_a + DL_W2_OFF + (tap % 3) * DL_W2_SLOT;
      h16x8 wf[4], xf[4][2];
      static_for<4>([&](auto kc) __attribute__((always_inline)) {
        constexpr int ks = decltype(kc)::value;
        ds_read_b128_at<0>(wf[ks], wb + w2a[ks]);
        ds_read_b128_at<(dy)*DL_MROW + dx * DL_MPITCH + ks * 32>(xf[ks][0], mid_a);
        ds_read_b128_at<(2 + dy) * DL_MROW + dx * DL_MPITCH + ks * 32>(xf[ks][1], mid_a);
      });
      static_for<4>([&](auto kc) __attribute__((always_inline)) {
        constexpr int ks = decltype(kc)::value;
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(wf[ks]), "+v"(xf[ks][0]), "+v"(xf[ks][1]) : "n"(3 * (3 - ks)));
        oacc[0] = mfma_32x32x16(wf[ks], xf[ks][0], oacc[0]);
        oacc[1] = mfma_32x32x16(wf[ks], xf[ks][1], oacc[1]);
      });
    });
  }
  DL_TICK(3);
  // no LDS-DMA is pending any more: plain LDS accesses and __syncthreads from here on
  __syncthreads();
  {
    // the two K halves of a row group: wave kh finishes pixel tile kh and hands over the other
    float4* xch = reinterpret_cast<float4*>(lds);
    const int give = kh ^ 1;
    f32x16 fin, out;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      fin[r] = kh ? oacc[1][r] : oacc[0][r];
      out[r] = kh ? oacc[0][r] : oacc[1][r];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
      xch[((rp * 2 + give) * 4 + g) * 64 + lane] = make_float4(out[4 * g], out[4 * g + 1], out[4 * g + 2], out[4 * g + 3]);
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = xch[((rp * 2 + kh) * 4 + g) * 64 + lane];
      fin[4 * g] += v.x; fin[4 * g + 1] += v.y; fin[4 * g + 2] += v.z; fin[4 * g + 3] += v.w;
    }
    DirectLane dl;
    direct_lane_init(q, 0, lane, dl);
    const int oy = y0 + 4 * rp + 2 * kh + (lrow >> 4);
    const int ox = x0 + (lrow & 15);
    const bool okp = oy < q.Ho && ox < q.Wo;
    const long long o_off = okp ? n * q.out_img_stride + (long long)oy * q.out_row_stride + (long long)ox * q.out_pix_stride : 0;
    const uint4 nores[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    direct_tile(q, dl, fin, okp, o_off, 0, false, nores, nullptr, nullptr, false);
  }
#ifdef MISEG_DL_PROFILE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // count the stores' round trip
  DL_TICK(4);
  if (lane == 0 && blockIdx.x < (1 << 14))
    for (int i = 0; i < 8; ++i) g_dl_prof[blockIdx.x * 4 + wave][i] = dl_t[i];
#endif
}

hipError_t launch_dense_layer_kernel(const ConvParams& c1, const ConvParams& c2, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL(dense_layer_kernel, grid, dim3(256), 0, s, c1, c2);
  return hipGetLastError();
}

int dense_layer_max_cin() { return DL_ST0_MAX / (2 * (int)sizeof(dl_coef_t)); }

}  // namespace MISEG_KNS
