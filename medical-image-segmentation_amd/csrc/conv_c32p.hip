// 3x3 stride-1 convolution with 32 input and <= 32 output channels at full resolution (the
// decoder's last block, training forward and data gradients; tf.keras Conv2D instantiated through
// /root/reference utils/model.py:176-186): 128 bytes of traffic per 18 KFLOP pixel, so the layer
// is bound by HBM and by how little of it a block keeps in flight.
//
// Persistent blocks, two per CU, walk over 8 x 32 output patches.
//  * Weights (9 taps x 32 k x 32 couts) stay in 72 registers of every wave for the life of the
//    block: nothing but pixels goes through LDS.
//  * The 10 x 34 pixel halo (64 B of data at an 80-byte pitch: conflict-free ds_read_b128 with
//    linear addressing) is double-buffered; the next patch's halo is requested by LDS-DMA before
//    the arithmetic of the current one.
//  * A wave owns two pixel rows: the four halo rows it needs give 24 fragments for 36 MFMAs, read
//    by hand one halo row ahead of the MFMAs that consume them.
//  * Epilogue straight from the accumulators (two 16-byte stores per tile complete the 64 bytes
//    of a pixel); BatchNorm statistics accumulate in registers over all patches of the block and
//    are flushed once.
//  * As a data gradient behind relu(bn(x)) (ConvParams::res_mask_c) the patch's x values are requested
//    right after the next patch's halo, by hand, and awaited by the wait that halo needs anyway: a
//    load the compiler sees would put a full vmcnt(0) in front of the arithmetic, i.e. behind the
//    halo prefetch.
//  * With a prologue (ConvParams::in_scale: the input is relu(in * in_scale + in_shift), the BatchNorm + ReLU
//    apply pass of the training forward that then never runs as a pass of its own) the halo is transformed
//    where it lies: thread t owns 16-byte slot t % 4 of pixels t / 4 + 64 k, so its 16 coefficients stay in
//    registers; pixels outside the image stay zero (they are the convolution's padding, not bn(0)); one
//    more barrier per patch.  LDS accesses by hand, as everywhere behind a pending LDS-DMA.
#include "conv_common.h"

namespace MISEG_KNS {

namespace {
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int CP_TH = 8, CP_TW = 32, CP_HWD = CP_TW + 2;
constexpr int CP_HROWS = (CP_TH + 2) * CP_HWD;     // 340 halo pixels
constexpr int CP_PITCH = 80;
constexpr int CP_NHI = 28;                          // halo wave-instructions (27,200 B -> 28 KB)
constexpr int CP_HALO_BYTES = CP_NHI * 1024;
constexpr int CP_LDS = 2 * CP_HALO_BYTES + 256;     // + scale[32] | shift[32]
static_assert(CP_HROWS * CP_PITCH <= CP_HALO_BYTES, "halo fits its buffer");
static_assert(2 * CP_LDS <= 160 * 1024, "two blocks per CU");
}  // namespace

#ifdef MISEG_CP_PROFILE
// phase timers of wave w of block b (100 MHz ticks): [0] top barrier, [1] halo requests,
// [2] arithmetic, [3] vmcnt wait, [4] epilogue, [6] patches
__device__ unsigned long long g_cp_prof[4096][8];
#define CP_TICK(i) do { const unsigned long long t__ = wall_clock64(); cp_t[i] += t__ - cp_last; cp_last = t__; } while (0)
#else
#define CP_TICK(i) do { } while (0)
#endif

template <bool BNM, bool PRO = false>
__global__ __launch_bounds__(256, 2) void conv3x3_c32p_kernel(const ConvParams p) {
#ifdef MISEG_CP_PROFILE
  unsigned long long cp_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cp_last = wall_clock64();
#endif
  constexpr int HWD = CP_HWD, PITCH = CP_PITCH, NHW = CP_NHI / 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[CP_LDS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per_img = p.tiles_y * p.tiles_x;
  const int lrow = lane & 31, lhalf = lane >> 5;
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const unsigned lds_a = (unsigned)(unsigned long long)(lds_byte*)lds;
  const unsigned char* zero_src = g_zero_line;

  // ---- weights: fragment [tap][kk] = cout lrow, k = kk * 16 + lhalf * 8 .. + 7 (rows past Cout
  //      repeat row 0: their results are never stored) ----
  h16x8 wreg[9][2];
  static_for<18>([&](auto ic) __attribute__((always_inline)) {
    constexpr int tap = decltype(ic)::value >> 1, kk = decltype(ic)::value & 1;
    const h16_t* wsrc = p.weight + ((lrow < p.Cout ? lrow : 0) * 9 + tap) * 32 + kk * 16 + lhalf * 8;
    wreg[tap][kk] = __builtin_bit_cast(h16x8, *reinterpret_cast<const uint4*>(wsrc));
  });
  // ---- epilogue coefficients wait in LDS between patches ----
  float* coef = reinterpret_cast<float*>(lds + 2 * CP_HALO_BYTES);
  if (tid < 64) {
    const int c = tid & 31;
    const float* src = tid < 32 ? p.scale : p.shift;
    coef[tid] = (src && c < (BNM ? p.res_mask_c : p.Cout)) ? src[c] : (tid < 32 ? 1.f : 0.f);
  }
  const unsigned coef_a = lds_a + 2 * CP_HALO_BYTES + lhalf * 32;
  DirectLane dl;
  dl.c2[0] = 8 * lhalf;
  dl.c2[1] = 16 + 8 * lhalf;
  dl.okc[0] = dl.c2[0] + 8 <= p.Cout;
  dl.okc[1] = dl.c2[1] + 8 <= p.Cout;
  dl.okm[0] = BNM && dl.c2[0] + 8 <= p.res_mask_c;
  dl.okm[1] = BNM && dl.c2[1] + 8 <= p.res_mask_c;
  constexpr bool bnm = BNM;
  dl.relu = p.act == MISEG_ACT_RELU;
  const bool head = p.act == MISEG_ACT_SOFTMAX || p.act == MISEG_ACT_SIGMOID;
  const bool stats = p.stats != nullptr && !head;
  float st[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) st[j] = 0.f;

  // ---- halo requests: request j of this wave fills LDS bytes [(4j + wave) * 1024, +1024) of a
  //      buffer; a lane's 16 bytes belong to pixel o / 80, slot (o % 80) / 16 (slot 4 = padding).
  //      One word per request: halo row | column << 8 | slot << 16, or -1 ----
  int hl[NHW];
#pragma unroll
  for (int j = 0; j < NHW; ++j) {
    const int o = (4 * j + wave) * 1024 + lane * 16;
    const int hr = o / PITCH;
    const int sl = (o - hr * PITCH) >> 4;
    const int hy = hr / HWD;
    const int hx = hr - hy * HWD;
    hl[j] = hr < CP_HROWS && sl < 4 ? (hy | (hx << 8) | (sl << 16)) : -1;
  }
  // ---- prologue: this thread's slot of pixels tid / 4 + 64 k; pixels past the halo are parked on the padding slot
  //      of their pixel modulo the halo (nobody reads it) ----
  constexpr int NPRO = (CP_HROWS + 63) / 64;
  float psc[8], psh[8];
  int pro_h[NPRO];
  unsigned pro_a[NPRO];
  if constexpr (PRO) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      psc[j] = p.in_scale[(tid & 3) * 8 + j];
      psh[j] = p.in_shift[(tid & 3) * 8 + j];
    }
#pragma unroll
    for (int k = 0; k < NPRO; ++k) {
      const int px = (tid >> 2) + 64 * k;
      const bool in_halo = px < CP_HROWS;
      const int hy = px / HWD, hx = px - hy * HWD;
      pro_h[k] = in_halo ? (hy | (hx << 8)) : -1;
      pro_a[k] = lds_a + (in_halo ? px * PITCH + (tid & 3) * 16 : (px - CP_HROWS) * PITCH + 64);
    }
  }
  // Patch order: blocks b, b + 8, .. share an XCD and its L2; each XCD walks its own contiguous
  // eighth of the patches, its blocks side by side.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nslot = (gridDim.x + 7 - xcd) >> 3;
  const int per_xcd = (p.mtiles + 7) >> 3;
  const int t_begin = xcd * per_xcd;
  const int t_end = t_begin + per_xcd < p.mtiles ? t_begin + per_xcd : p.mtiles;
  auto patch_origin = [&](int tile, int& n, int& y0, int& x0) {
    n = tile / per_img;
    const int trem = tile - n * per_img;
    const int tyi = trem / p.tiles_x;
    y0 = tyi * CP_TH;
    x0 = (trem - tyi * p.tiles_x) * CP_TW;
  };
  auto issue_halo = [&](int tile, int buf) {
    int n, y0, x0;
    patch_origin(tile, n, y0, x0);
    const unsigned char* base = reinterpret_cast<const unsigned char*>(p.in + n * p.in_img_stride) +
                                ((long long)(y0 - 1) * p.in_row_stride + (long long)(x0 - 1) * p.in_pix_stride) * 2;
#pragma unroll
    for (int j = 0; j < NHW; ++j) {
      int h = hl[j];
      asm volatile("" : "+v"(h));   // keep the unpacking inside the loop
      const int hy = h & 255, hx = (h >> 8) & 255, sl = h >> 16;
      // (channels from in_valid_c on do not exist in memory and read as zero)
      const bool ok = h >= 0 && (unsigned)(y0 - 1 + hy) < (unsigned)p.H && (unsigned)(x0 - 1 + hx) < (unsigned)p.W &&
                      sl * 8 < p.in_valid_c;
      const unsigned char* src = ok ? base + (hy * p.in_row_stride + hx * p.in_pix_stride + sl * 8) * 2 : zero_src;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(lds + buf * CP_HALO_BYTES + (4 * j + wave) * 1024), 16, 0, 0);
    }
  };
  // fragment (halo row r of this wave, column offset dx, sub-step kk) =
  //   xl + (r * HWD + dx) * PITCH + kk * 32
  const unsigned xl0 = lds_a + ((2 * wave) * HWD + lrow) * PITCH + lhalf * 16;

  __syncthreads();   // coefficients visible
  int tile = t_begin + slot;
  if (tile < t_end) issue_halo(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int cur = 0;
  for (; tile < t_end; tile += nslot, cur ^= 1) {
    // every wave has waited for its share of this patch's halo and is done with the other buffer
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    CP_TICK(0);
    if (tile + nslot < t_end) issue_halo(tile + nslot, cur ^ 1);
    const unsigned xl = xl0 + cur * CP_HALO_BYTES;
    int n, y0, x0;
    patch_origin(tile, n, y0, x0);
    if constexpr (PRO) {
      h16x8 v[NPRO];
      static_for<NPRO>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        ds_read_b128_at<0>(v[k], pro_a[k] + cur * CP_HALO_BYTES);
      });
      static_for<NPRO>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        lgkm_wait<NPRO - 1 - k>(v[k]);
        const int hy = pro_h[k] & 255, hx = (pro_h[k] >> 8) & 255;
        const bool ok = pro_h[k] >= 0 && (unsigned)(y0 - 1 + hy) < (unsigned)p.H && (unsigned)(x0 - 1 + hx) < (unsigned)p.W;
        const u32x4 in = __builtin_bit_cast(u32x4, v[k]);
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = fmaxf(h16lo_to_f32(in[j]) * psc[2 * j] + psh[2 * j], 0.f);
          const float hi = fmaxf(h16hi_to_f32(in[j]) * psc[2 * j + 1] + psh[2 * j + 1], 0.f);
          o[j] = ok ? pack_h16x2_raw(lo, hi) : 0u;
        }
        const unsigned dst = pro_a[k] + cur * CP_HALO_BYTES;
        asm volatile("ds_write_b128 %0, %1" ::"v"(dst), "v"(o) : "memory");
      });
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const int ox = x0 + lrow;
    // x of relu(bn(x)) at this lane's two output pixels, two octets each (lanes without a pixel or an
    // octet read the zero line, so that every lane's loads count alike)
    u32x4 xres[2][2];
    if constexpr (bnm) {
      static_for<4>([&](auto ic) __attribute__((always_inline)) {
        constexpr int b = decltype(ic)::value >> 1, q = decltype(ic)::value & 1;
        const int oy = y0 + 2 * wave + b;
        const bool ok = oy < p.Ho && ox < p.Wo && dl.okm[q];
        const h16_t* src = ok ? reinterpret_cast<const h16_t*>(p.residual) + n * p.res_img_stride +
                                    (long long)oy * p.res_row_stride + (long long)ox * p.res_pix_stride + dl.c2[q]
                              : reinterpret_cast<const h16_t*>(zero_src);
        u32x4& dst = xres[b][q];     // (asm operands do not capture by themselves)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
      });
    }
    CP_TICK(1);

    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    h16x8 xf[2][6];   // [halo row parity][dx * 2 + kk]
    auto read_row = [&](auto rc, h16x8 (&x)[6]) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;
      static_for<6>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        ds_read_b128_at<(r * HWD + (i >> 1)) * PITCH + (i & 1) * 32>(x[i], xl);
      });
    };
    read_row(std::integral_constant<int, 0>{}, xf[0]);
    read_row(std::integral_constant<int, 1>{}, xf[1]);
    static_for<4>([&](auto rc) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;     // halo row r feeds (b = 0, dy = r) and (b = 1, dy = r - 1)
      // rows arrive in order: all but the six reads of the row after this one have landed
      if constexpr (r < 3)
        asm volatile("s_waitcnt lgkmcnt(6)"
                     : "+v"(xf[r & 1][0]), "+v"(xf[r & 1][1]), "+v"(xf[r & 1][2]), "+v"(xf[r & 1][3]),
                       "+v"(xf[r & 1][4]), "+v"(xf[r & 1][5]));
      else
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(xf[r & 1][0]), "+v"(xf[r & 1][1]), "+v"(xf[r & 1][2]), "+v"(xf[r & 1][3]),
                       "+v"(xf[r & 1][4]), "+v"(xf[r & 1][5]));
      static_for<6>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr int dx = i >> 1, kk = i & 1;
        if constexpr (r < 3)
          acc[0] = mfma_32x32x16(wreg[r * 3 + dx][kk], xf[r & 1][i], acc[0]);
        if constexpr (r >= 1)
          acc[1] = mfma_32x32x16(wreg[(r - 1) * 3 + dx][kk], xf[r & 1][i], acc[1]);
      });
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (r + 2 < 4) read_row(std::integral_constant<int, r + 2>{}, xf[r & 1]);
    });
    // the next patch's halo has landed (it had the arithmetic phase, and the other block of the CU
    // was running meanwhile); only then are this patch's stores issued
    CP_TICK(2);
    if constexpr (bnm)
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(xres[0][0]), "+v"(xres[0][1]), "+v"(xres[1][0]), "+v"(xres[1][1])::"memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CP_TICK(3);
    {
      h16x8 c[8];
      ds_read_b128_at<0>(c[0], coef_a);
      ds_read_b128_at<16>(c[1], coef_a);
      ds_read_b128_at<64>(c[2], coef_a);
      ds_read_b128_at<80>(c[3], coef_a);
      ds_read_b128_at<128>(c[4], coef_a);
      ds_read_b128_at<144>(c[5], coef_a);
      ds_read_b128_at<192>(c[6], coef_a);
      ds_read_b128_at<208>(c[7], coef_a);
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 a = __builtin_bit_cast(float4, c[i]);
        const float4 b = __builtin_bit_cast(float4, c[4 + i]);
        dl.sc[4 * i] = a.x; dl.sc[4 * i + 1] = a.y; dl.sc[4 * i + 2] = a.z; dl.sc[4 * i + 3] = a.w;
        dl.sh[4 * i] = b.x; dl.sh[4 * i + 1] = b.y; dl.sh[4 * i + 2] = b.z; dl.sh[4 * i + 3] = b.w;
      }
    }
    const uint4 nores[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    if (head) {
      // softmax / sigmoid head (Cout <= 4, fp32 out): the lower half-wave's first four accumulators are
      // couts 0..3 of the lane's pixel; bias, activation and collapse in registers, 4 * Cout bytes per lane
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int oy = y0 + 2 * wave + b;
        if (lhalf == 0 && oy < p.Ho && ox < p.Wo) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[b][j] * dl.sc[j] + dl.sh[j];
          const long long o_off = n * p.out_img_stride + (long long)oy * p.out_row_stride + (long long)ox * p.out_pix_stride;
          head_store(p, v, reinterpret_cast<float*>(p.out) + o_off,
                     p.collapse_out ? p.collapse_out + (((long long)n * p.Ho + oy) * p.Wo + ox) * p.Cout : nullptr);
        }
      }
    } else {
      static_for<2>([&](auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
        const int oy = y0 + 2 * wave + b;
        const bool okp = oy < p.Ho && ox < p.Wo;
        const long long o_off = n * p.out_img_stride + (long long)oy * p.out_row_stride + (long long)ox * p.out_pix_stride;
        if constexpr (bnm) {
          const uint4 xr[2] = {__builtin_bit_cast(uint4, xres[b][0]), __builtin_bit_cast(uint4, xres[b][1])};
          direct_tile_mask(p, dl, acc[b], okp, okp ? o_off : 0, xr, st, stats);
        } else {
          direct_tile(p, dl, acc[b], okp, okp ? o_off : 0, 0, false, nores, nullptr, st, stats);
        }
      });
    }
    CP_TICK(4);
#ifdef MISEG_CP_PROFILE
    cp_t[6] += 1;
#endif
  }
#ifdef MISEG_CP_PROFILE
  if (lane == 0)
    for (int i = 0; i < 8; ++i) g_cp_prof[blockIdx.x * 4 + wave][i] = cp_t[i];
#endif
  if (stats) {
    // lanes of one half-wave hold the same couts: butterfly over the 32 pixel lanes, block sums in
    // LDS in fp64 (the order in which waves arrive must not show in the fp32 statistics derived
    // from them), one global fp64 atomic per channel and block
#pragma unroll
    for (int j = 0; j < 32; ++j)
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) st[j] += __shfl_xor(st[j], off, 64);
    double* sl = reinterpret_cast<double*>(lds);
    __syncthreads();
    if (tid < 64) sl[tid] = 0.0;
    __syncthreads();
    if (lrow == 0) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (dl.okc[q]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            unsafeAtomicAdd(&sl[dl.c2[q] + j], (double)st[8 * q + j]);
            unsafeAtomicAdd(&sl[32 + dl.c2[q] + j], (double)st[16 + 8 * q + j]);
          }
        }
    }
    __syncthreads();
    double* dst = p.stats + (long long)(blockIdx.x % p.stats_rep) * 2 * p.stats_stride;
    if (tid < 64) {
      const int which = tid >> 5, c = tid & 31;
      const double v = sl[tid];
      if (c < p.Cout && v != 0.0) unsafeAtomicAdd(dst + which * p.stats_stride + c, v);
    }
  }
}

// The layers this kernel takes over from conv3x3_c32_kernel: bf16 output whose 8-channel groups can be
// stored as 16 bytes, or the softmax / sigmoid head (fp32 out, Cout <= 4); no residual, no second output.
bool conv_c32p_ok(const ConvParams& p) {
  if ((p.residual && !p.res_mask_c) || p.out2 || !p.out) return false;
  if (p.in_scale && (MISEG_BUILD_F16 || p.res_mask_c || p.in_valid_c < 32)) return false;
  if (p.act == MISEG_ACT_SOFTMAX || p.act == MISEG_ACT_SIGMOID)
    return p.out_dtype == MISEG_DT_F32 && p.Cout <= 4 && !p.stats;
  return p.vec_ok && p.out_dtype == MISEG_DT_BF16 && (p.Cout & 7) == 0;
}

hipError_t launch_conv_c32p(const ConvParams& p, hipStream_t s) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int blocks = p.mtiles < 2 * cus ? p.mtiles : 2 * cus;
  if (p.in_scale)      // (conv_c32p_ok admits a prologue only where this instantiation exists)
    hipLaunchKernelGGL((conv3x3_c32p_kernel<false, !MISEG_BUILD_F16>), dim3(blocks), dim3(256), 0, s, p);
  else if (p.res_mask_c && !MISEG_BUILD_F16)
    hipLaunchKernelGGL(conv3x3_c32p_kernel<!MISEG_BUILD_F16>, dim3(blocks), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(conv3x3_c32p_kernel<false>, dim3(blocks), dim3(256), 0, s, p);
  return hipGetLastError();
}

}  // namespace MISEG_KNS
