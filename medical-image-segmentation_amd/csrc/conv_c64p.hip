// 3x3 stride-1 convolution with 64 input channels and 33..64 output channels (stage 1 of the ResNets at a quarter
// of the input resolution; tf.keras Conv2D instantiated through /root/reference utils/model.py:176-186): 18 KFLOP
// per pixel against 256-512 bytes of traffic, so the layer sits where the matrix pipes and HBM both matter, and
// the K-split form of the ring kernel (a block loads its halo, waits, computes, swaps half of its accumulators
// through LDS, stores) leaves both idle half of the time.
//
// One persistent block of 8 waves per CU walks over 8 x 32 output patches.
//  * The 10 x 34 pixel halo (128 B of data at a 144-byte pitch: conflict-free ds_read_b128 with linear
//    addressing) is double-buffered; the next patch's halo is requested by LDS-DMA before the arithmetic of the
//    current one, and so are the residual values of the current patch (by hand, awaited by the wait the halo
//    needs anyway).
//  * Wave (g, rp) owns cout group g (32 couts) of pixel rows 2 rp, 2 rp + 1: the four halo rows it needs give 48
//    fragments for 72 MFMAs.  Its weights of the middle tap row (dy = 1) stay in registers for the life of the
//    block; the fragments of the rows dy = 0 and dy = 2 wait in LDS as the lanes hold them (1 KB per fragment,
//    lane-linear): every (halo row, column) step then takes exactly one parked tap, and the kernel fits its
//    registers with the epilogue's 64 coefficients and the residual beside the weights.
//  * Epilogue straight from the accumulators (conv_common.h direct_tile: scale/shift, residual, ReLU, second
//    output); its coefficients wait in LDS between patches.
#include "conv_common.h"

namespace MISEG_KNS {

namespace {
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int C6_TH = 8, C6_TW = 32, C6_HWD = C6_TW + 2;
constexpr int C6_HROWS = (C6_TH + 2) * C6_HWD;      // 340 halo pixels
constexpr int C6_PITCH = 144;
constexpr int C6_NHI = 48;                          // halo wave-instructions (48,960 B -> 48 KB)
constexpr int C6_HALO_BYTES = C6_NHI * 1024;
static_assert(C6_HROWS * C6_PITCH <= C6_HALO_BYTES, "halo fits its buffer");
constexpr int C6_NREG = 3;                          // taps whose weight fragments stay in registers: the row dy = 1
constexpr int C6_NLDS = 9 - C6_NREG;                // parked in LDS: rows dy = 0 (index dx) and dy = 2 (3 + dx)
constexpr int C6_W_OFF = 2 * C6_HALO_BYTES;
constexpr int C6_W_BYTES = 2 * C6_NLDS * 4 * 1024;  // [cout group][tap][k step] fragments
constexpr int C6_COEF_OFF = C6_W_OFF + C6_W_BYTES;  // scale | shift | scale2 | shift2, 64 fp32 each
constexpr int C6_LDS = C6_COEF_OFF + 4 * 64 * 4;
constexpr int C6_AHEAD = 3;                         // fragment reads run this many (row, column, k-step) units ahead
                                                    // (5 and 7 measured the same: the kernel does not wait on LDS)
static_assert(C6_LDS <= 160 * 1024, "one block per CU");
}  // namespace

// STATS: the launch gathers the per-channel sum and sum of squares of what it stores (training forward); it then
// has no second output, whose 32 coefficient registers the running sums take.
// BNM (with STATS): a data gradient behind relu(bn(x)) (ConvParams::res_mask_c): `residual` is x, scale/shift
// are the BatchNorm's forward coefficients, the sums are sum(g) and sum(g * x) of the masked result.
template <bool STATS, bool BNM = false>
__global__ __launch_bounds__(512, 1) void conv3x3_c64p_kernel(const ConvParams p) {
  constexpr int HWD = C6_HWD, PITCH = C6_PITCH, NHW = C6_NHI / 8;
  __shared__ __attribute__((aligned(16))) unsigned char lds[C6_LDS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave & 1, rp = wave >> 1;
  const int per_img = p.tiles_y * p.tiles_x;
  const int lrow = lane & 31, lhalf = lane >> 5;
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const unsigned lds_a = (unsigned)(unsigned long long)(lds_byte*)lds;
  const unsigned char* zero_src = g_zero_line;

  // ---- weights: fragment [tap][kk] = cout 32 g + lrow, k = kk * 16 + lhalf * 8 .. + 7 (rows past Cout repeat
  //      row 0: their results are never stored) ----
  h16x8 wreg[C6_NREG][4];
  {
    const int co = 32 * g + lrow;
    const h16_t* wrow = p.weight + (long long)(co < p.Cout ? co : 0) * 9 * 64 + lhalf * 8;
    static_for<C6_NREG * 4>([&](auto ic) __attribute__((always_inline)) {
      constexpr int dx = decltype(ic)::value >> 2, kk = decltype(ic)::value & 3;
      wreg[dx][kk] = __builtin_bit_cast(h16x8, *reinterpret_cast<const uint4*>(wrow + (3 + dx) * 64 + kk * 16));
    });
    if (rp == 0) {       // one wave per cout group parks the other taps' fragments in LDS
      static_for<C6_NLDS * 4>([&](auto ic) __attribute__((always_inline)) {
        constexpr int t = decltype(ic)::value >> 2, kk = decltype(ic)::value & 3;
        constexpr int tap = t < 3 ? t : 3 + t;          // (0, dx) or (2, dx)
        *reinterpret_cast<uint4*>(lds + C6_W_OFF + ((g * C6_NLDS + t) * 4 + kk) * 1024 + lane * 16) =
            *reinterpret_cast<const uint4*>(wrow + tap * 64 + kk * 16);
      });
    }
  }
  const unsigned wl_a = lds_a + C6_W_OFF + g * C6_NLDS * 4096 + lane * 16;
  // ---- epilogue coefficients wait in LDS between patches ----
  {
    float* coef = reinterpret_cast<float*>(lds + C6_COEF_OFF);
    if (tid < 256) {
      const int which = tid >> 6, c = tid & 63;
      const float* src = which == 0 ? p.scale : which == 1 ? p.shift : which == 2 ? p.scale2 : p.shift2;
      coef[tid] = (src && c < (BNM ? p.res_mask_c : p.Cout)) ? src[c] : ((which & 1) ? 0.f : 1.f);
    }
  }
  const unsigned coef_a = lds_a + C6_COEF_OFF + (32 * g + 8 * lhalf) * 4;
  DirectLane dl;
  dl.c2[0] = 32 * g + 8 * lhalf;
  dl.c2[1] = 32 * g + 16 + 8 * lhalf;
  dl.okc[0] = dl.c2[0] + 8 <= p.Cout;
  dl.okc[1] = dl.c2[1] + 8 <= p.Cout;
  dl.okm[0] = BNM && dl.c2[0] + 8 <= p.res_mask_c;
  dl.okm[1] = BNM && dl.c2[1] + 8 <= p.res_mask_c;
  dl.relu = p.act == MISEG_ACT_RELU;
  const bool has_res = p.residual != nullptr;
  const bool has_out2 = !STATS && p.out2 != nullptr;
  float st[STATS ? 32 : 1];
#pragma unroll
  for (int j = 0; j < (STATS ? 32 : 1); ++j) st[j] = 0.f;

  // ---- halo requests: request j of this wave fills LDS bytes [(8j + wave) * 1024, +1024) of a buffer; a lane's
  //      16 bytes belong to pixel o / 144, slot (o % 144) / 16 (slot 8 = padding).  One word per request: halo
  //      row | column << 8 | slot << 16, or -1 ----
  int hl[NHW];
#pragma unroll
  for (int j = 0; j < NHW; ++j) {
    const int o = (8 * j + wave) * 1024 + lane * 16;
    const int hr = o / PITCH;
    const int sl = (o - hr * PITCH) >> 4;
    const int hy = hr / HWD;
    const int hx = hr - hy * HWD;
    hl[j] = hr < C6_HROWS && sl < 8 ? (hy | (hx << 8) | (sl << 16)) : -1;
  }
  // Patch order: blocks b, b + 8, .. share an XCD and its L2; each XCD walks its own contiguous eighth of the
  // patches, its blocks side by side.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nslot = (gridDim.x + 7 - xcd) >> 3;
  const int per_xcd = (p.mtiles + 7) >> 3;
  const int t_begin = xcd * per_xcd;
  const int t_end = t_begin + per_xcd < p.mtiles ? t_begin + per_xcd : p.mtiles;
  auto patch_origin = [&](int tile, int& n, int& y0, int& x0) {
    n = tile / per_img;
    const int trem = tile - n * per_img;
    const int tyi = trem / p.tiles_x;
    y0 = tyi * C6_TH;
    x0 = (trem - tyi * p.tiles_x) * C6_TW;
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
      const bool ok = h >= 0 && (unsigned)(y0 - 1 + hy) < (unsigned)p.H && (unsigned)(x0 - 1 + hx) < (unsigned)p.W;
      const unsigned char* src = ok ? base + (hy * p.in_row_stride + hx * p.in_pix_stride + sl * 8) * 2 : zero_src;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(lds + buf * C6_HALO_BYTES + (8 * j + wave) * 1024), 16, 0, 0);
    }
  };
  // fragment (halo row r of this wave, column offset dx, sub-step kk) = xl + (r * HWD + dx) * PITCH + kk * 32
  const unsigned xl0 = lds_a + ((2 * rp) * HWD + lrow) * PITCH + lhalf * 16;

  __syncthreads();   // coefficients and parked weights visible (no LDS-DMA is pending yet)
  int tile = t_begin + slot;
  if (tile < t_end) issue_halo(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int cur = 0;
  for (; tile < t_end; tile += nslot, cur ^= 1) {
    // every wave has waited for its share of this patch's halo and is done with the other buffer
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tile + nslot < t_end) issue_halo(tile + nslot, cur ^ 1);
    const unsigned xl = xl0 + cur * C6_HALO_BYTES;
    int n, y0, x0;
    patch_origin(tile, n, y0, x0);
    const int ox = x0 + lrow;
    // the residual at this lane's two output pixels, two octets each (lanes without a pixel, an octet or a
    // residual read the zero line, so that every lane's loads count alike)
    u32x4 rres[2][2];
    static_for<4>([&](auto ic) __attribute__((always_inline)) {
      constexpr int b = decltype(ic)::value >> 1, q = decltype(ic)::value & 1;
      const int oy = y0 + 2 * rp + b;
      const bool ok = has_res && oy < p.Ho && ox < p.Wo && (BNM ? dl.okm[q] : dl.okc[q]);
      const h16_t* src = ok ? reinterpret_cast<const h16_t*>(p.residual) + n * p.res_img_stride +
                                  (long long)oy * p.res_row_stride + (long long)ox * p.res_pix_stride + dl.c2[q]
                            : reinterpret_cast<const h16_t*>(zero_src);
      u32x4& dst = rres[b][q];     // (asm operands do not capture by themselves)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
    });

    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    // units (halo row r, column dx, sub-step kk), 48 of them: one pixel fragment and the weight fragment of the
    // unit's parked tap.  acc[0] takes tap (r, dx) when r < 3, acc[1] tap (r - 1, dx) when r >= 1, so halo rows 0
    // and 1 use the parked tap (0, dx) and rows 2 and 3 the parked tap (2, dx); the tap (1, dx) of rows 1 and 2
    // comes from registers.  Fragments are read AHEAD units ahead of the MFMAs that consume them, through rings
    // of AHEAD + 1 registers each.
    constexpr int AHEAD = C6_AHEAD, NU = 48, RING = C6_AHEAD + 1;
    h16x8 xq[RING], wq[RING];
    auto read_unit = [&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value, s = u >> 2, kk = u & 3, r = s / 3, dx = s % 3;
      ds_read_b128_at<(r * HWD + dx) * PITCH + kk * 32>(xq[u % RING], xl);
      ds_read_b128_at<(((r < 2 ? 0 : 3) + dx) * 4 + kk) * 1024>(wq[u % RING], wl_a);
    };
    static_for<AHEAD>([&](auto uc) __attribute__((always_inline)) { read_unit(uc); });
    static_for<NU>([&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value, s = u >> 2, kk = u & 3, r = s / 3, dx = s % 3;
      if constexpr (u + AHEAD < NU) read_unit(std::integral_constant<int, u + AHEAD>{});
      // reads return in order: all but the two of every unit requested after this one have landed
      constexpr int younger = 2 * ((u + AHEAD < NU ? u + AHEAD : NU - 1) - u);
      asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(xq[u % RING]), "+v"(wq[u % RING]) : "n"(younger));
      if constexpr (r == 0 || r == 2) acc[0] = mfma_32x32x16(wq[u % RING], xq[u % RING], acc[0]);
      if constexpr (r == 1) acc[0] = mfma_32x32x16(wreg[dx][kk], xq[u % RING], acc[0]);
      if constexpr (r == 1 || r == 3) acc[1] = mfma_32x32x16(wq[u % RING], xq[u % RING], acc[1]);
      if constexpr (r == 2) acc[1] = mfma_32x32x16(wreg[dx][kk], xq[u % RING], acc[1]);
    });
    __builtin_amdgcn_sched_barrier(0);
    // the next patch's halo and this patch's residual have landed (they had the arithmetic phase); only then
    // are this patch's stores issued
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(rres[0][0]), "+v"(rres[0][1]), "+v"(rres[1][0]), "+v"(rres[1][1])::"memory");
    {
      h16x8 c[8];
      ds_read_b128_at<0>(c[0], coef_a);
      ds_read_b128_at<16>(c[1], coef_a);
      ds_read_b128_at<64>(c[2], coef_a);
      ds_read_b128_at<80>(c[3], coef_a);
      ds_read_b128_at<256>(c[4], coef_a);
      ds_read_b128_at<272>(c[5], coef_a);
      ds_read_b128_at<320>(c[6], coef_a);
      ds_read_b128_at<336>(c[7], coef_a);
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
    DirectOut2 d2;
    if constexpr (!STATS) if (has_out2) {
      h16x8 c[8];
      ds_read_b128_at<512>(c[0], coef_a);
      ds_read_b128_at<528>(c[1], coef_a);
      ds_read_b128_at<576>(c[2], coef_a);
      ds_read_b128_at<592>(c[3], coef_a);
      ds_read_b128_at<768>(c[4], coef_a);
      ds_read_b128_at<784>(c[5], coef_a);
      ds_read_b128_at<832>(c[6], coef_a);
      ds_read_b128_at<848>(c[7], coef_a);
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 a = __builtin_bit_cast(float4, c[i]);
        const float4 b = __builtin_bit_cast(float4, c[4 + i]);
        d2.s2[4 * i] = a.x; d2.s2[4 * i + 1] = a.y; d2.s2[4 * i + 2] = a.z; d2.s2[4 * i + 3] = a.w;
        d2.t2[4 * i] = b.x; d2.t2[4 * i + 1] = b.y; d2.t2[4 * i + 2] = b.z; d2.t2[4 * i + 3] = b.w;
      }
    }
    static_for<2>([&](auto bc) __attribute__((always_inline)) {
      constexpr int b = decltype(bc)::value;
      const int oy = y0 + 2 * rp + b;
      const bool okp = oy < p.Ho && ox < p.Wo;
      const long long o_off = okp ? n * p.out_img_stride + (long long)oy * p.out_row_stride + (long long)ox * p.out_pix_stride : 0;
      const long long o2_off = okp && has_out2 ? n * p.out2_img_stride + (long long)oy * p.out2_row_stride +
                                                     (long long)ox * p.out2_pix_stride : 0;
      const uint4 rr[2] = {__builtin_bit_cast(uint4, rres[b][0]), __builtin_bit_cast(uint4, rres[b][1])};
      if constexpr (BNM)
        direct_tile_mask(p, dl, acc[b], okp, o_off, rr, st, true);
      else if (p.out)
        direct_tile<true>(p, dl, acc[b], okp, o_off, o2_off, has_res, rr, !STATS && has_out2 ? &d2 : nullptr, STATS ? st : nullptr, STATS);
      else
        direct_tile<false>(p, dl, acc[b], okp, o_off, o2_off, has_res, rr, !STATS && has_out2 ? &d2 : nullptr, STATS ? st : nullptr, STATS);
    });
  }
  if constexpr (STATS) {
    // lanes of one half-wave hold the same couts: butterfly over the 32 pixel lanes, block sums in LDS in fp64
    // (the order in which waves arrive must not show in the fp32 statistics derived from them), one global fp64
    // atomic per channel and block
#pragma unroll
    for (int j = 0; j < 32; ++j)
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) st[j] += __shfl_xor(st[j], off, 64);
    double* sl = reinterpret_cast<double*>(lds);
    __syncthreads();
    if (tid < 128) sl[tid] = 0.0;
    __syncthreads();
    if (lrow == 0) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (dl.okc[q]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            unsafeAtomicAdd(&sl[dl.c2[q] + j], (double)st[8 * q + j]);
            unsafeAtomicAdd(&sl[64 + dl.c2[q] + j], (double)st[16 + 8 * q + j]);
          }
        }
    }
    __syncthreads();
    double* dst = p.stats + (long long)(blockIdx.x % p.stats_rep) * 2 * p.stats_stride;
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      const double v = sl[tid];
      if (c < p.Cout && v != 0.0) unsafeAtomicAdd(dst + which * p.stats_stride + c, v);
    }
  }
}

// The layers this kernel takes over from the K-split ring kernel: 64 input channels, 33..64 output channels
// stored 16 bits wide in 16-byte groups, optionally with a 16-bit residual at the output resolution and either a
// second output or (bf16 storage) statistics, or (bf16 storage) as a masking data gradient; no prologue, no head.
bool conv_c64p_ok(const ConvParams& p) {
  if (p.transposed || p.dgrad_t || p.Cin != 64 || p.Cout <= 32 || p.Cout > 64 || (p.Cout & 7)) return false;
  if (p.KH != 3 || p.KW != 3 || p.stride_y != 1 || p.stride_x != 1 || p.pad_y != 1 || p.pad_x != 1) return false;
  if (p.Ho != p.H || p.Wo != p.W || !p.vec_ok || p.out_dtype != MISEG_DT_BF16 || (!p.out && !p.out2)) return false;
  if (p.in_scale || p.collapse_out || p.act == MISEG_ACT_SOFTMAX || p.act == MISEG_ACT_SIGMOID) return false;
  if (p.residual && (p.res_dtype != MISEG_DT_BF16 || p.res_shift)) return false;
  if (p.stats && (MISEG_BUILD_F16 || p.out2)) return false;
  if (p.res_mask_c && (MISEG_BUILD_F16 || !p.stats || !p.out || p.act != MISEG_ACT_NONE)) return false;
  // per-lane source offsets of the halo requests are 32-bit
  return (long long)p.H * p.in_row_stride * 2 < (1ll << 31);
}

hipError_t launch_conv_c64p(ConvParams p, hipStream_t s) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  p.tiles_y = (p.Ho + C6_TH - 1) / C6_TH;
  p.tiles_x = (p.Wo + C6_TW - 1) / C6_TW;
  p.mtiles = p.N * p.tiles_y * p.tiles_x;
  const int blocks = p.mtiles < cus ? p.mtiles : cus;
  if (p.res_mask_c)
    hipLaunchKernelGGL((conv3x3_c64p_kernel<!MISEG_BUILD_F16, !MISEG_BUILD_F16>), dim3(blocks), dim3(512), 0, s, p);
  else if (p.stats)
    hipLaunchKernelGGL(conv3x3_c64p_kernel<!MISEG_BUILD_F16>, dim3(blocks), dim3(512), 0, s, p);
  else
    hipLaunchKernelGGL(conv3x3_c64p_kernel<false>, dim3(blocks), dim3(512), 0, s, p);
  return hipGetLastError();
}

}  // namespace MISEG_KNS
