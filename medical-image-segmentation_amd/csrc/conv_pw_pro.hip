// 1x1 convolution whose operand gets BatchNorm + ReLU on the way in (gfx950): DenseNet's transition
// layers (pool{k}_bn, pool{k}_relu, pool{k}_conv of keras_applications `transition_block`, reached from
// /root/reference utils/model.py:176-211) and the 1x1 of a dense layer when it runs on its own.  The
// register-staged generic kernel applies the prologue between its global loads and its LDS stores and runs
// at 220-340 TFLOP/s on these layers; here the operand goes to LDS untouched by LDS-DMA and the prologue is
// applied to the pixel FRAGMENTS in registers after the LDS read (packed fp16 arithmetic with fp16 storage:
// 8 VALU operations per fragment, which feeds two matrix instructions), as in the dense-layer kernel.
//
// A block of 4 waves owns 192 consecutive pixels x 128 couts: K = Cin in chunks of 32 channels, a chunk's
// pixels [192][32] and weights [128][32] arrive in a three-slot ring two chunks ahead, one barrier per chunk.
// Wave (cg, ph): couts [64 cg, 64 cg + 64) x pixel tiles 3 ph .. 3 ph + 2.  The epilogue (scale/shift, ReLU,
// 16-bit stores) goes straight from the accumulators to memory.  Blocks that share pixels (the cout tiles of
// one pixel tile) follow each other on one XCD, so the operand is read from HBM once.
// LDS: 3 x 20 KB ring + 16 KB of input coefficients = 76 KB, two blocks per CU.
#include "conv_common.h"

namespace MISEG_KNS {

namespace {
constexpr int PP_XT = 6;                         // 32-pixel tiles per block
constexpr int PP_PIX = PP_XT * 32;
constexpr int PP_XST = PP_PIX * 64;              // pixel part of a ring stage: [192][32 ch]
constexpr int PP_WST = 128 * 64;                 // weight part: [128 couts][32 ch]
constexpr int PP_STAGE = PP_XST + PP_WST;
constexpr int PP_RING = 3 * PP_STAGE;
constexpr int PP_ST0_OFF = PP_RING;
constexpr int PP_ST0_BYTES = 16 * 1024;
constexpr int PP_LDS = PP_ST0_OFF + PP_ST0_BYTES;
static_assert(2 * PP_LDS <= 160 * 1024, "two blocks per CU");

#if MISEG_BUILD_F16
typedef _Float16 pp_coef_t;
#else
typedef float pp_coef_t;
#endif

__device__ __forceinline__ void pp_barrier() {
  // LDS hand-off between waves without draining vmcnt (pending LDS-DMA of later stages)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <int N>
__device__ __forceinline__ void pp_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pp_dma16(const void* src, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
}  // namespace

// p.mtiles pixel tiles x p.ntiles cout tiles; grid = 8 * ceil(mtiles / 8) * ntiles.
__global__ __launch_bounds__(256, 2) void conv_pw_pro_kernel(const ConvParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[PP_LDS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 31, lhalf = lane >> 5;
  const int C = p.Cin;
  const int nch = C >> 5;

  // XCD x takes a contiguous run of the pixel tiles; inside it the cout tiles of a pixel tile are neighbours
  int pt, ct;
  {
    const int ntot = p.mtiles, bid = blockIdx.x;
    const int qq = ntot >> 3, r = ntot & 7, x = bid & 7, loc = bid >> 3;
    const int cnt = qq + (x < r ? 1 : 0);
    const int first = x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq;
    const int pl = loc / p.ntiles;
    if (pl >= cnt) return;
    ct = loc - pl * p.ntiles;
    pt = first + pl;
  }
  const int m0 = pt * PP_PIX, cout0 = ct * 128;
  const int HW = p.H * p.W;

  // ---- input coefficients into LDS (plain stores: no LDS-DMA is pending yet) ----
  {
    pp_coef_t* st0 = reinterpret_cast<pp_coef_t*>(lds + PP_ST0_OFF);
    for (int i = tid * 4; i < C; i += 1024) {
      const float4 s = *reinterpret_cast<const float4*>(p.in_scale + i);
      const float4 t = *reinterpret_cast<const float4*>(p.in_shift + i);
      st0[i] = (pp_coef_t)s.x; st0[i + 1] = (pp_coef_t)s.y; st0[i + 2] = (pp_coef_t)s.z; st0[i + 3] = (pp_coef_t)s.w;
      st0[C + i] = (pp_coef_t)t.x; st0[C + i + 1] = (pp_coef_t)t.y; st0[C + i + 2] = (pp_coef_t)t.z;
      st0[C + i + 3] = (pp_coef_t)t.w;
    }
  }

  // ---- LDS-DMA sources ----
  const unsigned char* zero_src = g_zero_line + (lane & 3) * 16;
  const unsigned char* xsrc[3];    // pixel rows 16 i + lane / 4 of instruction i = wave, wave + 4, wave + 8
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int row = (wave + 4 * j) * 16 + (lane >> 2);
    const int m = m0 + row;
    const int ch = (lane & 3) ^ ((row >> 2) & 3);
    if (m < p.M) {
      const int n = m / HW, rem = m - n * HW;
      const int y = rem / p.W, x = rem - y * p.W;
      xsrc[j] = reinterpret_cast<const unsigned char*>(p.in + n * p.in_img_stride + (long long)y * p.in_row_stride +
                                                       (long long)x * p.in_pix_stride + ch * 8);
    } else {
      xsrc[j] = nullptr;
    }
  }
  const unsigned char* wsrc[2];    // cout rows of instruction i = wave, wave + 4
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave + 4 * j) * 16 + (lane >> 2);
    const int ch = (lane & 3) ^ ((row >> 2) & 3);
    wsrc[j] = cout0 + row < p.Cout
                  ? reinterpret_cast<const unsigned char*>(p.weight + (long long)(cout0 + row) * C + ch * 8)
                  : nullptr;
  }
  auto issue_stage = [&](int chunk, int slot) {
    unsigned char* st = lds + slot * PP_STAGE;
#pragma unroll
    for (int j = 0; j < 3; ++j) pp_dma16(xsrc[j] ? xsrc[j] + chunk * 64 : zero_src, st + (wave + 4 * j) * 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      pp_dma16(wsrc[j] ? wsrc[j] + chunk * 64 : zero_src, st + PP_XST + (wave + 4 * j) * 1024);
  };

  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const unsigned lds_a = (unsigned)(unsigned long long)(lds_byte*)lds;

  const int cg = wave & 1, ph = wave >> 1;
  f32x16 acc[2][3];
  zero_acc<3, 2>(acc);
  {
    unsigned xa[2][3], wa[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int b = 0; b < 3; ++b) xa[ks][b] = swz64((ph * 3 + b) * 32 + lrow, ks * 2 + lhalf);
#pragma unroll
      for (int a = 0; a < 2; ++a) wa[ks][a] = PP_XST + swz64(cg * 64 + a * 32 + lrow, ks * 2 + lhalf);
    }
    const unsigned st_a = lds_a + PP_ST0_OFF + lhalf * 8 * (unsigned)sizeof(pp_coef_t);
    const unsigned tofs = C * (unsigned)sizeof(pp_coef_t);

    issue_stage(0, 0);
    if (nch > 1) issue_stage(1, 1);
    int slot = 0;
    for (int ch = 0; ch < nch; ++ch) {
      if (ch + 1 < nch) pp_vmcnt<5>(); else pp_vmcnt<0>();
      pp_barrier();       // chunk ch has landed for every wave (and, first time round, the coefficients)
      if (ch + 2 < nch) issue_stage(ch + 2, slot >= 1 ? slot - 1 : 2);
      const unsigned sb = lds_a + slot * PP_STAGE;
      h16x8 xf[2][3], wf[2][2];
#if MISEG_BUILD_F16
      h16x8 cs[2], ct_[2];
#else
      h16x8 cs[2][2], ct_[2][2];     // 8 fp32 each, as raw 16-byte pairs
#endif
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const unsigned ca = st_a + (ch * 32 + ks * 16) * (unsigned)sizeof(pp_coef_t);
#if MISEG_BUILD_F16
        ds_read_b128_at<0>(cs[ks], ca);
        ds_read_b128_at<0>(ct_[ks], ca + tofs);
#else
        ds_read_b128_at<0>(cs[ks][0], ca);
        ds_read_b128_at<16>(cs[ks][1], ca);
        ds_read_b128_at<0>(ct_[ks][0], ca + tofs);
        ds_read_b128_at<16>(ct_[ks][1], ca + tofs);
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
                     : "+v"(cs[ks]), "+v"(ct_[ks]), "+v"(wf[ks][0]), "+v"(wf[ks][1]), "+v"(xf[ks][0]),
                       "+v"(xf[ks][1]), "+v"(xf[ks][2])
                     : "n"(later));
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          h16x8 v = xf[ks][b] * cs[ks] + ct_[ks];
          const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
          xf[ks][b] = __builtin_elementwise_max(v, z);
        }
#else
        constexpr int later = ks == 0 ? 9 : 0;
        asm volatile("s_waitcnt lgkmcnt(%9)"
                     : "+v"(cs[ks][0]), "+v"(cs[ks][1]), "+v"(ct_[ks][0]), "+v"(ct_[ks][1]), "+v"(wf[ks][0]),
                       "+v"(wf[ks][1]), "+v"(xf[ks][0]), "+v"(xf[ks][1]), "+v"(xf[ks][2])
                     : "n"(later));
        {
          const float4 s0 = __builtin_bit_cast(float4, cs[ks][0]), s1 = __builtin_bit_cast(float4, cs[ks][1]);
          const float4 t0 = __builtin_bit_cast(float4, ct_[ks][0]), t1 = __builtin_bit_cast(float4, ct_[ks][1]);
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

  // ---- epilogue: every lane finishes its pixel's 16 couts of each 32-cout group ----
  const uint4 nores[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
  bool okp[3];
  long long o_off[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const int m = m0 + (ph * 3 + b) * 32 + lrow;
    okp[b] = m < p.M;
    const int mm = okp[b] ? m : 0;
    const int n = mm / HW, rem = mm - n * HW;
    const int y = rem / p.W, x = rem - y * p.W;
    o_off[b] = n * p.out_img_stride + (long long)y * p.out_row_stride + (long long)x * p.out_pix_stride;
  }
  static_for<2>([&](auto ac) __attribute__((always_inline)) {
    constexpr int a = decltype(ac)::value;
    const int cbase = cout0 + cg * 64 + a * 32;
    if (cbase < p.Cout) {           // wave-uniform
      DirectLane dl;
      direct_lane_init(p, cbase, lane, dl);
      static_for<3>([&](auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
        direct_tile(p, dl, acc[a][b], okp[b], o_off[b], 0, false, nores, nullptr, nullptr, false);
      });
    }
  });
}

// 1x1, stride 1, operand prologue, 16-bit output through the direct epilogue, no residual / second output /
// statistics; the input coefficients of all Cin channels fit the LDS region kept for them.
bool conv_pw_pro_ok(const ConvParams& p) {
  return p.in_scale && p.in_shift && p.KH == 1 && p.KW == 1 && p.stride_y == 1 && p.stride_x == 1 && p.pad_y == 0 &&
         p.pad_x == 0 && !p.transposed && (p.Cin & 31) == 0 && p.Cin >= 64 &&
         2 * p.Cin * (int)sizeof(pp_coef_t) <= PP_ST0_BYTES && p.vec_ok && p.out && p.out_dtype == MISEG_DT_BF16 &&
         (p.Cout & 7) == 0 && p.act != MISEG_ACT_SOFTMAX && p.act != MISEG_ACT_SIGMOID && !p.residual && !p.out2 &&
         !p.stats && p.Cout >= 64 && ((uintptr_t)p.in_scale & 15) == 0 && ((uintptr_t)p.in_shift & 15) == 0;
}

hipError_t launch_conv_pw_pro(ConvParams p, hipStream_t s) {
  p.mtiles = (p.M + PP_PIX - 1) / PP_PIX;
  p.ntiles = (p.Cout + 127) / 128;
  const long long grid = 8ll * ((p.mtiles + 7) / 8) * p.ntiles;
  hipLaunchKernelGGL(conv_pw_pro_kernel, dim3((unsigned)grid), dim3(256), 0, s, p);
  return hipGetLastError();
}

}  // namespace MISEG_KNS
