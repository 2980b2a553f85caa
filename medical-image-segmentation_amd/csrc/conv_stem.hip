// The 7x7 stride-2 stem convolution (3 -> 64 channels; classification_models' `conv0`, Keras
// DenseNet's `conv1/conv`, instantiated through /root/reference utils/model.py:176-211) in the
// form the plan builder gives it: a 7x1 convolution with vertical stride 2 over a zero-padded
// input whose "pixels" are windows of 8 real pixels x 4 channels (32 values, 64 bytes) that
// start every 2 real pixels (in_pix_stride = 8 elements), K = 7 x 32 = 224.
//
// Persistent blocks, two per CU, walk over 8 x 32 output patches.  The 21 input rows x 560 bytes a
// patch needs are contiguous runs of global memory and land in LDS as they are (LDS-DMA, double-
// buffered, next patch requested before the arithmetic of the current one); fragment (output row
// b, tap row ky, half s of the window) of output pixel j is the 16 bytes at
// (2b + ky) * 576 + 16 j + 32 s + 16 (lane / 32): conflict-free ds_read_b128 at an immediate
// offset.  A wave owns two output rows and all 64 couts; its weights (14 K-slices x 2 cout
// groups) stay in 112 registers for the life of the block; the nine input rows it needs give 18
// fragments for 56 MFMAs, read one row ahead.  Epilogue straight from the accumulators: four
// consecutive stores complete the 128 bytes of a pixel.
#include "conv_common.h"

namespace MISEG_KNS {

namespace {
constexpr int ST_TH = 8, ST_TW = 32;
constexpr int ST_ROWS = 2 * ST_TH + 5;            // 21 input rows
constexpr int ST_ROWB = 576;                      // bytes per input row in LDS (560 used)
constexpr int ST_NHI = 12;                        // wave-instructions per patch (12,096 B -> 12 KB)
constexpr int ST_BUF = ST_NHI * 1024;
constexpr int ST_LDS = 2 * ST_BUF + 512;          // + scale[64] | shift[64]
static_assert(ST_ROWS * ST_ROWB <= ST_BUF, "region fits its buffer");
}  // namespace

__global__ __launch_bounds__(256, 2) void conv_stem7_kernel(const ConvParams p) {
  constexpr int NHW = ST_NHI / 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[ST_LDS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per_img = p.tiles_y * p.tiles_x;
  const int lrow = lane & 31, lhalf = lane >> 5;
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const unsigned lds_a = (unsigned)(unsigned long long)(lds_byte*)lds;
  const unsigned char* zero_src = g_zero_line;
  const int row_bytes = p.in_row_stride * 2;

  // ---- weights: fragment [ky][s][g] = cout 32 g + lrow, window values 16 s + 8 lhalf .. + 7 ----
  h16x8 wreg[7][2][2];
  static_for<28>([&](auto ic) __attribute__((always_inline)) {
    constexpr int ky = decltype(ic)::value >> 2, s = (decltype(ic)::value >> 1) & 1, g = decltype(ic)::value & 1;
    const int co = g * 32 + lrow;
    const h16_t* wsrc = p.weight + ((co < p.Cout ? co : 0) * 7 + ky) * 32 + s * 16 + lhalf * 8;
    wreg[ky][s][g] = __builtin_bit_cast(h16x8, *reinterpret_cast<const uint4*>(wsrc));
  });
  // ---- epilogue coefficients wait in LDS between patches: scale[64] | shift[64] ----
  float* coef = reinterpret_cast<float*>(lds + 2 * ST_BUF);
  if (tid < 128) {
    const int c = tid & 63;
    const float* src = tid < 64 ? p.scale : p.shift;
    coef[tid] = (src && c < p.Cout) ? src[c] : (tid < 64 ? 1.f : 0.f);
  }
  const unsigned coef_a = lds_a + 2 * ST_BUF + lhalf * 32;
  const bool relu = p.act == MISEG_ACT_RELU;

  // ---- region requests: request j of this wave fills LDS bytes [(4j + wave) * 1024, +1024); a
  //      lane's 16 bytes are bytes [c, c + 16) of region row r, o = r * 576 + c ----
  int hr[NHW], hc[NHW];
#pragma unroll
  for (int j = 0; j < NHW; ++j) {
    const int o = (4 * j + wave) * 1024 + lane * 16;
    hr[j] = o / ST_ROWB;
    hc[j] = o - hr[j] * ST_ROWB;
  }
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nslot = (gridDim.x + 7 - xcd) >> 3;
  const int per_xcd = (p.mtiles + 7) >> 3;
  const int t_begin = xcd * per_xcd;
  const int t_end = t_begin + per_xcd < p.mtiles ? t_begin + per_xcd : p.mtiles;
  auto patch_origin = [&](int tile, int& n, int& y0, int& x0) {
    n = tile / per_img;
    const int trem = tile - n * per_img;
    const int tyi = trem / p.tiles_x;
    y0 = tyi * ST_TH;
    x0 = (trem - tyi * p.tiles_x) * ST_TW;
  };
  auto issue_region = [&](int tile, int buf) {
    int n, y0, x0;
    patch_origin(tile, n, y0, x0);
    const int col0 = x0 * p.in_pix_stride * 2;       // byte offset of the patch's first window in a row
    const unsigned char* base = reinterpret_cast<const unsigned char*>(p.in + n * p.in_img_stride) +
                                (long long)(2 * y0) * row_bytes + col0;
#pragma unroll
    for (int j = 0; j < NHW; ++j) {
      const bool ok = hr[j] < ST_ROWS && 2 * y0 + hr[j] < p.H && col0 + hc[j] + 16 <= row_bytes;
      const unsigned char* src = ok ? base + (long long)hr[j] * row_bytes + hc[j] : zero_src;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(lds + buf * ST_BUF + (4 * j + wave) * 1024), 16, 0, 0);
    }
  };
  // fragment (region row r of this wave, half s) = xl + r * 576 + s * 32
  const unsigned xl0 = lds_a + (4 * wave) * ST_ROWB + lrow * 16 + lhalf * 16;

  __syncthreads();   // coefficients visible
  int tile = t_begin + slot;
  if (tile < t_end) issue_region(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int cur = 0;
  for (; tile < t_end; tile += nslot, cur ^= 1) {
    // every wave has waited for its share of this patch's region and is done with the other buffer
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tile + nslot < t_end) issue_region(tile + nslot, cur ^ 1);
    const unsigned xl = xl0 + cur * ST_BUF;

    f32x16 acc[2][2];   // [output row][cout group]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][g][r] = 0.f;
    h16x8 xf[2][2];   // [region row parity][s]
    auto read_row = [&](auto rc, h16x8 (&x)[2]) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;
      ds_read_b128_at<r * ST_ROWB>(x[0], xl);
      ds_read_b128_at<r * ST_ROWB + 32>(x[1], xl);
    };
    read_row(std::integral_constant<int, 0>{}, xf[0]);
    read_row(std::integral_constant<int, 1>{}, xf[1]);
    static_for<9>([&](auto rc) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;   // region row r feeds (b = 0, ky = r) and (b = 1, ky = r - 2)
      if constexpr (r < 8)
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xf[r & 1][0]), "+v"(xf[r & 1][1]));
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[r & 1][0]), "+v"(xf[r & 1][1]));
      static_for<4>([&](auto ic) __attribute__((always_inline)) {
        constexpr int s = decltype(ic)::value >> 1, g = decltype(ic)::value & 1;
        if constexpr (r <= 6)
          acc[0][g] = mfma_32x32x16(wreg[r][s][g], xf[r & 1][s], acc[0][g]);
        if constexpr (r >= 2)
          acc[1][g] = mfma_32x32x16(wreg[r - 2][s][g], xf[r & 1][s], acc[1][g]);
      });
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (r + 2 < 9) read_row(std::integral_constant<int, r + 2>{}, xf[r & 1]);
    });
    // the next patch's region has landed; only then are this patch's stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int n, y0, x0;
    patch_origin(tile, n, y0, x0);
    const int ox = x0 + lrow;
    const uint4 nores[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    static_for<2>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
      DirectLane dl;
      dl.c2[0] = 32 * g + 8 * lhalf;
      dl.c2[1] = 32 * g + 16 + 8 * lhalf;
      dl.okc[0] = dl.c2[0] + 8 <= p.Cout;
      dl.okc[1] = dl.c2[1] + 8 <= p.Cout;
      dl.relu = relu;
      {
        // scale | shift of couts [32g + 8h, + 8) and [32g + 16 + 8h, + 8)
        h16x8 c[8];
        ds_read_b128_at<g * 128 + 0>(c[0], coef_a);
        ds_read_b128_at<g * 128 + 16>(c[1], coef_a);
        ds_read_b128_at<g * 128 + 64>(c[2], coef_a);
        ds_read_b128_at<g * 128 + 80>(c[3], coef_a);
        ds_read_b128_at<256 + g * 128 + 0>(c[4], coef_a);
        ds_read_b128_at<256 + g * 128 + 16>(c[5], coef_a);
        ds_read_b128_at<256 + g * 128 + 64>(c[6], coef_a);
        ds_read_b128_at<256 + g * 128 + 80>(c[7], coef_a);
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
      static_for<2>([&](auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
        const int oy = y0 + 2 * wave + b;
        const bool okp = oy < p.Ho && ox < p.Wo;
        const long long o_off = n * p.out_img_stride + (long long)oy * p.out_row_stride + (long long)ox * p.out_pix_stride;
        direct_tile(p, dl, acc[b][g], okp, okp ? o_off : 0, 0, false, nores, nullptr, nullptr, false);
      });
    });
  }
}

// The launches this kernel takes: the plan builder's 7x1 / stride (2, 1) form of the stem with
// 8-pixel windows every 2 pixels, bf16 output in whole 8-channel groups, nothing but
// scale/shift/ReLU in the epilogue.
bool conv_stem7_ok(const ConvParams& p) {
  return p.KH == 7 && p.KW == 1 && p.stride_y == 2 && p.stride_x == 1 && p.pad_y == 0 && p.pad_x == 0 &&
         p.Cin == 32 && p.in_pix_stride == 8 && p.Cout <= 64 && (p.Cout & 7) == 0 && !p.transposed &&
         !p.in_scale && p.vec_ok && p.out && p.out_dtype == MISEG_DT_BF16 && !p.residual && !p.out2 &&
         !p.stats && p.act != MISEG_ACT_SOFTMAX && p.act != MISEG_ACT_SIGMOID &&
         (((uintptr_t)p.in | (uintptr_t)p.weight) & 15) == 0 && (p.in_row_stride & 7) == 0 &&
         (p.in_img_stride & 7) == 0 && p.Ho == (p.H - 7) / 2 + 1 && p.Wo == p.W &&
         (long long)(p.W - 1) * 8 + 32 <= p.in_row_stride;
}

hipError_t launch_conv_stem7(ConvParams p, hipStream_t s) {
  p.tiles_y = (p.Ho + ST_TH - 1) / ST_TH;
  p.tiles_x = (p.Wo + ST_TW - 1) / ST_TW;
  p.mtiles = p.N * p.tiles_y * p.tiles_x;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int blocks = p.mtiles < 2 * cus ? p.mtiles : 2 * cus;
  hipLaunchKernelGGL(conv_stem7_kernel, dim3(blocks), dim3(256), 0, s, p);
  return hipGetLastError();
}

}  // namespace MISEG_KNS
