// U-Net tail in one kernel: the last decoder convolution (3x3, 32 -> 32, BatchNorm + ReLU;
// segmentation_models `decoder_stage4b`) followed by the segmentation head (3x3, 32 -> classes,
// bias, softmax / sigmoid; `final_conv` + activation) and the collapse_probabilities mask
// (/root/reference utils/model.py:176-186, utils/utils.py:50-68).  The 32-channel full-resolution
// tensor between the two convolutions never leaves the CU: it is the widest tensor of the network
// per unit of arithmetic (64 B per pixel for 18 kMAC), so writing and re-reading it is what bounds
// the two layers when they run separately.
//
// Persistent blocks (two per CU) walk over 8 x 32 patches of output pixels; the body and head
// weights are brought into LDS once per block.  Per patch, 4 waves:
//   1. The (8+4) x (32+4) x 32 input halo arrives by LDS-DMA in one of two buffers; the halo of
//      the block's NEXT patch is requested before this patch's arithmetic starts, so global
//      loads are in flight during all of it.
//   2. Body: the (8+2) x (32+2) = 340 intermediate pixels are 11 flattened 32-pixel MFMA tiles
//      (v_mfma_f32_32x32x16_bf16, three tiles per wave, one weight fragment feeding all three);
//      scale/shift + ReLU, pixels outside the image forced to zero (they are the head's zero
//      padding), bf16, written straight from the accumulator registers over the halo buffer the
//      body has just finished with (a lane holds runs of 4 consecutive channels of one pixel).
//   3. Head: v_mfma_f32_16x16x32_bf16 with the head weights as the 16-row operand (rows >= 4
//      alias rows 0..3 and are ignored), one MFMA per tap per 16 pixels; lanes 0..15 end up with
//      the logits of their pixel and apply bias, softmax / sigmoid and the collapse in registers.
// LDS: 2 x 27 KB halo + 18 KB body weights + 2.3 KB head weights + 5 KB output staging = 79.3 KB,
// two blocks per CU.  Every LDS access pattern of the kernel is checked against the bank model of
// gfx950 in scripts/analysis/lds_banks.py.
#include "conv_common.h"

namespace MISEG_KNS {

namespace {
constexpr int HF_TH = 8, HF_TW = 32;
constexpr int HF_IW = HF_TW + 4;                    // input halo width
constexpr int HF_IROWS = (HF_TH + 4) * HF_IW;       // 432 input pixels
constexpr int HF_NII = (HF_IROWS + 15) / 16;        // wave-instructions of 16 rows x 64 B
constexpr int HF_IN_BYTES = HF_NII * 1024;
constexpr int HF_MW = HF_TW + 2;                    // intermediate width
// The body walks the (8+2) x (32+2) intermediate pixels in rows as wide as the input halo (36): pixel
// idx of that flattening sits over halo pixel idx, so the 32 pixels of an MFMA tile are 32 CONSECUTIVE
// halo pixels under every tap and the swizzled 64-byte rows are read without bank conflicts (a
// ds_read_b128 is served in four groups of 16 lanes, {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same
// +32, over 64 banks: a run broken by a row end costs 6.8 instead of 4 cycles).  Columns 34, 35 of a row
// are computed and dropped.
constexpr int HF_BROWS = (HF_TH + 2) * HF_IW;       // 360 flattened body pixels, 340 of them kept
static_assert((HF_BROWS + 31) / 32 <= 12, "three body tiles per wave cover the intermediate");
constexpr int HF_NWI = 9 * 32 / 16;                 // body weight wave-instructions
constexpr int HF_BW_BYTES = HF_NWI * 1024;
constexpr int HF_CW_BYTES = 9 * 4 * 64;
// Intermediate tile: rows of 40 pixels (34 used), 64 bytes per pixel, the 16-byte chunk c of pixel p at
// chunk slot c ^ ((p >> 1) & 3).  The head's lanes (pixel hn = lane & 15, chunk kg = lane >> 4) then hit
// 64 distinct banks in every 16-lane group of their ds_read_b128 (a group mixes two chunk indices, which
// no linear pitch survives), and every tap / tile offset is a multiple of 8 pixels apart from dx, so
// one base address per dx plus immediates addresses all 36 fragments.
constexpr int HF_TROW = 40;
constexpr int HF_MID_BYTES = (HF_TH + 2) * HF_TROW * 64;
static_assert(HF_MID_BYTES <= HF_IN_BYTES, "the intermediate tile overlays a halo buffer");
constexpr int HF_BW_OFF = 2 * HF_IN_BYTES;
constexpr int HF_CW_OFF = HF_BW_OFF + HF_BW_BYTES;
constexpr int HF_SCR_OFF = HF_CW_OFF + HF_CW_BYTES;   // per-wave staging of logits / probabilities
constexpr int HF_SCR_WAVE = 1024 + 256;
constexpr int HF_SS_OFF = HF_SCR_OFF + 4 * HF_SCR_WAVE;   // body scale[32], shift[32] (fp32)
constexpr int HF_LDS = HF_SS_OFF + 256;
static_assert(2 * HF_LDS <= 160 * 1024, "two blocks per CU");
typedef f32x4_t f32x4;

// Workgroup barrier for LDS hand-offs between waves: this wave's LDS operations have completed
// (lgkmcnt), then s_barrier.  __syncthreads() would also drain vmcnt -- the LDS-DMA requests of the
// NEXT patch and the global stores of the previous one -- and so serialise what the pipeline is
// built to overlap; the DMA hand-off is ordered by the explicit vmcnt wait instead.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
}  // namespace

#ifdef MISEG_HF_PROFILE
// phase timers of wave w of block b (100 MHz ticks): [0] top barrier, [1] halo requests, [2] body
// arithmetic, [3] barrier + intermediate tile, [4] barrier + head arithmetic, [5] vmcnt wait,
// [6] activation + stores, [7] patches
__device__ unsigned long long g_hf_prof[4096][8];
#define HF_TICK(i) do { const unsigned long long t__ = wall_clock64(); hf_t[i] += t__ - hf_last; hf_last = t__; } while (0)
#else
#define HF_TICK(i) do { } while (0)
#endif

// body: the 32 -> 32 convolution (in, weight, scale, shift; its out pointers are unused);
// head: the 32 -> classes convolution (weight, scale, shift, act, out, collapse_out, ...).
__global__ __launch_bounds__(256, 2) void conv3x3_c32_head_kernel(const ConvParams p, const ConvParams hd) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[HF_LDS];
#ifdef MISEG_HF_PROFILE
  unsigned long long hf_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hf_last = wall_clock64();
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per_img = p.tiles_y * p.tiles_x;
  const int r16 = lane >> 2;
  const int slot = lane & 3;
  const h16_t* zero_src = reinterpret_cast<const h16_t*>(g_zero_line) + slot * 8;
  const int lrow = lane & 31;
  const int lhalf = lane >> 5;

  // patch -> (image, origin)
  auto patch_origin = [&](int tile, int& n, int& y0, int& x0) {
    n = tile / per_img;
    const int trem = tile - n * per_img;
    const int tyi = trem / p.tiles_x;
    y0 = tyi * HF_TH;
    x0 = (trem - tyi * p.tiles_x) * HF_TW;
  };
  // halo sources relative to the halo's first pixel (element offsets; -1 = a row past the halo):
  // patches whose halo lies inside the image need no per-lane bounds checks
  constexpr int NHJ = (HF_NII + 3) / 4;
  int h_rel[NHJ];
#pragma unroll
  for (int j = 0; j < NHJ; ++j) {
    const int hr = (j * 4 + wave) * 16 + r16;
    const int hy = hr / HF_IW;
    const int hx = hr - hy * HF_IW;
    h_rel[j] = hr < HF_IROWS ? hy * p.in_row_stride + hx * p.in_pix_stride + (slot ^ ((hr >> 2) & 3)) * 8 : -1;
  }
  // LDS-DMA of the input halo of `tile` into buffer `buf`
  auto issue_halo = [&](int tile, int buf) {
    int n, y0, x0;
    patch_origin(tile, n, y0, x0);
    const h16_t* img = p.in + n * p.in_img_stride;
    if (y0 >= 2 && y0 + HF_TH + 2 <= p.H && x0 >= 2 && x0 + HF_TW + 2 <= p.W) {   // block-uniform
      const h16_t* org = img + (long long)(y0 - 2) * p.in_row_stride + (long long)(x0 - 2) * p.in_pix_stride;
#pragma unroll
      for (int j = 0; j < NHJ; ++j) {
        const int i = j * 4 + wave;
        if (i < HF_NII) {
          const h16_t* src = h_rel[j] >= 0 ? org + h_rel[j] : zero_src;
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)src,
              (__attribute__((address_space(3))) void*)(lds + buf * HF_IN_BYTES + i * 1024), 16, 0, 0);
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < (HF_NII + 3) / 4; ++j) {
      const int i = j * 4 + wave;
      if (i < HF_NII) {
        const int hr = i * 16 + r16;
        const int hy = hr / HF_IW;
        const int hx = hr - hy * HF_IW;
        const int iy = y0 - 2 + hy;
        const int ix = x0 - 2 + hx;
        const bool ok = hr < HF_IROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const int ch = slot ^ ((hr >> 2) & 3);
        const h16_t* src = ok ? img + (long long)iy * p.in_row_stride + (long long)ix * p.in_pix_stride + ch * 8
                               : zero_src;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)src,
            (__attribute__((address_space(3))) void*)(lds + buf * HF_IN_BYTES + i * 1024), 16, 0, 0);
      }
    }
  };

  // ---- once per block: body weights by LDS-DMA, head weights [class][3][3][32] -> [tap][4][32] ----
#pragma unroll
  for (int j = 0; j < (HF_NWI + 3) / 4; ++j) {
    const int i = j * 4 + wave;
    if (i < HF_NWI) {
      const int wr = i * 16 + r16;      // LDS row = tap * 32 + cout
      const int tap = wr >> 5;
      const int co = wr & 31;
      const int ch = slot ^ ((wr >> 2) & 3);
      const h16_t* src = co < p.Cout ? p.weight + ((long long)co * 9 + tap) * 32 + ch * 8 : zero_src;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(lds + HF_BW_OFF + i * 1024), 16, 0, 0);
    }
  }
  if (tid < 9 * 4 * 4) {
    const int tap = tid >> 4, row = (tid >> 2) & 3, chunk = tid & 3;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < hd.Cout)
      v = *reinterpret_cast<const uint4*>(hd.weight + ((long long)row * 9 + tap) * 32 + chunk * 8);
    *reinterpret_cast<uint4*>(lds + HF_CW_OFF + (tap * 4 + row) * 64 + chunk * 16) = v;
  }

  if (tid < 64) {
    float v = tid < 32 ? 1.f : 0.f;
    if (tid < 32 && p.scale) v = p.scale[tid];
    if (tid >= 32 && p.shift) v = p.shift[tid - 32];
    reinterpret_cast<float*>(lds + HF_SS_OFF)[tid] = v;
  }

  // per-lane constants of the body tiles wave, wave + 4, wave + 8 (of 11)
  constexpr int BT = 3;
  int q0[BT], mr[BT], mc[BT], moff[BT];
  bool mval[BT];
#pragma unroll
  for (int t = 0; t < BT; ++t) {
    const int idx = (wave + 4 * t) * 32 + lrow;
    q0[t] = idx < HF_BROWS ? idx : HF_BROWS - 1;     // input-halo pixel under tap (0, 0)
    mr[t] = q0[t] / HF_IW;
    mc[t] = q0[t] - mr[t] * HF_IW;
    mval[t] = idx < HF_BROWS && mc[t] < HF_MW;
    const int tp = mr[t] * HF_TROW + mc[t];          // pixel of the intermediate tile
    // byte offset of this lane's 8 bytes of chunk 0; chunk g sits at moff ^ (g << 4)
    moff[t] = tp * 64 + (((tp >> 1) & 3) << 4) + lhalf * 8;
  }
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const unsigned lds_a = (unsigned)(unsigned long long)(lds_byte*)lds;   // LDS byte address of lds[0]
  const int hn = lane & 15;       // head: pixel within the 16-pixel tile
  const int kg = lane >> 4;       // head: 8-channel group
  const int HoWo = p.Ho * p.Wo;
  float hscale[4], hshift[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hscale[j] = (hd.scale && j < hd.Cout) ? hd.scale[j] : 1.f;
    hshift[j] = (hd.shift && j < hd.Cout) ? hd.shift[j] : 0.f;
  }

  int tile = blockIdx.x;
  if (tile < p.mtiles) issue_halo(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // weights and the first halo
  int cur = 0;
  for (; tile < p.mtiles; tile += gridDim.x, cur ^= 1) {
    // every wave has waited for its share of this patch's halo (before the loop, or before the
    // stores of the previous patch) and has left the previous patch, whose buffer the next
    // request overwrites
    lds_barrier();
    HF_TICK(0);
    if (tile + (int)gridDim.x < p.mtiles) issue_halo(tile + gridDim.x, cur ^ 1);
    int n, y0, x0;
    patch_origin(tile, n, y0, x0);
    HF_TICK(1);

    // ---- body: 18 K-steps (tap x 16-deep half); the fragments of step s + 2 are requested
    // right after the MFMAs of step s, so two steps of reads are always in flight ----
    f32x16 acc[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    {
      const unsigned inb_a = lds_a + cur * HF_IN_BYTES;
      const unsigned wb_a[2] = {lds_a + HF_BW_OFF + swz64(lrow, lhalf), lds_a + HF_BW_OFF + swz64(lrow, 2 + lhalf)};
      h16x8 fw[2], fx[2][BT];
      auto issue = [&](auto sc_) __attribute__((always_inline)) {
        constexpr int st = decltype(sc_)::value;
        constexpr int tap = st >> 1, kk = st & 1, dy = tap / 3, dx = tap % 3;
        const int ch = kk * 2 + lhalf;
        ds_read_b128_at<tap * 2048>(fw[st & 1], wb_a[kk]);   // swz64(tap * 32 + lrow, ch)
#pragma unroll
        for (int t = 0; t < BT; ++t)
          ds_read_b128_at<0>(fx[st & 1][t], inb_a + swz64(q0[t] + dy * HF_IW + dx, ch));
      };
      issue(std::integral_constant<int, 0>{});
      issue(std::integral_constant<int, 1>{});
      static_for<18>([&](auto sc_) __attribute__((always_inline)) {
        constexpr int st = decltype(sc_)::value;
        if constexpr (st + 1 < 18)
          lgkm_wait<BT + 1>(fw[st & 1], fx[st & 1][0], fx[st & 1][1], fx[st & 1][2]);
        else
          lgkm_wait<0>(fw[st & 1], fx[st & 1][0], fx[st & 1][1], fx[st & 1][2]);
#pragma unroll
        for (int t = 0; t < BT; ++t)
          acc[t] = mfma_32x32x16(fw[st & 1], fx[st & 1][t], acc[t]);
        if constexpr (st + 2 < 18) issue(std::integral_constant<int, st + 2>{});
      });
    }
    HF_TICK(2);
    lds_barrier();   // every wave is done reading the halo: the intermediate tile overlays it
    // scale/shift + ReLU -> bf16; lane: pixel lrow of the tile, channels 8g + 4*lhalf + i
    // (read by hand: a compiler-visible LDS read here would first drain vmcnt, i.e. wait for the
    // halo of the NEXT patch, because it cannot tell the read apart from the LDS-DMA's target)
    float4 sc[4], sh[4];
    {
      h16x8 raw[8];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        ds_read_b128_at<0>(raw[g], lds_a + HF_SS_OFF + (8 * g + 4 * lhalf) * 4);
        ds_read_b128_at<0>(raw[4 + g], lds_a + HF_SS_OFF + 128 + (8 * g + 4 * lhalf) * 4);
      }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]),
                     "+v"(raw[6]), "+v"(raw[7]));
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        sc[g] = __builtin_bit_cast(float4, raw[g]);
        sh[g] = __builtin_bit_cast(float4, raw[4 + g]);
      }
    }
    const unsigned mid_base = lds_a + cur * HF_IN_BYTES;
#pragma unroll
    for (int t = 0; t < BT; ++t) {
      if (mval[t]) {
        const bool inside = (unsigned)(y0 - 1 + mr[t]) < (unsigned)p.H && (unsigned)(x0 - 1 + mc[t]) < (unsigned)p.W;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v0 = acc[t][4 * g] * sc[g].x + sh[g].x;
          float v1 = acc[t][4 * g + 1] * sc[g].y + sh[g].y;
          float v2 = acc[t][4 * g + 2] * sc[g].z + sh[g].z;
          float v3 = acc[t][4 * g + 3] * sc[g].w + sh[g].w;
          const bool relu = p.act == MISEG_ACT_RELU;
          v0 = h16_act(v0, relu); v1 = h16_act(v1, relu); v2 = h16_act(v2, relu); v3 = h16_act(v3, relu);
          uint2 o = make_uint2(pack_h16x2_raw(v0, v1), pack_h16x2_raw(v2, v3));
          if (!inside) o = make_uint2(0, 0);
          // by hand for the same reason as the scale/shift reads above
          asm volatile("ds_write_b64 %0, %1" ::"v"(mid_base + (moff[t] ^ (g << 4))), "v"(o) : "memory");
        }
      }
    }
    HF_TICK(3);
    lds_barrier();

    // ---- head: wave -> output rows 2*wave, 2*wave + 1, two 16-pixel tiles per row ----
    f32x4 hacc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) hacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      // this lane's pixel (row 2*wave, x = hn + dx) and 8-channel group of the intermediate tile
      unsigned mid_a[3];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int tp = 2 * wave * HF_TROW + hn + dx;
        mid_a[dx] = lds_a + cur * HF_IN_BYTES + tp * 64 + ((kg ^ ((tp >> 1) & 3)) << 4);
      }
      const unsigned cw_a = lds_a + HF_CW_OFF + (lane & 3) * 64 + kg * 16;
      h16x8 fw[2], fx[2][4];
      auto issue = [&](auto tc_) __attribute__((always_inline)) {
        constexpr int tap = decltype(tc_)::value;
        constexpr int dy = tap / 3, dx = tap % 3;
        ds_read_b128_at<tap * 256>(fw[tap & 1], cw_a);
        static_for<4>([&](auto t_) __attribute__((always_inline)) {
          constexpr int t = decltype(t_)::value;
          ds_read_b128_at<(((t >> 1) + dy) * HF_TROW + (t & 1) * 16) * 64>(fx[tap & 1][t], mid_a[dx]);
        });
      };
      issue(std::integral_constant<int, 0>{});
      issue(std::integral_constant<int, 1>{});
      static_for<9>([&](auto tc_) __attribute__((always_inline)) {
        constexpr int tap = decltype(tc_)::value;
        if constexpr (tap + 1 < 9)
          lgkm_wait<5>(fw[tap & 1], fx[tap & 1][0], fx[tap & 1][1], fx[tap & 1][2], fx[tap & 1][3]);
        else
          lgkm_wait<0>(fw[tap & 1], fx[tap & 1][0], fx[tap & 1][1], fx[tap & 1][2], fx[tap & 1][3]);
#pragma unroll
        for (int t = 0; t < 4; ++t)
          hacc[t] = mfma_16x16x32(fw[tap & 1], fx[tap & 1][t], hacc[t]);
        if constexpr (tap + 2 < 9) issue(std::integral_constant<int, tap + 2>{});
      });
    }
    // Lanes 0..15 hold the logits of the 4 x 16 pixels of this wave (two output rows).  Through
    // LDS every lane takes one pixel (row = lane >> 5, x = lane & 31), evaluates the activation
    // and the collapse, and -- when the patch is whole and the rows are 16-byte aligned -- the
    // results go through LDS once more so that each row leaves as contiguous 16-byte stores.
    // The next patch's halo, requested two stages ago, must have landed before the loop comes
    // round; waiting here, BEFORE this patch's stores are issued, keeps the slow store
    // acknowledgements out of that wait (vmcnt retires in order).
    HF_TICK(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HF_TICK(5);
    {
      float* sw = reinterpret_cast<float*>(lds + HF_SCR_OFF + wave * HF_SCR_WAVE);
      uint8_t* sm = reinterpret_cast<uint8_t*>(sw) + 1024;
      if (lane < 16) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          *reinterpret_cast<float4*>(sw + (t * 16 + hn) * 4) =
              make_float4(hacc[t][0], hacc[t][1], hacc[t][2], hacc[t][3]);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const float4 lg = *reinterpret_cast<const float4*>(sw + lane * 4);
      const int C = hd.Cout;
      const int orow = y0 + 2 * wave + (lane >> 5);
      const int ox = x0 + (lane & 31);
      float v[4] = {lg.x, lg.y, lg.z, lg.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = v[j] * hscale[j] + hshift[j];
      float pr[4], cv[4];
      head_eval(hd, v, pr, cv);
      const long long o_row0 = n * hd.out_img_stride + (long long)(y0 + 2 * wave) * hd.out_row_stride +
                               (long long)x0 * hd.out_pix_stride;
      const long long m_row0 = ((long long)n * HoWo + (long long)(y0 + 2 * wave) * p.Wo + x0) * C;
      float* out = reinterpret_cast<float*>(hd.out);
      const bool whole = y0 + HF_TH <= p.Ho && x0 + HF_TW <= p.Wo && hd.out_pix_stride == C &&
                         (((uintptr_t)(out + o_row0) | ((uintptr_t)hd.out_row_stride * 4)) & 15) == 0 &&
                         (!hd.collapse_out ||
                          (((uintptr_t)(hd.collapse_out + m_row0) | ((uintptr_t)p.Wo * C)) & 15) == 0);
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
          *reinterpret_cast<float4*>(out + o_row0 + (long long)r * hd.out_row_stride + c * 4) =
              *reinterpret_cast<const float4*>(sw + r * 32 * C + c * 4);
        }
        const int cpm = 2 * C;             // 16-byte chunks per row of the mask
        if (hd.collapse_out && lane < 2 * cpm) {
          const int r = lane / cpm, c = lane - r * cpm;
          *reinterpret_cast<uint4*>(hd.collapse_out + m_row0 + (long long)r * p.Wo * C + c * 16) =
              *reinterpret_cast<const uint4*>(sm + r * 32 * C + c * 16);
        }
      } else if (orow < p.Ho && ox < p.Wo) {
        float* o = out + n * hd.out_img_stride + (long long)orow * hd.out_row_stride +
                   (long long)ox * hd.out_pix_stride;
        uint8_t* c8 = hd.collapse_out ? hd.collapse_out + ((long long)n * HoWo + (long long)orow * p.Wo + ox) * C
                                      : nullptr;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < C) {
            o[j] = pr[j];
            if (c8) c8[j] = (uint8_t)cv[j];
          }
      }
    }
    HF_TICK(6);
#ifdef MISEG_HF_PROFILE
    hf_t[7] += 1;
#endif
  }
#ifdef MISEG_HF_PROFILE
  if (lane == 0 && blockIdx.x < 1024)
    for (int i = 0; i < 8; ++i) g_hf_prof[blockIdx.x * 4 + wave][i] = hf_t[i];
#endif
}

hipError_t launch_c32_head(const ConvParams& body, const ConvParams& head, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL(conv3x3_c32_head_kernel, grid, dim3(256), 0, s, body, head);
  return hipGetLastError();
}

}  // namespace MISEG_KNS
