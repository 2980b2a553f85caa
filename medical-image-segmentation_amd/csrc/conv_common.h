// Pieces shared by the convolution kernels of libmiseg.so: launch parameters, the XCD-aware
// tile order, LDS swizzles and the epilogue (scale/shift, ReLU, residual, second output,
// softmax/sigmoid head with the collapse mask).
#pragma once
#include <utility>

#include "common.h"

namespace MISEG_KNS {

// Compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>).  Keeps
// accumulator arrays indexed by constants only (a runtime index would move them to scratch).
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

typedef f32x16_t f32x16;

struct ConvParams {
  const h16_t* in;
  const h16_t* weight;
  void* out;
  h16_t* out2;
  const void* residual;
  const float* scale;
  const float* shift;
  const float* scale2;
  const float* shift2;
  const float* in_scale;
  const float* in_shift;
  uint8_t* collapse_out;
  int N, H, W, Cin;
  int Ho, Wo, Cout;
  int KH, KW, stride_y, stride_x, pad_y, pad_x;
  int in_pix_stride, in_row_stride;
  long long in_img_stride;
  int out_pix_stride, out_row_stride;
  long long out_img_stride;
  int out2_pix_stride, out2_row_stride;
  long long out2_img_stride;
  int res_pix_stride, res_row_stride;
  long long res_img_stride;
  int out_dtype, res_dtype, act, transposed;
  float inv_temperature;
  int collapse_value;
  int M;        // N*Ho*Wo
  int mtiles;   // ceil(M / BM)
  int ntiles;   // ceil(Cout / BN)
  int vec_ok;   // strides/pointers allow 16-byte epilogue accesses
  int tiles_y, tiles_x;  // halo kernel: output patches per image
  int out2_linear;       // the second output without its ReLU: out2 = v * scale2 + shift2
  double* stats;         // optional per-channel sum / sum of squares of the stored bf16 output
  int stats_rep, stats_stride;
  int direct;            // generic kernels: finish large launches with the LDS-free epilogue
  int res_shift;         // the residual is read at (oy >> res_shift, ox >> res_shift): nearest up-sampling
  int res_after;         // the residual is added after the activation (out = act(v) + residual)
  int dgrad_t;           // data gradient of Conv2DTranspose(4, 2): `in` is dY, taps run over its four phase lattices
  int in_valid_c;        // channels of an input pixel that exist in memory (the rest read as zero); Cin when all do
  int res_mask_c;        // > 0: MISEG_RES_BN_RELU_MASK -- `residual` is x, scale/shift are the BatchNorm's forward
                         // coefficients (not applied); couts below res_mask_c are zeroed where x * scale + shift <= 0
                         // and the statistics are sum(out), sum(out * x); 0: the residual is added
};

// 16 zero bytes x 8: the source of LDS-DMA loads for taps that fall outside the image.
static __device__ __attribute__((aligned(128))) unsigned char g_zero_line[128];

// LDS tile addressing.  Rows of 64 B (BK=32) hold 4 sixteen-byte chunks, rows of 128 B
// (BK=64) hold 8; the chunk index is XOR-swizzled with row bits so that the 16 rows a
// ds_read_b128 lane group touches land on 16 distinct 16-B slots of the 256-B bank row.
__device__ __forceinline__ int swz64(int row, int chunk) {
  return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}
__device__ __forceinline__ int swz128(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

struct TileCoord {
  int m0, cout0, ph_a, ph_b, pad_y, pad_x;
  const h16_t* wbase;
};

// PHASES = false: a transpose-conv kernel that computes all four phases of a tile in one block
// (one work item per tile; the caller sets phase and weight base per wave).
template <int BM, int BN, bool PHASES = true>
__device__ __forceinline__ TileCoord tile_coord(const ConvParams& p) {
  // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each of the 8
  // groups a contiguous run of tiles, with the N-tiles of one M-tile adjacent.
  TileCoord t;
  // For transpose-conv the four phases of one tile are consecutive work items, so they run
  // on the same XCD at about the same time and their interleaved output pixels merge in L2.
  const bool phased = PHASES && p.transposed;
  const int nph = phased ? 4 : 1;
  const int ntot = p.mtiles * p.ntiles * nph;
  const int bid = blockIdx.x;
  const int q = ntot >> 3, r = ntot & 7;
  const int x = bid & 7, loc = bid >> 3;
  const int item = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
  const int phase = phased ? (item & 3) : 0;
  const int tile = phased ? (item >> 2) : item;
  const int mtile = tile / p.ntiles;
  const int ntile = tile - mtile * p.ntiles;
  t.m0 = mtile * BM;
  t.cout0 = ntile * BN;
  t.pad_y = p.pad_y;
  t.pad_x = p.pad_x;
  t.ph_a = t.ph_b = 0;
  t.wbase = p.weight;
  if (phased) {
    t.ph_a = phase >> 1;
    t.ph_b = phase & 1;
    t.pad_y = 1 - t.ph_a;
    t.pad_x = 1 - t.ph_b;
    t.wbase += (long long)phase * p.Cout * 4 * p.Cin;
  }
  return t;
}

// Softmax (with temperature) or sigmoid over the <= 4 logits of one pixel, fp32 probabilities to
// `o`, and the reference's collapse_probabilities mask to `c8` (utils/utils.py:50-68): channel i
// is compared in place with the already collapsed values (0 or collapse_value) of channels j < i.
__device__ __forceinline__ void head_eval(const ConvParams& p, const float* v, float (&pr)[4], float (&cv)[4]) {
  const int C = p.Cout;
  if (p.act == MISEG_ACT_SIGMOID) {
#pragma unroll
    for (int j = 0; j < 4; ++j) pr[j] = 1.f / (1.f + __expf(-v[j]));
  } else {
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < C) mx = fmaxf(mx, v[j] * p.inv_temperature);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pr[j] = j < C ? __expf(v[j] * p.inv_temperature - mx) : 0.f;
      sum += pr[j];
    }
    const float inv = 1.f / sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) pr[j] *= inv;
  }
  const float val = (float)p.collapse_value;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bool win = i < C;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j == i || j >= C) continue;
      const float other = j < i ? cv[j] : pr[j];
      win = win && (pr[i] > other);
    }
    cv[i] = win ? val : 0.f;
  }
}

__device__ __forceinline__ void head_store(const ConvParams& p, const float* v, float* o, uint8_t* c8) {
  const int C = p.Cout;
  float pr[4], cv[4];
  head_eval(p, v, pr, cv);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < C) o[j] = pr[j];
  if (c8) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < C) c8[j] = (uint8_t)cv[j];
  }
}

// ---------------------------------------------------------------------------------
// Epilogue.  The accumulators hold D[cout][pixel]; each wave transposes one 32-pixel
// tile at a time through its own LDS region ([32 pixels][NT*32 couts] fp32, rows padded
// by 16 B) so that every lane then owns 8 consecutive output channels of one pixel:
// 16-byte bf16 stores (32-byte fp32), residual loads of the same shape, and the
// channels of a pixel written by adjacent lanes.
// ---------------------------------------------------------------------------------
// Maps a pixel index inside a linear (row-major over N*Ho*Wo) block tile to coordinates.
struct LinearDecode {
  int m0, M, HoWo, Wo;
  __device__ __forceinline__ bool operator()(int idx, int& n, int& oy, int& ox, int& rem) const {
    const int m = m0 + idx;
    if (m >= M) return false;
    n = m / HoWo;
    rem = m - n * HoWo;
    oy = rem / Wo;
    ox = rem - oy * Wo;
    return true;
  }
};

template <int WARPS_N, int MT, int NT, class Decode, bool BNM = false>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, const TileCoord& tc,
                                              const Decode& decode, f32x16 (&acc)[NT][MT],
                                              unsigned char* lds, int wm, int wn, int lane,
                                              int ep_slot = -1) {
  constexpr int TNW = NT * 32;       // couts per wave tile
  constexpr int ROWF = TNW + 4;      // floats per LDS row
  constexpr int LPP = TNW / 8;       // lanes per pixel when reading back
  constexpr int PPP = 64 / LPP;      // pixels per pass
  constexpr int PASSES = 32 / PPP;
  const int wave = ep_slot >= 0 ? ep_slot : wm * WARPS_N + wn;   // which LDS region
  float* ep = reinterpret_cast<float*>(lds) + wave * (32 * ROWF);
  const int lrow = lane & 31, lhalf = lane >> 5;
  const int HoWo = p.Ho * p.Wo;
  const bool has_res = p.residual != nullptr && !BNM;
  const int px_l = lane / LPP;
  const int co_l = (lane % LPP) * 8;
  const int co = tc.cout0 + wn * TNW + co_l;
  const bool full = co + 8 <= p.Cout && p.vec_ok;
  const bool head = p.act == MISEG_ACT_SOFTMAX || p.act == MISEG_ACT_SIGMOID;

  // statistics of the stored values for a following BatchNormalization (this lane's 8 channels)
  float st_s[8], st_q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) st_s[j] = st_q[j] = 0.f;

  static_for<MT>([&](auto bc) __attribute__((always_inline)) {
    constexpr int b = decltype(bc)::value;
    // The region is private to this wave and a wave's LDS operations execute in order, so
    // no workgroup barrier is needed between the transposing writes and the reads below,
    // nor before the next tile overwrites the region: only the compiler must keep the order.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(ep + lrow * ROWF + a * 32 + 8 * g + 4 * lhalf) =
            make_float4(acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2],
                        acc[a][b][4 * g + 3]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      int px = ps * PPP + px_l;
      int co_lx = co_l;
      if (head) {
        // softmax / sigmoid head (Cout <= 4): one pixel per lane, one pass
        if (ps != 0 || lane >= 32) continue;
        px = lane;
        co_lx = 0;
      }
      const int cox = tc.cout0 + wn * TNW + co_lx;
      const bool fullx = !head && full;
      int n, oy, ox, rem;
      if (cox >= p.Cout || !decode(wm * (MT * 32) + b * 32 + px, n, oy, ox, rem)) continue;
      if (p.transposed) {
        oy = 2 * oy + tc.ph_a;
        ox = 2 * ox + tc.ph_b;
      }
      float v[8];
      {
        const float4 v0 = *reinterpret_cast<const float4*>(ep + px * ROWF + co_lx);
        const float4 v1 = *reinterpret_cast<const float4*>(ep + px * ROWF + co_lx + 4);
        v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w;
        v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
      }
      const long long o_off = n * p.out_img_stride + (long long)oy * p.out_row_stride +
                              (long long)ox * p.out_pix_stride + cox;
      const long long o2_off = n * p.out2_img_stride + (long long)oy * p.out2_row_stride +
                               (long long)ox * p.out2_pix_stride + cox;
      const long long r_off = n * p.res_img_stride + (long long)(oy >> p.res_shift) * p.res_row_stride +
                              (long long)(ox >> p.res_shift) * p.res_pix_stride + cox;
      if constexpr (BNM) {
        // data gradient behind relu(bn(x)) (ConvParams::res_mask_c; full, aligned channel groups are guaranteed
        // by the launcher): mask with the sign of x * scale + shift, reduce sum(g), sum(g * x)
        if (!fullx) continue;
        uint4 pk = make_uint4(pack_h16x2_raw(v[0], v[1]), pack_h16x2_raw(v[2], v[3]), pack_h16x2_raw(v[4], v[5]),
                              pack_h16x2_raw(v[6], v[7]));
        if (cox + 8 <= p.res_mask_c) {
          const uint4 r = *reinterpret_cast<const uint4*>(reinterpret_cast<const h16_t*>(p.residual) + r_off);
          const float xr[8] = {h16lo_to_f32(r.x), h16hi_to_f32(r.x), h16lo_to_f32(r.y), h16hi_to_f32(r.y),
                               h16lo_to_f32(r.z), h16hi_to_f32(r.z), h16lo_to_f32(r.w), h16hi_to_f32(r.w)};
          const float4 s0 = *reinterpret_cast<const float4*>(p.scale + cox);
          const float4 s1 = *reinterpret_cast<const float4*>(p.scale + cox + 4);
          const float4 t0 = *reinterpret_cast<const float4*>(p.shift + cox);
          const float4 t1 = *reinterpret_cast<const float4*>(p.shift + cox + 4);
          const float sc8[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
          const float sh8[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (xr[j] * sc8[j] + sh8[j] <= 0.f) v[j] = 0.f;
          pk = make_uint4(pack_h16x2_raw(v[0], v[1]), pack_h16x2_raw(v[2], v[3]), pack_h16x2_raw(v[4], v[5]),
                          pack_h16x2_raw(v[6], v[7]));
          if (p.stats) {
            const float g8[8] = {h16lo_to_f32(pk.x), h16hi_to_f32(pk.x), h16lo_to_f32(pk.y), h16hi_to_f32(pk.y),
                                 h16lo_to_f32(pk.z), h16hi_to_f32(pk.z), h16lo_to_f32(pk.w), h16hi_to_f32(pk.w)};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              st_s[j] += g8[j];
              st_q[j] += g8[j] * xr[j];
            }
          }
        }
        *reinterpret_cast<uint4*>(reinterpret_cast<h16_t*>(p.out) + o_off) = pk;
        continue;
      }
      if (fullx) {
        if (p.scale) {
          const float4 s0 = *reinterpret_cast<const float4*>(p.scale + cox);
          const float4 s1 = *reinterpret_cast<const float4*>(p.scale + cox + 4);
          v[0] *= s0.x; v[1] *= s0.y; v[2] *= s0.z; v[3] *= s0.w;
          v[4] *= s1.x; v[5] *= s1.y; v[6] *= s1.z; v[7] *= s1.w;
        }
        if (p.shift) {
          const float4 s0 = *reinterpret_cast<const float4*>(p.shift + cox);
          const float4 s1 = *reinterpret_cast<const float4*>(p.shift + cox + 4);
          v[0] += s0.x; v[1] += s0.y; v[2] += s0.z; v[3] += s0.w;
          v[4] += s1.x; v[5] += s1.y; v[6] += s1.z; v[7] += s1.w;
        }
        float rr[8];
        if (has_res) {
          if (p.res_dtype == MISEG_DT_F32) {
            const float* rp = reinterpret_cast<const float*>(p.residual) + r_off;
            const float4 r0 = *reinterpret_cast<const float4*>(rp);
            const float4 r1 = *reinterpret_cast<const float4*>(rp + 4);
            rr[0] = r0.x; rr[1] = r0.y; rr[2] = r0.z; rr[3] = r0.w;
            rr[4] = r1.x; rr[5] = r1.y; rr[6] = r1.z; rr[7] = r1.w;
          } else {
            const uint4 r = *reinterpret_cast<const uint4*>(
                reinterpret_cast<const h16_t*>(p.residual) + r_off);
            rr[0] = h16lo_to_f32(r.x); rr[1] = h16hi_to_f32(r.x);
            rr[2] = h16lo_to_f32(r.y); rr[3] = h16hi_to_f32(r.y);
            rr[4] = h16lo_to_f32(r.z); rr[5] = h16hi_to_f32(r.z);
            rr[6] = h16lo_to_f32(r.w); rr[7] = h16hi_to_f32(r.w);
          }
          if (!p.res_after) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rr[j];
          }
        }
        const bool radd = has_res && p.res_after;
        if (p.out) {
          float w[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) w[j] = (p.act == MISEG_ACT_RELU ? fmaxf(v[j], 0.f) : v[j]) + (radd ? rr[j] : 0.f);
          if (p.out_dtype == MISEG_DT_F32) {
            float* o = reinterpret_cast<float*>(p.out) + o_off;
            *reinterpret_cast<float4*>(o) = make_float4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(w[4], w[5], w[6], w[7]);
          } else {
            float u[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) u[j] = h16_act(v[j], p.act == MISEG_ACT_RELU);
            if (radd) {
#pragma unroll
              for (int j = 0; j < 8; ++j) u[j] = h16_act(u[j] + rr[j], false);
            }
            const uint4 pk = make_uint4(pack_h16x2_raw(u[0], u[1]), pack_h16x2_raw(u[2], u[3]),
                                        pack_h16x2_raw(u[4], u[5]), pack_h16x2_raw(u[6], u[7]));
            *reinterpret_cast<uint4*>(reinterpret_cast<h16_t*>(p.out) + o_off) = pk;
            if (p.stats) {
              const float r[8] = {h16lo_to_f32(pk.x), h16hi_to_f32(pk.x), h16lo_to_f32(pk.y),
                                  h16hi_to_f32(pk.y), h16lo_to_f32(pk.z), h16hi_to_f32(pk.z),
                                  h16lo_to_f32(pk.w), h16hi_to_f32(pk.w)};
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                st_s[j] += r[j];
                st_q[j] += r[j] * r[j];
              }
            }
          }
        }
        if (p.out2) {
          const float4 s0 = *reinterpret_cast<const float4*>(p.scale2 + cox);
          const float4 s1 = *reinterpret_cast<const float4*>(p.scale2 + cox + 4);
          const float4 t0 = *reinterpret_cast<const float4*>(p.shift2 + cox);
          const float4 t1 = *reinterpret_cast<const float4*>(p.shift2 + cox + 4);
          const bool r2 = !p.out2_linear;
          *reinterpret_cast<uint4*>(p.out2 + o2_off) = make_uint4(
              pack_h16x2_raw(h16_act(v[0] * s0.x + t0.x, r2), h16_act(v[1] * s0.y + t0.y, r2)),
              pack_h16x2_raw(h16_act(v[2] * s0.z + t0.z, r2), h16_act(v[3] * s0.w + t0.w, r2)),
              pack_h16x2_raw(h16_act(v[4] * s1.x + t1.x, r2), h16_act(v[5] * s1.y + t1.y, r2)),
              pack_h16x2_raw(h16_act(v[6] * s1.z + t1.z, r2), h16_act(v[7] * s1.w + t1.w, r2)));
        }
        continue;
      }
      // ---- partial channel group (Cout tail, unaligned strides, or the softmax head) ----
      float rp[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (cox + j < p.Cout) {
          if (p.scale) v[j] *= p.scale[cox + j];
          if (p.shift) v[j] += p.shift[cox + j];
          if (has_res) {
            rp[j] = p.res_dtype == MISEG_DT_F32
                        ? reinterpret_cast<const float*>(p.residual)[r_off + j]
                        : h16_to_f32(reinterpret_cast<const h16_t*>(p.residual)[r_off + j]);
            if (!p.res_after) v[j] += rp[j];
          }
        }
      }
      if (p.act == MISEG_ACT_SOFTMAX || p.act == MISEG_ACT_SIGMOID) {
        // Cout <= 4: the whole channel vector of this pixel sits in this lane (cox == 0).
        head_store(p, v, reinterpret_cast<float*>(p.out) + o_off,
                   p.collapse_out ? p.collapse_out + ((long long)n * HoWo + rem) * p.Cout : nullptr);
        continue;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (cox + j >= p.Cout) continue;
        if (p.out) {
          const float w = (p.act == MISEG_ACT_RELU ? fmaxf(v[j], 0.f) : v[j]) + ((has_res && p.res_after) ? rp[j] : 0.f);
          if (p.out_dtype == MISEG_DT_F32)
            reinterpret_cast<float*>(p.out)[o_off + j] = w;
          else
            reinterpret_cast<h16_t*>(p.out)[o_off + j] = f32_to_h16(w);
        }
        if (p.out2)
          p.out2[o2_off + j] = f32_to_h16(p.out2_linear ? v[j] * p.scale2[cox + j] + p.shift2[cox + j]
                                                         : fmaxf(v[j] * p.scale2[cox + j] + p.shift2[cox + j], 0.f));
      }
    }
  });
  if (p.stats) {
    // lanes with equal lane % LPP hold the same channels: butterfly over the pixel lanes, then the
    // first LPP lanes add the wave's totals to this block's replica
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int off = LPP; off < 64; off <<= 1) {
        st_s[j] += __shfl_xor(st_s[j], off, 64);
        st_q[j] += __shfl_xor(st_q[j], off, 64);
      }
    }
    // block-level sums in LDS (the transposing regions are free now), then one global fp64 atomic
    // per channel and block, issued by full waves.  The LDS sums are fp64 as well: the order in
    // which waves arrive must not show in the fp32 statistics derived from them.
    constexpr int BNB = WARPS_N * TNW;
    double* sl = reinterpret_cast<double*>(lds);
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * BNB; i += blockDim.x) sl[i] = 0.0;
    __syncthreads();
    if (lane < LPP) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsafeAtomicAdd(&sl[wn * TNW + co_l + j], (double)st_s[j]);
        unsafeAtomicAdd(&sl[BNB + wn * TNW + co_l + j], (double)st_q[j]);
      }
    }
    __syncthreads();
    double* dst = p.stats + (long long)(blockIdx.x % p.stats_rep) * 2 * p.stats_stride;
    for (int i = threadIdx.x; i < 2 * BNB; i += blockDim.x) {
      const int which = i / BNB, c = tc.cout0 + (i - which * BNB);
      const double v = sl[i];
      if (c < p.Cout && v != 0.0) unsafeAtomicAdd(dst + which * p.stats_stride + c, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Epilogue straight from the accumulators, without LDS, for one 32-cout group per wave and bf16
// outputs.  A 32x32 accumulator tile leaves lane (pixel j, half h) with couts 8g + 4h + r
// (g, r = 0..3); four v_permlane32_swap per 8 values trade the g = 1 (3) quads of the lower
// half-wave for the g = 0 (2) quads of the upper one, after which lane (j, h) owns couts
// [8h, 8h + 8) and [16 + 8h, 16 + 8h + 8) of its pixel: two 16-byte stores, and a store
// instruction writes 32 contiguous bytes per pixel.  Nothing here waits on anything but the
// residual loads, which are requested four pixel rows at a time.
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ bool direct_epilogue_ok(const ConvParams& p) {
  return p.vec_ok && p.out && p.out_dtype == MISEG_DT_BF16 && (p.Cout & 7) == 0 &&
         (!p.residual || p.res_dtype == MISEG_DT_BF16) && p.act != MISEG_ACT_SOFTMAX &&
         p.act != MISEG_ACT_SIGMOID;
}

// coefficients of couts [c, c + 8) (or `fill` when the vector is absent / the group is past Cout)
__device__ __forceinline__ void load_coef8(const float* src, int c, int Cout, float* d, float fill) {
  if (src && c + 8 <= Cout) {
    const float4 a = *reinterpret_cast<const float4*>(src + c);
    const float4 b = *reinterpret_cast<const float4*>(src + c + 4);
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w;
    d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = fill;
  }
}

// What a lane needs to finish accumulator tiles of one 32-cout group: its two cout octets and the
// coefficients of those 16 couts.
struct DirectLane {
  int c2[2];          // first cout of each octet
  bool okc[2];        // octet inside Cout
  bool okm[2];        // MISEG_RES_BN_RELU_MASK: octet below res_mask_c (masked and reduced)
  float sc[16], sh[16];
  bool relu;
};
// coefficients of the second output (BatchNorm + ReLU of the first) for the same couts
struct DirectOut2 {
  float s2[16], t2[16];
};
// BNM: the launch is a data gradient behind relu(bn(x)) (ConvParams::res_mask_c); a compile-time switch, so
// that the kernels' other instantiations carry none of it.
template <bool BNM = false>
__device__ __forceinline__ void direct_lane_init(const ConvParams& p, int cbase, int lane, DirectLane& d) {
  const int lhalf = lane >> 5;
  d.c2[0] = cbase + 8 * lhalf;
  d.c2[1] = cbase + 16 + 8 * lhalf;
  d.relu = p.act == MISEG_ACT_RELU;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    d.okc[q] = d.c2[q] + 8 <= p.Cout;
    d.okm[q] = BNM && d.c2[q] + 8 <= p.res_mask_c;
    // (as a BatchNorm-ReLU mask the vectors hold res_mask_c entries only)
    const int ncoef = BNM ? p.res_mask_c : p.Cout;
    load_coef8(p.scale, d.c2[q], ncoef, d.sc + 8 * q, 1.f);
    load_coef8(p.shift, d.c2[q], ncoef, d.sh + 8 * q, 0.f);
  }
}
__device__ __forceinline__ void direct_out2_init(const ConvParams& p, const DirectLane& d, DirectOut2& o) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    load_coef8(p.scale2, d.c2[q], p.Cout, o.s2 + 8 * q, 1.f);
    load_coef8(p.shift2, d.c2[q], p.Cout, o.t2 + 8 * q, 0.f);
  }
}

// The same for the data gradient behind relu(bn(x)) (ConvParams::res_mask_c): `res` holds x, d.sc / d.sh the
// BatchNorm's forward coefficients; octet by octet, so that x lives in registers only while its octet is finished.
__device__ __forceinline__ void direct_tile_mask(const ConvParams& p, const DirectLane& d, const f32x16& a, bool okp,
                                                 long long o_off, const uint4 (&res)[2], float* st, bool do_stats) {
  float v[16];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[8 * q + r]),
                                                       __float_as_uint(a[8 * q + 4 + r]), false, false);
      v[8 * q + r] = __uint_as_float(sw[0]);
      v[8 * q + 4 + r] = __uint_as_float(sw[1]);
    }
  static_for<2>([&](auto qc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value;
    if (okp && d.okc[q]) {
      float xr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = v[8 * q + j];
      if (d.okm[q]) {
        const uint4 r = res[q];
        xr[0] = h16lo_to_f32(r.x); xr[1] = h16hi_to_f32(r.x); xr[2] = h16lo_to_f32(r.y); xr[3] = h16hi_to_f32(r.y);
        xr[4] = h16lo_to_f32(r.z); xr[5] = h16hi_to_f32(r.z); xr[6] = h16lo_to_f32(r.w); xr[7] = h16hi_to_f32(r.w);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          w[j] = xr[j] * d.sc[8 * q + j] + d.sh[8 * q + j] <= 0.f ? 0.f : w[j];
      }
      const uint4 pk = make_uint4(pack_h16x2_raw(w[0], w[1]), pack_h16x2_raw(w[2], w[3]), pack_h16x2_raw(w[4], w[5]),
                                  pack_h16x2_raw(w[6], w[7]));
      *reinterpret_cast<uint4*>(reinterpret_cast<h16_t*>(p.out) + o_off + d.c2[q]) = pk;
      if (do_stats && d.okm[q]) {
        const float g[8] = {h16lo_to_f32(pk.x), h16hi_to_f32(pk.x), h16lo_to_f32(pk.y), h16hi_to_f32(pk.y),
                            h16lo_to_f32(pk.z), h16hi_to_f32(pk.z), h16lo_to_f32(pk.w), h16hi_to_f32(pk.w)};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          st[8 * q + j] += g[j];
          st[16 + 8 * q + j] += g[j] * xr[j];
        }
      }
    }
  });
}

// One 32x32 accumulator tile -> memory.  EVERY lane of the wave must call this (the swaps are
// cross-lane); `okp` says whether this lane's pixel exists.  o_off / o2_off: element offsets of
// the pixel in out / out2; res: the residual's two octets (used when has_res); d2: null when
// there is no second output; st: null, or this lane's running sum [0, 16) and sum of squares
// [16, 32) of the stored (bf16-rounded) values of its 16 couts.
// OUT1 = false: the launch has no first output (p.out == nullptr), only the second one.
template <bool OUT1 = true>
__device__ __forceinline__ void direct_tile(const ConvParams& p, const DirectLane& d, const f32x16& a,
                                            bool okp, long long o_off, long long o2_off, bool has_res,
                                            const uint4 (&res)[2], const DirectOut2* d2, float* st, bool do_stats) {
  float v[16];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[8 * q + r]),
                                                       __float_as_uint(a[8 * q + 4 + r]), false, false);
      v[8 * q + r] = __uint_as_float(sw[0]);
      v[8 * q + 4 + r] = __uint_as_float(sw[1]);
    }
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = v[j] * d.sc[j] + d.sh[j];
  float rr[16];
  const bool radd = has_res && p.res_after;
  if (has_res) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint4 r = res[q];
      rr[8 * q + 0] = h16lo_to_f32(r.x); rr[8 * q + 1] = h16hi_to_f32(r.x);
      rr[8 * q + 2] = h16lo_to_f32(r.y); rr[8 * q + 3] = h16hi_to_f32(r.y);
      rr[8 * q + 4] = h16lo_to_f32(r.z); rr[8 * q + 5] = h16hi_to_f32(r.z);
      rr[8 * q + 6] = h16lo_to_f32(r.w); rr[8 * q + 7] = h16hi_to_f32(r.w);
    }
    if (!radd) {
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] += rr[j];
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (!(okp && d.okc[q])) continue;
    const float* w = v + 8 * q;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = h16_act(w[j], d.relu);
    if (radd) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = h16_act(o[j] + rr[8 * q + j], false);
    }
    const uint4 pk = make_uint4(pack_h16x2_raw(o[0], o[1]), pack_h16x2_raw(o[2], o[3]), pack_h16x2_raw(o[4], o[5]),
                                pack_h16x2_raw(o[6], o[7]));
    if constexpr (OUT1) *reinterpret_cast<uint4*>(reinterpret_cast<h16_t*>(p.out) + o_off + d.c2[q]) = pk;
    if (do_stats) {
      const float r[8] = {h16lo_to_f32(pk.x), h16hi_to_f32(pk.x), h16lo_to_f32(pk.y), h16hi_to_f32(pk.y),
                          h16lo_to_f32(pk.z), h16hi_to_f32(pk.z), h16lo_to_f32(pk.w), h16hi_to_f32(pk.w)};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        st[8 * q + j] += r[j];
        st[16 + 8 * q + j] += r[j] * r[j];
      }
    }
    if (d2) {
      const float* s = d2->s2 + 8 * q;
      const float* t = d2->t2 + 8 * q;
      const bool r2 = !p.out2_linear;
      *reinterpret_cast<uint4*>(p.out2 + o2_off + d.c2[q]) = make_uint4(
          pack_h16x2_raw(h16_act(w[0] * s[0] + t[0], r2), h16_act(w[1] * s[1] + t[1], r2)),
          pack_h16x2_raw(h16_act(w[2] * s[2] + t[2], r2), h16_act(w[3] * s[3] + t[3], r2)),
          pack_h16x2_raw(h16_act(w[4] * s[4] + t[4], r2), h16_act(w[5] * s[5] + t[5], r2)),
          pack_h16x2_raw(h16_act(w[6] * s[6] + t[6], r2), h16_act(w[7] * s[7] + t[7], r2)));
    }
  }
}

// All MT pixel rows of a wave (pixel index = wm * MT * 32 + b * 32 + lane % 32 through `decode`).
template <int MT, class Decode, bool BNM = false, bool OUT1 = true>
__device__ __forceinline__ void conv_epilogue_direct(const ConvParams& p, const TileCoord& tc,
                                                     const Decode& decode, f32x16 (&acc)[MT],
                                                     int wm, int lane, const DirectLane& d,
                                                     float (&st)[32], bool do_stats) {
  const int lrow = lane & 31;
  const bool has_res = p.residual != nullptr;
  DirectOut2 o2;
  if (p.out2) direct_out2_init(p, d, o2);
  constexpr int RG = MT < 2 ? MT : 2;   // pixel rows per residual request group
  static_assert(MT % RG == 0, "row groups");
  static_for<MT / RG>([&](auto gc) __attribute__((always_inline)) {
    constexpr int b0 = decltype(gc)::value * RG;
    bool okp[RG];
    long long o_off[RG], o2_off[RG];
    uint4 res[RG][2];
#pragma unroll
    for (int i = 0; i < RG; ++i) {
      int n, oy, ox, rem;
      okp[i] = decode(wm * (MT * 32) + (b0 + i) * 32 + lrow, n, oy, ox, rem);
      if (!okp[i]) n = oy = ox = 0;
      if (p.transposed) {
        oy = 2 * oy + tc.ph_a;
        ox = 2 * ox + tc.ph_b;
      }
      o_off[i] = n * p.out_img_stride + (long long)oy * p.out_row_stride + (long long)ox * p.out_pix_stride;
      o2_off[i] = n * p.out2_img_stride + (long long)oy * p.out2_row_stride + (long long)ox * p.out2_pix_stride;
      if (has_res) {
        const long long r_off = n * p.res_img_stride + (long long)(oy >> p.res_shift) * p.res_row_stride +
                                (long long)(ox >> p.res_shift) * p.res_pix_stride;
#pragma unroll
        for (int q = 0; q < 2; ++q)
          res[i][q] = okp[i] && d.okc[q] && (!BNM || d.okm[q])
                          ? *reinterpret_cast<const uint4*>(reinterpret_cast<const h16_t*>(p.residual) + r_off + d.c2[q])
                          : make_uint4(0, 0, 0, 0);
      }
    }
    static_for<RG>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      if constexpr (BNM)
        direct_tile_mask(p, d, acc[b0 + i], okp[i], o_off[i], res[i], st, do_stats);
      else
        direct_tile<OUT1>(p, d, acc[b0 + i], okp[i], o_off[i], o2_off[i], has_res, res[i], p.out2 ? &o2 : nullptr, st, do_stats);
    });
  });
}

// A wave tile of NT cout groups: group a covers couts [cbase + 32 a, cbase + 32 a + 32).  BNB =
// couts per block (for the statistics' block-level sums in `lds`, which every wave of the block
// must reach).
template <int MT, int NT, int BNB, class Decode, bool BNM = false, bool OUT1 = true>
__device__ __forceinline__ void conv_epilogue_direct_groups(const ConvParams& p, const TileCoord& tc,
                                                            const Decode& decode, f32x16 (&acc)[NT][MT],
                                                            int wm, int cbase, int lane, unsigned char* lds) {
  const bool stats = p.stats != nullptr;
  float st[NT][32];
  static_for<NT>([&](auto ac) __attribute__((always_inline)) {
    constexpr int a = decltype(ac)::value;
#pragma unroll
    for (int j = 0; j < 32; ++j) st[a][j] = 0.f;
    if (cbase + a * 32 < p.Cout) {
      DirectLane dl;
      direct_lane_init<BNM>(p, cbase + a * 32, lane, dl);
      conv_epilogue_direct<MT, Decode, BNM, OUT1>(p, tc, decode, acc[a], wm, lane, dl, st[a], stats);
    }
  });
  if (stats) {
    // lanes of one half-wave hold the same couts: butterfly over the 32 pixel lanes, then block-
    // level fp64 sums in LDS (the order in which waves arrive must not show in the fp32
    // statistics derived from them) and one global fp64 atomic per channel and block
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int j = 0; j < 32; ++j)
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) st[a][j] += __shfl_xor(st[a][j], off, 64);
    double* sl = reinterpret_cast<double*>(lds);
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * BNB; i += blockDim.x) sl[i] = 0.0;
    __syncthreads();
    if ((lane & 31) == 0) {
      const int lhalf = lane >> 5;
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int c = cbase + a * 32 + 16 * q + 8 * lhalf - tc.cout0;
          if (c + tc.cout0 + 8 <= p.Cout && c + 8 <= BNB) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              unsafeAtomicAdd(&sl[c + j], (double)st[a][8 * q + j]);
              unsafeAtomicAdd(&sl[BNB + c + j], (double)st[a][16 + 8 * q + j]);
            }
          }
        }
    }
    __syncthreads();
    double* dst = p.stats + (long long)(blockIdx.x % p.stats_rep) * 2 * p.stats_stride;
    for (int i = threadIdx.x; i < 2 * BNB; i += blockDim.x) {
      const int which = i / BNB, c = tc.cout0 + (i - which * BNB);
      const double v = sl[i];
      if (c < p.Cout && v != 0.0) unsafeAtomicAdd(dst + which * p.stats_stride + c, v);
    }
  }
}

template <int MT, int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NT][MT]) {
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int TH, int TW>
struct PatchDecode {
  int n, y0, x0, Ho, Wo;
  __device__ __forceinline__ bool operator()(int idx, int& n_, int& oy, int& ox, int& rem) const {
    oy = y0 + idx / TW;
    ox = x0 + idx % TW;
    if (oy >= Ho || ox >= Wo) return false;
    n_ = n;
    rem = oy * Wo + ox;
    return true;
  }
};

// LDS read the compiler does not track: the caller waits with lgkm_wait before using dst.
template <int OFF>
__device__ __forceinline__ void ds_read_b128_at(h16x8& dst, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read immediate offset is 16 bits");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// Waits until at most CNT LDS operations are outstanding; `frag` is threaded through so that
// its consumer cannot be scheduled above the wait.
template <int CNT>
__device__ __forceinline__ void lgkm_wait(h16x8& frag) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(CNT));
}


// Waits until at most CNT LDS operations are outstanding, threading several fragments through.
template <int CNT>
__device__ __forceinline__ void lgkm_wait(h16x8& a, h16x8& b, h16x8& c, h16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(CNT));
}
template <int CNT>
__device__ __forceinline__ void lgkm_wait(h16x8& a, h16x8& b, h16x8& c, h16x8& d, h16x8& e) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(CNT));
}

// Waits for every outstanding LDS operation; the 12 fragments of a K step are threaded through.
__device__ __forceinline__ void lgkm_wait12(h16x8 (&x)[12]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]),
                 "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]));
}

}  // namespace MISEG_KNS
