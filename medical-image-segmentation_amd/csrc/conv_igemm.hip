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
// Block = 256 threads = 4 waves.  Per K-step (one tap x 32 input channels) the block
// stages a [BM pixels][32] and a [BN couts][32] bf16 tile through registers into
// double-buffered LDS (one barrier per K-step, next tile's global loads issued before
// the current tile's MFMAs).  Rows are 64 B; the 16-B chunk index is XOR-swizzled with
// row bits 2..3 so ds_read_b128 fragment reads and ds_write_b128 staging writes are
// bank-conflict free.
#include "common.h"

namespace miseg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct ConvParams {
  const bf16_t* in;
  const bf16_t* weight;
  void* out;
  bf16_t* out2;
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
};

__device__ __forceinline__ int swz(int row, int chunk) {
  return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}

template <int WARPS_M, int WARPS_N, int MT, int NT, bool PRO>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
  constexpr int BM = WARPS_M * MT * 32;
  constexpr int BN = WARPS_N * NT * 32;
  constexpr int PA = BM / 64;
  constexpr int PB = (BN + 63) / 64;
  static_assert(WARPS_M * WARPS_N == 4, "4 waves per block");
  static_assert(BM % 64 == 0, "BM multiple of 64");

  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (BM + BN) * 64];
  constexpr int STAGE = (BM + BN) * 64;  // bytes per pipeline stage: A tile then B tile
  constexpr int BOFF = BM * 64;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WARPS_N;
  const int wn = wave % WARPS_N;

  // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each of
  // the 8 groups a contiguous run of tiles, N-tiles of one M-tile adjacent.
  const int ntot = p.mtiles * p.ntiles;
  int tile;
  {
    const int bid = blockIdx.x;
    const int q = ntot >> 3, r = ntot & 7;
    const int x = bid & 7, loc = bid >> 3;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
  }
  const int mtile = tile / p.ntiles;
  const int ntile = tile - mtile * p.ntiles;
  const int m0 = mtile * BM;
  const int cout0 = ntile * BN;

  int pad_y = p.pad_y, pad_x = p.pad_x;
  int ph_a = 0, ph_b = 0;
  const bf16_t* wbase = p.weight;
  if (p.transposed) {
    const int phase = blockIdx.y;
    ph_a = phase >> 1;
    ph_b = phase & 1;
    pad_y = 1 - ph_a;
    pad_x = 1 - ph_b;
    wbase += (long long)phase * p.Cout * 4 * p.Cin;
  }

  const int KK = p.KH * p.KW;
  const int nchunk = p.Cin >> 5;
  const int ksteps = KK * nchunk;
  const int HoWo = p.Ho * p.Wo;

  // ---- staging assignment: thread -> (row, 16-byte chunk) ----
  const int srow = tid >> 2;
  const int chunk = tid & 3;

  const bf16_t* a_ptr[PA];
  int a_iy0[PA], a_ix0[PA];
  bool a_ok[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = m0 + i * 64 + srow;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    const int n = mm / HoWo;
    const int rem = mm - n * HoWo;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    a_iy0[i] = oy * p.stride_y - pad_y;
    a_ix0[i] = ox * p.stride_x - pad_x;
    a_ptr[i] = p.in + n * p.in_img_stride + (long long)a_iy0[i] * p.in_row_stride +
               (long long)a_ix0[i] * p.in_pix_stride + chunk * 8;
  }
  const bf16_t* w_ptr[PB];
  bool w_ok[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int co = cout0 + i * 64 + srow;
    w_ok[i] = (BN >= 64 || srow < BN) && co < p.Cout;
    w_ptr[i] = wbase + (long long)(w_ok[i] ? co : 0) * KK * p.Cin + chunk * 8;
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
          v.x = pack_bf16x2(fmaxf(bf16lo_to_f32(v.x) * s0.x + t0.x, 0.f),
                            fmaxf(bf16hi_to_f32(v.x) * s0.y + t0.y, 0.f));
          v.y = pack_bf16x2(fmaxf(bf16lo_to_f32(v.y) * s0.z + t0.z, 0.f),
                            fmaxf(bf16hi_to_f32(v.y) * s0.w + t0.w, 0.f));
          v.z = pack_bf16x2(fmaxf(bf16lo_to_f32(v.z) * s1.x + t1.x, 0.f),
                            fmaxf(bf16hi_to_f32(v.z) * s1.y + t1.y, 0.f));
          v.w = pack_bf16x2(fmaxf(bf16lo_to_f32(v.w) * s1.z + t1.z, 0.f),
                            fmaxf(bf16hi_to_f32(v.w) * s1.w + t1.w, 0.f));
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
    // advance cursor
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
    for (int i = 0; i < PA; ++i) {
      const int row = i * 64 + srow;
      *reinterpret_cast<uint4*>(lds + buf * STAGE + swz(row, chunk)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int row = i * 64 + srow;
      if (BN >= 64 || srow < BN) *reinterpret_cast<uint4*>(lds + buf * STAGE + BOFF + swz(row, chunk)) = rb[i];
    }
  };

  f32x16 acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int lrow = lane & 31;
  const int lhalf = lane >> 5;

  auto compute = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = kk * 2 + lhalf;
      bf16x8 xf[MT], wf[NT];
#pragma unroll
      for (int b = 0; b < MT; ++b) {
        const int row = wm * (MT * 32) + b * 32 + lrow;
        xf[b] = *reinterpret_cast<const bf16x8*>(lds + buf * STAGE + swz(row, ch));
      }
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        const int row = wn * (NT * 32) + a * 32 + lrow;
        wf[a] = *reinterpret_cast<const bf16x8*>(lds + buf * STAGE + BOFF + swz(row, ch));
      }
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[a], xf[b], acc[a][b], 0, 0, 0);
    }
  };

  // ---- main loop ----
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

  // ---- epilogue: each lane owns runs of 4 consecutive couts of one pixel ----
  const bool has_res = p.residual != nullptr;
#pragma unroll
  for (int b = 0; b < MT; ++b) {
    const int m = m0 + wm * (MT * 32) + b * 32 + lrow;
    if (m >= p.M) continue;
    const int n = m / HoWo;
    const int rem = m - n * HoWo;
    int oy = rem / p.Wo;
    int ox = rem - oy * p.Wo;
    if (p.transposed) {
      oy = 2 * oy + ph_a;
      ox = 2 * ox + ph_b;
    }
    const long long o_off = n * p.out_img_stride + (long long)oy * p.out_row_stride +
                            (long long)ox * p.out_pix_stride;
    const long long o2_off = n * p.out2_img_stride + (long long)oy * p.out2_row_stride +
                             (long long)ox * p.out2_pix_stride;
    const long long r_off = n * p.res_img_stride + (long long)oy * p.res_row_stride +
                            (long long)ox * p.res_pix_stride;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = cout0 + wn * (NT * 32) + a * 32 + 8 * g + 4 * lhalf;
        if (co >= p.Cout) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[a][b][4 * g + j];
        const bool full = co + 4 <= p.Cout;
        if (full) {
          if (p.scale) {
            const float4 s = *reinterpret_cast<const float4*>(p.scale + co);
            v[0] *= s.x; v[1] *= s.y; v[2] *= s.z; v[3] *= s.w;
          }
          if (p.shift) {
            const float4 s = *reinterpret_cast<const float4*>(p.shift + co);
            v[0] += s.x; v[1] += s.y; v[2] += s.z; v[3] += s.w;
          }
          if (has_res) {
            if (p.res_dtype == MISEG_DT_F32) {
              const float4 r = *reinterpret_cast<const float4*>(
                  reinterpret_cast<const float*>(p.residual) + r_off + co);
              v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
            } else {
              const uint2 r = *reinterpret_cast<const uint2*>(
                  reinterpret_cast<const bf16_t*>(p.residual) + r_off + co);
              v[0] += bf16lo_to_f32(r.x); v[1] += bf16hi_to_f32(r.x);
              v[2] += bf16lo_to_f32(r.y); v[3] += bf16hi_to_f32(r.y);
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (co + j < p.Cout) {
              if (p.scale) v[j] *= p.scale[co + j];
              if (p.shift) v[j] += p.shift[co + j];
              if (has_res) {
                v[j] += p.res_dtype == MISEG_DT_F32
                            ? reinterpret_cast<const float*>(p.residual)[r_off + co + j]
                            : bf16_to_f32(reinterpret_cast<const bf16_t*>(p.residual)[r_off + co + j]);
              }
            }
          }
        }
        if (p.act == MISEG_ACT_SOFTMAX || p.act == MISEG_ACT_SIGMOID) {
          // Cout <= 4: the whole channel vector of this pixel sits in this lane.
          const int C = p.Cout;
          float pr[4];
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
          float* o = reinterpret_cast<float*>(p.out) + o_off;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < C) o[j] = pr[j];
          if (p.collapse_out) {
            // in-place semantics of the reference loop: channel i is compared with the
            // already collapsed values (0 or collapse_value) of channels j < i.
            float cv[4];
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
            uint8_t* c8 = p.collapse_out + ((long long)n * HoWo + rem) * C;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < C) c8[j] = (uint8_t)cv[j];
          }
          continue;
        }
        if (p.out) {
          float w[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = p.act == MISEG_ACT_RELU ? fmaxf(v[j], 0.f) : v[j];
          if (p.out_dtype == MISEG_DT_F32) {
            float* o = reinterpret_cast<float*>(p.out) + o_off + co;
            if (full) {
              *reinterpret_cast<float4*>(o) = make_float4(w[0], w[1], w[2], w[3]);
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (co + j < p.Cout) o[j] = w[j];
            }
          } else {
            bf16_t* o = reinterpret_cast<bf16_t*>(p.out) + o_off + co;
            if (full) {
              *reinterpret_cast<uint2*>(o) =
                  make_uint2(pack_bf16x2(w[0], w[1]), pack_bf16x2(w[2], w[3]));
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (co + j < p.Cout) o[j] = f32_to_bf16(w[j]);
            }
          }
        }
        if (p.out2) {
          float w[4];
          if (full) {
            const float4 s = *reinterpret_cast<const float4*>(p.scale2 + co);
            const float4 t = *reinterpret_cast<const float4*>(p.shift2 + co);
            w[0] = fmaxf(v[0] * s.x + t.x, 0.f);
            w[1] = fmaxf(v[1] * s.y + t.y, 0.f);
            w[2] = fmaxf(v[2] * s.z + t.z, 0.f);
            w[3] = fmaxf(v[3] * s.w + t.w, 0.f);
            *reinterpret_cast<uint2*>(p.out2 + o2_off + co) =
                make_uint2(pack_bf16x2(w[0], w[1]), pack_bf16x2(w[2], w[3]));
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (co + j < p.Cout)
                p.out2[o2_off + co + j] =
                    f32_to_bf16(fmaxf(v[j] * p.scale2[co + j] + p.shift2[co + j], 0.f));
          }
        }
      }
    }
  }
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
static hipError_t launch_cfg(const ConvParams& p, dim3 grid, bool pro, hipStream_t s) {
  if (pro)
    hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, MT, NT, true>), grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, MT, NT, false>), grid, dim3(256), 0, s, p);
  return hipGetLastError();
}

static int choose_cfg(int M, int Cout) {
  // Prefer the widest cout tile the layer fills, then the tallest pixel tile that still
  // leaves at least ~2 blocks per CU.
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
  MISEG_CHECK_ARG((d->in_pix_stride & 7) == 0 && (d->in_row_stride & 7) == 0 &&
                      (d->in_img_stride & 7) == 0,
                  "conv: input strides must be multiples of 8 elements (16-byte loads)");
  MISEG_CHECK_ARG(((uintptr_t)d->in & 15) == 0 && ((uintptr_t)d->weight & 15) == 0,
                  "conv: input/weight must be 16-byte aligned");
  if (d->transposed) {
    MISEG_CHECK_ARG(d->KH == 2 && d->KW == 2 && d->stride_y == 1 && d->stride_x == 1 &&
                        d->Ho == d->H && d->Wo == d->W,
                    "conv: transposed mode needs KH=KW=2, stride 1, Ho=H, Wo=W");
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
  if (d->in_scale || d->in_shift) {
    MISEG_CHECK_ARG(d->in_scale && d->in_shift, "conv: prologue needs both in_scale and in_shift");
    MISEG_CHECK_ARG(((uintptr_t)d->in_scale & 15) == 0 && ((uintptr_t)d->in_shift & 15) == 0,
                    "conv: prologue vectors must be 16-byte aligned");
  }
  if (d->scale) MISEG_CHECK_ARG(((uintptr_t)d->scale & 15) == 0, "conv: scale misaligned");
  if (d->shift) MISEG_CHECK_ARG(((uintptr_t)d->shift & 15) == 0, "conv: shift misaligned");
  const long long M = (long long)d->N * d->Ho * d->Wo;
  if (M >= (1ll << 31)) {
    set_error("conv: N*Ho*Wo=%lld exceeds 2^31", M);
    return MISEG_E_UNSUPPORTED_SHAPE;
  }
  MISEG_CHECK_ARG(d->tile_hint >= 0 && d->tile_hint <= (int)(sizeof(kCfgs) / sizeof(kCfgs[0])),
                  "conv: bad tile_hint");
  return MISEG_OK;
}

int launch_conv(const miseg_conv_desc* d, hipStream_t stream) {
  int rc = validate_conv(d);
  if (rc != MISEG_OK) return rc;
  ConvParams p;
  p.in = (const bf16_t*)d->in;
  p.weight = (const bf16_t*)d->weight;
  p.out = d->out;
  p.out2 = (bf16_t*)d->out2;
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
  p.M = d->N * d->Ho * d->Wo;
  const int cfg = d->tile_hint ? d->tile_hint : choose_cfg(p.M, d->Cout);
  const TileCfg tc = kCfgs[cfg - 1];
  p.mtiles = (p.M + tc.bm - 1) / tc.bm;
  p.ntiles = (d->Cout + tc.bn - 1) / tc.bn;
  const dim3 grid(p.mtiles * p.ntiles, d->transposed ? 4 : 1, 1);
  const bool pro = d->in_scale != nullptr;
  hipError_t e;
  switch (cfg) {
    case 1: e = launch_cfg<2, 2, 2, 2>(p, grid, pro, stream); break;
    case 2: e = launch_cfg<4, 1, 2, 2>(p, grid, pro, stream); break;
    case 3: e = launch_cfg<4, 1, 2, 1>(p, grid, pro, stream); break;
    case 4: e = launch_cfg<2, 2, 1, 1>(p, grid, pro, stream); break;
    case 5: e = launch_cfg<4, 1, 1, 2>(p, grid, pro, stream); break;
    case 6: e = launch_cfg<4, 1, 1, 1>(p, grid, pro, stream); break;
    default: e = launch_cfg<2, 2, 1, 2>(p, grid, pro, stream); break;
  }
  if (e != hipSuccess) {
    set_error("conv launch failed: %s", hipGetErrorString(e));
    return MISEG_E_HIP;
  }
  return MISEG_OK;
}

}  // namespace miseg
