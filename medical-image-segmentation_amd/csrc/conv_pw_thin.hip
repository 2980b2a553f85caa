// 1x1 convolutions with a short contraction (Cin = 64, 128, 192 or 256) and 16-bit output (gfx950): the
// LinkNet decoder's C -> C/4 and C/4 -> C layers with their skip additions (segmentation_models
// DecoderBlock), FPN's lateral layers with the up-sampled top-down addend (FPNBlock; both reached from
// /root/reference utils/model.py:188-211) and, in training, the data gradients of DenseNet's C -> 128 layers
// (a 128 -> C layer that accumulates onto the gradient the concatenation buffer already holds).
// Such a layer writes (and, with a residual, reads) Cout 16-bit values per pixel for Cin multiply-adds per
// output value: it is bound by HBM traffic, and the tiled kernels, which spend a block on one 128 x 128
// output tile with a single K step, run it at 1.5-2 TB/s.  Here the blocks are persistent:
//   * a block owns one tile of 128 couts; its weights [128][Cin] go to LDS once, epilogue coefficients too;
//   * a wave walks over 32-pixel tiles.  In the matrix instruction's B operand a lane holds 8 consecutive
//     channels of one pixel, which is 16 contiguous bytes of the NHWC tensor: the pixel fragments are
//     loaded from memory straight into registers (no LDS staging), the next tile's while the current
//     tile's matrix instructions and stores run;
//   * the epilogue goes from the accumulators to memory (scale/shift, ReLU, residual before or after the
//     activation, optionally read at (y >> s, x >> s), second output), all residual loads of a tile
//     requested before the first is used.
// The blocks of the cout tiles that share pixels follow each other on one XCD, so the operand comes from
// HBM once.  LDS: 128 x (2 Cin + 16) bytes of weights + 2 KB of coefficients; two blocks per CU (the epilogue's
// registers: three would spill a hundred of them).
#include "conv_common.h"

namespace MISEG_KNS {

namespace {
constexpr int PT_COUTS = 128;
constexpr int PT_MAXK = 256;
constexpr int PT_W_BYTES = PT_COUTS * (PT_MAXK * 2 + 16);
constexpr int PT_LDS = PT_W_BYTES + 4 * PT_COUTS * 4;
static_assert(2 * PT_LDS <= 160 * 1024, "two blocks per CU");
}  // namespace

// KS = Cin / 16 matrix-instruction K steps.  gridDim.x = 8 * workers per XCD * p.ntiles.
template <int KS, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_pw_thin_kernel(const ConvParams p) {
  constexpr int K = KS * 16;
  constexpr int PITCH = K * 2 + 16;          // odd number of 16-byte pieces: rows spread over the banks
  constexpr bool PREFETCH = KS <= 8 && OCC <= 2;   // the next tile's fragments fit beside the current ones
  constexpr int W_BYTES = PT_COUTS * PITCH;
  __shared__ __attribute__((aligned(16))) unsigned char lds[W_BYTES + 4 * PT_COUTS * 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 31, lhalf = lane >> 5;

  // XCD x = blockIdx % 8 runs the workers x, x + 8, ...; the cout tiles of a worker are neighbours there
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int ct = loc % p.ntiles;
  const int wk = (loc / p.ntiles) * 8 + xcd;
  const int nworkers = (gridDim.x >> 3) / p.ntiles * 8;
  const int cout0 = ct * PT_COUTS;

  // ---- weights and coefficients of this cout tile into LDS ----
  for (int i = tid; i < PT_COUTS * (K / 8); i += 256) {
    const int row = i / (K / 8), pc = i - row * (K / 8);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (cout0 + row < p.Cout) v = *reinterpret_cast<const uint4*>(p.weight + (long long)(cout0 + row) * K + pc * 8);
    *reinterpret_cast<uint4*>(lds + row * PITCH + pc * 16) = v;
  }
  float* cf = reinterpret_cast<float*>(lds + W_BYTES);     // scale | shift | scale2 | shift2, 128 each
  for (int i = tid; i < 4 * PT_COUTS; i += 256) {
    const int which = i >> 7, c = cout0 + (i & 127);
    const float* src = which == 0 ? p.scale : which == 1 ? p.shift : which == 2 ? p.scale2 : p.shift2;
    cf[i] = (src && c < p.Cout) ? src[c] : ((which & 1) ? 0.f : 1.f);
  }
  __syncthreads();

  const int HW = p.H * p.W;
  const int ntile = (p.M + 31) >> 5;
  const int step = nworkers * 4;
  const bool has_res = p.residual != nullptr;
  const bool relu = p.act == MISEG_ACT_RELU;

  auto load_x = [&](int t, h16x8 (&xf)[KS]) {
    const int m = t * 32 + lrow;
    if (m < p.M) {
      const int n = m / HW, rem = m - n * HW;
      const int y = rem / p.W, x = rem - y * p.W;
      const h16_t* src = p.in + n * p.in_img_stride + (long long)y * p.in_row_stride + (long long)x * p.in_pix_stride +
                         lhalf * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const h16x8*>(src + ks * 16);
    } else {
      const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xf[ks] = z;
    }
  };

  const unsigned char* wrow = lds + lrow * PITCH + lhalf * 16;
  int t = wk * 4 + wave;
  h16x8 xf[KS];
  if (t < ntile) load_x(t, xf);
  while (t < ntile) {
    const int tn = t + step;
    h16x8 xn[PREFETCH ? KS : 1];
    if constexpr (PREFETCH) {
      if (tn < ntile) load_x(tn, xn);
    }
    f32x16 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    // with a long contraction the weight fragments must be read again for every tile: hoisted out of the
    // tile loop they would need 4 * KS * 4 registers (the compiler did hoist them, and spilled)
    unsigned wofs = 0;
    if constexpr (KS > 4) asm volatile("" : "+v"(wofs));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const h16x8 wf = *reinterpret_cast<const h16x8*>(wrow + wofs + g * 32 * PITCH + ks * 32);
        acc[g] = mfma_32x32x16(wf, xf[ks], acc[g]);
      }

    // ---- epilogue of the 32 pixels x 128 couts ----
    const int m = t * 32 + lrow;
    const bool okp = m < p.M;
    int n = 0, y = 0, x = 0;
    if (okp) {
      n = m / HW;
      const int rem = m - n * HW;
      y = rem / p.W;
      x = rem - y * p.W;
    }
    const long long o_off = n * p.out_img_stride + (long long)y * p.out_row_stride + (long long)x * p.out_pix_stride;
    const long long o2_off = n * p.out2_img_stride + (long long)y * p.out2_row_stride + (long long)x * p.out2_pix_stride;
    // (the same for the epilogue coefficients: 4 x 64 floats per lane if they were hoisted)
    unsigned cofs = 0;
    asm volatile("" : "+v"(cofs));
    const float* cft = cf + cofs;
    uint4 res[4][2];
    if (has_res) {
      const h16_t* rsrc = reinterpret_cast<const h16_t*>(p.residual) + n * p.res_img_stride +
                          (long long)(y >> p.res_shift) * p.res_row_stride +
                          (long long)(x >> p.res_shift) * p.res_pix_stride;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int c = cout0 + g * 32 + 16 * q + 8 * lhalf;
          res[g][q] = okp && c + 8 <= p.Cout ? *reinterpret_cast<const uint4*>(rsrc + c) : make_uint4(0, 0, 0, 0);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (cout0 + g * 32 >= p.Cout) continue;
      DirectLane dl;
      DirectOut2 o2;
      dl.relu = relu;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int cl = g * 32 + 16 * q + 8 * lhalf;
        dl.c2[q] = cout0 + cl;
        dl.okc[q] = dl.c2[q] + 8 <= p.Cout;
#pragma unroll
        for (int j = 0; j < 8; j += 4) {
          const float4 s = *reinterpret_cast<const float4*>(cft + cl + j);
          const float4 h = *reinterpret_cast<const float4*>(cft + PT_COUTS + cl + j);
          dl.sc[8 * q + j] = s.x; dl.sc[8 * q + j + 1] = s.y; dl.sc[8 * q + j + 2] = s.z; dl.sc[8 * q + j + 3] = s.w;
          dl.sh[8 * q + j] = h.x; dl.sh[8 * q + j + 1] = h.y; dl.sh[8 * q + j + 2] = h.z; dl.sh[8 * q + j + 3] = h.w;
          if (p.out2) {
            const float4 s2 = *reinterpret_cast<const float4*>(cft + 2 * PT_COUTS + cl + j);
            const float4 h2 = *reinterpret_cast<const float4*>(cft + 3 * PT_COUTS + cl + j);
            o2.s2[8 * q + j] = s2.x; o2.s2[8 * q + j + 1] = s2.y; o2.s2[8 * q + j + 2] = s2.z; o2.s2[8 * q + j + 3] = s2.w;
            o2.t2[8 * q + j] = h2.x; o2.t2[8 * q + j + 1] = h2.y; o2.t2[8 * q + j + 2] = h2.z; o2.t2[8 * q + j + 3] = h2.w;
          }
        }
      }
      // two call sites: a pointer chosen at run time would pin `o2` to scratch memory
      if (p.out2)
        direct_tile(p, dl, acc[g], okp, o_off, o2_off, has_res, res[g], &o2, nullptr, false);
      else
        direct_tile(p, dl, acc[g], okp, o_off, o2_off, has_res, res[g], nullptr, nullptr, false);
    }

    if constexpr (PREFETCH) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xf[ks] = xn[ks];
    } else {
      if (tn < ntile) load_x(tn, xf);
    }
    t = tn;
  }
}

bool conv_pw_thin_ok(const ConvParams& p) {
  return !p.in_scale && p.KH == 1 && p.KW == 1 && p.stride_y == 1 && p.stride_x == 1 && p.pad_y == 0 && p.pad_x == 0 &&
         !p.transposed && !p.dgrad_t && (p.Cin == 64 || p.Cin == 128 || p.Cin == 192 || p.Cin == 256) &&
         p.in_valid_c == p.Cin && p.vec_ok && p.out && p.out_dtype == MISEG_DT_BF16 && (p.Cout & 7) == 0 &&
         (!p.residual || p.res_dtype == MISEG_DT_BF16) && p.act != MISEG_ACT_SOFTMAX && p.act != MISEG_ACT_SIGMOID &&
         !p.stats && p.Ho == p.H && p.Wo == p.W &&
         ((uintptr_t)p.in & 15) == 0 && (p.in_pix_stride & 7) == 0 && (p.in_row_stride & 7) == 0 &&
         (p.in_img_stride & 7) == 0 && ((uintptr_t)p.weight & 15) == 0;
}

template <int KS, int OCC>
static void launch_thin(const ConvParams& p, int wpx, hipStream_t s) {
  hipLaunchKernelGGL((conv_pw_thin_kernel<KS, OCC>), dim3(8u * wpx * p.ntiles), dim3(256), 0, s, p);
}

hipError_t launch_conv_pw_thin(ConvParams p, hipStream_t s) {
  p.ntiles = (p.Cout + PT_COUTS - 1) / PT_COUTS;
  const int ntile = (p.M + 31) >> 5;
  int wpx = 64 / p.ntiles;                       // workers per XCD: about two blocks per CU in all
  if (wpx < 1) wpx = 1;
  if (wpx * 32 > ntile) wpx = ntile / 32 > 0 ? ntile / 32 : 1;   // every wave gets a tile
  switch (p.Cin) {
    case 64: launch_thin<4, 2>(p, wpx, s); break;
    case 128: launch_thin<8, 2>(p, wpx, s); break;
    case 192: launch_thin<12, 2>(p, wpx, s); break;
    default: launch_thin<16, 2>(p, wpx, s); break;
  }
  return hipGetLastError();
}

}  // namespace MISEG_KNS
