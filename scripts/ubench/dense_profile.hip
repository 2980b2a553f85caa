// Phase timers of the fused DenseNet layer kernel at 8 x 480 x 640 (block 1 of DenseNet-169 at
// 1920x2560, 8 images) for a given prefix width C.  Build:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMISEG_BUILD_F16=1 -Iinclude -Imedical-image-segmentation_amd/csrc \
//         -o exp_tmp/dense_profile scripts/ubench/dense_profile.hip
#define MISEG_DL_PROFILE 1
#include "dense_layer.hip"
#include <vector>
#include <string.h>
#include <stdlib.h>
namespace miseg { void set_error(const char*, ...) {} }
using namespace MISEG_KNS;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 64;
  const int N = argc > 2 ? atoi(argv[2]) : 8, H = argc > 3 ? atoi(argv[3]) : 480, W = argc > 4 ? atoi(argv[4]) : 640;
  const int CT = 256 + 1408 * (C > 224);
  const size_t ne = (size_t)N * H * W * CT;
  h16_t *cat, *w1, *w2;
  float *cf;
  CK(hipMalloc(&cat, ne * 2)); CK(hipMalloc(&w1, 128 * C * 2)); CK(hipMalloc(&w2, 32 * 9 * 128 * 2));
  CK(hipMalloc(&cf, 4096 * 4));
  std::vector<h16_t> h(1 << 22);
  unsigned s = 1;
  for (size_t i = 0; i < h.size(); ++i) { s = s * 1664525u + 1013904223u; h[i] = 0x2c00 + ((s >> 20) & 0xff); }
  for (size_t o = 0; o < ne; o += h.size()) CK(hipMemcpy(cat + o, h.data(), (ne - o < h.size() ? ne - o : h.size()) * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(w1, h.data(), 128 * C * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(w2, h.data(), 32 * 9 * 128 * 2, hipMemcpyHostToDevice));
  std::vector<float> one(4096, 1.f);
  CK(hipMemcpy(cf, one.data(), 4096 * 4, hipMemcpyHostToDevice));
  ConvParams p, q;
  memset(&p, 0, sizeof p);
  p.in = cat; p.weight = w1; p.in_scale = cf; p.in_shift = cf; p.scale = cf; p.shift = cf;
  p.N = N; p.H = H; p.W = W; p.Cin = C; p.Ho = H; p.Wo = W; p.Cout = 128;
  p.KH = p.KW = 1; p.stride_y = p.stride_x = 1;
  p.in_pix_stride = CT; p.in_row_stride = CT * W; p.in_img_stride = (long long)CT * W * H;
  p.act = MISEG_ACT_RELU; p.inv_temperature = 1.f;
  p.tiles_y = (H + 7) / 8; p.tiles_x = (W + 15) / 16; p.mtiles = N * p.tiles_y * p.tiles_x; p.ntiles = 1;
  q = p;
  q.in = nullptr; q.weight = w2; q.Cin = 128; q.Cout = 32; q.KH = q.KW = 3; q.pad_y = q.pad_x = 1;
  q.in_scale = q.in_shift = q.scale = q.shift = nullptr; q.act = MISEG_ACT_NONE;
  q.out = cat + C; q.out_pix_stride = CT; q.out_row_stride = CT * W; q.out_img_stride = (long long)CT * W * H;
  q.out_dtype = MISEG_DT_BF16; q.vec_ok = 1;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const dim3 grid(p.mtiles);
  for (int it = 0; it < 2; ++it) CK(launch_dense_layer_kernel(p, q, grid, 0));
  CK(hipEventRecord(e0, 0));
  for (int it = 0; it < 5; ++it) CK(launch_dense_layer_kernel(p, q, grid, 0));
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 2.0 * N * H * W * (128.0 * C + 9 * 128 * 32);
  printf("dense layer C=%d, %d x %d x %d: %.3f ms per launch, %.0f TFLOP/s of the layer's own arithmetic, %d patches\n", C, N, H, W,
         ms / 5, flop / (ms / 5) * 1e-9, p.mtiles);
  static unsigned long long prof[1 << 16][8];
  CK(hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_dl_prof), sizeof prof));
  const char* names[] = {"coefficients + first requests", "1x1 chunk loop", "barrier + tile write", "3x3 tap loop", "exchange + stores"};
  double t[8] = {0};
  const int nb = p.mtiles < (1 << 14) ? p.mtiles : (1 << 14);
  for (int b = 0; b < nb * 4; ++b) for (int i = 0; i < 8; ++i) t[i] += prof[b][i];
  printf("us per patch and wave:");
  for (int i = 0; i < 5; ++i) printf("  %s %.2f", names[i], t[i] / (nb * 4) * 0.01);
  printf("\n");
  return 0;
}
