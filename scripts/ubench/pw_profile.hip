// Phase timers of the wave-private-ring 3x3 kernel on encoder/decoder layer shapes of the
// ResNet-34 U-Net at 1920x2560, batch 4 (bf16 output, scale/shift/ReLU epilogue).  Build:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Imedical-image-segmentation_amd/csrc \
//         -o exp_tmp/pw_profile scripts/ubench/pw_profile.hip
#define MISEG_PW_PROFILE 1
#include "conv_halo_pw.hip"
#include <vector>
#include <string.h>
namespace miseg { void set_error(const char*, ...) {} }
using namespace miseg;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static int run(int N, int H, int W, int Cin, int Cout) {
  const size_t in_e = (size_t)N * H * W * Cin, out_e = (size_t)N * H * W * Cout, w_e = (size_t)Cout * 9 * Cin;
  bf16_t *in, *wt, *out;
  float *sc, *sh;
  CK(hipMalloc(&in, in_e * 2)); CK(hipMalloc(&wt, w_e * 2)); CK(hipMalloc(&out, out_e * 2));
  CK(hipMalloc(&sc, Cout * 4)); CK(hipMalloc(&sh, Cout * 4));
  std::vector<bf16_t> h(in_e > w_e ? in_e : w_e);
  unsigned s = 1;
  for (size_t i = 0; i < h.size(); ++i) { s = s * 1664525u + 1013904223u; h[i] = 0x3c00 + ((s >> 20) & 0xff); }
  CK(hipMemcpy(in, h.data(), in_e * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(wt, h.data(), w_e * 2, hipMemcpyHostToDevice));
  std::vector<float> ones(Cout, 1.f);
  CK(hipMemcpy(sc, ones.data(), Cout * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sh, ones.data(), Cout * 4, hipMemcpyHostToDevice));
  ConvParams p;
  memset(&p, 0, sizeof p);
  p.in = in; p.weight = wt; p.out = out; p.scale = sc; p.shift = sh;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = H; p.Wo = W; p.Cout = Cout;
  p.KH = p.KW = 3; p.stride_y = p.stride_x = 1; p.pad_y = p.pad_x = 1;
  p.in_pix_stride = Cin; p.in_row_stride = Cin * W; p.in_img_stride = (long long)Cin * W * H;
  p.out_pix_stride = Cout; p.out_row_stride = Cout * W; p.out_img_stride = (long long)Cout * W * H;
  p.out_dtype = MISEG_DT_BF16; p.act = MISEG_ACT_RELU; p.inv_temperature = 1.f;
  p.M = N * H * W; p.vec_ok = 1;
  p.tiles_y = (H + 7) / 8; p.tiles_x = (W + 31) / 32; p.mtiles = N * p.tiles_y * p.tiles_x; p.ntiles = (Cout + 127) / 128;
  const int blocks = p.mtiles * p.ntiles;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 3; ++it) CK(launch_halo_pw(p, dim3(blocks), 0));
  CK(hipEventRecord(e0, 0));
  for (int it = 0; it < 10; ++it) CK(launch_halo_pw(p, dim3(blocks), 0));
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 2.0 * N * H * W * 9.0 * Cin * Cout;
  printf("%dx%dx%d Cin %d Cout %d: %.3f ms per launch, %.0f TFLOP/s, %d blocks\n", N, H, W, Cin, Cout, ms / 10,
         flop / (ms / 10) * 1e-9, blocks);
  static unsigned long long prof[1 << 16][4];
  CK(hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_pw_prof), sizeof prof));
  const int nb = blocks < (1 << 14) ? blocks : (1 << 14);
  double t[4] = {0};
  for (int b = 0; b < nb * 4; ++b) for (int i = 0; i < 4; ++i) t[i] += prof[b][i];
  const char* names[] = {"prologue", "arithmetic", "halo swaps", "epilogue"};
  printf("   us per block (wave average):");
  for (int i = 0; i < 4; ++i) printf("  %s %.2f", names[i], t[i] / (nb * 4) * 0.01);
  printf("   (arithmetic at 1.5 GHz would be %.2f)\n", (Cin / 64) * 9 * 32 * 32 / 1.5e3);
  hipFree(in); hipFree(wt); hipFree(out); hipFree(sc); hipFree(sh);
  return 0;
}
int main() {
  if (run(4, 240, 320, 128, 128)) return 1;    // stage 2
  if (run(4, 120, 160, 256, 256)) return 1;    // stage 3
  if (run(4, 60, 80, 512, 512)) return 1;      // stage 4
  if (run(4, 240, 320, 256, 128)) return 1;    // decoder 2b
  if (run(4, 120, 160, 512, 256)) return 1;    // decoder 1b
  return 0;
}
