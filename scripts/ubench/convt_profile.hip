// Phase timers of the persistent 64 -> 32 transpose-convolution kernel on the headline layer
// (4 x 960 x 1280 x 64 -> 4 x 1920 x 2560 x 32).  Build:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Imedical-image-segmentation_amd/csrc \
//         -o exp_tmp/convt_profile scripts/ubench/convt_profile.hip
#define MISEG_CT_PROFILE 1
#include "conv_halo_pw.hip"
#include <vector>
#include <string.h>
namespace miseg { void set_error(const char*, ...) {} }
using namespace miseg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int N = 4, H = 960, W = 1280, Cin = 64, Cout = 32;
  const size_t in_e = (size_t)N * H * W * Cin, out_e = (size_t)N * 4 * H * W * Cout, w_e = 4 * 32 * 4 * 64;
  bf16_t *in, *wt, *out;
  CK(hipMalloc(&in, in_e * 2)); CK(hipMalloc(&wt, w_e * 2)); CK(hipMalloc(&out, out_e * 2));
  std::vector<bf16_t> h(in_e);
  unsigned s = 1;
  for (size_t i = 0; i < in_e; ++i) { s = s * 1664525u + 1013904223u; h[i] = 0x3c00 + ((s >> 20) & 0xff); }
  CK(hipMemcpy(in, h.data(), in_e * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(wt, h.data(), w_e * 2, hipMemcpyHostToDevice));
  ConvParams p;
  memset(&p, 0, sizeof p);
  p.in = in; p.weight = wt; p.out = out;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = H; p.Wo = W; p.Cout = Cout;
  p.KH = p.KW = 2; p.stride_y = p.stride_x = 1;
  p.in_pix_stride = Cin; p.in_row_stride = Cin * W; p.in_img_stride = (long long)Cin * W * H;
  p.out_pix_stride = Cout; p.out_row_stride = Cout * 2 * W; p.out_img_stride = (long long)Cout * 4 * W * H;
  p.out_dtype = MISEG_DT_BF16; p.act = MISEG_ACT_RELU; p.transposed = 1; p.inv_temperature = 1.f;
  p.M = N * H * W; p.vec_ok = 1;
  p.tiles_y = (H + 7) / 8; p.tiles_x = (W + 31) / 32; p.mtiles = N * p.tiles_y * p.tiles_x; p.ntiles = 1;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 4; ++mode) {
  p.collapse_value = mode;   // profiling switches: 1 = no stores, 2 = no halo requests after the first
  printf("mode %d (%s%s)\n", mode, mode & 1 ? "no stores " : "", mode & 2 ? "no halo requests" : "");
  for (int it = 0; it < 3; ++it) CK(launch_halo_pw(p, dim3(p.mtiles), 0));
  CK(hipEventRecord(e0, 0));
  for (int it = 0; it < 10; ++it) CK(launch_halo_pw(p, dim3(p.mtiles), 0));
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("convT 64->32 %.3f ms per launch, %.2f TB/s\n", ms / 10, (in_e + out_e) * 2 / (ms / 10) * 1e-9);
  static unsigned long long prof[1024][8];
  CK(hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_ct_prof), sizeof prof));
  const char* names[] = {"top barrier", "halo issue", "arithmetic", "vmcnt wait", "epilogue", "-"};
  for (int w = 0; w < 4; ++w) {
    double t[8] = {0};
    for (int b = 0; b < 256; ++b) for (int i = 0; i < 8; ++i) t[i] += prof[b * 4 + w][i];
    printf("wave %d: patches/block %.1f; us per patch:", w, t[6] / 256);
    for (int i = 0; i < 6; ++i) printf("  %s %.2f", names[i], t[i] / t[6] * 0.01);
    printf("\n");
  }
  }
  return 0;
}
