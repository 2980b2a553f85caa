// Store-rate microbenchmark: the output of a 2x-upsampling transpose convolution
// (4 x 1920 x 2560 x 32 bf16, 1.26 GB) written by blocks of 4 waves, one block per 8 x 32 input
// patch, with the lane -> address patterns an epilogue can produce:
//   0  wave = phase; a lane pair writes 32 contiguous bytes of a pixel, pixels 128 B apart
//   1  wave = phase; four lanes write the 64 bytes of a pixel, pixels 128 B apart
//   2  wave = (row phase, row half), both column phases; 32-byte pieces, four instructions fill a line
//   3  wave = (row phase, row half); every instruction writes 1 KB contiguous
// Build: hipcc -O3 --offload-arch=gfx950 -o exp_tmp/store_patterns scripts/ubench/store_patterns.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 4, H = 960, W = 1280, C = 32;
constexpr long long ROW = 2ll * W * C * 2;           // bytes per output row
constexpr long long IMG = 2ll * H * ROW;

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(unsigned char* out, int tiles_x, int per_img, int ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint4 val = make_uint4(lane, wave, blockIdx.x, 7);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n = tile / per_img, trem = tile - n * per_img;
    const int y0 = (trem / tiles_x) * 8, x0 = (trem % tiles_x) * 32;
    unsigned char* img = out + n * IMG;
    if (MODE == 0 || MODE == 1) {
      const int pa = wave >> 1, pb = wave & 1;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        unsigned char* row = img + (2 * (y0 + b) + pa) * ROW;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          long long o;
          if (MODE == 0) o = (2 * (x0 + (lane & 31)) + pb) * 64 + q * 32 + (lane >> 5) * 16;
          else o = (2 * (x0 + q * 16 + (lane >> 2)) + pb) * 64 + (lane & 3) * 16;
          *reinterpret_cast<uint4*>(row + o) = val;
        }
      }
    } else {
      const int pa = wave >> 1, half = wave & 1;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        unsigned char* row = img + (2 * (y0 + half * 4 + b) + pa) * ROW + (long long)x0 * 128;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          long long o;
          if (MODE == 2) o = (lane & 31) * 128 + k * 32 + (lane >> 5) * 16;
          else o = k * 1024 + lane * 16;
          *reinterpret_cast<uint4*>(row + o) = val;
        }
      }
    }
  }
}

template <int MODE>
int run(unsigned char* out, int blocks, const char* what) {
  const int tiles_x = W / 32, per_img = (H / 8) * tiles_x, ntiles = N * per_img;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, tiles_x, per_img, ntiles);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, tiles_x, per_img, ntiles);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("mode %d blocks %5d: %.3f ms, %.2f TB/s  (%s)\n", MODE, blocks, ms / 10, N * IMG / (ms / 10) * 1e-9, what);
  return 0;
}
int main() {
  unsigned char* out;
  CK(hipMalloc(&out, N * IMG));
  for (int blocks : {256, 512, 1024, 19200}) {
    if (run<0>(out, blocks, "32 B pieces, wave = phase")) return 1;
    if (run<1>(out, blocks, "64 B pieces, wave = phase")) return 1;
    if (run<2>(out, blocks, "32 B pieces, 4 instr fill lines")) return 1;
    if (run<3>(out, blocks, "dense 1 KB")) return 1;
  }
  return 0;
}
