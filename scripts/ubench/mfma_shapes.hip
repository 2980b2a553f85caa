// Microbenchmark: LDS-fed bf16 MFMA loops, 32x32x16 against 16x16x32, same FLOPs and the same LDS
// read traffic per iteration (9 ds_read_b128 per 262144 FLOP per wave), random operands, two
// 4-wave blocks per CU.  Build: hipcc -O3 --offload-arch=gfx950 mfma_shapes.hip -o mfma_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void loop_kernel(const unsigned short* src, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
  for (int i = threadIdx.x; i < 64 * 1024 / 16; i += 256)
    reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(src)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned char* base = lds + wave * 8192 + lane * 16;
  f32x16 acc32[8];
  f32x4 acc16[32];
  for (int a = 0; a < 8; ++a)
    for (int r = 0; r < 16; ++r) acc32[a][r] = 0.f;
  for (int a = 0; a < 32; ++a)
    for (int r = 0; r < 4; ++r) acc16[a][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    const unsigned char* p = base + (it & 7) * 1024;
    bf16x8 w = *reinterpret_cast<const bf16x8*>(p);
    bf16x8 x[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) x[b] = *reinterpret_cast<const bf16x8*>(p + 1024 * (b + 1) + (b & 1) * 32768 % 49152);
    if (SHAPE == 32) {
#pragma unroll
      for (int b = 0; b < 8; ++b) acc32[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x[b], acc32[b], 0, 0, 0);
    } else {
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        acc16[2 * b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x[b], acc16[2 * b], 0, 0, 0);
        acc16[2 * b + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[b], w, acc16[2 * b + 1], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int a = 0; a < 8; ++a)
    for (int r = 0; r < 16; ++r) s += acc32[a][r];
  for (int a = 0; a < 32; ++a)
    for (int r = 0; r < 4; ++r) s += acc16[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int blocks = 512, iters = 40000;
  std::vector<unsigned short> h(32 * 1024);
  srand(1);
  for (auto& v : h) {   // random bf16 in about [-1, 1]
    float f = (float)rand() / RAND_MAX * 2.f - 1.f;
    unsigned u;
    memcpy(&u, &f, 4);
    v = (unsigned short)(u >> 16);
  }
  unsigned short* d;
  float* o;
  hipMalloc(&d, h.size() * 2);
  hipMalloc(&o, blocks * 256 * 4);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep)
    for (int shape : {32, 16}) {
      hipEventRecord(e0);
      if (shape == 32)
        hipLaunchKernelGGL(loop_kernel<32>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
      else
        hipLaunchKernelGGL(loop_kernel<16>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)blocks * 4 * iters * 8 * 32768.0;
      printf("shape %2d: %.3f ms  %.1f TFLOP/s\n", shape, ms, flop / ms / 1e9);
    }
  return 0;
}
