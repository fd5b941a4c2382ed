// Sustained MFMA rate of the two bf16 shapes with register-resident operands (no LDS, no memory):
// what the matrix pipe delivers once the chip's power limit has set the clock.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_power.hip -o tools/micro/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(256) void burn(const unsigned* seed, float* out, int iters) {
  unsigned s = seed[threadIdx.x & 63] * 2654435761u + threadIdx.x;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 ra[4], rb[6];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { s = s * 1664525u + 1013904223u; ra[i][j] = (s & 0x807f807fu) | 0x3f003f00u; }
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 4; ++j) { s = s * 1664525u + 1013904223u; rb[i][j] = (s & 0x807f807fu) | 0x3f003f00u; }
  float sum = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[4][6];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ra[i]), __builtin_bit_cast(bf16x8, rb[j]), acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) sum += acc[i][j][0] + acc[i][j][3];
  } else {
    f32x16 acc[2][3];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[2 * k + i]), __builtin_bit_cast(bf16x8, rb[3 * k + j]), acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) sum += acc[i][j][0] + acc[i][j][15];
  }
  if (sum == 12345.678f) out[threadIdx.x] = sum;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  unsigned* seed; float* out;
  hipMalloc(&seed, 256); hipMalloc(&out, 1024);
  unsigned h[64]; for (int i = 0; i < 64; ++i) h[i] = 12345u + 977u * i;
  hipMemcpy(seed, h, 256, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int shape : {16, 32, 16, 32}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (shape == 16) hipLaunchKernelGGL(burn<16>, dim3(512), dim3(256), 0, 0, seed, out, iters);
      else hipLaunchKernelGGL(burn<32>, dim3(512), dim3(256), 0, 0, seed, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // per iteration and wave: 24 MFMAs of 16x16x32 (16384 flops) or 12 of 32x32x16 (32768 flops): the same flops
      const double flops = 512.0 * 4 * iters * 24 * 16384.0;
      if (rep) printf("shape %dx%d: %.2f ms, %.0f TFLOP/s\n", shape, shape, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
