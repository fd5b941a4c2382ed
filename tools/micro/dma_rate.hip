// LDS-DMA (global_load_lds_dwordx4) delivery rate per CU against the shape of the source addresses.
//   hipcc -O3 --offload-arch=gfx950 dma_rate.hip -o dma_rate && ./dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// mode: bytes of each contiguous run (64, 96->64 of 96, 1024) and the stride between runs
__global__ __launch_bounds__(512) void dma_kernel(const char* src, long span, int seg_stride, int lanes_per_seg, int iters,
                                                  long* cycles) {
  __shared__ __attribute__((aligned(1024))) char lds[64 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lane -> segment (lane / lanes_per_seg), 16 bytes inside it
  const long lane_off = (long)(lane / lanes_per_seg) * seg_stride + (lane % lanes_per_seg) * 16;
  const long piece_stride = (long)(64 / lanes_per_seg) * seg_stride;
  long base = ((long)blockIdx.x * 8 + wave) * piece_stride * iters;
  const long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    long off = (base + (long)i * piece_stride) % span + lane_off;
    __builtin_amdgcn_global_load_lds(src + off, LDS_PTR(void, lds + wave * 8192 + (i & 7) * 1024), 16, 0, 0);
    if ((i & 7) == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  const long bytes = 1L << 30;
  char* src; long* cyc;
  hipMalloc(&src, bytes + (1 << 20)); hipMemset(src, 1, bytes + (1 << 20));
  hipMalloc(&cyc, 256 * sizeof(long));
  struct { const char* name; int stride, lps; long span; } modes[] = {
    {"contiguous 1 KiB pieces, 1 GiB footprint", 1024, 64, bytes},
    {"contiguous 1 KiB pieces, 32 MiB footprint (L2/MALL)", 1024, 64, 32L << 20},
    {"64-B runs at stride 96 B, 1 GiB", 96, 4, bytes},
    {"64-B runs at stride 288 B, 1 GiB", 288, 4, bytes},
    {"64-B runs at stride 64 B (contiguous, 4 lanes a run), 1 GiB", 64, 4, bytes},
    {"64-B runs at stride 288 B, 32 MiB", 288, 4, 32L << 20},
    {"128-B runs at stride 288 B, 1 GiB", 288, 8, bytes},
    {"64-B runs at stride 192 B, 32 MiB", 192, 4, 32L << 20},
    {"64-B runs at stride 384 B, 32 MiB", 384, 4, 32L << 20},
    {"128-B runs at stride 384 B, 32 MiB", 384, 8, 32L << 20},
    {"256-B runs at stride 768 B, 32 MiB", 768, 16, 32L << 20},
    {"64-B runs at stride 128 B, 32 MiB", 128, 4, 32L << 20},
    {"96-B runs at stride 192 B (6 lanes a run), 32 MiB", 192, 6, 32L << 20},
  };
  for (auto& m : modes) {
    const int iters = 2048;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(dma_kernel, dim3(256), dim3(512), 0, 0, src, m.span, m.stride, m.lps, iters, cyc);
      hipDeviceSynchronize();
    }
    long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
    printf("%-62s %7.1f cycles per piece per CU  = %5.1f B/clk/CU\n", m.name, avg / (8.0 * iters), 1024.0 * 8 * iters / avg);
  }
  return 0;
}
