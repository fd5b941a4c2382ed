// Cost per wave-instruction (cycles of the CU, eight waves issuing) of 16-byte-per-lane vector memory
// operations against the shape of the lane addresses: LDS-DMA loads, register loads and stores.
//   hipcc -O3 --offload-arch=gfx950 addr_rate.hip -o addr_rate && ./addr_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// op 0: global_load_lds 16 B; 1: global_load_dwordx4 to registers; 2: global_store_dwordx4
template <int OP>
__global__ __launch_bounds__(512) void rate_kernel(char* src, long span, int seg_stride, int lanes_per_seg, int misalign, int perm,
                                                   int iters, long* cycles, u32x4* sink) {
  __shared__ __attribute__((aligned(1024))) char lds[64 * 1024];
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // perm < 64: XOR mask; perm >= 64: rotate rows of 12 lanes (a 192-byte pixel) by interleaving 0,8,1,9.. over 16 pixels is not expressible: use bit swaps
  const int lane = perm < 64 ? (lane0 ^ perm) : (perm == 64 ? ((lane0 & 0x30) | ((lane0 & 7) << 1) | ((lane0 >> 3) & 1)) : ((lane0 * 5) & 63));
  const long lane_off = (long)(lane / lanes_per_seg) * seg_stride + (lane % lanes_per_seg) * 16 + misalign;
  const long piece_stride = (long)(64 / lanes_per_seg) * seg_stride;
  long base = ((long)blockIdx.x * 8 + wave) * piece_stride * iters;
  u32x4 acc = {0, 0, 0, 0};
  const long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    long off = (base + (long)i * piece_stride) % span + lane_off;
    if (OP == 0) {
      __builtin_amdgcn_global_load_lds(src + off, LDS_PTR(void, lds + wave * 8192 + (i & 7) * 1024), 16, 0, 0);
      if ((i & 7) == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else if (OP == 1) {
      u32x4 v = *reinterpret_cast<const u32x4*>(src + off);
      acc += v;
    } else {
      *reinterpret_cast<u32x4*>(src + off) = u32x4{(unsigned)i, 1u, 2u, 3u};
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if (OP == 1 && acc.x == 0x12345678u) sink[threadIdx.x] = acc;
}

int main() {
  const long bytes = 1L << 30;
  char* src; long* cyc; u32x4* sink;
  hipMalloc(&src, bytes + (4 << 20)); hipMemset(src, 1, bytes + (4 << 20));
  hipMalloc(&cyc, 256 * sizeof(long)); hipMalloc(&sink, 512 * 16);
  struct { const char* name; int stride, lps, mis; long span; int perm; } modes[] = {
    {"1 KiB consecutive, aligned, 32 MiB", 1024, 64, 0, 32L << 20},
    {"1 KiB consecutive, lanes ^1", 1024, 64, 0, 32L << 20, 1},
    {"1 KiB consecutive, lanes ^2", 1024, 64, 0, 32L << 20, 2},
    {"1 KiB consecutive, lanes ^4", 1024, 64, 0, 32L << 20, 4},
    {"1 KiB consecutive, lanes ^8", 1024, 64, 0, 32L << 20, 8},
    {"1 KiB consecutive, lanes ^16", 1024, 64, 0, 32L << 20, 16},
    {"1 KiB consecutive, lanes ^32", 1024, 64, 0, 32L << 20, 32},
    {"1 KiB consecutive, lanes ^63 (reversed)", 1024, 64, 0, 32L << 20, 63},
    {"1 KiB consecutive, 16-lane interleave 0,8,1,9..", 1024, 64, 0, 32L << 20, 64},
    {"1 KiB consecutive, lane*5 mod 64", 1024, 64, 0, 32L << 20, 65},
    {"2 x 512-B runs adjacent rows of 2 KiB pitch", 2048, 32, 0, 32L << 20, 0},
    {"1 KiB consecutive, aligned, 1 GiB, lanes ^2", 1024, 64, 0, 1L << 30, 2},
    {"1 KiB consecutive, +64 B, 32 MiB", 1024, 64, 64, 32L << 20},
    {"1 KiB consecutive, +192 B, 32 MiB", 1024, 64, 192, 32L << 20},
    {"1 KiB consecutive, +16 B, 32 MiB", 1024, 64, 16, 32L << 20},
    {"512-B runs at stride 768 B, 32 MiB", 768, 32, 0, 32L << 20},
    {"512-B runs at stride 1536 B, 32 MiB", 1536, 32, 0, 32L << 20},
    {"256-B runs at stride 768 B, 32 MiB", 768, 16, 0, 32L << 20},
    {"128-B runs at stride 384 B, 32 MiB", 384, 8, 0, 32L << 20},
    {"64-B runs at stride 192 B, 32 MiB", 192, 4, 0, 32L << 20},
    {"64-B runs at stride 64 B (consecutive), 32 MiB", 64, 4, 0, 32L << 20},
    {"192-B runs at stride 192 B (consecutive, 12 lanes), 32 MiB", 192, 12, 0, 32L << 20},
    {"1 KiB consecutive, aligned, 1 GiB", 1024, 64, 0, bytes},
    {"64-B runs at stride 192 B, 1 GiB", 192, 4, 0, bytes},
  };
  const char* ops[3] = {"LDS-DMA load", "register load", "store"};
  for (int op = 0; op < 3; ++op) {
    printf("== %s\n", ops[op]);
    for (auto& m : modes) {
      const int iters = 2048;
      for (int rep = 0; rep < 2; ++rep) {
        if (op == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(256), dim3(512), 0, 0, src, m.span, m.stride, m.lps, m.mis, m.perm, iters, cyc, sink);
        if (op == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(256), dim3(512), 0, 0, src, m.span, m.stride, m.lps, m.mis, m.perm, iters, cyc, sink);
        if (op == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(256), dim3(512), 0, 0, src, m.span, m.stride, m.lps, m.mis, m.perm, iters, cyc, sink);
        hipDeviceSynchronize();
      }
      long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
      printf("%-62s %7.1f cycles per instr per CU  = %5.1f B/clk/CU\n", m.name, avg / (8.0 * iters), 1024.0 * 8 * iters / avg);
    }
  }
  return 0;
}
