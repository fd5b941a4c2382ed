// MFMA fragment helpers shared by the GEMM-shaped kernels (gfx950).
// An operand tile is staged in LDS as rows of K-contiguous elements in 64-byte chunks
// (32 bf16 or 16 f32); load() returns the lane's fragment of one chunk for the 16-row tile whose
// row pointer it is given: A[row = lane&15][k...] and B[k...][col = lane&15] use the same form.
#pragma once
#include "common.h"

namespace {

template <typename T> struct MMA;
template <> struct MMA<bf16> {
  static constexpr int KC = 32;      // channels per 64-byte chunk row
  static constexpr int KSTEPS = 1;   // MFMA k-steps per chunk
  using frag = bf16x8;
  static __device__ __forceinline__ frag load(const char* row, int ks, int lane) {
    return *reinterpret_cast<const frag*>(row + (lane >> 4) * 16);
  }
  // sw = swz(row index): the 16-byte slot of a 64-byte row is XORed with it
  static __device__ __forceinline__ frag load_swz(const char* row, int ks, int lane, int sw) {
    return *reinterpret_cast<const frag*>(row + (((lane >> 4) ^ sw) * 16));
  }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct MMA<float> {
  static constexpr int KC = 16;
  static constexpr int KSTEPS = 4;
  using frag = float;
  static __device__ __forceinline__ frag load(const char* row, int ks, int lane) {
    return *reinterpret_cast<const float*>(row + (ks * 4 + (lane >> 4)) * 4);
  }
  static __device__ __forceinline__ frag load_swz(const char* row, int ks, int lane, int sw) {
    return *reinterpret_cast<const float*>(row + ((ks ^ sw) * 4 + (lane >> 4)) * 4);
  }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

// Swizzle of unpadded 64-byte LDS rows: slot ^= 2 on every other group of four rows.  With it a
// ds_read_b128 fragment read of 16 consecutive rows (any starting row) touches 16 distinct 16-byte
// slots of the 256-byte bank row in each of the instruction's four lane groups: conflict-free.
__device__ __forceinline__ int swz(int row) { return ((row >> 2) & 1) << 1; }

}  // namespace
