// Shared device helpers for the gfx950 kernels of libunetk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

#include "../../include/unetk.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// 16-byte vector of T: 4 floats or 8 bf16.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  float v[4];
};
template <> struct Vec16<bf16> {
  static constexpr int N = 8;
  bf16 v[8];
};

template <typename T> __device__ __forceinline__ Vec16<T> load16(const T* p) {
  u32x4 r = *reinterpret_cast<const u32x4*>(p);
  return __builtin_bit_cast(Vec16<T>, r);
}
template <typename T> __device__ __forceinline__ void store16(T* p, const Vec16<T>& v) {
  *reinterpret_cast<u32x4*>(p) = __builtin_bit_cast(u32x4, v);
}
template <typename T> __device__ __forceinline__ Vec16<T> zero16() {
  u32x4 r = {0u, 0u, 0u, 0u};
  return __builtin_bit_cast(Vec16<T>, r);
}

// Sum over the 64 lanes of a wave.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{}): a loop whose
// index is a constant expression in the body (template arguments, builtin immediates).
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// "At most VM of this wave's vector-memory operations still outstanding" (and, with LGKM0, none of its LDS / scalar
// ones), then the workgroup barrier.  The wait goes through the s_waitcnt BUILTIN: the compiler's own wait insertion
// reads it and knows which LDS-DMA transfers have landed.  Written as inline asm the wait is opaque to it, and it puts
// a vmcnt(0) of its own in front of the first LDS read of a buffer it believes a DMA may still be filling - which
// drains the newer transfers that were meant to stay in flight.  (gfx9 encoding: vmcnt in bits 3:0 and 15:14,
// expcnt 6:4, lgkmcnt 11:8.)
template <int VM, bool LGKM0>
__device__ __forceinline__ void wait_vm_then_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_waitcnt((VM & 15) | (7 << 4) | ((LGKM0 ? 0 : 15) << 8) | ((VM >> 4) << 14));
  asm volatile("s_barrier" ::: "memory");
#endif
}

// Error plumbing shared by the C-ABI entry points.
void unetk_set_error(const char* fmt, ...);
#define UNETK_CHECK_ARG(cond, ...)                       \
  do {                                                   \
    if (!(cond)) {                                       \
      unetk_set_error(__VA_ARGS__);                      \
      return UNETK_EINVAL;                               \
    }                                                    \
  } while (0)
#define UNETK_CHECK_LAUNCH()                                          \
  do {                                                                \
    hipError_t e_ = hipGetLastError();                                \
    if (e_ != hipSuccess) {                                           \
      unetk_set_error("HIP launch failed: %s", hipGetErrorString(e_)); \
      return UNETK_EHIP;                                              \
    }                                                                 \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
