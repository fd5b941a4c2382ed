// Epilogue arithmetic and the LDS-DMA wrapper shared by the token GEMM kernels (gfx950).
#pragma once
#include "common.h"

namespace {

// LDS-DMA: 16 bytes per lane from `src` to (wave-uniform) `lds_dst` + lane*16.  The builtin exists only in the
// device pass; the host pass just needs the kernels' signatures.
__device__ __forceinline__ void glds16e(const void* src, char* lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(src, LDS_PTR(void, lds_dst), 16, 0, 0);
#endif
}

// GELU (erf form) and its derivative for the bf16 epilogues.  Phi(v) = 0.5 * (1 + erf(v / sqrt 2)) through
// Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the 2^-9 rounding of the bf16 result): one
// reciprocal, one exponential and five multiply-adds, against the several dozen instructions and
// branches of the library erff.  The exponential exp(-v^2 / 2) is also the Gaussian of the derivative.
struct PhiExp {
  float Phi, e;     // Phi(v), exp(-v*v/2)
};
__device__ __forceinline__ PhiExp phi_exp(float v) {
  const float z = fabsf(v) * 0.70710678118654752440f;
  const float t = __frcp_rn(1.0f + 0.3275911f * z);
  const float e = __expf(-z * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;
  PhiExp r;
  r.Phi = 0.5f * (1.0f + (v < 0.f ? -erf_abs : erf_abs));
  r.e = e;
  return r;
}
__device__ __forceinline__ float gelu_erf_g(float v) { return v * phi_exp(v).Phi; }
// d/dv [ v * Phi(v) ] = Phi(v) + v * phi(v)
__device__ __forceinline__ float gelu_grad_g(float v) {
  const PhiExp p = phi_exp(v);
  return p.Phi + v * 0.3989422804014327f * p.e;
}

}  // namespace
