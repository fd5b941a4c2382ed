// Optimizer tail of the training step on the GPU (gfx950): global gradient norm + clip coefficient
// and AdamW over a list of tensors in ONE launch each (reference: clip_grad_norm_(1.0) then
// optim.AdamW, enhanced_training_multi_loss.py:758-763, 827-829; swin_unet_model.py:836-842).
//
// Tensors are addressed through device arrays of pointers; a chunk table (tensor index, chunk index)
// built by the host spreads them over workgroups.  HBM-bound: the norm reads every gradient once,
// AdamW reads g, p, m, v and writes p, m, v (and g when it applies the clip).  The norm is reduced
// from per-chunk partials in a fixed order (double), so it is bitwise reproducible.
#include "common.h"

namespace {

constexpr int OPT_THREADS = 256;
constexpr int OPT_CHUNK = 16384;   // elements per workgroup

__global__ __launch_bounds__(OPT_THREADS) void grad_sqnorm_kernel(const float* const* __restrict__ grads,
                                                                  const int64_t* __restrict__ sizes,
                                                                  const int32_t* __restrict__ chunks,
                                                                  float* __restrict__ partials) {
  __shared__ float red[OPT_THREADS / 64];
  const int t = chunks[2 * blockIdx.x], c = chunks[2 * blockIdx.x + 1];
  const float* g = grads[t];
  const int64_t n = sizes[t];
  const int64_t beg = (int64_t)c * OPT_CHUNK;
  const int64_t end = beg + OPT_CHUNK < n ? beg + OPT_CHUNK : n;
  float s = 0.f;
  for (int64_t i = beg + threadIdx.x; i < end; i += OPT_THREADS) {
    const float v = g[i];
    s += v * v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int w = 0; w < OPT_THREADS / 64; ++w) tot += red[w];
    partials[blockIdx.x] = tot;
  }
}

// out[0] = total L2 norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))  (torch's rule)
__global__ __launch_bounds__(256) void grad_norm_finalize_kernel(const float* __restrict__ partials, int n,
                                                                 float max_norm, float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[i];
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < 256; ++i) tot += red[i];
    const float norm = (float)sqrt(tot);
    out[0] = norm;
    float coef = max_norm / (norm + 1e-6f);
    out[1] = (max_norm > 0.f && coef < 1.f) ? coef : 1.f;
  }
}

__global__ __launch_bounds__(OPT_THREADS) void scale_list_kernel(float* const* __restrict__ tensors,
                                                                 const int64_t* __restrict__ sizes,
                                                                 const int32_t* __restrict__ chunks,
                                                                 const float* __restrict__ coef) {
  const float k = coef[0];
  if (k == 1.f) return;
  const int t = chunks[2 * blockIdx.x], c = chunks[2 * blockIdx.x + 1];
  float* g = tensors[t];
  const int64_t n = sizes[t];
  const int64_t beg = (int64_t)c * OPT_CHUNK;
  const int64_t end = beg + OPT_CHUNK < n ? beg + OPT_CHUNK : n;
  for (int64_t i = beg + threadIdx.x; i < end; i += OPT_THREADS) g[i] *= k;
}

struct AdamArgs {
  float* const* params;
  float* const* grads;
  float* const* m;
  float* const* v;
  const int64_t* sizes;
  const int32_t* chunks;
  const float* coef;      // device clip coefficient or null
  const float* dyn;       // device {decay, step_size, bc2_sqrt} (graph-replayable step) or null: the values below
  float decay, omb1, beta2, omb2, eps, step_size, bc2_sqrt;   // derived in double on the host
  int write_grad;
};

// torch.optim.AdamW (single-tensor formulation), in its order of operations:
//   p *= 1 - lr*wd;  m = m + (g - m)*(1 - b1);  v = v*b2 + g*g*(1 - b2);
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(OPT_THREADS) void adamw_kernel(AdamArgs a) {
  const int t = a.chunks[2 * blockIdx.x], c = a.chunks[2 * blockIdx.x + 1];
  float* p = a.params[t];
  float* g = a.grads[t];
  float* m = a.m[t];
  float* v = a.v[t];
  const int64_t n = a.sizes[t];
  const int64_t beg = (int64_t)c * OPT_CHUNK;
  const int64_t end = beg + OPT_CHUNK < n ? beg + OPT_CHUNK : n;
  const float k = a.coef ? a.coef[0] : 1.f;
  const float decay = a.dyn ? a.dyn[0] : a.decay;
  const float step_size = a.dyn ? a.dyn[1] : a.step_size;
  const float bc2_sqrt = a.dyn ? a.dyn[2] : a.bc2_sqrt;
  auto one = [&](float& gi, float& pi, float& mi, float& vi) __attribute__((always_inline)) {
    gi *= k;          // (k == 1 leaves the bits alone)
    pi *= decay;
    mi = mi + (gi - mi) * a.omb1;
    vi = vi * a.beta2 + a.omb2 * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + a.eps;
    pi -= step_size * (mi / denom);
  };
  // 16 bytes a lane where the four tensors allow it (the chunk starts on a multiple of four elements): four times
  // the bytes in flight per load instruction of this purely HBM-bound pass; the same operations per element
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                     reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  int64_t i0 = beg;
  if (vec) {
    const int64_t nvec = (end - beg) / 4;
    for (int64_t q = threadIdx.x; q < nvec; q += OPT_THREADS) {
      const int64_t i = beg + 4 * q;
      f32x4 gv = *reinterpret_cast<const f32x4*>(g + i), pv = *reinterpret_cast<const f32x4*>(p + i);
      f32x4 mv = *reinterpret_cast<const f32x4*>(m + i), vv = *reinterpret_cast<const f32x4*>(v + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float gi = gv[e], pi = pv[e], mi = mv[e], vi = vv[e];
        one(gi, pi, mi, vi);
        gv[e] = gi; pv[e] = pi; mv[e] = mi; vv[e] = vi;
      }
      if (k != 1.f && a.write_grad) *reinterpret_cast<f32x4*>(g + i) = gv;
      *reinterpret_cast<f32x4*>(p + i) = pv;
      *reinterpret_cast<f32x4*>(m + i) = mv;
      *reinterpret_cast<f32x4*>(v + i) = vv;
    }
    i0 = beg + 4 * nvec;
  }
  for (int64_t i = i0 + threadIdx.x; i < end; i += OPT_THREADS) {
    float gi = g[i], pi = p[i], mi = m[i], vi = v[i];
    one(gi, pi, mi, vi);
    if (k != 1.f && a.write_grad) g[i] = gi;
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

// One thread: advances the step counter kept on the device and derives the step's scalars from it, in
// double as the host path does.  hyper: {step, lr, weight_decay, beta1, beta2}; dyn: {decay, step_size, bc2_sqrt}.
__global__ void adamw_advance_kernel(double* __restrict__ hyper, float* __restrict__ dyn) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double step = hyper[0] + 1.0;
  hyper[0] = step;
  const double lr = hyper[1], wd = hyper[2], b1 = hyper[3], b2 = hyper[4];
  dyn[0] = (float)(1.0 - lr * wd);
  dyn[1] = (float)(lr / (1.0 - pow(b1, step)));
  dyn[2] = (float)sqrt(1.0 - pow(b2, step));
}

}  // namespace

extern "C" {

int unetk_optim_chunk_elems(void) { return OPT_CHUNK; }

int unetk_grad_norm(const float* const* grads, const int64_t* sizes, const int32_t* chunks, int nchunks,
                    float max_norm, float* partials, float* out, hipStream_t stream) {
  UNETK_CHECK_ARG(grads && sizes && chunks && partials && out && nchunks > 0, "grad_norm: bad arguments");
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(nchunks), dim3(OPT_THREADS), 0, stream, grads, sizes, chunks, partials);
  UNETK_CHECK_LAUNCH();
  hipLaunchKernelGGL(grad_norm_finalize_kernel, dim3(1), dim3(256), 0, stream, partials, nchunks, max_norm, out);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_scale_list(float* const* tensors, const int64_t* sizes, const int32_t* chunks, int nchunks,
                     const float* coef, hipStream_t stream) {
  UNETK_CHECK_ARG(tensors && sizes && chunks && coef && nchunks > 0, "scale_list: bad arguments");
  hipLaunchKernelGGL(scale_list_kernel, dim3(nchunks), dim3(OPT_THREADS), 0, stream, tensors, sizes, chunks, coef);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_adamw_step(float* const* params, float* const* grads, float* const* m, float* const* v,
                     const int64_t* sizes, const int32_t* chunks, int nchunks, double lr, double beta1,
                     double beta2, double eps, double weight_decay, int64_t step, const float* clip_coef,
                     int write_grad, hipStream_t stream) {
  UNETK_CHECK_ARG(params && grads && m && v && sizes && chunks && nchunks > 0 && step > 0, "adamw_step: bad arguments");
  AdamArgs a;
  a.params = params; a.grads = grads; a.m = m; a.v = v; a.sizes = sizes; a.chunks = chunks; a.coef = clip_coef;
  a.decay = (float)(1.0 - lr * weight_decay);
  a.omb1 = (float)(1.0 - beta1);
  a.beta2 = (float)beta2;
  a.omb2 = (float)(1.0 - beta2);
  a.eps = (float)eps;
  a.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
  a.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
  a.write_grad = write_grad;
  a.dyn = nullptr;
  hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(OPT_THREADS), 0, stream, a);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_adamw_step_device(float* const* params, float* const* grads, float* const* m, float* const* v,
                            const int64_t* sizes, const int32_t* chunks, int nchunks, double* hyper, float* dyn,
                            double eps, const float* clip_coef, int write_grad, double beta1, double beta2,
                            hipStream_t stream) {
  UNETK_CHECK_ARG(params && grads && m && v && sizes && chunks && nchunks > 0 && hyper && dyn,
                  "adamw_step_device: bad arguments");
  hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(64), 0, stream, hyper, dyn);
  UNETK_CHECK_LAUNCH();
  AdamArgs a;
  a.params = params; a.grads = grads; a.m = m; a.v = v; a.sizes = sizes; a.chunks = chunks; a.coef = clip_coef;
  a.dyn = dyn;
  a.decay = 1.f; a.step_size = 0.f; a.bc2_sqrt = 1.f;
  a.omb1 = (float)(1.0 - beta1);
  a.beta2 = (float)beta2;
  a.omb2 = (float)(1.0 - beta2);
  a.eps = (float)eps;
  a.write_grad = write_grad;
  hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(OPT_THREADS), 0, stream, a);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
