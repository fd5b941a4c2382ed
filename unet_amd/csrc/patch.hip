// Training-patch extraction on the GPU (gfx950): the sliding-window quality rules of
// EnhancedDataPreparation.extract_patches (enhanced_training_multi_loss.py:633-720) as one
// reduction kernel over all candidate windows, and a batched gather of the windows that passed
// (optionally with the NaN/inf -> 0 clean-up and the flips of the dataset's __getitem__, :470-513).
// HBM-bound integer/compare work: every candidate window is read once (overlapping windows hit L2).
#include "common.h"

namespace {

constexpr int PQ_THREADS = 256;

__device__ __forceinline__ double block_sum_d(double v, double* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = PQ_THREADS / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// One workgroup per candidate window (row-major over the candidate grid).  flags[c] = 1 when the
// window passes every rule, in the reference's order: label valid ratio, label nanstd >= 1e-6,
// positive-weight ratio >= min_valid/2, every feature's valid ratio.
__global__ __launch_bounds__(PQ_THREADS) void patch_qc_kernel(const float* __restrict__ feats, int nvar,
                                                              const float* __restrict__ labels,
                                                              const float* __restrict__ weights, int lat, int lon,
                                                              int P, int S, int margin, int ncol, double min_valid,
                                                              int* __restrict__ flags) {
  __shared__ double red[PQ_THREADS];
  const int c = blockIdx.x;
  const int i0 = margin + (c / ncol) * S, j0 = margin + (c % ncol) * S;
  const int size = P * P;
  const long plane = (long)lat * lon;
  double n_lab = 0, s_lab = 0, n_w = 0;
  for (int e = threadIdx.x; e < size; e += PQ_THREADS) {
    const long off = (long)(i0 + e / P) * lon + j0 + e % P;
    const float l = labels[off];
    if (!isnan(l)) {
      n_lab += 1;
      s_lab += (double)l;
    }
    if (weights[off] > 0.f) n_w += 1;
  }
  n_lab = block_sum_d(n_lab, red);
  s_lab = block_sum_d(s_lab, red);
  n_w = block_sum_d(n_w, red);
  bool ok = !(n_lab / size < min_valid);
  if (ok) {
    // population standard deviation over the non-NaN cells (np.nanstd), second pass around the mean
    const double mean = s_lab / n_lab;
    double ss = 0;
    for (int e = threadIdx.x; e < size; e += PQ_THREADS) {
      const float l = labels[(long)(i0 + e / P) * lon + j0 + e % P];
      if (!isnan(l)) {
        const double d = (double)l - mean;
        ss += d * d;
      }
    }
    ss = block_sum_d(ss, red);
    const double sd = sqrt(ss / n_lab);          // NaN when the window has no valid cell: passes, as in numpy
    if (sd < 1e-6) ok = false;
  }
  if (ok && n_w / size < min_valid * 0.5) ok = false;
  for (int v = 0; v < nvar && ok; ++v) {           // ok is uniform over the workgroup
    double n_f = 0;
    for (int e = threadIdx.x; e < size; e += PQ_THREADS)
      if (!isnan(feats[v * plane + (long)(i0 + e / P) * lon + j0 + e % P])) n_f += 1;
    n_f = block_sum_d(n_f, red);
    if (n_f / size < min_valid) ok = false;
  }
  if (threadIdx.x == 0) flags[c] = ok ? 1 : 0;
}

__device__ __forceinline__ float clean(float v, int sanitize) { return (sanitize && !isfinite(v)) ? 0.f : v; }

// out_f (n, nvar, P, P), out_l (n, 1, P, P), out_w (n, 1, P, P) from the windows at pos[k] = (lat, lon).
// flips (nullable): per window 0 none, 1 last axis reversed, 2 rows reversed.
__global__ void patch_gather_kernel(const float* __restrict__ feats, int nvar, const float* __restrict__ labels,
                                    const float* __restrict__ weights, int lat, int lon,
                                    const int* __restrict__ pos, const int* __restrict__ flips, int n, int P,
                                    int sanitize, float* __restrict__ out_f, float* __restrict__ out_l,
                                    float* __restrict__ out_w) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int size = P * P;
  const long total = (long)n * (nvar + 2) * size;
  if (idx >= total) return;
  const int e = idx % size;
  const int ch = (idx / size) % (nvar + 2);
  const int k = idx / ((long)size * (nvar + 2));
  int y = e / P, x = e % P;
  const int fl = flips ? flips[k] : 0;
  const int sy = fl == 2 ? P - 1 - y : y, sx = fl == 1 ? P - 1 - x : x;
  const long off = (long)(pos[2 * k] + sy) * lon + pos[2 * k + 1] + sx;
  if (ch < nvar) out_f[((long)k * nvar + ch) * size + e] = clean(feats[(long)ch * lat * lon + off], sanitize);
  else if (ch == nvar) out_l[(long)k * size + e] = clean(labels[off], sanitize);
  else out_w[(long)k * size + e] = clean(weights[off], sanitize);
}

}  // namespace

extern "C" {

int unetk_patch_qc(const float* feats, int nvar, const float* labels, const float* weights, int lat, int lon,
                   int P, int S, double min_valid, int* flags, hipStream_t stream) {
  UNETK_CHECK_ARG(feats && labels && weights && flags, "patch_qc: null pointer");
  UNETK_CHECK_ARG(P > 0 && S > 0 && nvar > 0, "patch_qc: bad patch size / stride");
  const int margin = P / 8;
  const int nrow = lat - P - 2 * margin >= 0 ? (lat - P - 2 * margin) / S + 1 : 0;
  const int ncol = lon - P - 2 * margin >= 0 ? (lon - P - 2 * margin) / S + 1 : 0;
  if (nrow == 0 || ncol == 0) return UNETK_OK;
  hipLaunchKernelGGL(patch_qc_kernel, dim3((unsigned)(nrow * ncol)), dim3(PQ_THREADS), 0, stream, feats, nvar, labels,
                     weights, lat, lon, P, S, margin, ncol, min_valid, flags);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_patch_gather(const float* feats, int nvar, const float* labels, const float* weights, int lat, int lon,
                       const int* pos, const int* flips, int n, int P, int sanitize, float* out_f, float* out_l,
                       float* out_w, hipStream_t stream) {
  UNETK_CHECK_ARG(feats && labels && weights && pos && out_f && out_l && out_w, "patch_gather: null pointer");
  if (n <= 0) return UNETK_OK;
  const long total = (long)n * (nvar + 2) * P * P;
  hipLaunchKernelGGL(patch_gather_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, feats, nvar, labels,
                     weights, lat, lon, pos, flips, n, P, sanitize, out_f, out_l, out_w);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
