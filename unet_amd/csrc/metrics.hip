// Validation metrics as streaming reductions on the GPU (gfx950): Pearson correlation over the whole
// validation set (enhanced_training_multi_loss.py:883-889 concatenates every prediction on the host
// and calls np.corrcoef), MSE and MAE (predict.py:188-200).  Each batch adds eight double sums
// (count, Σp, Σt, Σp², Σt², Σpt, Σ|p−t|, Σ(p−t)²) to a device-resident state; nothing leaves the
// device until the caller asks for the result.  Fixed reduction order: bitwise reproducible.
#include "common.h"

namespace {

constexpr int MT_THREADS = 256;
constexpr int MT_SUMS = 8;

__global__ __launch_bounds__(MT_THREADS) void metrics_partial_kernel(const float* __restrict__ p,
                                                                     const float* __restrict__ t, long n,
                                                                     double* __restrict__ partials) {
  __shared__ double red[MT_SUMS][MT_THREADS];
  double s[MT_SUMS] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long i = (long)blockIdx.x * MT_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * MT_THREADS) {
    const double a = (double)p[i], b = (double)t[i], d = a - b;
    s[0] += 1.0;
    s[1] += a;
    s[2] += b;
    s[3] += a * a;
    s[4] += b * b;
    s[5] += a * b;
    s[6] += fabs(d);
    s[7] += d * d;
  }
#pragma unroll
  for (int k = 0; k < MT_SUMS; ++k) red[k][threadIdx.x] = s[k];
  __syncthreads();
  if (threadIdx.x < MT_SUMS) {
    double tot = 0.0;
    for (int i = 0; i < MT_THREADS; ++i) tot += red[threadIdx.x][i];
    partials[(long)blockIdx.x * MT_SUMS + threadIdx.x] = tot;
  }
}

__global__ void metrics_accumulate_kernel(const double* __restrict__ partials, int nblk, double* __restrict__ state) {
  const int k = threadIdx.x;
  if (k >= MT_SUMS) return;
  double tot = 0.0;
  for (int b = 0; b < nblk; ++b) tot += partials[(long)b * MT_SUMS + k];
  state[k] += tot;
}

}  // namespace

extern "C" {

int unetk_metrics_num_partials(int64_t n) {
  long b = (n + MT_THREADS * 8 - 1) / (MT_THREADS * 8);
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

int unetk_metrics_update(const float* pred, const float* target, int64_t n, double* partials, double* state,
                         hipStream_t stream) {
  UNETK_CHECK_ARG(pred && target && partials && state && n > 0, "metrics_update: bad arguments");
  const int nblk = unetk_metrics_num_partials(n);
  hipLaunchKernelGGL(metrics_partial_kernel, dim3(nblk), dim3(MT_THREADS), 0, stream, pred, target, (long)n, partials);
  UNETK_CHECK_LAUNCH();
  hipLaunchKernelGGL(metrics_accumulate_kernel, dim3(1), dim3(64), 0, stream, partials, nblk, state);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
