// Separable Gaussian low-pass of a float64 grid, scipy.ndimage.gaussian_filter semantics
// (mode='reflect', truncate 4 sigma), as the reference's long-wavelength background uses it
// (data_preparation.py:392-414: NaN -> 0 before the filter, NaN restored after it).
//
// One kernel filters along the contiguous axis; the other axis is reached by transposing, which
// keeps every global access coalesced.  A workgroup owns 16 rows x 256 columns: the 256 + 2r input
// columns of each row (reflected at the borders, NaNs zeroed) are staged in LDS, then each thread
// produces 16 consecutive outputs of one row from a window of inputs that slides through
// registers: per 16 taps it reads 16 new inputs from LDS and issues 256 FMAs, so the pass is bound
// by the double-precision FMA rate, not by LDS or HBM.  The weights are read through the scalar
// path (uniform index).  Lanes of a wave run along the ROWS of the tile and the row pitch is an
// odd number of 8-byte words, so the LDS reads are conflict-free.  The tile is sized by the radius
// (dynamic LDS): 66 KB at radius 124, two workgroups per CU.
#include "common.h"

namespace {

constexpr int FT_ROWS = 16;       // rows per workgroup
constexpr int FT_COLS = 256;      // output columns per workgroup
constexpr int FT_R = 16;          // outputs per thread (and taps per register block)
constexpr int FT_RMAX = 256;      // largest radius (sigma up to 64 pixels)
// doubles per staged row: the columns the register windows read, plus one so the pitch is odd
__host__ __device__ inline int ft_pitch(int radius) { return FT_COLS + (2 * radius + 1 + FT_R - 1) / FT_R * FT_R + 1; }

// scipy 'reflect' (half-sample symmetric): ... c b a | a b c d | d c b ...
__device__ __forceinline__ int reflect_index(int i, int n) {
  while (i < 0 || i >= n) i = i < 0 ? -i - 1 : 2 * n - 1 - i;
  return i;
}

// out[y][x] = sum_t w[t] * in[y][reflect(x - r + t)], t = 0..2r.  wpad: the 2r+1 weights followed
// by zeros up to a multiple of FT_R.  nan_src (same shape as out, or null): where it holds NaN the
// output is NaN.
__global__ __launch_bounds__(256) void gauss_rows_kernel(const double* __restrict__ in, double* __restrict__ out,
                                                         const double* __restrict__ nan_src, int rows, int cols,
                                                         const double* __restrict__ wpad, int radius, int zero_nan) {
  extern __shared__ double tile[];                        // FT_ROWS x pitch, sized by the radius at launch
  const int FT_PITCH = ft_pitch(radius);
  const int x0 = blockIdx.x * FT_COLS, y0 = blockIdx.y * FT_ROWS;
  const int width = FT_COLS + 2 * radius;                 // staged inputs that carry weight
  const int nblk = (2 * radius + 1 + FT_R - 1) / FT_R;    // tap blocks
  const int span = FT_COLS + nblk * FT_R;                 // columns the register windows read
  for (int ry = 0; ry < FT_ROWS; ++ry) {
    const int y = y0 + ry;
    for (int i = threadIdx.x; i < span; i += 256) {
      double v = 0.0;
      if (y < rows && i < width) {
        v = in[(size_t)y * cols + reflect_index(x0 - radius + i, cols)];
        if (zero_nan && v != v) v = 0.0;
      }
      tile[ry * FT_PITCH + i] = v;
    }
  }
  __syncthreads();
  const int ry = threadIdx.x % FT_ROWS, xg = threadIdx.x / FT_ROWS;
  const double* row = tile + ry * FT_PITCH + xg * FT_R;
  double acc[FT_R], xw[2 * FT_R];
#pragma unroll
  for (int o = 0; o < FT_R; ++o) {
    acc[o] = 0.0;
    xw[o] = row[o];
  }
  for (int tb = 0; tb < nblk * FT_R; tb += FT_R) {
#pragma unroll
    for (int o = 0; o < FT_R; ++o) xw[FT_R + o] = row[tb + FT_R + o];
#pragma unroll
    for (int tt = 0; tt < FT_R; ++tt) {
      const double wt = wpad[tb + tt];
#pragma unroll
      for (int o = 0; o < FT_R; ++o) acc[o] = fma(wt, xw[o + tt], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < FT_R; ++o) xw[o] = xw[FT_R + o];
  }
  const int y = y0 + ry;
  if (y >= rows) return;
#pragma unroll
  for (int o = 0; o < FT_R; ++o) {
    const int x = x0 + xg * FT_R + o;
    if (x < cols) {
      double v = acc[o];
      if (nan_src != nullptr) {
        const double s = nan_src[(size_t)y * cols + x];
        if (s != s) v = s;
      }
      out[(size_t)y * cols + x] = v;
    }
  }
}

__global__ __launch_bounds__(256) void transpose_f64_kernel(const double* __restrict__ in, double* __restrict__ out,
                                                            int rows, int cols) {
  __shared__ double t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;      // 32 x 8
  for (int j = ty; j < 32; j += 8)
    if (by + j < rows && bx + tx < cols) t[j][tx] = in[(size_t)(by + j) * cols + bx + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (bx + j < cols && by + tx < rows) out[(size_t)(bx + j) * rows + by + tx] = t[tx][j];
}

}  // namespace

extern "C" {

int unetk_gauss_rows_max_radius(void) { return FT_RMAX; }

int unetk_gauss_rows_f64(const double* in, double* out, const double* nan_src, int rows, int cols,
                         const double* wpad, int radius, int zero_nan, hipStream_t stream) {
  UNETK_CHECK_ARG(in && out && wpad && in != out, "gauss_rows: null or aliased pointer");
  UNETK_CHECK_ARG(rows > 0 && cols > 0, "gauss_rows: bad shape %d x %d", rows, cols);
  UNETK_CHECK_ARG(radius >= 0 && radius <= FT_RMAX, "gauss_rows: radius %d (0..%d)", radius, FT_RMAX);
  dim3 grid((unsigned)cdiv(cols, FT_COLS), (unsigned)cdiv(rows, FT_ROWS));
  UNETK_CHECK_ARG(grid.y < 65536u, "gauss_rows: too many rows");
  const size_t lds = (size_t)FT_ROWS * ft_pitch(radius) * sizeof(double);
  if (lds > 64 * 1024) {                                  // opt in on the current device (cheap, idempotent)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_rows_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      unetk_set_error("gauss_rows: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
      return UNETK_EHIP;
    }
  }
  hipLaunchKernelGGL(gauss_rows_kernel, grid, dim3(256), lds, stream, in, out, nan_src, rows, cols, wpad, radius,
                     zero_nan);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_transpose_f64(const double* in, double* out, int rows, int cols, hipStream_t stream) {
  UNETK_CHECK_ARG(in && out && in != out && rows > 0 && cols > 0, "transpose: bad arguments");
  dim3 grid((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32));
  UNETK_CHECK_ARG(grid.y < 65536u, "transpose: too many rows");
  hipLaunchKernelGGL(transpose_f64_kernel, grid, dim3(256), 0, stream, in, out, rows, cols);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
