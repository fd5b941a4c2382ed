// Multi-loss reductions (gfx950): TID-weighted pixel loss, Sobel / forward-difference gradient
// loss and the post-FFT part of the spectral loss, forward and backward, as wavefront-reduction
// kernels.  Arithmetic follows reference enhanced_training_multi_loss.py:143-344.
//
// All inputs are fp32 (B,1,H,W).  Forward kernels write per-workgroup partial sums which a
// one-workgroup kernel adds in a fixed order in double (bitwise reproducible, no atomics) and
// turns into the loss parts plus the normalisation coefficients the backward kernels need; those
// stay on the device in a small result record, so a training step needs no host round trip
// between forward and backward.  HBM-bound; the tensors are ~1/24 of one UNet activation.
//
// Result record (floats):
//   [0] pixel  [1] gradient  [2] spectral  [3] total
//   [4] c_pix  [5] pix_weighted   d pixel/d pred = c_pix * (pix_weighted ? w : 1) * dL/dd, w>0 only when weighted
//   [6] c_grad [7] grad_weighted
//   [8] c_dx [9] dx_weighted [10] c_dy [11] dy_weighted     (forward-difference method)
#include <mutex>

#include "common.h"

namespace {

constexpr int NQ = 11;   // S_p, S_pw, N_pos, S_g, S_gw, S_dx, S_dxw, Nx_pos, S_dy, S_dyw, Ny_pos
constexpr int LT = 256;

struct LossCfg {
  int B, H, W;
  int pixel_mae;      // 0 mse, 1 mae
  int grad_method;    // 0 none, 1 sobel, 2 diff
  int pixel_use_w;    // weights present AND use_tid_weights AND combination != baseline
  int has_w;          // weights present (gradient / spectral use them whenever present)
  float tid_scale;
  float w_pixel, w_grad, w_spec;
  float hf_factor;
};

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < LT / 64; ++i) t += red[i];
  return t;
}

__device__ __forceinline__ float dval(const float* __restrict__ p, const float* __restrict__ t, int H, int W,
                                      int y, int x) {
  if (y < 0 || y >= H || x < 0 || x >= W) return 0.f;   // zero padding of F.conv2d(padding=1)
  return p[y * W + x] - t[y * W + x];
}

// One thread per pixel; grid-stride over all B*H*W pixels.
__global__ __launch_bounds__(LT) void pixgrad_fwd_kernel(const float* __restrict__ pred,
                                                         const float* __restrict__ target,
                                                         const float* __restrict__ w, LossCfg c,
                                                         float* __restrict__ gx_out, float* __restrict__ gy_out,
                                                         float* __restrict__ partials) {
  __shared__ float red[LT / 64];
  float s[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) s[i] = 0.f;
  const long npix = (long)c.B * c.H * c.W;
  const int hw = c.H * c.W;
  for (long idx = (long)blockIdx.x * LT + threadIdx.x; idx < npix; idx += (long)gridDim.x * LT) {
    const int b = idx / hw;
    const int r = idx - (long)b * hw;
    const int y = r / c.W, x = r - y * c.W;
    const float* p = pred + (long)b * hw;
    const float* t = target + (long)b * hw;
    const float d = p[r] - t[r];
    const float L = c.pixel_mae ? fabsf(d) : d * d;
    const float wv = c.has_w ? w[idx] : 0.f;
    s[0] += L;
    if (c.has_w && wv > 0.f) {
      s[1] += L * (wv * c.tid_scale);
      s[2] += 1.f;
    }
    if (c.grad_method == 1) {
      const float d00 = dval(p, t, c.H, c.W, y - 1, x - 1), d01 = dval(p, t, c.H, c.W, y - 1, x),
                  d02 = dval(p, t, c.H, c.W, y - 1, x + 1);
      const float d10 = dval(p, t, c.H, c.W, y, x - 1), d12 = dval(p, t, c.H, c.W, y, x + 1);
      const float d20 = dval(p, t, c.H, c.W, y + 1, x - 1), d21 = dval(p, t, c.H, c.W, y + 1, x),
                  d22 = dval(p, t, c.H, c.W, y + 1, x + 1);
      // cross-correlation with [[-1,0,1],[-2,0,2],[-1,0,1]] and [[-1,-2,-1],[0,0,0],[1,2,1]]
      const float gx = (d02 - d00) + 2.f * (d12 - d10) + (d22 - d20);
      const float gy = (d20 - d00) + 2.f * (d21 - d01) + (d22 - d02);
      const float g = gx * gx + gy * gy;
      gx_out[idx] = gx;
      gy_out[idx] = gy;
      s[3] += g;
      if (c.has_w && wv > 0.f) s[4] += g * wv;
    } else if (c.grad_method == 2) {
      if (x + 1 < c.W) {
        const float e = (p[r + 1] - t[r + 1]) - d;
        const float wx = c.has_w ? w[idx + 1] : 0.f;
        s[5] += e * e;
        if (c.has_w && wx > 0.f) {
          s[6] += e * e * wx;
          s[7] += 1.f;
        }
      }
      if (y + 1 < c.H) {
        const float e = (p[r + c.W] - t[r + c.W]) - d;
        const float wy = c.has_w ? w[idx + c.W] : 0.f;
        s[8] += e * e;
        if (c.has_w && wy > 0.f) {
          s[9] += e * e * wy;
          s[10] += 1.f;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const float t = block_sum(s[i], red);
    if (threadIdx.x == 0) partials[(long)blockIdx.x * NQ + i] = t;
  }
}

// Four consecutive pixels of a row per thread (W % 4 == 0, 16-byte aligned images, no gradient part or the Sobel
// one): a row of the 3x3 stencil is one 16-byte load and two edge values per image instead of three loads per pixel.
// Every pixel's values come from the expressions of pixgrad_fwd_kernel; only the order in which a thread adds its
// pixels into the sums differs (another assignment of pixels to threads).
struct Row6 {
  float v[6];
};
__device__ __forceinline__ Row6 diff_row6(const float* __restrict__ p, const float* __restrict__ t, int H, int W,
                                          int y, int x0) {
  Row6 r;
  if (y < 0 || y >= H) {      // zero padding of F.conv2d(padding=1)
#pragma unroll
    for (int k = 0; k < 6; ++k) r.v[k] = 0.f;
    return r;
  }
  const long o = (long)y * W + x0;
  const f32x4 a = *reinterpret_cast<const f32x4*>(p + o), b = *reinterpret_cast<const f32x4*>(t + o);
#pragma unroll
  for (int k = 0; k < 4; ++k) r.v[k + 1] = a[k] - b[k];
  r.v[0] = x0 > 0 ? p[o - 1] - t[o - 1] : 0.f;
  r.v[5] = x0 + 4 < W ? p[o + 4] - t[o + 4] : 0.f;
  return r;
}

__global__ __launch_bounds__(LT) void pixgrad_fwd_vec4_kernel(const float* __restrict__ pred,
                                                              const float* __restrict__ target,
                                                              const float* __restrict__ w, LossCfg c,
                                                              float* __restrict__ gx_out, float* __restrict__ gy_out,
                                                              float* __restrict__ partials) {
  __shared__ float red[LT / 64];
  float s[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) s[i] = 0.f;
  const long nquad = (long)c.B * c.H * c.W / 4;
  const int hw = c.H * c.W, wq = c.W / 4;
  for (long q = (long)blockIdx.x * LT + threadIdx.x; q < nquad; q += (long)gridDim.x * LT) {
    const long idx = q * 4;
    const int b = idx / hw;
    const int r = idx - (long)b * hw;
    const int y = r / c.W, x0 = r - y * c.W;
    const float* p = pred + (long)b * hw;
    const float* t = target + (long)b * hw;
    const Row6 d1 = diff_row6(p, t, c.H, c.W, y, x0);
    f32x4 wv = {0.f, 0.f, 0.f, 0.f};
    if (c.has_w) wv = *reinterpret_cast<const f32x4*>(w + idx);
    float gxv[4] = {0.f, 0.f, 0.f, 0.f}, gyv[4] = {0.f, 0.f, 0.f, 0.f};
    if (c.grad_method == 1) {
      const Row6 d0 = diff_row6(p, t, c.H, c.W, y - 1, x0), d2 = diff_row6(p, t, c.H, c.W, y + 1, x0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d00 = d0.v[k], d01 = d0.v[k + 1], d02 = d0.v[k + 2];
        const float d10 = d1.v[k], d12 = d1.v[k + 2];
        const float d20 = d2.v[k], d21 = d2.v[k + 1], d22 = d2.v[k + 2];
        gxv[k] = (d02 - d00) + 2.f * (d12 - d10) + (d22 - d20);
        gyv[k] = (d20 - d00) + 2.f * (d21 - d01) + (d22 - d02);
      }
      *reinterpret_cast<f32x4*>(gx_out + idx) = f32x4{gxv[0], gxv[1], gxv[2], gxv[3]};
      *reinterpret_cast<f32x4*>(gy_out + idx) = f32x4{gyv[0], gyv[1], gyv[2], gyv[3]};
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = d1.v[k + 1];
      const float L = c.pixel_mae ? fabsf(d) : d * d;
      s[0] += L;
      if (c.has_w && wv[k] > 0.f) {
        s[1] += L * (wv[k] * c.tid_scale);
        s[2] += 1.f;
      }
      if (c.grad_method == 1) {
        const float g = gxv[k] * gxv[k] + gyv[k] * gyv[k];
        s[3] += g;
        if (c.has_w && wv[k] > 0.f) s[4] += g * wv[k];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const float t = block_sum(s[i], red);
    if (threadIdx.x == 0) partials[(long)blockIdx.x * NQ + i] = t;
  }
}

// the one-pixel-per-thread kernel or, where the images allow it, the four-pixel one (the choice depends on the
// shape, the gradient method and the alignment of the images only)
inline bool loss_vec4_ok(const LossCfg& c, const void* a, const void* b, const void* w, const void* g0, const void* g1) {
  auto al = [](const void* p) { return p == nullptr || ((uintptr_t)p % 16) == 0; };
  return c.W % 4 == 0 && c.grad_method != 2 && al(a) && al(b) && al(w) && al(g0) && al(g1);
}
inline void launch_pixgrad_fwd(const float* pred, const float* target, const float* w, const LossCfg& c, float* gx,
                               float* gy, float* partials, int nblk, hipStream_t stream) {
  if (loss_vec4_ok(c, pred, target, w, gx, gy))
    hipLaunchKernelGGL(pixgrad_fwd_vec4_kernel, dim3(nblk), dim3(LT), 0, stream, pred, target, w, c, gx, gy, partials);
  else
    hipLaunchKernelGGL(pixgrad_fwd_kernel, dim3(nblk), dim3(LT), 0, stream, pred, target, w, c, gx, gy, partials);
}

// Single workgroup: sum the partials in double, emit loss parts and backward coefficients.
__global__ __launch_bounds__(LT) void pixgrad_finalize_kernel(const float* __restrict__ partials, int nblk,
                                                              LossCfg c, float* __restrict__ res) {
  // all NQ quantities in one sweep: a thread sums its rows of every quantity in row order, then thread q adds the LT
  // thread sums of quantity q in thread order (one quantity after the other, with two barriers each, gave the same
  // bits in five times the time: this launch sits between the loss kernels of every step)
  __shared__ double red[NQ][LT];
  __shared__ double tot[NQ];
  double s[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) s[q] = 0.0;
  for (int i = threadIdx.x; i < nblk; i += LT) {
    float v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = partials[(long)i * NQ + q];
#pragma unroll
    for (int q = 0; q < NQ; ++q) s[q] += (double)v[q];
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) red[q][threadIdx.x] = s[q];
  __syncthreads();
  if (threadIdx.x < NQ) {
    double t = 0.0;
    for (int i = 0; i < LT; ++i) t += red[threadIdx.x][i];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double N = (double)c.B * c.H * c.W;
  const double npos = tot[2];
  // pixel (:169-182)
  const bool pw = c.pixel_use_w && npos > 0.0;
  res[0] = (float)(pw ? tot[1] / npos : tot[0] / N);
  res[4] = (float)(pw ? (double)c.tid_scale / npos : 1.0 / N);
  res[5] = pw ? 1.f : 0.f;
  // gradient (:211-269)
  res[1] = 0.f;
  for (int i = 6; i < 12; ++i) res[i] = 0.f;
  if (c.grad_method == 1) {
    const bool gw = c.has_w && npos > 0.0;
    res[1] = (float)(gw ? tot[4] / npos : tot[3] / N);
    res[6] = (float)(gw ? 1.0 / npos : 1.0 / N);
    res[7] = gw ? 1.f : 0.f;
  } else if (c.grad_method == 2) {
    const double Nx = (double)c.B * c.H * (c.W - 1), Ny = (double)c.B * (c.H - 1) * c.W;
    const bool xw = c.has_w && tot[7] > 0.0, yw = c.has_w && tot[10] > 0.0;
    const double lx = xw ? tot[6] / tot[7] : tot[5] / Nx;
    const double ly = yw ? tot[9] / tot[10] : tot[8] / Ny;
    res[1] = (float)((lx + ly) / 2.0);
    res[8] = (float)(0.5 * (xw ? 1.0 / tot[7] : 1.0 / Nx));
    res[9] = xw ? 1.f : 0.f;
    res[10] = (float)(0.5 * (yw ? 1.0 / tot[10] : 1.0 / Ny));
    res[11] = yw ? 1.f : 0.f;
  }
}

// total = w_pixel*pixel + w_grad*gradient + w_spec*spectral
__global__ void combine_kernel(float* res, LossCfg c, int use_grad, int use_spec) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (!use_grad) res[1] = 0.f;
  if (!use_spec) res[2] = 0.f;
  res[3] = c.w_pixel * res[0] + (use_grad ? c.w_grad * res[1] : 0.f) + (use_spec ? c.w_spec * res[2] : 0.f);
}

// dpred (+)= gout * (w_pixel * d pixel + w_grad * d gradient)
__global__ void pixgrad_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                   const float* __restrict__ w, LossCfg c, const float* __restrict__ gx,
                                   const float* __restrict__ gy, const float* __restrict__ res,
                                   const float* __restrict__ gout, int use_grad, float* __restrict__ dpred,
                                   int accumulate) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long npix = (long)c.B * c.H * c.W;
  if (idx >= npix) return;
  const int hw = c.H * c.W;
  const int b = idx / hw;
  const int r = idx - (long)b * hw;
  const int y = r / c.W, x = r - y * c.W;
  const float go = gout[0];
  const float d = pred[idx] - target[idx];
  float g = 0.f;
  {   // pixel
    const float dl = c.pixel_mae ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d;
    float m = 1.f;
    if (res[5] != 0.f) m = (w[idx] > 0.f) ? w[idx] : 0.f;
    g += c.w_pixel * res[4] * m * dl;
  }
  if (use_grad && c.grad_method == 1) {
    // transposed stencil: pixel q receives from each neighbour n = q + (a,b) the term
    // 2*m(n)*(gx(n)*Sx[-a,-b] + gy(n)*Sy[-a,-b])
    const bool weighted = res[7] != 0.f;
    float acc = 0.f;
#pragma unroll
    for (int a = -1; a <= 1; ++a)
#pragma unroll
      for (int bb = -1; bb <= 1; ++bb) {
        const int ny = y + a, nx = x + bb;
        if (ny < 0 || ny >= c.H || nx < 0 || nx >= c.W) continue;
        const long n = (long)b * hw + ny * c.W + nx;
        float m = 1.f;
        if (weighted) m = (w[n] > 0.f) ? w[n] : 0.f;
        // kernel tap that pixel n applies to q: offset (q - n) = (-a, -bb)
        const int ka = -a, kb = -bb;
        const float sx = (float)(kb) * (ka == 0 ? 2.f : 1.f);        // Sx[ka][kb] = kb * (2 if ka==0 else 1)
        const float sy = (float)(ka) * (kb == 0 ? 2.f : 1.f);        // Sy[ka][kb] = ka * (2 if kb==0 else 1)
        acc += m * (gx[n] * sx + gy[n] * sy);
      }
    g += c.w_grad * res[6] * 2.f * acc;
  } else if (use_grad && c.grad_method == 2) {
    const float* p = pred + (long)b * hw;
    const float* t = target + (long)b * hw;
    float acc = 0.f;
    const bool xw = res[9] != 0.f, yw = res[11] != 0.f;
    if (x + 1 < c.W) {   // e = d[x+1]-d[x]: d/dd[x] = -1
      const float e = (p[r + 1] - t[r + 1]) - d;
      float m = 1.f;
      if (xw) m = (w[idx + 1] > 0.f) ? w[idx + 1] : 0.f;
      acc -= res[8] * m * 2.f * e;
    }
    if (x > 0) {         // e = d[x]-d[x-1]: d/dd[x] = +1, weight indexed at x
      const float e = d - (p[r - 1] - t[r - 1]);
      float m = 1.f;
      if (xw) m = (w[idx] > 0.f) ? w[idx] : 0.f;
      acc += res[8] * m * 2.f * e;
    }
    if (y + 1 < c.H) {
      const float e = (p[r + c.W] - t[r + c.W]) - d;
      float m = 1.f;
      if (yw) m = (w[idx + c.W] > 0.f) ? w[idx + c.W] : 0.f;
      acc -= res[10] * m * 2.f * e;
    }
    if (y > 0) {
      const float e = d - (p[r - c.W] - t[r - c.W]);
      float m = 1.f;
      if (yw) m = (w[idx] > 0.f) ? w[idx] : 0.f;
      acc += res[10] * m * 2.f * e;
    }
    g += c.w_grad * acc;
  }
  g *= go;
  dpred[idx] = accumulate ? dpred[idx] + g : g;
}

// The same backward for four consecutive pixels of a row per thread (W % 4 == 0, 16-byte aligned images, no gradient
// part or the Sobel one): the 3x3 neighbourhoods of the Sobel maps and the weights are three 16-byte loads and six
// edge values per image.  Every pixel's terms are added in the order of pixgrad_bwd_kernel.
__device__ __forceinline__ Row6 load_row6(const float* __restrict__ a, int H, int W, int y, int x0) {
  Row6 r;
  if (y < 0 || y >= H) {
#pragma unroll
    for (int k = 0; k < 6; ++k) r.v[k] = 0.f;
    return r;
  }
  const long o = (long)y * W + x0;
  const f32x4 v = *reinterpret_cast<const f32x4*>(a + o);
#pragma unroll
  for (int k = 0; k < 4; ++k) r.v[k + 1] = v[k];
  r.v[0] = x0 > 0 ? a[o - 1] : 0.f;
  r.v[5] = x0 + 4 < W ? a[o + 4] : 0.f;
  return r;
}

__global__ __launch_bounds__(256) void pixgrad_bwd_vec4_kernel(const float* __restrict__ pred,
                                                               const float* __restrict__ target,
                                                               const float* __restrict__ w, LossCfg c,
                                                               const float* __restrict__ gx,
                                                               const float* __restrict__ gy,
                                                               const float* __restrict__ res,
                                                               const float* __restrict__ gout, int use_grad,
                                                               float* __restrict__ dpred, int accumulate) {
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long npix = (long)c.B * c.H * c.W;
  const long idx = q * 4;
  if (idx >= npix) return;
  const int hw = c.H * c.W;
  const int b = idx / hw;
  const int r = idx - (long)b * hw;
  const int y = r / c.W, x0 = r - y * c.W;
  const float go = gout[0];
  const f32x4 pv = *reinterpret_cast<const f32x4*>(pred + idx), tv = *reinterpret_cast<const f32x4*>(target + idx);
  const bool pixel_w = res[5] != 0.f;
  const bool sobel = use_grad && c.grad_method == 1;
  const bool weighted = sobel && res[7] != 0.f;
  const long base = (long)b * hw;
  Row6 W6[3], GX[3], GY[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int k = 0; k < 6; ++k) W6[a].v[k] = GX[a].v[k] = GY[a].v[k] = 0.f;
  }
  if (pixel_w || weighted) W6[1] = load_row6(w + base, c.H, c.W, y, x0);
  if (sobel) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      GX[a] = load_row6(gx + base, c.H, c.W, y + a - 1, x0);
      GY[a] = load_row6(gy + base, c.H, c.W, y + a - 1, x0);
      if (weighted && a != 1) W6[a] = load_row6(w + base, c.H, c.W, y + a - 1, x0);
    }
  }
  float out[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float d = pv[k] - tv[k];
    float g = 0.f;
    {   // pixel
      const float dl = c.pixel_mae ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d;
      float m = 1.f;
      if (pixel_w) m = (W6[1].v[k + 1] > 0.f) ? W6[1].v[k + 1] : 0.f;
      g += c.w_pixel * res[4] * m * dl;
    }
    if (sobel) {
      float acc = 0.f;
#pragma unroll
      for (int a = -1; a <= 1; ++a)
#pragma unroll
        for (int bb = -1; bb <= 1; ++bb) {
          const int ny = y + a, nx = x0 + k + bb;
          if (ny < 0 || ny >= c.H || nx < 0 || nx >= c.W) continue;
          const float wn = W6[a + 1].v[k + 1 + bb];
          float m = 1.f;
          if (weighted) m = (wn > 0.f) ? wn : 0.f;
          const int ka = -a, kb = -bb;
          const float sx = (float)(kb) * (ka == 0 ? 2.f : 1.f);
          const float sy = (float)(ka) * (kb == 0 ? 2.f : 1.f);
          acc += m * (GX[a + 1].v[k + 1 + bb] * sx + GY[a + 1].v[k + 1 + bb] * sy);
        }
      g += c.w_grad * res[6] * 2.f * acc;
    }
    out[k] = g * go;
  }
  f32x4 o = {out[0], out[1], out[2], out[3]};
  if (accumulate) o += *reinterpret_cast<const f32x4*>(dpred + idx);
  *reinterpret_cast<f32x4*>(dpred + idx) = o;
}

// ---------------------------------------------------------------- spectral loss (after the FFTs)
// fw(y,x) = 1 + |(y,x)-(H/2,W/2)| / |(H/2,W/2)| * (factor-1) on unshifted bins (:313-326)
__device__ __forceinline__ float freq_weight(int y, int x, int H, int W, float factor) {
  const int ch = H / 2, cw = W / 2;
  const float dist = sqrtf((float)((y - ch) * (y - ch) + (x - cw) * (x - cw)));
  const float dmax = sqrtf((float)(ch * ch + cw * cw));
  return 1.0f + (dist / dmax) * (factor - 1.0f);
}

// grid (blocks per sample, B).  partials[b][blk] = sum_k (log(|Fp|+eps)-log(|Ft|+eps))^2 * fw ;
// wsum[b][blk] = sum of the sample's TID weights
__global__ __launch_bounds__(LT) void spectral_fwd_kernel(const float2* __restrict__ Fp,
                                                          const float2* __restrict__ Ft,
                                                          const float* __restrict__ w, int H, int W,
                                                          float factor, float* __restrict__ partials,
                                                          float* __restrict__ wsum) {
  __shared__ float red[LT / 64];
  const int b = blockIdx.y;
  const int hw = H * W;
  float s = 0.f, ws = 0.f;
  for (int r = blockIdx.x * LT + threadIdx.x; r < hw; r += gridDim.x * LT) {
    const float2 a = Fp[(long)b * hw + r], t = Ft[(long)b * hw + r];
    const float lp = logf(hypotf(a.x, a.y) + 1e-8f), lt = logf(hypotf(t.x, t.y) + 1e-8f);
    const float df = lp - lt;
    const int y = r / W, x = r - y * W;
    s += df * df * freq_weight(y, x, H, W, factor);
    if (w != nullptr) ws += w[(long)b * hw + r];
  }
  const float ts = block_sum(s, red);
  const float tw = block_sum(ws, red);
  if (threadIdx.x == 0) {
    partials[b * gridDim.x + blockIdx.x] = ts;
    wsum[b * gridDim.x + blockIdx.x] = tw;
  }
}

// spectral = 0.01 * mean_b [ (sum_b / HW) * wbar_b / HW ];  coef[b] = 0.01/B * wbar_b / HW^2
__global__ __launch_bounds__(256) void spectral_finalize_kernel(const float* __restrict__ partials,
                                                                const float* __restrict__ wsum, int nblk, int B,
                                                                int H, int W, int has_w, float* __restrict__ res,
                                                                float* __restrict__ coef) {
  // one thread per sample sums that sample's partials in order; thread 0 then adds the samples in order
  __shared__ double term[256];
  const double hw = (double)H * W;
  double total = 0.0;
  for (int b0 = 0; b0 < B; b0 += 256) {
    const int b = b0 + threadIdx.x;
    if (b < B) {
      double s = 0.0, ws = 0.0;
      int i = 0;
      for (; i + 8 <= nblk; i += 8) {          // eight loads of each list in flight, added in order
        float v[8], u[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          v[k] = partials[b * nblk + i + k];
          u[k] = wsum[b * nblk + i + k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          s += (double)v[k];
          ws += (double)u[k];
        }
      }
      for (; i < nblk; ++i) {
        s += (double)partials[b * nblk + i];
        ws += (double)wsum[b * nblk + i];
      }
      const double wbar = has_w ? ws / hw : 1.0;
      term[threadIdx.x] = (s / hw) * wbar / hw;
      coef[b] = (float)(0.01 / B * wbar / (hw * hw));
    }
    __syncthreads();
    if (threadIdx.x == 0)
      for (int k = 0; k < 256 && b0 + k < B; ++k) total += term[k];
    __syncthreads();
  }
  if (threadIdx.x == 0) res[2] = (float)(0.01 * total / B);
}

// Z[k] = gscale * coef[b] * 2*(lp-lt)*fw / (|Fp|+eps) * conj(Fp)/|Fp|   (0 where |Fp| == 0)
// d spectral / d pred = Re(fft2(Z))
__global__ void spectral_bwd_kernel(const float2* __restrict__ Fp, const float2* __restrict__ Ft, int B, int H,
                                    int W, float factor, const float* __restrict__ coef, float w_spec,
                                    const float* __restrict__ gout, float2* __restrict__ Z) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int hw = H * W;
  if (idx >= (long)B * hw) return;
  const int b = idx / hw;
  const int r = idx - (long)b * hw;
  const int y = r / W, x = r - y * W;
  const float2 a = Fp[idx], t = Ft[idx];
  const float ma = hypotf(a.x, a.y), mt = hypotf(t.x, t.y);
  float2 z = make_float2(0.f, 0.f);
  if (ma > 0.f) {
    const float df = logf(ma + 1e-8f) - logf(mt + 1e-8f);
    const float k = gout[0] * w_spec * coef[b] * 2.f * df * freq_weight(y, x, H, W, factor) / (ma + 1e-8f) / ma;
    z.x = k * a.x;
    z.y = -k * a.y;
  }
  Z[idx] = z;
}

inline int loss_blocks(long npix) {
  long n = (npix + LT * 4 - 1) / (LT * 4);
  if (n > 1024) n = 1024;
  if (n < 1) n = 1;
  return (int)n;
}

inline LossCfg make_cfg(int B, int H, int W, int pixel_mae, int grad_method, int pixel_use_w, int has_w,
                        float tid_scale, float w_pixel, float w_grad, float w_spec, float hf) {
  LossCfg c;
  c.B = B; c.H = H; c.W = W; c.pixel_mae = pixel_mae; c.grad_method = grad_method;
  c.pixel_use_w = pixel_use_w; c.has_w = has_w; c.tid_scale = tid_scale;
  c.w_pixel = w_pixel; c.w_grad = w_grad; c.w_spec = w_spec; c.hf_factor = hf;
  return c;
}

// Partial sums of unetk_loss_pixgrad_fwd_static: device memory that belongs to the library, one
// copy per GPU, sized for the largest launch (loss_blocks never exceeds MAX_LOSS_BLOCKS).
constexpr int MAX_LOSS_BLOCKS = 1024;
constexpr long MAX_STATIC_PIXELS = 1L << 30;
static __device__ __align__(256) float scratch[MAX_LOSS_BLOCKS * NQ];

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
float* g_scratch_ptr[MAX_DEVICES];      // device address of `scratch` on each GPU, looked up once

// The array's name is only a host-side handle: kernels must be given its device address.
// The address belongs to the CURRENT device; a stream of another GPU would hand that GPU's kernels this GPU's
// copy of the array (a peer access, or a fault without peer access), so such a call is refused.
int scratch_address(float** out, hipStream_t stream) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
    unetk_set_error("loss_pixgrad_fwd_static: hipGetDevice failed or device index %d outside [0,%d)", dev,
                    MAX_DEVICES);
    return UNETK_EHIP;
  }
  if (stream != nullptr) {
    int sdev = -1;
    e = hipStreamGetDevice(stream, &sdev);
    if (e != hipSuccess) {
      unetk_set_error("loss_pixgrad_fwd_static: hipStreamGetDevice failed: %s", hipGetErrorString(e));
      return UNETK_EHIP;
    }
    if (sdev != dev) {
      unetk_set_error("loss_pixgrad_fwd_static: the stream belongs to device %d but device %d is current; make the "
                      "stream's device current before the call", sdev, dev);
      return UNETK_EINVAL;
    }
  }
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (g_scratch_ptr[dev] == nullptr) {
    float* p = nullptr;
    e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr) {
      unetk_set_error("loss_pixgrad_fwd_static: hipGetSymbolAddress failed on device %d: %s", dev,
                      hipGetErrorString(e));
      return UNETK_EHIP;
    }
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return UNETK_OK;
}

}  // namespace

extern "C" {

int unetk_loss_num_partials(int B, int H, int W) { return loss_blocks((long)B * H * W); }
int unetk_loss_spectral_blocks(int H, int W) {
  int n = (H * W + LT * 4 - 1) / (LT * 4);
  return n > 64 ? 64 : (n < 1 ? 1 : n);
}

int unetk_loss_pixgrad_fwd(const float* pred, const float* target, const float* w, int B, int H, int W,
                           int pixel_mae, int grad_method, int pixel_use_w, float tid_scale, float* gx,
                           float* gy, float* partials, float* res, hipStream_t stream) {
  UNETK_CHECK_ARG(pred && target && partials && res, "loss_pixgrad_fwd: null pointer");
  UNETK_CHECK_ARG(grad_method >= 0 && grad_method <= 2, "loss_pixgrad_fwd: gradient method %d", grad_method);
  UNETK_CHECK_ARG(grad_method != 1 || (gx && gy), "loss_pixgrad_fwd: Sobel maps required");
  UNETK_CHECK_ARG(B > 0 && H > 0 && W > 0, "loss_pixgrad_fwd: bad shape");
  LossCfg c = make_cfg(B, H, W, pixel_mae, grad_method, pixel_use_w && w != nullptr, w != nullptr, tid_scale,
                       0.f, 0.f, 0.f, 0.f);
  const int nblk = loss_blocks((long)B * H * W);
  launch_pixgrad_fwd(pred, target, w, c, gx, gy, partials, nblk, stream);
  UNETK_CHECK_LAUNCH();
  hipLaunchKernelGGL(pixgrad_finalize_kernel, dim3(1), dim3(LT), 0, stream, partials, nblk, c, res);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int64_t unetk_loss_static_max_pixels(void) { return MAX_STATIC_PIXELS; }

int unetk_loss_pixgrad_fwd_static(const float* pred, const float* target, const float* w, int B, int H, int W,
                                  int pixel_mae, int grad_method, int pixel_use_w, float tid_scale, float* gx,
                                  float* gy, float* res, hipStream_t stream) {
  UNETK_CHECK_ARG(pred && target && res, "loss_pixgrad_fwd_static: null pointer");
  UNETK_CHECK_ARG(grad_method >= 0 && grad_method <= 2, "loss_pixgrad_fwd_static: gradient method %d", grad_method);
  UNETK_CHECK_ARG(grad_method != 1 || (gx && gy), "loss_pixgrad_fwd_static: Sobel maps required");
  UNETK_CHECK_ARG(B > 0 && H > 0 && W > 0, "loss_pixgrad_fwd_static: bad shape");
  const long npix = (long)B * H * W;
  UNETK_CHECK_ARG(npix <= MAX_STATIC_PIXELS,
                  "loss_pixgrad_fwd_static: %ld pixels exceed the %ld this entry point is built for; use "
                  "unetk_loss_pixgrad_fwd with a caller-owned partials buffer", npix, MAX_STATIC_PIXELS);
  const int nblk = loss_blocks(npix);
  UNETK_CHECK_ARG(nblk <= MAX_LOSS_BLOCKS, "loss_pixgrad_fwd_static: %d partial sums exceed the library's %d", nblk,
                  MAX_LOSS_BLOCKS);
  float* partials = nullptr;
  const int rc = scratch_address(&partials, stream);
  if (rc != UNETK_OK) return rc;
  LossCfg c = make_cfg(B, H, W, pixel_mae, grad_method, pixel_use_w && w != nullptr, w != nullptr, tid_scale,
                       0.f, 0.f, 0.f, 0.f);
  // every partial the finalize kernel reads is written by this launch first: nothing to clear
  launch_pixgrad_fwd(pred, target, w, c, gx, gy, partials, nblk, stream);
  UNETK_CHECK_LAUNCH();
  hipLaunchKernelGGL(pixgrad_finalize_kernel, dim3(1), dim3(LT), 0, stream, partials, nblk, c, res);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_loss_spectral_fwd(const void* Fp, const void* Ft, const float* w, int B, int H, int W, float factor,
                            float* partials, float* wsum, float* coef, float* res, hipStream_t stream) {
  UNETK_CHECK_ARG(Fp && Ft && partials && wsum && coef && res, "loss_spectral_fwd: null pointer");
  const int nblk = unetk_loss_spectral_blocks(H, W);
  hipLaunchKernelGGL(spectral_fwd_kernel, dim3(nblk, B), dim3(LT), 0, stream, (const float2*)Fp,
                     (const float2*)Ft, w, H, W, factor, partials, wsum);
  UNETK_CHECK_LAUNCH();
  hipLaunchKernelGGL(spectral_finalize_kernel, dim3(1), dim3(256), 0, stream, partials, wsum, nblk, B, H, W,
                     w != nullptr, res, coef);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_loss_combine(float* res, float w_pixel, float w_grad, float w_spec, int use_grad, int use_spec,
                       hipStream_t stream) {
  UNETK_CHECK_ARG(res, "loss_combine: null pointer");
  LossCfg c = make_cfg(1, 1, 1, 0, 0, 0, 0, 1.f, w_pixel, w_grad, w_spec, 0.f);
  hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, stream, res, c, use_grad, use_spec);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_loss_pixgrad_bwd(const float* pred, const float* target, const float* w, int B, int H, int W,
                           int pixel_mae, int grad_method, const float* gx, const float* gy, const float* res,
                           const float* gout, float w_pixel, float w_grad, int use_grad, float* dpred,
                           int accumulate, hipStream_t stream) {
  UNETK_CHECK_ARG(pred && target && res && gout && dpred, "loss_pixgrad_bwd: null pointer");
  LossCfg c = make_cfg(B, H, W, pixel_mae, grad_method, 0, w != nullptr, 1.f, w_pixel, w_grad, 0.f, 0.f);
  const long npix = (long)B * H * W;
  if (loss_vec4_ok(c, pred, target, w, gx, gy) && ((uintptr_t)dpred % 16) == 0)
    hipLaunchKernelGGL(pixgrad_bwd_vec4_kernel, dim3(cdiv(npix / 4, 256)), dim3(256), 0, stream, pred, target, w, c, gx,
                       gy, res, gout, use_grad, dpred, accumulate);
  else
    hipLaunchKernelGGL(pixgrad_bwd_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, stream, pred, target, w, c, gx, gy,
                       res, gout, use_grad, dpred, accumulate);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_loss_spectral_bwd(const void* Fp, const void* Ft, int B, int H, int W, float factor,
                            const float* coef, float w_spec, const float* gout, void* Z, hipStream_t stream) {
  UNETK_CHECK_ARG(Fp && Ft && coef && gout && Z, "loss_spectral_bwd: null pointer");
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(spectral_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, (const float2*)Fp,
                     (const float2*)Ft, B, H, W, factor, coef, w_spec, gout, (float2*)Z);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
