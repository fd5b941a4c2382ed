// Tiled large-area inference helpers (gfx950): tile validity scan, batched tile gather and the
// deterministic overlap-weighted blend.  Arithmetic follows reference
// large_area_predictor.py:151-216.
//
// The blend is a GATHER: every output pixel adds its covering tiles' pred*weight (and weight) in
// the reference's tile order (lat-major, then lon), in fp32 with separately rounded multiply and
// add, so the result is bit-identical to numpy's sequential `+=` over tiles.  Tiles arrive in
// batches; each launch adds the tiles [t0, t1) of the compacted list, launches are stream-ordered,
// so the per-pixel order is the global tile order.  HBM-bound: features are read once per covering
// tile, predictions once.
#include <float.h>

#include "common.h"

namespace {

// A product and a sum that round separately (numpy's "acc += pred * w"): without the pragma the compiler is free
// to contract them into one fused multiply-add, whose result differs in the last bit.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}

__device__ __forceinline__ float nan_to_num(float v) {
  // numpy.nan_to_num defaults: nan -> 0, +inf -> largest finite, -inf -> most negative finite
  if (v != v) return 0.f;
  if (v == INFINITY) return FLT_MAX;
  if (v == -INFINITY) return -FLT_MAX;
  return v;
}

// valid[tile] = 1 unless one of the nvar variables is all-NaN inside the tile window
__global__ __launch_bounds__(256) void tile_valid_kernel(const float* __restrict__ feats, int nvar, int lat,
                                                         int lon, int P, int S, int ntj,
                                                         int* __restrict__ valid) {
  __shared__ int any_finite[8];
  const int tile = blockIdx.x;
  const int i0 = (tile / ntj) * S, j0 = (tile % ntj) * S;
  if (threadIdx.x < 8) any_finite[threadIdx.x] = 0;
  __syncthreads();
  for (int v = 0; v < nvar; ++v) {
    const float* f = feats + (long)v * lat * lon;
    int found = 0;
    for (int r = threadIdx.x; r < P * P && !found; r += 256) {
      const int y = r / P, x = r - y * P;
      const float val = f[(long)(i0 + y) * lon + j0 + x];
      if (val == val) found = 1;
    }
    if (found) any_finite[v] = 1;   // benign race: every writer stores 1
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int ok = 1;
    for (int v = 0; v < nvar; ++v) ok &= any_finite[v];
    valid[tile] = ok;
  }
}

// The usual shape of tile_gather (one 16-byte pixel, tile columns and row pitch multiples of four floats): a thread
// moves four neighbouring pixels, one float4 per variable in and four 16-byte pixels out.  A tile whose first column
// is not a multiple of four takes the scalar loads.
template <typename T>
__global__ void tile_gather_quad_kernel(const float* __restrict__ feats, int nvar, int lat, int lon,
                                        const int* __restrict__ pos, int n, int P, T* __restrict__ out) {
  constexpr int E = Vec16<T>::N;
  const int Q = P / 4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * P * Q) return;
  const int t = idx / ((long)P * Q);
  const int r = idx - (long)t * P * Q;
  const int y = r / Q, x = (r - y * Q) * 4;
  const int j0 = pos[2 * t + 1];
  const long src = (long)(pos[2 * t] + y) * lon + j0 + x;
  float v[E][4];
#pragma unroll
  for (int c = 0; c < E; ++c) {
    if (c < nvar) {
      const float* f = feats + (long)c * lat * lon + src;
      if ((j0 & 3) == 0) {
        const float4 q = *reinterpret_cast<const float4*>(f);
        v[c][0] = q.x; v[c][1] = q.y; v[c][2] = q.z; v[c][3] = q.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[c][k] = f[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[c][k] = 0.f;
    }
  }
  T* o = out + (((long)t * P + y) * P + x) * E;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    Vec16<T> pix;
#pragma unroll
    for (int c = 0; c < E; ++c) pix.v[c] = from_f32<T>(c < nvar ? nan_to_num(v[c][k]) : 0.f);
    store16(o + k * E, pix);
  }
}

// out[n][y][x][0:nvar] = nan_to_num(feats[v][i0+y][j0+x]); channels nvar..ld-1 are zero
template <typename T>
__global__ void tile_gather_kernel(const float* __restrict__ feats, int nvar, int lat, int lon,
                                   const int* __restrict__ pos, int n, int P, T* __restrict__ out, int ld) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * P * P) return;
  const int t = idx / ((long)P * P);
  const int r = idx - (long)t * P * P;
  const int y = r / P, x = r - y * P;
  const long src = (long)(pos[2 * t] + y) * lon + pos[2 * t + 1] + x;
  T* o = out + idx * ld;
  if (ld == Vec16<T>::N && ((uintptr_t)out & 15) == 0) {     // the usual case (4 variables): one 16-byte store per pixel
    Vec16<T> pix;
#pragma unroll
    for (int c = 0; c < Vec16<T>::N; ++c) {
      float v = 0.f;
      if (c < nvar) v = nan_to_num(feats[(long)c * lat * lon + src]);
      pix.v[c] = from_f32<T>(v);
    }
    store16(o, pix);
    return;
  }
  for (int c = 0; c < ld; ++c) {
    float v = 0.f;
    if (c < nvar) v = nan_to_num(feats[(long)c * lat * lon + src]);
    o[c] = from_f32<T>(v);
  }
}

// acc[p] += pred*wk and cnt[p] += wk over the tiles of [t0,t1) covering p, in tile order.
__global__ void blend_accumulate_kernel(const float* __restrict__ pred, const float* __restrict__ wk,
                                        const int* __restrict__ slot, int t0, int t1, int nti, int ntj, int P,
                                        int S, float* __restrict__ acc, float* __restrict__ cnt, int lon,
                                        int row0, int row1) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)(row1 - row0) * lon;
  if (idx >= total) return;
  const int y = row0 + (int)(idx / lon), x = (int)(idx % lon);
  int ti_lo = (y - P + S) / S;        // ceil((y-P+1)/S) for y-P+1 > 0
  if (y - P + 1 <= 0) ti_lo = 0;
  int ti_hi = y / S;
  if (ti_hi > nti - 1) ti_hi = nti - 1;
  int tj_lo = (x - P + S) / S;
  if (x - P + 1 <= 0) tj_lo = 0;
  int tj_hi = x / S;
  if (tj_hi > ntj - 1) tj_hi = ntj - 1;
  const long o = (long)y * lon + x;
  float a = acc[o], c = cnt[o];
  bool touched = false;
  for (int ti = ti_lo; ti <= ti_hi; ++ti)
    for (int tj = tj_lo; tj <= tj_hi; ++tj) {
      const int s = slot[ti * ntj + tj];
      if (s < t0 || s >= t1) continue;
      const int py = y - ti * S, px = x - tj * S;
      const float w = wk[py * P + px];
      const float pv = nan_to_num(pred[((long)(s - t0) * P + py) * P + px]);
      a = add_rn(a, mul_rn(pv, w));
      c = add_rn(c, w);
      touched = true;
    }
  if (touched) {
    acc[o] = a;
    cnt[o] = c;
  }
}

// Four neighbouring pixels per thread (P, S and lon multiples of four: the four share their covering tiles and
// every access is one aligned float4); per pixel the same operations in the same order as the kernel above.
__global__ void blend_accumulate_quad_kernel(const float* __restrict__ pred, const float* __restrict__ wk,
                                             const int* __restrict__ slot, int t0, int t1, int nti, int ntj, int P,
                                             int S, float* __restrict__ acc, float* __restrict__ cnt, int lon,
                                             int row0, int row1) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int lq = lon >> 2;
  if (idx >= (long)(row1 - row0) * lq) return;
  const int y = row0 + (int)(idx / lq), x = (int)(idx % lq) * 4;
  int ti_lo = (y - P + S) / S;
  if (y - P + 1 <= 0) ti_lo = 0;
  int ti_hi = y / S;
  if (ti_hi > nti - 1) ti_hi = nti - 1;
  int tj_lo = (x - P + S) / S;
  if (x - P + 1 <= 0) tj_lo = 0;
  int tj_hi = x / S;
  if (tj_hi > ntj - 1) tj_hi = ntj - 1;
  const long o = (long)y * lon + x;
  float4 a = *reinterpret_cast<const float4*>(acc + o), c = *reinterpret_cast<const float4*>(cnt + o);
  bool touched = false;
  for (int ti = ti_lo; ti <= ti_hi; ++ti)
    for (int tj = tj_lo; tj <= tj_hi; ++tj) {
      const int s = slot[ti * ntj + tj];
      if (s < t0 || s >= t1) continue;
      const int py = y - ti * S, px = x - tj * S;
      const float4 w = *reinterpret_cast<const float4*>(wk + py * P + px);
      const float4 pv = *reinterpret_cast<const float4*>(pred + ((long)(s - t0) * P + py) * P + px);
      a.x = add_rn(a.x, mul_rn(nan_to_num(pv.x), w.x)); c.x = add_rn(c.x, w.x);
      a.y = add_rn(a.y, mul_rn(nan_to_num(pv.y), w.y)); c.y = add_rn(c.y, w.y);
      a.z = add_rn(a.z, mul_rn(nan_to_num(pv.z), w.z)); c.z = add_rn(c.z, w.z);
      a.w = add_rn(a.w, mul_rn(nan_to_num(pv.w), w.w)); c.w = add_rn(c.w, w.w);
      touched = true;
    }
  if (touched) {
    *reinterpret_cast<float4*>(acc + o) = a;
    *reinterpret_cast<float4*>(cnt + o) = c;
  }
}

__global__ void blend_normalize_kernel(const float* __restrict__ acc, const float* __restrict__ cnt,
                                       float* __restrict__ out, long n) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float c = cnt[idx];
  out[idx] = __fdiv_rn(acc[idx], c > 1e-9f ? c : 1.0f);
}

}  // namespace

extern "C" {

int unetk_tile_valid(const float* feats, int nvar, int lat, int lon, int P, int S, int nti, int ntj,
                     int* valid, hipStream_t stream) {
  UNETK_CHECK_ARG(feats && valid, "tile_valid: null pointer");
  UNETK_CHECK_ARG(nvar >= 1 && nvar <= 8 && P > 0 && S > 0 && nti > 0 && ntj > 0, "tile_valid: bad arguments");
  UNETK_CHECK_ARG((long)(nti - 1) * S + P <= lat && (long)(ntj - 1) * S + P <= lon,
                  "tile_valid: tile grid exceeds the feature grid");
  hipLaunchKernelGGL(tile_valid_kernel, dim3(nti * ntj), dim3(256), 0, stream, feats, nvar, lat, lon, P, S, ntj,
                     valid);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_tile_gather(const float* feats, int nvar, int lat, int lon, const int* pos, int n, int P, void* out,
                      int ld, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(feats && pos && out && n > 0 && ld >= nvar, "tile_gather: bad arguments");
  UNETK_CHECK_ARG(dtype == UNETK_F32 || dtype == UNETK_BF16, "tile_gather: bad dtype %d", dtype);
  const long total = (long)n * P * P;
  const int epp = dtype == UNETK_BF16 ? 8 : 4;
  if (ld == epp && nvar <= epp && P % 4 == 0 && lon % 4 == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)feats % 16) == 0 &&
      ((long)lat * lon) % 4 == 0) {
    if (dtype == UNETK_BF16)
      hipLaunchKernelGGL(tile_gather_quad_kernel<bf16>, dim3(cdiv(total / 4, 256)), dim3(256), 0, stream, feats, nvar, lat,
                         lon, pos, n, P, (bf16*)out);
    else
      hipLaunchKernelGGL(tile_gather_quad_kernel<float>, dim3(cdiv(total / 4, 256)), dim3(256), 0, stream, feats, nvar, lat,
                         lon, pos, n, P, (float*)out);
    UNETK_CHECK_LAUNCH();
    return UNETK_OK;
  }
  if (dtype == UNETK_BF16)
    hipLaunchKernelGGL(tile_gather_kernel<bf16>, dim3(cdiv(total, 256)), dim3(256), 0, stream, feats, nvar, lat,
                       lon, pos, n, P, (bf16*)out, ld);
  else
    hipLaunchKernelGGL(tile_gather_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, stream, feats, nvar, lat,
                       lon, pos, n, P, (float*)out, ld);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_blend_accumulate(const float* pred, const float* wk, const int* slot, int t0, int t1, int nti,
                           int ntj, int P, int S, float* acc, float* cnt, int lat, int lon, int row0, int row1,
                           hipStream_t stream) {
  UNETK_CHECK_ARG(pred && wk && slot && acc && cnt, "blend_accumulate: null pointer");
  UNETK_CHECK_ARG(t1 > t0 && row0 >= 0 && row1 <= lat && row1 > row0, "blend_accumulate: bad range");
  const long total = (long)(row1 - row0) * lon;
  const uintptr_t bits = (uintptr_t)pred | (uintptr_t)wk | (uintptr_t)acc | (uintptr_t)cnt;
  if (P % 4 == 0 && S % 4 == 0 && lon % 4 == 0 && (bits & 15) == 0 && getenv("UNETK_BLEND_SCALAR") == nullptr)
    hipLaunchKernelGGL(blend_accumulate_quad_kernel, dim3(cdiv(total / 4, 256)), dim3(256), 0, stream, pred, wk, slot,
                       t0, t1, nti, ntj, P, S, acc, cnt, lon, row0, row1);
  else
    hipLaunchKernelGGL(blend_accumulate_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, pred, wk, slot, t0,
                       t1, nti, ntj, P, S, acc, cnt, lon, row0, row1);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_blend_normalize(const float* acc, const float* cnt, float* out, int64_t n, hipStream_t stream) {
  UNETK_CHECK_ARG(acc && cnt && out && n > 0, "blend_normalize: bad arguments");
  hipLaunchKernelGGL(blend_normalize_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, acc, cnt, out, (long)n);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
