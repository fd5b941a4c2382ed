// Shifted-window attention for ANY window side and head dimension (reference swin_unet_model.py:161-200, 241-269,
// 339-360): the path SwinUNet takes when the configuration is not the one the MFMA kernels of attn.hip / swin.hip
// are written for (window 8, head dimension 32).  Correctness first: one workgroup per (window, head), one thread
// per query (or key) row, K / V / Q / dO of the window as fp32 rows in LDS, every score recomputed where it is
// needed instead of stored (a 144 x 144 window would not fit beside its operands).  The softmax statistics, the
// products and the sums are fp32 whatever the storage type.  No atomics: the bias-table gradient of a window is
// summed entry by entry, one thread per table entry, in a fixed order; the windows' rows are added by the caller.
#include <math.h>

#include "common.h"

namespace {

struct AnyGeom {
  int B, He, We, C, heads, ws, hd, shift, use_mask;
  float scale;
};

__device__ __forceinline__ int any_region(int s, int extent, int ws) {
  return s < extent - ws ? 0 : (s < extent - ws / 2 ? 1 : 2);       // the reference's slices (0,-ws), (-ws,-shift), (-shift,)
}

// flat token (b, y, x) of row n of window (wy, wx) in the frame rolled by -shift
__device__ __forceinline__ long any_token(const AnyGeom& g, int b, int wy, int wx, int n) {
  int ty = wy * g.ws + n / g.ws + g.shift, tx = wx * g.ws + n % g.ws + g.shift;
  if (ty >= g.He) ty -= g.He;
  if (tx >= g.We) tx -= g.We;
  return ((long)b * g.He + ty) * g.We + tx;
}

struct AnyItem {
  int b, wy, wx, head;
};

__device__ __forceinline__ AnyItem any_item(const AnyGeom& g) {
  const int nwx = g.We / g.ws, nwy = g.He / g.ws;
  int t = blockIdx.x;
  AnyItem it;
  it.head = t % g.heads;
  t /= g.heads;
  it.wx = t % nwx;
  t /= nwx;
  it.wy = t % nwy;
  it.b = t / nwy;
  return it;
}

// What every phase needs of the window besides the operands: the head's bias table and the tokens' region ids.
__device__ __forceinline__ void any_prepare(const AnyGeom& g, const AnyItem& it, const float* table, float* tbl, int* ids) {
  const int N = g.ws * g.ws, T2 = (2 * g.ws - 1) * (2 * g.ws - 1);
  for (int i = threadIdx.x; i < T2; i += blockDim.x) tbl[i] = table[(long)i * g.heads + it.head];
  for (int n = threadIdx.x; n < N; n += blockDim.x)
    ids[n] = 3 * any_region(it.wy * g.ws + n / g.ws, g.He, g.ws) + any_region(it.wx * g.ws + n % g.ws, g.We, g.ws);
}

// bias + mask of the score of (query i, key j)
__device__ __forceinline__ float any_bias(const AnyGeom& g, const float* tbl, const int* ids, int i, int j) {
  const int ws = g.ws;
  const int yi = i / ws, xi = i - yi * ws, yj = j / ws, xj = j - yj * ws;
  float v = tbl[(yi - yj + ws - 1) * (2 * ws - 1) + (xi - xj + ws - 1)];
  if (g.use_mask && ids[i] != ids[j]) v += -100.0f;
  return v;
}

template <typename T>
__device__ __forceinline__ void any_stage(const T* src, float* row, int hd) {
  for (int d = 0; d < hd; ++d) row[d] = to_f32(src[d]);
}

template <int DP> __device__ __forceinline__ float any_dot(const float (&a)[DP], const float* b, int hd) {
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < DP; ++d)
    if (d < hd) s += a[d] * b[d];
  return s;
}

template <typename T, int DP>
__global__ __launch_bounds__(256) void attn_any_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                           const float* __restrict__ table, AnyGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = g.ws * g.ws, hd = g.hd, ld = hd + 1, T2 = (2 * g.ws - 1) * (2 * g.ws - 1);
  float* Ks = smem;
  float* Vs = Ks + N * ld;
  float* tbl = Vs + N * ld;
  int* ids = reinterpret_cast<int*>(tbl + T2);
  const AnyItem it = any_item(g);
  const int i = threadIdx.x;
  any_prepare(g, it, table, tbl, ids);
  float q[DP];
  long tok = 0;
  if (i < N) {
    tok = any_token(g, it.b, it.wy, it.wx, i);
    const T* base = qkv + tok * 3 * g.C + it.head * hd;
#pragma unroll
    for (int d = 0; d < DP; ++d) q[d] = d < hd ? to_f32(base[d]) : 0.f;
    any_stage<T>(base + g.C, Ks + i * ld, hd);
    any_stage<T>(base + 2 * g.C, Vs + i * ld, hd);
  }
  __syncthreads();
  if (i >= N) return;
  float mx = -INFINITY;
  for (int j = 0; j < N; ++j) mx = fmaxf(mx, any_dot<DP>(q, Ks + j * ld, hd) * g.scale + any_bias(g, tbl, ids, i, j));
  float sum = 0.f, o[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) o[d] = 0.f;
  for (int j = 0; j < N; ++j) {
    const float e = expf(any_dot<DP>(q, Ks + j * ld, hd) * g.scale + any_bias(g, tbl, ids, i, j) - mx);
    sum += e;
    const float* v = Vs + j * ld;
#pragma unroll
    for (int d = 0; d < DP; ++d)
      if (d < hd) o[d] += e * v[d];
  }
  const float inv = 1.0f / sum;
  T* dst = out + tok * g.C + it.head * hd;
#pragma unroll
  for (int d = 0; d < DP; ++d)
    if (d < hd) dst[d] = from_f32<T>(o[d] * inv);
}

// Backward of one (window, head).  With P = softmax(S), dP = dO V^T, D_i = sum_j P_ij dP_ij, dS = P (dP - D):
// dQ = scale dS K, dK = scale dS^T Q, dV = P^T dO, and the window's row of the bias-table gradient.
template <typename T, int DP>
__global__ __launch_bounds__(256) void attn_any_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                           T* __restrict__ dqkv, const float* __restrict__ table,
                                                           float* __restrict__ dtable_partials, AnyGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = g.ws * g.ws, hd = g.hd, ld = hd + 1, ws = g.ws, T2 = (2 * ws - 1) * (2 * ws - 1);
  float* Qs = smem;
  float* Ks = Qs + N * ld;
  float* Vs = Ks + N * ld;
  float* Gs = Vs + N * ld;          // dO
  float* rmax = Gs + N * ld;
  float* rsum = rmax + N;
  float* rdd = rsum + N;            // D_i
  float* tbl = rdd + N;
  int* ids = reinterpret_cast<int*>(tbl + T2);
  const AnyItem it = any_item(g);
  const int i = threadIdx.x;
  any_prepare(g, it, table, tbl, ids);
  long tok = 0;
  if (i < N) {
    tok = any_token(g, it.b, it.wy, it.wx, i);
    const T* base = qkv + tok * 3 * g.C + it.head * hd;
    any_stage<T>(base, Qs + i * ld, hd);
    any_stage<T>(base + g.C, Ks + i * ld, hd);
    any_stage<T>(base + 2 * g.C, Vs + i * ld, hd);
    any_stage<T>(dout + tok * g.C + it.head * hd, Gs + i * ld, hd);
  }
  __syncthreads();
  auto score = [&](const float* qrow, int qi, int kj) {
    float s = 0.f;
    const float* k = Ks + kj * ld;
    for (int d = 0; d < hd; ++d) s += qrow[d] * k[d];
    return s * g.scale + any_bias(g, tbl, ids, qi, kj);
  };
  auto dotg = [&](const float* grow, int kj) {
    float s = 0.f;
    const float* v = Vs + kj * ld;
    for (int d = 0; d < hd; ++d) s += grow[d] * v[d];
    return s;
  };
  // ---- phase A, thread = query row: the row's statistics and dQ
  if (i < N) {
    const float* qrow = Qs + i * ld;
    const float* grow = Gs + i * ld;
    float mx = -INFINITY;
    for (int j = 0; j < N; ++j) mx = fmaxf(mx, score(qrow, i, j));
    float sum = 0.f, dnum = 0.f;
    for (int j = 0; j < N; ++j) {
      const float e = expf(score(qrow, i, j) - mx);
      sum += e;
      dnum += e * dotg(grow, j);
    }
    const float inv = 1.0f / sum, dd = dnum * inv;
    float dq[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) dq[d] = 0.f;
    for (int j = 0; j < N; ++j) {
      const float p = expf(score(qrow, i, j) - mx) * inv;
      const float ds = p * (dotg(grow, j) - dd);
      const float* k = Ks + j * ld;
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < hd) dq[d] += ds * k[d];
    }
    T* dst = dqkv + tok * 3 * g.C + it.head * hd;
#pragma unroll
    for (int d = 0; d < DP; ++d)
      if (d < hd) dst[d] = from_f32<T>(dq[d] * g.scale);
    rmax[i] = mx;
    rsum[i] = inv;
    rdd[i] = dd;
  }
  __syncthreads();
  // ---- phase B, thread = key row: dK, then dV
  if (i < N) {
    const int j = i;
    float acc[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) acc[d] = 0.f;
    for (int qi = 0; qi < N; ++qi) {
      const float* qrow = Qs + qi * ld;
      const float p = expf(score(qrow, qi, j) - rmax[qi]) * rsum[qi];
      const float ds = p * (dotg(Gs + qi * ld, j) - rdd[qi]);
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < hd) acc[d] += ds * qrow[d];
    }
    T* dst = dqkv + tok * 3 * g.C + g.C + it.head * hd;
#pragma unroll
    for (int d = 0; d < DP; ++d)
      if (d < hd) dst[d] = from_f32<T>(acc[d] * g.scale);
#pragma unroll
    for (int d = 0; d < DP; ++d) acc[d] = 0.f;
    for (int qi = 0; qi < N; ++qi) {
      const float p = expf(score(Qs + qi * ld, qi, j) - rmax[qi]) * rsum[qi];
      const float* grow = Gs + qi * ld;
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < hd) acc[d] += p * grow[d];
    }
    dst += g.C;
#pragma unroll
    for (int d = 0; d < DP; ++d)
      if (d < hd) dst[d] = from_f32<T>(acc[d]);
  }
  // ---- phase C, thread = entry of the bias table: the sum of dS over the (query, key) pairs at that offset
  float* part = dtable_partials + (long)blockIdx.x * T2;
  for (int r = threadIdx.x; r < T2; r += blockDim.x) {
    const int oy = r / (2 * ws - 1) - (ws - 1), ox = r % (2 * ws - 1) - (ws - 1);
    float s = 0.f;
    for (int qi = 0; qi < N; ++qi) {
      const int yi = qi / ws, xi = qi - yi * ws, yj = yi - oy, xj = xi - ox;
      if (yj < 0 || yj >= ws || xj < 0 || xj >= ws) continue;
      const int j = yj * ws + xj;
      const float p = expf(score(Qs + qi * ld, qi, j) - rmax[qi]) * rsum[qi];
      s += p * (dotg(Gs + qi * ld, j) - rdd[qi]);
    }
    part[r] = s;
  }
}

int any_check(const char* name, int He, int We, int C, int heads, int ws, int shift, int dtype) {
  UNETK_CHECK_ARG(dtype == UNETK_F32 || dtype == UNETK_BF16, "%s: bad dtype %d", name, dtype);
  UNETK_CHECK_ARG(heads > 0 && C > 0 && C % heads == 0, "%s: %d channels do not split into %d heads", name, C, heads);
  UNETK_CHECK_ARG(unetk_window_attn_any_supported(ws, C / heads), "%s: window %d with head dimension %d is beyond the general "
                  "kernels (window side 2...16, head dimension up to 64, window tokens x head dimension up to 9216)",
                  name, ws, C / heads);
  UNETK_CHECK_ARG(He > 0 && We > 0 && He % ws == 0 && We % ws == 0, "%s: token grid %dx%d not a multiple of the window side %d",
                  name, He, We, ws);
  UNETK_CHECK_ARG(shift >= 0 && shift < ws, "%s: bad shift %d", name, shift);
  return UNETK_OK;
}

template <typename K> int any_lds(K kernel, size_t bytes) {
  if (bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      unetk_set_error("window_attn_any: %zu bytes of LDS refused: %s", bytes, hipGetErrorString(e));
      return UNETK_EHIP;
    }
  }
  return UNETK_OK;
}

}  // namespace

extern "C" {

int unetk_window_attn_any_supported(int ws, int hd) {
  return ws >= 2 && ws <= 16 && hd >= 1 && hd <= 64 && ws * ws * hd <= 9216;
}

int unetk_window_attn_any_fwd(const void* qkv, void* out, const float* table, int B, int He, int We, int C, int heads,
                              int ws, int shift, int use_mask, float scale, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(qkv && out && table && B > 0, "window_attn_any_fwd: bad arguments");
  if (int rc = any_check("window_attn_any_fwd", He, We, C, heads, ws, shift, dtype)) return rc;
  const int hd = C / heads, N = ws * ws, T2 = (2 * ws - 1) * (2 * ws - 1);
  AnyGeom g{B, He, We, C, heads, ws, hd, shift, use_mask, scale};
  const long nblk = (long)B * (He / ws) * (We / ws) * heads;
  UNETK_CHECK_ARG(nblk < (1L << 31), "window_attn_any_fwd: too many windows");
  const size_t lds = ((size_t)2 * N * (hd + 1) + T2 + N) * 4;
  const int threads = (N + 63) / 64 * 64;
#define UNETK_ANY_FWD(T, DP)                                                                                   \
  do {                                                                                                         \
    if (int rc = any_lds(attn_any_fwd_kernel<T, DP>, lds)) return rc;                                          \
    hipLaunchKernelGGL((attn_any_fwd_kernel<T, DP>), dim3((unsigned)nblk), dim3(threads), lds, stream, (const T*)qkv, \
                       (T*)out, table, g);                                                                     \
  } while (0)
  if (dtype == UNETK_BF16) {
    if (hd <= 32) UNETK_ANY_FWD(bf16, 32); else UNETK_ANY_FWD(bf16, 64);
  } else {
    if (hd <= 32) UNETK_ANY_FWD(float, 32); else UNETK_ANY_FWD(float, 64);
  }
#undef UNETK_ANY_FWD
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_window_attn_any_bwd(const void* qkv, const void* dout, void* dqkv, const float* table, float* dtable_partials,
                              int B, int He, int We, int C, int heads, int ws, int shift, int use_mask, float scale,
                              int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(qkv && dout && dqkv && table && dtable_partials && B > 0, "window_attn_any_bwd: bad arguments");
  if (int rc = any_check("window_attn_any_bwd", He, We, C, heads, ws, shift, dtype)) return rc;
  const int hd = C / heads, N = ws * ws, T2 = (2 * ws - 1) * (2 * ws - 1);
  AnyGeom g{B, He, We, C, heads, ws, hd, shift, use_mask, scale};
  const long nblk = (long)B * (He / ws) * (We / ws) * heads;
  UNETK_CHECK_ARG(nblk < (1L << 31), "window_attn_any_bwd: too many windows");
  const size_t lds = ((size_t)4 * N * (hd + 1) + 3 * N + T2 + N) * 4;
  const int threads = (N + 63) / 64 * 64;
#define UNETK_ANY_BWD(T, DP)                                                                                   \
  do {                                                                                                         \
    if (int rc = any_lds(attn_any_bwd_kernel<T, DP>, lds)) return rc;                                          \
    hipLaunchKernelGGL((attn_any_bwd_kernel<T, DP>), dim3((unsigned)nblk), dim3(threads), lds, stream, (const T*)qkv, \
                       (const T*)dout, (T*)dqkv, table, dtable_partials, g);                                   \
  } while (0)
  if (dtype == UNETK_BF16) {
    if (hd <= 32) UNETK_ANY_BWD(bf16, 32); else UNETK_ANY_BWD(bf16, 64);
  } else {
    if (hd <= 32) UNETK_ANY_BWD(float, 32); else UNETK_ANY_BWD(float, 64);
  }
#undef UNETK_ANY_BWD
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
