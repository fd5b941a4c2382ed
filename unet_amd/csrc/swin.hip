// Swin-UNet specific kernels (gfx950): LayerNorm, GELU, shifted-window attention on MFMA,
// patch embedding / patch merging gathers, the ConvTranspose2d(2,2) pixel shuffle, DropPath scale
// and per-channel sums.  Arithmetic follows reference swin_unet_model.py:139-535.
//
// Tokens are (B, H, W, C) channel-innermost in the compute dtype.  Window attention is one
// wavefront per (window, head): the cyclic shift, the window partition and its reverse are index
// arithmetic on the token grid (no copies), the relative-position bias comes from a 225-entry
// table staged in LDS, and the shift mask (-100 between different regions) is recomputed from the
// window coordinates.  QK^T and PV (and the five products of the backward) run on MFMA from
// LDS-staged operands; softmax is done on the accumulator registers.
#include "attn.h"
#include "common.h"
#include "mma.h"

namespace {

constexpr int WS = 8;        // window side
constexpr int NTOK = 64;     // tokens per window
constexpr int HD = 32;       // head dimension

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

// All-reduce over the 16 lanes of a DPP row (the lanes that share lane>>4) without touching LDS:
// swap inside pairs, inside quads, mirror the 8-lane halves, mirror the row.  Every lane combines
// the same pairs in the same tree, so all 16 end with the identical value.
template <int CTRL> __device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v));    // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_move<0x4E>(v));    // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_move<0x141>(v));   // row_half_mirror
  v = fmaxf(v, dpp_move<0x140>(v));   // row_mirror
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  return v;
}

// ------------------------------------------------------------------ LayerNorm over channels
// One wavefront per token.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, long ntok, int C,
                                                            float eps) {
  constexpr int E = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= ntok) return;
  const T* xr = x + tok * C;
  float s = 0.f;
  for (int c = lane * E; c < C; c += 64 * E) {
    Vec16<T> v = load16(xr + c);
#pragma unroll
    for (int e = 0; e < E; ++e) s += to_f32(v.v[e]);
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int c = lane * E; c < C; c += 64 * E) {
    Vec16<T> v = load16(xr + c);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float d = to_f32(v.v[e]) - mean;
      q += d * d;
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  for (int c = lane * E; c < C; c += 64 * E) {
    Vec16<T> v = load16(xr + c);
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < E; ++e)
      o.v[e] = from_f32<T>((to_f32(v.v[e]) - mean) * rstd * gamma[c + e] + beta[c + e]);
    store16(y + tok * C + c, o);
  }
  if (lane == 0) {
    mean_out[tok] = mean;
    rstd_out[tok] = rstd;
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)) + add,  g = dy * gamma
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const T* __restrict__ add, T* __restrict__ dx,
                                                            long ntok, int C) {
  constexpr int E = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= ntok) return;
  const float mu = mean[tok], rs = rstd[tok];
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane * E; c < C; c += 64 * E) {
    Vec16<T> dv = load16(dy + tok * C + c), xv = load16(x + tok * C + c);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float g = to_f32(dv.v[e]) * gamma[c + e];
      s1 += g;
      s2 += g * (to_f32(xv.v[e]) - mu) * rs;
    }
  }
  s1 = wave_sum(s1) / (float)C;
  s2 = wave_sum(s2) / (float)C;
  for (int c = lane * E; c < C; c += 64 * E) {
    Vec16<T> dv = load16(dy + tok * C + c), xv = load16(x + tok * C + c);
    Vec16<T> av = zero16<T>();
    if (add != nullptr) av = load16(add + tok * C + c);
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float g = to_f32(dv.v[e]) * gamma[c + e];
      const float xh = (to_f32(xv.v[e]) - mu) * rs;
      o.v[e] = from_f32<T>(rs * (g - s1 - xh * s2) + (add ? to_f32(av.v[e]) : 0.f));
    }
    store16(dx + tok * C + c, o);
  }
}

// ---- LayerNorm with a group of G lanes per token (G = 16, 32 or 64: the smallest that holds the
// token's 16-byte vectors in at most LN_MAXV per lane).  A 96-channel bf16 token is 12 vectors: one
// wave takes four tokens instead of one, every vector is read once and kept in registers, and the
// group reductions stay inside a DPP row for G = 16.  V = vectors per lane (registers are sized by it).
constexpr int LN_MAXV = 6;

template <int G> __device__ __forceinline__ float group_sum(float v) {
  v = row16_sum(v);
  if (G >= 32) v += __shfl_xor(v, 16, 64);
  if (G >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

template <typename T, int G, int V>
__global__ __launch_bounds__(256) void layernorm_fwd_group_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta,
                                                                  float* __restrict__ mean_out,
                                                                  float* __restrict__ rstd_out, long ntok, int C,
                                                                  float eps) {
  constexpr int E = Vec16<T>::N;
  const int CG = C / E;
  const int gl = threadIdx.x % G;
  const long tok = (long)blockIdx.x * (256 / G) + threadIdx.x / G;
  const bool live = tok < ntok;                      // (dead groups still take part in the shuffles)
  const T* xr = x + (live ? tok : 0) * C;
  Vec16<T> v[V];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const int cg = k * G + gl;
    if (cg < CG) {
      v[k] = load16(xr + cg * E);
#pragma unroll
      for (int e = 0; e < E; ++e) s += to_f32(v[k].v[e]);
    }
  }
  const float mean = group_sum<G>(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < V; ++k)
    if (k * G + gl < CG) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float d = to_f32(v[k].v[e]) - mean;
        q += d * d;
      }
    }
  const float rstd = 1.0f / sqrtf(group_sum<G>(q) / (float)C + eps);
  if (!live) return;
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const int cg = k * G + gl;
    if (cg < CG) {
      Vec16<T> o;
#pragma unroll
      for (int e = 0; e < E; ++e)
        o.v[e] = from_f32<T>((to_f32(v[k].v[e]) - mean) * rstd * gamma[cg * E + e] + beta[cg * E + e]);
      store16(y + tok * C + cg * E, o);
    }
  }
  if (gl == 0) {
    mean_out[tok] = mean;
    rstd_out[tok] = rstd;
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)) + add, g = dy * gamma; and, in the same pass, this
// workgroup's share of dgamma = sum dy*xhat and dbeta = sum dy: workgroup b walks tokens
// [b*chunk, (b+1)*chunk), each lane keeps the sums of its channel vectors in registers, the token
// groups of the workgroup are added through LDS in a fixed order -> partials[b][2][C].
template <typename T, int G, int V>
__global__ __launch_bounds__(256) void layernorm_bwd_group_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd,
                                                                  const T* __restrict__ add, T* __restrict__ dx,
                                                                  long ntok, int C, int chunk,
                                                                  float* __restrict__ partials,
                                                                  const float* __restrict__ keep, long tok_per_sample,
                                                                  T* __restrict__ dxs) {
  // dxs (optional): dx as stored, times the per-sample DropPath factor keep[token / tok_per_sample] of the branch
  // that consumes it next - what scale_sample_kernel would compute from dx in a pass of its own, same bits
  constexpr int E = Vec16<T>::N;
  constexpr int TPW = 256 / G;
  __shared__ float red[2][256 * E];
  const int CG = C / E;
  const int gl = threadIdx.x % G, tg = threadIdx.x / G;
  const long pbeg = (long)blockIdx.x * chunk;
  long pend = pbeg + chunk;
  if (pend > ntok) pend = ntok;
  float ag[V][E], ab[V][E];
#pragma unroll
  for (int k = 0; k < V; ++k)
#pragma unroll
    for (int e = 0; e < E; ++e) ag[k][e] = ab[k][e] = 0.f;
  // gamma of this thread's channels, kept in registers
  float gm[V][E];
#pragma unroll
  for (int k = 0; k < V; ++k)
#pragma unroll
    for (int e = 0; e < E; ++e) gm[k][e] = (k * G + gl) < CG ? gamma[(k * G + gl) * E + e] : 0.f;
  for (long t0 = pbeg; t0 < pend; t0 += TPW) {        // (uniform trip count: the shuffles need whole waves)
    const long tok = t0 + tg;
    const bool live = tok < pend;
    const long tk = live ? tok : pbeg;
    const float mu = mean[tk], rs = rstd[tk];
    Vec16<T> dv[V], xv[V], av[V];
    float s1 = 0.f, s2 = 0.f;
    // every load of the token first (the residual gradient too: it does not wait for the sums)
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const int cg = k * G + gl;
      av[k] = zero16<T>();
      if (cg < CG) {
        dv[k] = load16(dy + tk * C + cg * E);
        xv[k] = load16(x + tk * C + cg * E);
        if (add != nullptr) av[k] = load16(add + tk * C + cg * E);
      }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const int cg = k * G + gl;
      if (cg < CG) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const float d = live ? to_f32(dv[k].v[e]) : 0.f;
          const float xh = (to_f32(xv[k].v[e]) - mu) * rs;
          const float g = d * gm[k][e];
          s1 += g;
          s2 += g * xh;
          ag[k][e] += d * xh;
          ab[k][e] += d;
        }
      }
    }
    s1 = group_sum<G>(s1) / (float)C;
    s2 = group_sum<G>(s2) / (float)C;
    if (live) {
      const float kf = dxs != nullptr ? keep[tok / tok_per_sample] : 1.f;
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const int cg = k * G + gl;
        if (cg < CG) {
          Vec16<T> o;
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const float g = to_f32(dv[k].v[e]) * gm[k][e];
            const float xh = (to_f32(xv[k].v[e]) - mu) * rs;
            o.v[e] = from_f32<T>(rs * (g - s1 - xh * s2) + (add ? to_f32(av[k].v[e]) : 0.f));
          }
          store16(dx + tok * C + cg * E, o);
          if (dxs != nullptr) {
            Vec16<T> os;
#pragma unroll
            for (int e = 0; e < E; ++e) os.v[e] = from_f32<T>(to_f32(o.v[e]) * kf + 0.f);
            store16(dxs + tok * C + cg * E, os);
          }
        }
      }
    }
  }
  if (partials == nullptr) return;
#pragma unroll
  for (int k = 0; k < V; ++k) {
    if (k * G >= CG) break;                             // (uniform)
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) {
      red[0][threadIdx.x * E + e] = ag[k][e];
      red[1][threadIdx.x * E + e] = ab[k][e];
    }
    __syncthreads();
    // G*E channels of this vector slot, TPW token groups each
    for (int i = threadIdx.x; i < G * E; i += 256) {
      const int l = i / E, e = i - l * E;
      const int c = (k * G + l) * E + e;
      if (c < C) {
        float t0 = 0.f, t1 = 0.f;
        for (int g2 = 0; g2 < TPW; ++g2) {
          t0 += red[0][(g2 * G + l) * E + e];
          t1 += red[1][(g2 * G + l) * E + e];
        }
        partials[((long)blockIdx.x * 2) * C + c] = t0;
        partials[((long)blockIdx.x * 2 + 1) * C + c] = t1;
      }
    }
  }
}

// Per-channel sums over tokens.  MODE 0: s0 = sum x.  MODE 1: s0 = sum dy*xhat, s1 = sum dy (LayerNorm).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void token_reduce_kernel(const T* __restrict__ a, int lda,
                                                           const T* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, long ntok, int C,
                                                           int CS, int chunk, float* __restrict__ partials) {
  // blockIdx.y selects a slab of CS channels (CS/E <= 256 channel groups per workgroup)
  constexpr int E = Vec16<T>::N;
  __shared__ float red[2][256 * E];
  const int CG = CS / E;
  const int lanes = 256 / CG;
  const int tid = threadIdx.x;
  const int cg = tid % CG, pl = tid / CG;
  const int cbase = blockIdx.y * CS;
  const int c0 = cbase + cg * E;
  float s0[E], s1[E];
#pragma unroll
  for (int e = 0; e < E; ++e) s0[e] = s1[e] = 0.f;
  if (pl < lanes) {
    const long pbeg = (long)blockIdx.x * chunk;
    long pend = pbeg + chunk;
    if (pend > ntok) pend = ntok;
    for (long p = pbeg + pl; p < pend; p += lanes) {
      Vec16<T> av = load16(a + p * lda + c0);
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < E; ++e) s0[e] += to_f32(av.v[e]);
      } else {
        Vec16<T> xv = load16(x + p * C + c0);
        const float mu = mean[p], rs = rstd[p];
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const float d = to_f32(av.v[e]);
          s0[e] += d * (to_f32(xv.v[e]) - mu) * rs;
          s1[e] += d;
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    red[0][tid * E + e] = s0[e];
    red[1][tid * E + e] = s1[e];
  }
  __syncthreads();
  for (int c = tid; c < CS; c += 256) {
    const int g = c / E, e = c - g * E;
    float t0 = 0.f, t1 = 0.f;
    for (int l = 0; l < lanes; ++l) {
      t0 += red[0][(l * CG + g) * E + e];
      t1 += red[1][(l * CG + g) * E + e];
    }
    partials[((long)blockIdx.x * 2) * C + cbase + c] = t0;
    partials[((long)blockIdx.x * 2 + 1) * C + cbase + c] = t1;
  }
}

// One workgroup = 16 channels x 16 row lanes: a wave reads four rows of 64 consecutive bytes, a lane keeps eight rows
// in flight and adds them in row order (double), then the 16 lane sums are added in lane order.
__device__ __forceinline__ void partial_sum2_body(const float* __restrict__ partials, int nparts, int C,
                                                  float* out0, float* out1) {
  __shared__ double red[2][16][16];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s = 0.0, q = 0.0;
  if (c < C) {
    const float* p = partials + c;
    const long row = 2L * C;
    int r = rl;
    for (; r + 16 * 7 < nparts; r += 16 * 8) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[u] = p[(long)(r + 16 * u) * row];
        b[u] = p[(long)(r + 16 * u) * row + C];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s += (double)a[u];
        q += (double)b[u];
      }
    }
    for (; r < nparts; r += 16) {
      s += (double)p[(long)r * row];
      q += (double)p[(long)r * row + C];
    }
  }
  red[0][rl][cl] = s;
  red[1][rl][cl] = q;
  __syncthreads();
  if (rl == 0 && c < C) {
    double ss = 0.0, qq = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      ss += red[0][r][cl];
      qq += red[1][r][cl];
    }
    if (out0) out0[c] = (float)ss;
    if (out1) out1[c] = (float)qq;
  }
}

__global__ __launch_bounds__(256) void partial_sum2_kernel(const float* __restrict__ partials, int nparts, int C,
                                                           float* out0, float* out1) {
  partial_sum2_body(partials, nparts, C, out0, out1);
}

// Many of those sums in one launch (the parameter gradients of every LayerNorm of a backward pass, taken at its
// end): blockIdx.y = job, five int64 per job = partials, rows, C, out0, out1 (0 = not wanted).  Same arithmetic,
// same bits as one launch per job.
__global__ __launch_bounds__(256) void partial_sum2_batched_kernel(const long* __restrict__ jobs) {
  const long* j = jobs + 5L * blockIdx.y;
  const int C = (int)j[2];
  if ((int)blockIdx.x * 16 >= C) return;
  partial_sum2_body(reinterpret_cast<const float*>(j[0]), (int)j[1], C, reinterpret_cast<float*>(j[3]),
                    reinterpret_cast<float*>(j[4]));
}

// ------------------------------------------------------------------ elementwise
template <typename T>
__global__ void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long nvec) {
  constexpr int E = Vec16<T>::N;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvec) return;
  Vec16<T> v = load16(x + i * E), o;
#pragma unroll
  for (int e = 0; e < E; ++e) o.v[e] = from_f32<T>(gelu_f(to_f32(v.v[e])));
  store16(y + i * E, o);
}

template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx, long nvec) {
  constexpr int E = Vec16<T>::N;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvec) return;
  Vec16<T> d = load16(dy + i * E), v = load16(x + i * E), o;
#pragma unroll
  for (int e = 0; e < E; ++e) o.v[e] = from_f32<T>(to_f32(d.v[e]) * gelu_grad(to_f32(v.v[e])));
  store16(dx + i * E, o);
}

// y = x * keep[b]  (+ add)
template <typename T>
__global__ void scale_sample_kernel(const T* __restrict__ x, const float* __restrict__ keep,
                                    const T* __restrict__ add, T* __restrict__ y, long nvec, long vec_per_sample) {
  constexpr int E = Vec16<T>::N;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvec) return;
  const float k = keep ? keep[i / vec_per_sample] : 1.f;
  Vec16<T> v = load16(x + i * E), o;
  Vec16<T> a = zero16<T>();
  if (add) a = load16(add + i * E);
#pragma unroll
  for (int e = 0; e < E; ++e) o.v[e] = from_f32<T>(to_f32(v.v[e]) * k + (add ? to_f32(a.v[e]) : 0.f));
  store16(y + i * E, o);
}

// ------------------------------------------------------------------ gathers
// tokens[b,yp,xp, c*ps*ps + ky*ps + kx] = x[b,c,ps*yp+ky,ps*xp+kx]   (x NCHW fp32)
template <typename T>
__global__ void patchify_kernel(const float* __restrict__ x, T* __restrict__ tok, int B, int Cin, int H, int W,
                                int ps, int Hp, int Wp, int ld) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * Hp * Wp * ld;
  if (idx >= total) return;
  const int k = idx % ld;
  long p = idx / ld;
  const int xp = p % Wp;
  p /= Wp;
  const int yp = p % Hp;
  const int b = p / Hp;
  float v = 0.f;
  if (k < Cin * ps * ps) {
    const int c = k / (ps * ps), r = k - c * ps * ps;
    const int ky = r / ps, kx = r - ky * ps;
    v = x[(((long)b * Cin + c) * H + ps * yp + ky) * W + ps * xp + kx];
  }
  tok[idx] = from_f32<T>(v);
}

// The same for 4x4 patches (W % 4 == 0, ld % 4 == 0, 16-byte aligned images): one thread per (token, four consecutive
// k) = one 16-byte read of a patch row and one 8- or 16-byte write; a wave writes the 64-wide tokens of four patches
// as one run.
template <typename T>
__global__ void patchify4_kernel(const float* __restrict__ x, T* __restrict__ tok, int B, int Cin, int H, int W,
                                 int Hp, int Wp, int ld) {
  const int kq = ld / 4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * Hp * Wp * kq) return;
  const int k = (int)(idx % kq) * 4;
  long p = idx / kq;
  const long token = p;
  const int xp = p % Wp;
  p /= Wp;
  const int yp = p % Hp;
  const int b = p / Hp;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (k < Cin * 16) {
    const int c = k >> 4, ky = (k >> 2) & 3;
    v = *reinterpret_cast<const f32x4*>(x + (((long)b * Cin + c) * H + 4 * yp + ky) * W + 4 * xp);
  }
  T* d = tok + token * ld + k;
  if constexpr (sizeof(T) == 2) {
    union {
      T h[4];
      uint2 u;
    } o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o.h[i] = from_f32<T>(v[i]);
    *reinterpret_cast<uint2*>(d) = o.u;
  } else {
    *reinterpret_cast<f32x4*>(d) = v;
  }
}

template <typename T>
__global__ void unpatchify_kernel(const T* __restrict__ dtok, float* __restrict__ dx, int B, int Cin, int H, int W,
                                  int ps, int Hp, int Wp, int ld) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * Cin * H * W;
  if (idx >= total) return;
  const int xx = idx % W;
  long p = idx / W;
  const int yy = p % H;
  p /= H;
  const int c = p % Cin;
  const int b = p / Cin;
  float v = 0.f;
  const int yp = yy / ps, xp = xx / ps;
  if (yp < Hp && xp < Wp)
    v = to_f32(dtok[(((long)b * Hp + yp) * Wp + xp) * ld + c * ps * ps + (yy % ps) * ps + (xx % ps)]);
  dx[idx] = v;
}

// out[b,y2,x2, q*C + c] = x[b, 2*y2 + (q&1), 2*x2 + (q>>1), c]  (zero outside: odd H/W padding)
template <typename T>
__global__ void merge_gather_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H, int W, int C) {
  constexpr int E = Vec16<T>::N;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2, CG = C / E;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H2 * W2 * 4 * CG) return;
  const int cg = idx % CG;
  long p = idx / CG;
  const int q = p % 4;
  p /= 4;
  const int x2 = p % W2;
  p /= W2;
  const int y2 = p % H2;
  const int b = p / H2;
  const int yy = 2 * y2 + (q & 1), xx = 2 * x2 + (q >> 1);
  Vec16<T> v = zero16<T>();
  if (yy < H && xx < W) v = load16(x + (((long)b * H + yy) * W + xx) * C + cg * E);
  store16(out + ((((long)b * H2 + y2) * W2 + x2) * 4 + q) * C + cg * E, v);
}

template <typename T>
__global__ void merge_scatter_kernel(const T* __restrict__ dout, T* __restrict__ dx, int B, int H, int W, int C) {
  constexpr int E = Vec16<T>::N;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2, CG = C / E;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H * W * CG) return;
  const int cg = idx % CG;
  long p = idx / CG;
  const int xx = p % W;
  p /= W;
  const int yy = p % H;
  const int b = p / H;
  const int q = (xx & 1) * 2 + (yy & 1);
  store16(dx + (((long)b * H + yy) * W + xx) * C + cg * E,
          load16(dout + ((((long)b * H2 + (yy >> 1)) * W2 + (xx >> 1)) * 4 + q) * C + cg * E));
}

// ConvTranspose2d(k=2,s=2) after its GEMM: tmp[b,y,x,(ky*2+kx)*Co+co] (+bias) -> cat[b,2y+ky,2x+kx,co]
template <typename T, bool BWD>
__global__ void pixel_shuffle_kernel(T* __restrict__ tmp, T* __restrict__ cat, int ldcat,
                                     const float* __restrict__ bias, int B, int h, int w, int Co) {
  constexpr int E = Vec16<T>::N;
  const int CG = Co / E;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * h * w * 4 * CG) return;
  const int cg = idx % CG;
  long p = idx / CG;
  const int k = p % 4;
  p /= 4;
  const int x = p % w;
  p /= w;
  const int y = p % h;
  const int b = p / h;
  T* tp = tmp + ((((long)b * h + y) * w + x) * 4 + k) * Co + cg * E;
  T* cp = cat + (((long)b * 2 * h + 2 * y + (k >> 1)) * 2 * w + 2 * x + (k & 1)) * ldcat + cg * E;
  if (BWD) {
    store16(tp, load16(cp));
  } else {
    Vec16<T> v = load16(tp);
    if (bias != nullptr) {
#pragma unroll
      for (int e = 0; e < E; ++e) v.v[e] = from_f32<T>(to_f32(v.v[e]) + bias[cg * E + e]);
    }
    store16(cp, v);
  }
}

// ------------------------------------------------------------------ window attention
template <typename T> struct AttnLds {
  static constexpr int R32 = HD * (int)sizeof(T) + 16;     // row of 32 elements + pad
  static constexpr int R64 = NTOK * (int)sizeof(T) + 16;   // row of 64 elements + pad
  static constexpr int C32 = HD * (int)sizeof(T) / 64;     // 64-byte chunks per row
  static constexpr int C64 = NTOK * (int)sizeof(T) / 64;
};

struct AttnGeom {
  int B, He, We, C, heads, shift, use_mask;
  float scale;
};

__device__ __forceinline__ int region_of(int s, int extent) {
  return s < extent - WS ? 0 : (s < extent - WS / 2 ? 1 : 2);
}

// Four consecutive channels of one token from a C-layout accumulator register (rows = channels)
template <typename T> __device__ __forceinline__ void store4(T* p, const f32x4& v, float scale = 1.0f) {
  if constexpr (sizeof(T) == 2) {
    typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
    u16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = __builtin_bit_cast(unsigned short, from_f32<T>(v[r] * scale));
    *reinterpret_cast<u16x4*>(p) = o;
  } else {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale};
  }
}

// acc[mt][nt] += A(rows mt*16.., K) * B(rows nt*16.., K)^T with both operands row-major in LDS
template <typename T, int MT, int NT, int CHUNKS>
__device__ __forceinline__ void lds_gemm(const char* A, int lda, const char* Bm, int ldb, int lane,
                                         f32x4 (&acc)[MT][NT]) {
  using M = MMA<T>;
#pragma unroll
  for (int ch = 0; ch < CHUNKS; ++ch)
#pragma unroll
    for (int ks = 0; ks < M::KSTEPS; ++ks) {
      typename M::frag af[MT], bf_[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) af[m] = M::load(A + (m * 16 + (lane & 15)) * lda + ch * 64, ks, lane);
#pragma unroll
      for (int n = 0; n < NT; ++n) bf_[n] = M::load(Bm + (n * 16 + (lane & 15)) * ldb + ch * 64, ks, lane);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = M::mma(af[m], bf_[n], acc[m][n]);
    }
}

// softmax(scale*S + bias + mask) in place on the 64x64 accumulator tile set (C layout)
__device__ __forceinline__ void window_softmax(f32x4 (&s)[4][4], const float* tbl, const AttnGeom& g, int wy, int wx,
                                               int lane) {
  const int cl = lane & 15, pg = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = mt * 16 + 4 * pg + r;
      const int yi = i >> 3, xi = i & 7;
      const int idi = 3 * region_of(wy * WS + yi, g.He) + region_of(wx * WS + xi, g.We);
      float mx = -INFINITY;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int j = nt * 16 + cl;
        const int yj = j >> 3, xj = j & 7;
        float v = s[mt][nt][r] * g.scale + tbl[(yi - yj + WS - 1) * (2 * WS - 1) + (xi - xj + WS - 1)];
        if (g.use_mask) {
          const int idj = 3 * region_of(wy * WS + yj, g.He) + region_of(wx * WS + xj, g.We);
          if (idi != idj) v += -100.0f;
        }
        s[mt][nt][r] = v;
        mx = fmaxf(mx, v);
      }
      mx = row16_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float e = __expf(s[mt][nt][r] - mx);
        s[mt][nt][r] = e;
        sum += e;
      }
      sum = row16_sum(sum);
      const float inv = 1.0f / sum;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) s[mt][nt][r] *= inv;
    }
}

__device__ __forceinline__ long window_token(const AttnGeom& g, int b, int wy, int wx, int n) {
  int ty = wy * WS + (n >> 3) + g.shift, tx = wx * WS + (n & 7) + g.shift;
  if (ty >= g.He) ty -= g.He;
  if (tx >= g.We) tx -= g.We;
  return ((long)b * g.He + ty) * g.We + tx;
}

template <typename T>
__device__ __forceinline__ void stage_rows(const T* src, char* rows, int ldr, char* tr, int ldt, int n) {
  // this lane's token: 32 elements -> row n of `rows`, and column n of the transposed image `tr`
  constexpr int E = Vec16<T>::N;
#pragma unroll
  for (int p = 0; p < HD / E; ++p) {
    Vec16<T> v = load16(src + p * E);
    if (rows) store16(reinterpret_cast<T*>(rows + n * ldr) + p * E, v);
    if (tr) {
#pragma unroll
      for (int e = 0; e < E; ++e) reinterpret_cast<T*>(tr + (p * E + e) * ldt)[n] = v.v[e];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(64) void window_attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                             const float* __restrict__ table, AttnGeom g) {
  using L = AttnLds<T>;
  // (one wave per workgroup: P is written over Q and K, which are dead once S is in registers)
  static_assert(NTOK * L::R64 <= 2 * NTOK * L::R32, "P fits the storage of Q and K");
  __shared__ __attribute__((aligned(16))) char QK[2 * NTOK * L::R32];
  __shared__ __attribute__((aligned(16))) char Vt[HD * L::R64];
  __shared__ float tbl[(2 * WS - 1) * (2 * WS - 1)];
  char* const Qs = QK;
  char* const Ks = QK + NTOK * L::R32;
  char* const Ps = QK;
  const int lane = threadIdx.x;
  const int nwx = g.We / WS, nwy = g.He / WS;
  int t = blockIdx.x;
  const int head = t % g.heads;
  t /= g.heads;
  const int wx = t % nwx;
  t /= nwx;
  const int wy = t % nwy;
  const int b = t / nwy;
  for (int i = lane; i < 225; i += 64) tbl[i] = table[i * g.heads + head];
  const long tok = window_token(g, b, wy, wx, lane);
  const T* base = qkv + tok * 3 * g.C + head * HD;
  stage_rows<T>(base, Qs, L::R32, nullptr, 0, lane);
  stage_rows<T>(base + g.C, Ks, L::R32, nullptr, 0, lane);
  stage_rows<T>(base + 2 * g.C, nullptr, 0, Vt, L::R64, lane);
  __syncthreads();
  f32x4 s[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) s[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  lds_gemm<T, 4, 4, L::C32>(Qs, L::R32, Ks, L::R32, lane, s);
  window_softmax(s, tbl, g, wy, wx, lane);
  const int cl = lane & 15, pg = lane >> 4;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        reinterpret_cast<T*>(Ps + (m * 16 + 4 * pg + r) * L::R64)[n * 16 + cl] = from_f32<T>(s[m][n][r]);
  __syncthreads();
  // O^T[d][query] = sum_key V^T[d][key] P[query][key]: in the C layout a lane then holds four
  // consecutive channels of one token, i.e. one 8-byte (bf16) store instead of four 2-byte ones
  f32x4 o[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) o[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  lds_gemm<T, 2, 4, L::C64>(Vt, L::R64, Ps, L::R64, lane, o);
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const long tk = window_token(g, b, wy, wx, n * 16 + cl);
#pragma unroll
    for (int m = 0; m < 2; ++m) store4<T>(out + tk * g.C + head * HD + m * 16 + 4 * pg, o[m][n]);
  }
}

// Backward of one (window, head): recomputes P, then dV = P^T dO, dP = dO V^T,
// dS = P*(dP - rowsum(dP*P)), dQ = scale*dS K, dK = scale*dS^T Q, and the window's contribution to
// the relative-position-bias table gradient (225 bins, summed in a fixed order).
template <typename T>
__global__ __launch_bounds__(64) void window_attn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                             T* __restrict__ dqkv, const float* __restrict__ table,
                                                             float* __restrict__ dtable_partials, AttnGeom g) {
  using L = AttnLds<T>;
  // One wave per workgroup, so the LDS footprint IS the occupancy: buffers share storage by
  // lifetime.  Q, K (rows) are dead once S = QK^T is in registers: P^T, later dS, lives there.
  // V, dO (rows) are dead after dP = dO V^T: dS^T lives there.  (One wave: LDS operations complete
  // in issue order, so reuse needs no more than the barriers already between the phases.)
  constexpr int ROWS2 = 2 * NTOK * L::R32;
  static_assert(NTOK * L::R64 <= ROWS2, "a 64x64 tile fits the storage of two 64x32 operands");
  __shared__ __attribute__((aligned(16))) char bufA[ROWS2];
  __shared__ __attribute__((aligned(16))) char bufB[ROWS2];
  __shared__ __attribute__((aligned(16))) char Qt[HD * L::R64];
  __shared__ __attribute__((aligned(16))) char Kt[HD * L::R64];
  __shared__ __attribute__((aligned(16))) char Dt[HD * L::R64];
  __shared__ float tbl[(2 * WS - 1) * (2 * WS - 1)];
  char* const Qs = bufA;
  char* const Ks = bufA + NTOK * L::R32;
  char* const PA = bufA;                       // P^T, later dS
  char* const Vs = bufB;
  char* const Ds = bufB + NTOK * L::R32;
  char* const PB = bufB;                       // dS^T
  const int lane = threadIdx.x;
  const int cl = lane & 15, pg = lane >> 4;
  const int nwx = g.We / WS, nwy = g.He / WS;
  int t = blockIdx.x;
  const int head = t % g.heads;
  t /= g.heads;
  const int wx = t % nwx;
  t /= nwx;
  const int wy = t % nwy;
  const int b = t / nwy;
  for (int i = lane; i < 225; i += 64) tbl[i] = table[i * g.heads + head];
  const long tok = window_token(g, b, wy, wx, lane);
  const T* base = qkv + tok * 3 * g.C + head * HD;
  stage_rows<T>(base, Qs, L::R32, Qt, L::R64, lane);
  stage_rows<T>(base + g.C, Ks, L::R32, Kt, L::R64, lane);
  stage_rows<T>(base + 2 * g.C, Vs, L::R32, nullptr, 0, lane);
  stage_rows<T>(dout + tok * g.C + head * HD, Ds, L::R32, Dt, L::R64, lane);
  __syncthreads();
  f32x4 p[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) p[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  lds_gemm<T, 4, 4, L::C32>(Qs, L::R32, Ks, L::R32, lane, p);
  window_softmax(p, tbl, g, wy, wx, lane);
  // P^T -> PA
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        reinterpret_cast<T*>(PA + (n * 16 + cl) * L::R64)[m * 16 + 4 * pg + r] = from_f32<T>(p[m][n][r]);
  __syncthreads();
  {   // dV^T[d][key] = sum_query dO^T[d][query] * P^T[key][query]  (a lane gets four channels of one token)
    f32x4 dv[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) dv[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    lds_gemm<T, 2, 4, L::C64>(Dt, L::R64, PA, L::R64, lane, dv);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const long tk = window_token(g, b, wy, wx, n * 16 + cl);
#pragma unroll
      for (int m = 0; m < 2; ++m) store4<T>(dqkv + tk * 3 * g.C + 2 * g.C + head * HD + m * 16 + 4 * pg, dv[m][n]);
    }
  }
  // dP[query][key] = sum_d dO[query][d] * V[key][d]
  f32x4 dp[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) dp[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  lds_gemm<T, 4, 4, L::C32>(Ds, L::R32, Vs, L::R32, lane, dp);
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float rs = 0.f;
#pragma unroll
      for (int n = 0; n < 4; ++n) rs += dp[m][n][r] * p[m][n][r];
      rs = row16_sum(rs);
#pragma unroll
      for (int n = 0; n < 4; ++n) dp[m][n][r] = p[m][n][r] * (dp[m][n][r] - rs);   // dS
    }
  __syncthreads();   // PA (P^T) no longer read
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = m * 16 + 4 * pg + r, j = n * 16 + cl;
        const float v = dp[m][n][r];
        reinterpret_cast<T*>(PA + i * L::R64)[j] = from_f32<T>(v);
        reinterpret_cast<T*>(PB + j * L::R64)[i] = from_f32<T>(v);
      }
  __syncthreads();
  {   // dQ^T[d][query] = scale * sum_key K^T[d][key] * dS[query][key];  dK^T[d][key] = scale * sum_query Q^T * dS^T
    f32x4 dq[2][4], dk[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) dq[m][n] = dk[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    lds_gemm<T, 2, 4, L::C64>(Kt, L::R64, PA, L::R64, lane, dq);
    lds_gemm<T, 2, 4, L::C64>(Qt, L::R64, PB, L::R64, lane, dk);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const long tk = window_token(g, b, wy, wx, n * 16 + cl);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        store4<T>(dqkv + tk * 3 * g.C + head * HD + m * 16 + 4 * pg, dq[m][n], g.scale);
        store4<T>(dqkv + tk * 3 * g.C + g.C + head * HD + m * 16 + 4 * pg, dk[m][n], g.scale);
      }
    }
  }
  // Bias-table gradient of this (window, head): bin (dy,dx) sums dS[i][j] (as stored for the two
  // products above, i.e. rounded to T) over yi-yj = dy, xi-xj = dx.  Two steps, fixed order:
  // lane (yi,yj) reads its 8x8 block of dS (rows 8yi.., columns 8yj..: 16-byte reads) and sums the
  // block's 15 diagonals in registers; then each bin adds the up to eight blocks with yi-yj = dy.
  {
    constexpr int E = Vec16<T>::N;
    const int yi = lane >> 3, yj = lane & 7;
    float diag[2 * WS - 1];
#pragma unroll
    for (int k = 0; k < 2 * WS - 1; ++k) diag[k] = 0.f;
#pragma unroll
    for (int xi = 0; xi < WS; ++xi) {
      const T* rowp = reinterpret_cast<const T*>(PA + (yi * WS + xi) * L::R64) + yj * WS;
#pragma unroll
      for (int part = 0; part < WS / E; ++part) {
        const Vec16<T> v = load16(rowp + part * E);
#pragma unroll
        for (int e = 0; e < E; ++e) diag[xi - (part * E + e) + WS - 1] += to_f32(v.v[e]);
      }
    }
    __syncthreads();                 // everyone has read dS: the dS^T tile is free to hold the block sums
    float* blk = reinterpret_cast<float*>(PB);                 // [yi][yj][16]
    static_assert(NTOK * 16 * 4 <= NTOK * L::R64, "block sums fit the dS^T tile");
#pragma unroll
    for (int k = 0; k < 2 * WS - 1; ++k) blk[lane * 16 + k] = diag[k];
    __syncthreads();
    for (int bin = lane; bin < 225; bin += 64) {
      const int dy = bin / 15 - 7, dxk = bin % 15;
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < WS; ++j) {
        const int i = j + dy;
        const bool ok = i >= 0 && i < WS;
        const float v = blk[((ok ? i : 0) * WS + j) * 16 + dxk];
        acc += ok ? v : 0.f;
      }
      dtable_partials[(long)blockIdx.x * 225 + bin] = acc;
    }
  }
}

// dtable[bin][head] = sum over (b, window) of the partials, fixed order.  One workgroup = 16 bins x 16 window lanes
// (a wave reads 64 consecutive bytes of four windows); a lane keeps eight windows in flight and adds them in window
// order, then the 16 lane sums are added in lane order.
constexpr int DT_BIN_GROUPS = (225 + 15) / 16;
__device__ __forceinline__ void dtable_reduce_body(const float* __restrict__ partials, int nwin, int heads,
                                                   float* __restrict__ dtable) {
  __shared__ float red[16][16];
  const int bl = threadIdx.x & 15, wl = threadIdx.x >> 4;
  const int bin = blockIdx.x * 16 + bl, head = blockIdx.y;
  float s = 0.f;
  if (bin < 225) {
    const float* p = partials + (long)head * 225 + bin;
    const long stride = (long)heads * 225;
    int w = wl;
    for (; w + 16 * 7 < nwin; w += 16 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(long)(w + 16 * u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; w < nwin; w += 16) s += p[(long)w * stride];
  }
  red[wl][bl] = s;
  __syncthreads();
  if (wl == 0 && bin < 225) {
    float t = red[0][bl];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += red[k][bl];
    dtable[bin * heads + head] = t;
  }
}
__global__ __launch_bounds__(256) void dtable_reduce_kernel(const float* __restrict__ partials, int nwin, int heads,
                                                            float* __restrict__ dtable) {
  dtable_reduce_body(partials, nwin, heads, dtable);
}
// The same for every attention layer of a backward pass in one launch: blockIdx.z = job, four int64 per job =
// partials, windows (B * nW), heads, dtable.
__global__ __launch_bounds__(256) void dtable_reduce_batched_kernel(const long* __restrict__ jobs) {
  const long* j = jobs + 4L * blockIdx.z;
  const int heads = (int)j[2];
  if ((int)blockIdx.y >= heads) return;
  dtable_reduce_body(reinterpret_cast<const float*>(j[0]), (int)j[1], heads, reinterpret_cast<float*>(j[3]));
}

// channels per workgroup slab: the largest divisor split of C with at most 256 vector groups
inline int slab_channels(int C, int E) {
  int n = 1;
  while ((C / n) / E > 256 || C % n != 0 || (C / n) % E != 0) ++n;
  return C / n;
}

// lanes per token for the grouped LayerNorm kernels (0: too many vectors per token for them)
inline int ln_group(int cg) {
  for (int g = 16; g <= 64; g *= 2)
    if (cg <= g * (g < 64 ? 1 : LN_MAXV)) return g;
  return 0;
}

inline int tok_blocks(long ntok) {
  long n = (ntok + 15) / 16;       // at least 16 tokens per workgroup, at most 1024 workgroups
  if (n > 1024) n = 1024;
  if (n < 1) n = 1;
  return (int)n;
}

}  // namespace

#define DISPATCH_T(dtype, ...)                       \
  do {                                               \
    if ((dtype) == UNETK_BF16) {                     \
      using T = bf16;                                \
      __VA_ARGS__;                                   \
    } else {                                         \
      using T = float;                               \
      __VA_ARGS__;                                   \
    }                                                \
  } while (0)
#define CHECK_DT(name, dtype) \
  UNETK_CHECK_ARG((dtype) == UNETK_F32 || (dtype) == UNETK_BF16, name ": bad dtype %d", (dtype))
#define VEC_OF(dtype) ((dtype) == UNETK_BF16 ? 8 : 4)

extern "C" {

int unetk_layernorm_fwd(const void* x, void* y, const float* gamma, const float* beta, float* mean, float* rstd,
                        int64_t ntok, int C, float eps, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(x && y && gamma && beta && mean && rstd && ntok > 0, "layernorm_fwd: bad arguments");
  CHECK_DT("layernorm_fwd", dtype);
  UNETK_CHECK_ARG(C % VEC_OF(dtype) == 0, "layernorm_fwd: C=%d not a multiple of %d", C, VEC_OF(dtype));
  const int G = ln_group(C / VEC_OF(dtype));
  const int V = G ? cdiv(C / VEC_OF(dtype), G) : 0;
#define LN_FWD(G_, V_)                                                                                          \
  DISPATCH_T(dtype, hipLaunchKernelGGL((layernorm_fwd_group_kernel<T, G_, V_>), dim3(cdiv(ntok, 256 / G_)), dim3(256), \
                                       0, stream, (const T*)x, (T*)y, gamma, beta, mean, rstd, (long)ntok, C, eps))
  if (G == 16) LN_FWD(16, 1);
  else if (G == 32) LN_FWD(32, 1);
  else if (G == 64 && V == 1) LN_FWD(64, 1);
  else if (G == 64 && V == 2) LN_FWD(64, 2);
  else if (G == 64 && V == 3) LN_FWD(64, 3);
  else if (G == 64) LN_FWD(64, 6);
  else   // more vectors per token than the grouped kernel keeps in registers: one wave per token
    DISPATCH_T(dtype, hipLaunchKernelGGL(layernorm_fwd_kernel<T>, dim3(cdiv(ntok, 4)), dim3(256), 0, stream,
                                         (const T*)x, (T*)y, gamma, beta, mean, rstd, (long)ntok, C, eps));
#undef LN_FWD
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_token_reduce_num_partials(int64_t ntok) { return tok_blocks(ntok); }

/* Rows of partials that unetk_layernorm_bwd_partials writes for this shape; 0 when the shape takes the two-pass
 * form, which has no such split. */
int unetk_layernorm_bwd_rows(int64_t ntok, int C, int dtype) {
  if ((dtype != UNETK_F32 && dtype != UNETK_BF16) || ntok <= 0 || C <= 0 || C % VEC_OF(dtype) != 0) return 0;
  if (ln_group(C / VEC_OF(dtype)) == 0) return 0;
  const int nb = tok_blocks(ntok);
  return (int)cdiv(ntok, cdiv(ntok, nb));
}

static int layernorm_bwd_impl(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                              const void* add, void* dx, float* dgamma, float* dbeta, float* partials, int64_t ntok,
                              int C, int dtype, bool partials_only, const float* keep, int64_t tok_per_sample,
                              void* dx_scaled, hipStream_t stream);

/* dx = LN'(dy) + add;  dgamma, dbeta = sums over tokens.  partials: [num_partials][2][C] */
int unetk_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                        const void* add, void* dx, float* dgamma, float* dbeta, float* partials, int64_t ntok,
                        int C, int dtype, hipStream_t stream) {
  return layernorm_bwd_impl(dy, x, gamma, mean, rstd, add, dx, dgamma, dbeta, partials, ntok, C, dtype, false, nullptr, 1,
                            nullptr, stream);
}

/* The first half only: dx, and unetk_layernorm_bwd_rows(ntok, C, dtype) rows of partials [row][2][C] whose column
 * sums are dgamma and dbeta (unetk_partial_sums_batched adds them up, for many layers in one launch). */
int unetk_layernorm_bwd_partials(const void* dy, const void* x, const float* gamma, const float* mean,
                                 const float* rstd, const void* add, void* dx, float* partials, int64_t ntok, int C,
                                 int dtype, const float* keep, int64_t tok_per_sample, void* dx_scaled,
                                 hipStream_t stream) {
  UNETK_CHECK_ARG(unetk_layernorm_bwd_rows(ntok, C, dtype) > 0, "layernorm_bwd_partials: no one-pass form for C=%d", C);
  UNETK_CHECK_ARG(dx_scaled == nullptr || (keep != nullptr && tok_per_sample > 0 && ntok % tok_per_sample == 0),
                  "layernorm_bwd_partials: the scaled copy needs the per-sample factors and whole samples");
  return layernorm_bwd_impl(dy, x, gamma, mean, rstd, add, dx, nullptr, nullptr, partials, ntok, C, dtype, true, keep,
                            tok_per_sample > 0 ? tok_per_sample : 1, dx_scaled, stream);
}

int unetk_partial_sums_batched(const int64_t* jobs, int njobs, int max_C, hipStream_t stream) {
  UNETK_CHECK_ARG(jobs && njobs > 0 && njobs <= 65535 && max_C > 0, "partial_sums_batched: bad arguments");
  hipLaunchKernelGGL(partial_sum2_batched_kernel, dim3(cdiv(max_C, 16), njobs), dim3(256), 0, stream,
                     reinterpret_cast<const long*>(jobs));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

static int layernorm_bwd_impl(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                              const void* add, void* dx, float* dgamma, float* dbeta, float* partials, int64_t ntok,
                              int C, int dtype, bool partials_only, const float* keep, int64_t tok_per_sample,
                              void* dx_scaled, hipStream_t stream) {
  UNETK_CHECK_ARG(dy && x && gamma && mean && rstd && dx && partials && ntok > 0, "layernorm_bwd: bad arguments");
  CHECK_DT("layernorm_bwd", dtype);
  UNETK_CHECK_ARG(C % VEC_OF(dtype) == 0, "layernorm_bwd: unsupported C=%d", C);
  const int G = ln_group(C / VEC_OF(dtype));
  const int nb = tok_blocks(ntok);
  const int chunk = cdiv(ntok, nb);
  const bool want_sums = dgamma || dbeta;
  if (G != 0) {
    // dx and the workgroups' dgamma / dbeta partials in one pass over dy and x
    const int V = cdiv(C / VEC_OF(dtype), G);
#define LN_BWD(G_, V_)                                                                                         \
  DISPATCH_T(dtype, hipLaunchKernelGGL((layernorm_bwd_group_kernel<T, G_, V_>), dim3(cdiv(ntok, chunk)), dim3(256), \
                                       0, stream, (const T*)dy, (const T*)x, gamma, mean, rstd, (const T*)add,   \
                                       (T*)dx, (long)ntok, C, chunk, (want_sums || partials_only) ? partials : nullptr, \
                                       keep, (long)tok_per_sample, (T*)dx_scaled))
    if (G == 16) LN_BWD(16, 1);
    else if (G == 32) LN_BWD(32, 1);
    else if (V == 1) LN_BWD(64, 1);
    else if (V == 2) LN_BWD(64, 2);
    else if (V == 3) LN_BWD(64, 3);
    else LN_BWD(64, 6);
#undef LN_BWD
    UNETK_CHECK_LAUNCH();
    if (want_sums) {
      hipLaunchKernelGGL(partial_sum2_kernel, dim3(cdiv(C, 16)), dim3(256), 0, stream, partials, cdiv(ntok, chunk), C,
                         dgamma, dbeta);
      UNETK_CHECK_LAUNCH();
    }
    return UNETK_OK;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(layernorm_bwd_kernel<T>, dim3(cdiv(ntok, 4)), dim3(256), 0, stream,
                                       (const T*)dy, (const T*)x, gamma, mean, rstd, (const T*)add, (T*)dx,
                                       (long)ntok, C));
  UNETK_CHECK_LAUNCH();
  if (want_sums) {
    const int CS = slab_channels(C, VEC_OF(dtype));
    DISPATCH_T(dtype, hipLaunchKernelGGL((token_reduce_kernel<T, 1>), dim3(nb, C / CS), dim3(256), 0, stream,
                                         (const T*)dy, C, (const T*)x, mean, rstd, (long)ntok, C, CS, chunk,
                                         partials));
    UNETK_CHECK_LAUNCH();
    hipLaunchKernelGGL(partial_sum2_kernel, dim3(cdiv(C, 16)), dim3(256), 0, stream, partials, nb, C, dgamma, dbeta);
    UNETK_CHECK_LAUNCH();
  }
  return UNETK_OK;
}

/* out[c] = sum over tokens of x[tok][c]  (bias gradients) */
int unetk_channel_sum(const void* x, int ld, int64_t ntok, int C, float* out, float* partials, int dtype,
                      hipStream_t stream) {
  UNETK_CHECK_ARG(x && out && partials && ntok > 0, "channel_sum: bad arguments");
  CHECK_DT("channel_sum", dtype);
  UNETK_CHECK_ARG(C % VEC_OF(dtype) == 0 && ld % VEC_OF(dtype) == 0, "channel_sum: unsupported C=%d ld=%d", C, ld);
  const int nb = tok_blocks(ntok);
  const int chunk = cdiv(ntok, nb);
  const int CS = slab_channels(C, VEC_OF(dtype));
  DISPATCH_T(dtype, hipLaunchKernelGGL((token_reduce_kernel<T, 0>), dim3(nb, C / CS), dim3(256), 0, stream,
                                       (const T*)x, ld, (const T*)nullptr, (const float*)nullptr,
                                       (const float*)nullptr, (long)ntok, C, CS, chunk, partials));
  UNETK_CHECK_LAUNCH();
  hipLaunchKernelGGL(partial_sum2_kernel, dim3(cdiv(C, 16)), dim3(256), 0, stream, partials, nb, C, out,
                     (float*)nullptr);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_gelu_fwd(const void* x, void* y, int64_t n, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(x && y && n > 0, "gelu_fwd: bad arguments");
  CHECK_DT("gelu_fwd", dtype);
  UNETK_CHECK_ARG(n % VEC_OF(dtype) == 0, "gelu_fwd: length not a multiple of the vector width");
  const long nv = n / VEC_OF(dtype);
  DISPATCH_T(dtype, hipLaunchKernelGGL(gelu_fwd_kernel<T>, dim3(cdiv(nv, 256)), dim3(256), 0, stream, (const T*)x,
                                       (T*)y, nv));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(dy && x && dx && n > 0, "gelu_bwd: bad arguments");
  CHECK_DT("gelu_bwd", dtype);
  UNETK_CHECK_ARG(n % VEC_OF(dtype) == 0, "gelu_bwd: length not a multiple of the vector width");
  const long nv = n / VEC_OF(dtype);
  DISPATCH_T(dtype, hipLaunchKernelGGL(gelu_bwd_kernel<T>, dim3(cdiv(nv, 256)), dim3(256), 0, stream, (const T*)dy,
                                       (const T*)x, (T*)dx, nv));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

/* y = x * keep[b] + add   (keep, add nullable) */
int unetk_scale_sample(const void* x, const float* keep, const void* add, void* y, int B, int64_t per_sample,
                       int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(x && y && B > 0 && per_sample > 0, "scale_sample: bad arguments");
  CHECK_DT("scale_sample", dtype);
  UNETK_CHECK_ARG(per_sample % VEC_OF(dtype) == 0, "scale_sample: sample size not a multiple of the vector width");
  const long vps = per_sample / VEC_OF(dtype);
  const long nv = vps * B;
  DISPATCH_T(dtype, hipLaunchKernelGGL(scale_sample_kernel<T>, dim3(cdiv(nv, 256)), dim3(256), 0, stream,
                                       (const T*)x, keep, (const T*)add, (T*)y, nv, vps));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_patchify(const float* x, void* tok, int B, int Cin, int H, int W, int ps, int ld, int dtype,
                   hipStream_t stream) {
  UNETK_CHECK_ARG(x && tok && ps > 0 && H >= ps && W >= ps && ld >= Cin * ps * ps, "patchify: bad arguments");
  CHECK_DT("patchify", dtype);
  const int Hp = H / ps, Wp = W / ps;
  const long total = (long)B * Hp * Wp * ld;
  if (ps == 4 && W % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)tok % 16) == 0) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(patchify4_kernel<T>, dim3(cdiv(total / 4, 256)), dim3(256), 0, stream, x,
                                         (T*)tok, B, Cin, H, W, Hp, Wp, ld));
    UNETK_CHECK_LAUNCH();
    return UNETK_OK;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(patchify_kernel<T>, dim3(cdiv(total, 256)), dim3(256), 0, stream, x, (T*)tok,
                                       B, Cin, H, W, ps, Hp, Wp, ld));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_unpatchify(const void* dtok, float* dx, int B, int Cin, int H, int W, int ps, int ld, int dtype,
                     hipStream_t stream) {
  UNETK_CHECK_ARG(dtok && dx && ps > 0, "unpatchify: bad arguments");
  CHECK_DT("unpatchify", dtype);
  const long total = (long)B * Cin * H * W;
  DISPATCH_T(dtype, hipLaunchKernelGGL(unpatchify_kernel<T>, dim3(cdiv(total, 256)), dim3(256), 0, stream,
                                       (const T*)dtok, dx, B, Cin, H, W, ps, H / ps, W / ps, ld));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_merge_gather(const void* x, void* out, int B, int H, int W, int C, int backward, int dtype,
                       hipStream_t stream) {
  UNETK_CHECK_ARG(x && out, "merge_gather: null pointer");
  CHECK_DT("merge_gather", dtype);
  UNETK_CHECK_ARG(C % VEC_OF(dtype) == 0, "merge_gather: C=%d not a multiple of %d", C, VEC_OF(dtype));
  const int E = VEC_OF(dtype);
  if (!backward) {
    const long total = (long)B * ((H + 1) / 2) * ((W + 1) / 2) * 4 * (C / E);
    DISPATCH_T(dtype, hipLaunchKernelGGL(merge_gather_kernel<T>, dim3(cdiv(total, 256)), dim3(256), 0, stream,
                                         (const T*)x, (T*)out, B, H, W, C));
  } else {   // x = gradient of the merged tensor, out = gradient of the (B,H,W,C) input
    const long total = (long)B * H * W * (C / E);
    DISPATCH_T(dtype, hipLaunchKernelGGL(merge_scatter_kernel<T>, dim3(cdiv(total, 256)), dim3(256), 0, stream,
                                         (const T*)x, (T*)out, B, H, W, C));
  }
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_pixel_shuffle(void* tmp, void* cat, int ldcat, const float* bias, int B, int h, int w, int Co,
                        int backward, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(tmp && cat, "pixel_shuffle: null pointer");
  CHECK_DT("pixel_shuffle", dtype);
  const int E = VEC_OF(dtype);
  UNETK_CHECK_ARG(Co % E == 0 && ldcat % E == 0, "pixel_shuffle: channels not a multiple of %d", E);
  const long total = (long)B * h * w * 4 * (Co / E);
  if (backward) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((pixel_shuffle_kernel<T, true>), dim3(cdiv(total, 256)), dim3(256), 0,
                                         stream, (T*)tmp, (T*)cat, ldcat, bias, B, h, w, Co));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((pixel_shuffle_kernel<T, false>), dim3(cdiv(total, 256)), dim3(256), 0,
                                         stream, (T*)tmp, (T*)cat, ldcat, bias, B, h, w, Co));
  }
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

static int check_attn(const char* name, int He, int We, int C, int heads, int shift) {
  UNETK_CHECK_ARG(heads > 0 && C == heads * HD, "%s: head dimension must be %d (C=%d heads=%d)", name, HD, C, heads);
  UNETK_CHECK_ARG(He % WS == 0 && We % WS == 0 && He > 0 && We > 0, "%s: token grid %dx%d not a multiple of %d", name,
                  He, We, WS);
  UNETK_CHECK_ARG(shift >= 0 && shift < WS, "%s: bad shift %d", name, shift);
  return UNETK_OK;
}

int unetk_window_attn_fwd(const void* qkv, void* out, const float* table, int B, int He, int We, int C, int heads,
                          int shift, int use_mask, float scale, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(qkv && out && table && B > 0, "window_attn_fwd: bad arguments");
  CHECK_DT("window_attn_fwd", dtype);
  if (int rc = check_attn("window_attn_fwd", He, We, C, heads, shift)) return rc;
  if (dtype == UNETK_BF16) {      // persistent double-buffered kernels of attn.hip
    const AttnLaunch a{qkv, nullptr, out, nullptr, table, nullptr, B, He, We, C, heads, shift, use_mask, scale};
    return attn_bf16_fwd(a, stream);
  }
  AttnGeom g{B, He, We, C, heads, shift, use_mask, scale};
  const int nblk = B * (He / WS) * (We / WS) * heads;
  DISPATCH_T(dtype, hipLaunchKernelGGL(window_attn_fwd_kernel<T>, dim3(nblk), dim3(64), 0, stream, (const T*)qkv,
                                       (T*)out, table, g));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

/* dtable_partials: [B*nW*heads][225] floats */
int unetk_window_attn_bwd(const void* qkv, const void* dout, void* dqkv, const float* table, float* dtable,
                          float* dtable_partials, int B, int He, int We, int C, int heads, int shift, int use_mask,
                          float scale, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(qkv && dout && dqkv && table && dtable_partials && B > 0, "window_attn_bwd: bad arguments");
  CHECK_DT("window_attn_bwd", dtype);
  if (int rc = check_attn("window_attn_bwd", He, We, C, heads, shift)) return rc;
  AttnGeom g{B, He, We, C, heads, shift, use_mask, scale};
  const int nwin = B * (He / WS) * (We / WS);
  if (dtype == UNETK_BF16) {
    const AttnLaunch a{qkv, dout, nullptr, dqkv, table, dtable_partials, B, He, We, C, heads, shift, use_mask, scale};
    if (int rc = attn_bf16_bwd(a, stream)) return rc;
  } else {
    hipLaunchKernelGGL(window_attn_bwd_kernel<float>, dim3(nwin * heads), dim3(64), 0, stream, (const float*)qkv,
                       (const float*)dout, (float*)dqkv, table, dtable_partials, g);
    UNETK_CHECK_LAUNCH();
  }
  if (dtable != nullptr) {      // (null: the caller sums the partials later, unetk_dtable_reduce_batched)
    hipLaunchKernelGGL(dtable_reduce_kernel, dim3(DT_BIN_GROUPS, heads), dim3(256), 0, stream, dtable_partials, nwin, heads,
                       dtable);
    UNETK_CHECK_LAUNCH();
  }
  return UNETK_OK;
}

int unetk_dtable_reduce_batched(const int64_t* jobs, int njobs, int max_heads, hipStream_t stream) {
  UNETK_CHECK_ARG(jobs && njobs > 0 && njobs <= 65535 && max_heads > 0 && max_heads <= 65535,
                  "dtable_reduce_batched: bad arguments");
  hipLaunchKernelGGL(dtable_reduce_batched_kernel, dim3(DT_BIN_GROUPS, max_heads, njobs), dim3(256), 0, stream,
                     reinterpret_cast<const long*>(jobs));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
