// HBM-bound kernels around the convolutions (gfx950): layout changes, BatchNorm statistics,
// BatchNorm+ReLU(+Dropout2d mask)(+2x2 max-pool) forward and backward, bilinear resize fused
// with the skip concat and its backward, and the final 1x1 convolution.
//
// Activations are NHWC with the channel axis innermost, so every thread moves 16-byte
// vectors (8 bf16 / 4 f32) of consecutive channels and consecutive threads cover
// consecutive channel groups of one pixel: fully coalesced.  All arithmetic is fp32.
// Per-channel reductions write per-workgroup partials that a second kernel sums in a
// fixed order (in double), so results are bitwise reproducible; there are no atomics.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int RED_THREADS = 256;

// ------------------------------------------------------------------ layout changes
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int C,
                                    int H, int W, int ld) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * H * W * ld;
  if (idx >= total) return;
  const int c = idx % ld;
  const long p = idx / ld;
  const long hw = p % ((long)H * W);
  const int b = p / ((long)H * W);
  dst[idx] = from_f32<T>(c < C ? src[((long)b * C + c) * H * W + hw] : 0.f);
}

// The same for a destination pixel of exactly one 16-byte vector (network inputs: a few planes,
// padded to 8 bf16 / 4 fp32 channels): one thread per pixel, so each plane is read as consecutive
// floats across the wave and the pixel leaves as one vector store.
template <typename T>
__global__ void nchw_to_nhwc_vec_kernel(const float* __restrict__ src, T* __restrict__ dst, long npix, int C,
                                        long hw) {
  constexpr int E = Vec16<T>::N;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const long b = p / hw, r = p - b * hw;
  const float* s = src + b * C * hw + r;
  Vec16<T> o;
#pragma unroll
  for (int e = 0; e < E; ++e) o.v[e] = from_f32<T>(e < C ? s[(long)e * hw] : 0.f);
  store16(dst + p * E, o);
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int B, int C,
                                    int H, int W, int ld) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * C * H * W;
  if (idx >= total) return;
  const long hw = idx % ((long)H * W);
  const int c = (idx / ((long)H * W)) % C;
  const int b = idx / ((long)H * W * C);
  dst[idx] = to_f32(src[((long)b * H * W + hw) * ld + c]);
}

// ------------------------------------------------------------------ BatchNorm statistics
// First level for long partial lists: workgroup (g, column block) sums rows [g*G, (g+1)*G) of its
// columns in double and leaves the result in place as a float pair (high part in row g*G, remainder
// in row g*G+1).  Each thread touches only its own column, reads before it writes.
__global__ __launch_bounds__(256) void bn_prereduce_kernel(float* partials, int nparts, int ncol, int G) {
  const int col = blockIdx.y * 256 + threadIdx.x;
  if (col >= ncol) return;
  const long r0 = (long)blockIdx.x * G;
  long r1 = r0 + G;
  if (r1 > nparts) r1 = nparts;
  if (r1 - r0 < 2) return;
  // eight independent chains (rows r0+k, r0+k+8, ...) keep eight loads in flight; combined in a fixed order
  double c8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  long r = r0;
  for (; r + 8 <= r1; r += 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = partials[(r + k) * ncol + col];
#pragma unroll
    for (int k = 0; k < 8; ++k) c8[k] += (double)v[k];
  }
  for (int k = 0; r < r1; ++r, ++k) c8[k] += (double)partials[r * ncol + col];
  const double s = ((c8[0] + c8[1]) + (c8[2] + c8[3])) + ((c8[4] + c8[5]) + (c8[6] + c8[7]));
  const float hi = (float)s;
  partials[r0 * ncol + col] = hi;
  partials[(r0 + 1) * ncol + col] = (float)(s - (double)hi);
}

// partials: [nparts][2][C] (sum, sum of squares).  One workgroup per 4 channels.  With G > 0 only
// rows g*G and g*G+1 are read (the output of bn_prereduce_kernel).
__global__ __launch_bounds__(256) void bn_finalize_kernel(
    const float* __restrict__ partials, int nparts, int G, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, float* running_mean, float* running_var,
    float* __restrict__ mean_out, float* __restrict__ invstd_out, float* __restrict__ scale_out,
    float* __restrict__ shift_out) {
  __shared__ double red[2][64][4];
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  double s = 0.0, q = 0.0;
  if (c < C) {
    const int nrows = G > 0 ? 2 * ((nparts + G - 1) / G) : nparts;
    for (int i = rl; i < nrows; i += 64) {
      const long r = G > 0 ? (long)(i >> 1) * G + (i & 1) : i;
      if (r < nparts) {
        s += (double)partials[(r * 2) * C + c];
        q += (double)partials[(r * 2 + 1) * C + c];
      }
    }
  }
  red[0][rl][cl] = s;
  red[1][rl][cl] = q;
  __syncthreads();
  if (threadIdx.x < 4 && blockIdx.x * 4 + threadIdx.x < C) {
    const int cc = blockIdx.x * 4 + threadIdx.x;
    double ss = 0.0, qq = 0.0;
    for (int r = 0; r < 64; ++r) {
      ss += red[0][r][threadIdx.x];
      qq += red[1][r][threadIdx.x];
    }
    const double mean = ss / count;
    double var = qq / count - mean * mean;   // biased (normalisation) variance
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[cc] : 1.f, bt = beta ? beta[cc] : 0.f;
    mean_out[cc] = (float)mean;
    invstd_out[cc] = invstd;
    scale_out[cc] = g * invstd;
    shift_out[cc] = bt - (float)mean * g * invstd;
    if (running_mean != nullptr) {
      // nn.BatchNorm2d: running = (1-m)*running + m*batch, with the UNBIASED batch variance
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[cc] = (1.f - momentum) * running_mean[cc] + momentum * (float)mean;
      running_var[cc] = (1.f - momentum) * running_var[cc] + momentum * (float)unbiased;
    }
  }
}

__global__ void bn_fold_eval_kernel(const float* gamma, const float* beta, const float* rmean,
                                    const float* rvar, float eps, const float* conv_bias, float* scale,
                                    float* shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(rvar[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float sc = g * invstd;
  scale[c] = sc;
  shift[c] = b + ((conv_bias ? conv_bias[c] : 0.f) - rmean[c]) * sc;
}

// ------------------------------------------------------------------ BN + ReLU (+mask)(+pool) forward
// One thread per (2x2 pixel quad, 16-byte channel group).  `a` may be a channel slice of a wider
// buffer (row stride lda); argmax (nullable) receives, per pooled element, the position 0..3
// (row-major in the quad) of its first maximum, which is all the pooling backward needs.
template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ y, int ldy, T* __restrict__ a, int lda,
                                                         T* __restrict__ pooled,
                                                         unsigned char* __restrict__ argmax, int B, int H, int W, int C,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ keep, int relu, unsigned chunk) {
  // Threads are (quad lane, channel group): a thread keeps its channel group's scale and shift in
  // registers and walks the quads of the workgroup's chunk, sample by sample (the Dropout2d keep
  // factors are per sample and channel).
  constexpr int E = Vec16<T>::N;
  const int CG = C / E;
  const int lanes = 256 / CG;
  const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
  if (pl >= lanes) return;
  const int QH = (H + 1) >> 1, QW = (W + 1) >> 1;
  const unsigned qhw = (unsigned)QH * QW, nquads = (unsigned)B * qhw;
  const int c0 = cg * E;
  float sc[E], sh[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    sc[e] = scale ? scale[c0 + e] : 1.f;
    sh[e] = shift ? shift[c0 + e] : 0.f;
  }
  const unsigned qbeg = blockIdx.x * chunk;
  unsigned qend = qbeg + chunk;
  if (qend > nquads) qend = nquads;
  unsigned seg = qbeg;
  while (seg < qend) {
    const unsigned b = seg / qhw;
    unsigned seg_end = (b + 1) * qhw;
    if (seg_end > qend || seg_end < seg) seg_end = qend;
    float kp[E];
#pragma unroll
    for (int e = 0; e < E; ++e) kp[e] = keep ? keep[(size_t)b * C + c0 + e] : 1.f;
    for (unsigned qi = seg + pl; qi < seg_end; qi += lanes) {
      const unsigned r = qi - b * qhw;
      const int qy = r / QW, qx = r - qy * QW;
      const bool row1 = 2 * qy + 1 < H, col1 = 2 * qx + 1 < W;
      const long p00 = ((long)b * H + 2 * qy) * W + 2 * qx;
      // the (up to) four pixels of the quad: all loads first
      Vec16<T> v[4];
      v[0] = load16(y + p00 * ldy + c0);
      if (col1) v[1] = load16(y + (p00 + 1) * ldy + c0);
      if (row1) v[2] = load16(y + (p00 + W) * ldy + c0);
      if (row1 && col1) v[3] = load16(y + (p00 + W + 1) * ldy + c0);
      float mx[E];
      unsigned char arg[E];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ok = (k & 1 ? col1 : true) && (k & 2 ? row1 : true);
        if (!ok) continue;
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          float z = fmaf(to_f32(v[k].v[e]), sc[e], sh[e]);
          if (relu) z = fmaxf(z, 0.f);
          z *= kp[e];
          o.v[e] = from_f32<T>(z);
          const float zr = to_f32(o.v[e]);
          if (k == 0 || zr > mx[e]) {         // strict '>': the first maximum in scan order
            mx[e] = zr;
            arg[e] = (unsigned char)k;
          }
        }
        if (a != nullptr) store16(a + (p00 + (k & 1) + (k & 2 ? W : 0)) * lda + c0, o);
      }
      if (pooled != nullptr && row1 && col1) {
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < E; ++e) o.v[e] = from_f32<T>(mx[e]);
        const long q = (((long)b * (H >> 1) + qy) * (W >> 1) + qx) * C + c0;
        store16(pooled + q, o);
        if (argmax != nullptr) {
#pragma unroll
          for (int e = 0; e < E; e += 4)
            *reinterpret_cast<unsigned int*>(argmax + q + e) =
                arg[e] | (arg[e + 1] << 8) | (arg[e + 2] << 16) | ((unsigned)arg[e + 3] << 24);
        }
      }
    }
    seg = seg_end;
  }
}

// The same without pooling, as a stream: threads are (pixel lane, channel group), each keeps its
// channel group's scale/shift in registers and walks the pixels of the workgroup's chunk, so every
// wave instruction moves one contiguous kilobyte.
template <typename T>
__global__ __launch_bounds__(256) void bn_act_stream_kernel(const T* __restrict__ y, int ldy, T* __restrict__ a,
                                                            int lda, unsigned npix, unsigned hw, int C,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ keep, int relu,
                                                            unsigned chunk) {
  constexpr int E = Vec16<T>::N;
  const int CG = C / E;
  const int lanes = 256 / CG;
  const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
  if (pl >= lanes) return;
  const int c0 = cg * E;
  float sc[E], sh[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    sc[e] = scale ? scale[c0 + e] : 1.f;
    sh[e] = shift ? shift[c0 + e] : 0.f;
  }
  const unsigned pbeg = blockIdx.x * chunk;
  unsigned pend = pbeg + chunk;
  if (pend > npix) pend = npix;
  auto one = [&](unsigned p, const Vec16<T>& v, const float (&kp)[E]) {
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float z = fmaf(to_f32(v.v[e]), sc[e], sh[e]);
      if (relu) z = fmaxf(z, 0.f);
      o.v[e] = from_f32<T>(z * kp[e]);
    }
    store16(a + (size_t)p * lda + c0, o);
  };
  // sample by sample: the Dropout2d keep factors are per (sample, channel), loaded once per segment
  unsigned seg = pbeg;
  while (seg < pend) {
    const unsigned b = seg / hw;
    unsigned seg_end = (b + 1) * hw;
    if (seg_end > pend || seg_end < seg) seg_end = pend;
    float kp[E];
#pragma unroll
    for (int e = 0; e < E; ++e) kp[e] = keep ? keep[(size_t)b * C + c0 + e] : 1.f;
    unsigned p = seg + pl;
    for (; p + 3 * lanes < seg_end; p += 4 * lanes) {     // four pixels in flight
      const Vec16<T> v0 = load16(y + (size_t)p * ldy + c0), v1 = load16(y + (size_t)(p + lanes) * ldy + c0);
      const Vec16<T> v2 = load16(y + (size_t)(p + 2 * lanes) * ldy + c0), v3 = load16(y + (size_t)(p + 3 * lanes) * ldy + c0);
      one(p, v0, kp);
      one(p + lanes, v1, kp);
      one(p + 2 * lanes, v2, kp);
      one(p + 3 * lanes, v3, kp);
    }
    for (; p + lanes < seg_end; p += 2 * lanes) {
      const Vec16<T> v0 = load16(y + (size_t)p * ldy + c0), v1 = load16(y + (size_t)(p + lanes) * ldy + c0);
      one(p, v0, kp);
      one(p + lanes, v1, kp);
    }
    if (p < seg_end) one(p, load16(y + (size_t)p * ldy + c0), kp);
    seg = seg_end;
  }
}

// ------------------------------------------------------------------ per-channel reductions
// Generic NHWC reducer: workgroup k covers pixels [k*chunk, (k+1)*chunk); threads are laid out as
// (pixel lane, channel group).  MODE 0: BatchNorm backward sums (sum dz, sum dz*xhat).
// MODE 1: 1x1-conv weight gradient (sum d*x, sum d) with a per-pixel scalar d (fp32).
struct RedArgs {
  const void* x;        // MODE 0: conv output y ; MODE 1: input activations
  const void* g;        // MODE 0: grad wrt activation (T) ; MODE 1: per-pixel scalar (float)
  const float* scale;   // MODE 0
  const float* shift;
  const float* mean;
  const float* invstd;
  const float* keep;    // [B][C] or null
  float* partials;      // [nblk][2][C]
  const float* g1_d;    // MODE 0, rank-one gradient: g[p][c] = round_T(g1_d[p] * g1_w[c]) instead of reading g
  const float* g1_w;
  const void* dpooled;  // MODE 0, gradient through a 2x2 max-pool formed on the fly: g[p][c] = round_T(g[p][c] +
  const unsigned char* argmax;   // (argmax[quad][c] == place of p in its quad ? dpooled[quad][c] : 0)), see PoolSrc
  int H, W;
  long npix, hw;
  int C, ldx, ldg, chunk, relu;
};

// Gradient of the input of a 2x2 max-pool whose input also feeds a skip connection, for one pixel and one channel
// group, as pool_bwd_kernel would have stored it: the skip gradient plus the pooled gradient of the pixel's quad
// where the forward's argmax map names this pixel, rounded to T.  The BatchNorm-backward passes read it through
// this loader instead of a stored tensor (a tensor written once and read twice otherwise).
template <typename T>
struct PoolSrc {
  static constexpr int E = Vec16<T>::N;
  const T* dskip;
  const T* dpooled;
  const unsigned char* argmax;
  int ldskip, H, W, C, c0;
  __device__ __forceinline__ Vec16<T> at(unsigned p, unsigned b, unsigned hw) const {
    const unsigned rem = p - b * hw;
    const unsigned yy = rem / (unsigned)W, xx = rem - yy * (unsigned)W;
    Vec16<T> sk = load16(dskip + (size_t)p * ldskip + c0);
    if ((int)(yy | 1) < H && (int)(xx | 1) < W) {
      const size_t q = (((size_t)b * (H >> 1) + (yy >> 1)) * (W >> 1) + (xx >> 1)) * C + c0;
      const unsigned k = ((yy & 1) << 1) | (xx & 1);
      const Vec16<T> d = load16(dpooled + q);
#pragma unroll
      for (int e = 0; e < E; e += 4) {
        const unsigned packed = *reinterpret_cast<const unsigned*>(argmax + q + e);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (((packed >> (8 * i)) & 255u) == k) sk.v[e + i] = from_f32<T>(to_f32(sk.v[e + i]) + to_f32(d.v[e + i]));
      }
    }
    return sk;
  }
};

template <typename T, int MODE>
__global__ __launch_bounds__(RED_THREADS) void channel_reduce_kernel(RedArgs a) {
  constexpr int E = Vec16<T>::N;
  __shared__ float red[2][RED_THREADS * E];
  const int CG = a.C / E;
  const int lanes = RED_THREADS / CG;
  const int tid = threadIdx.x;
  const int cg = tid % CG, pl = tid / CG;
  const int c0 = cg * E;
  // The BatchNorm-backward sums walk the tensor LAST CHUNK FIRST (same chunks, same partial rows: same bits).  The
  // gradient was written first-to-last by the kernel before, so of a tensor larger than the last-level cache
  // (256 MB) it is the tail that is still there; walking first-to-last would find nothing and evict the tail on the
  // way.  The apply pass that follows walks forward again and finds the heads of g and y that this pass read last.
  // (-0.15 ms of the 17.5 ms UNet step at batch 64; reversing both, or neither, loses it.)
  const unsigned bid = MODE == 0 ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  float s0[E], s1[E];
#pragma unroll
  for (int e = 0; e < E; ++e) s0[e] = s1[e] = 0.f;
  if (pl < lanes) {
    float sc[E], sh[E], mu[E], is[E];
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        sc[e] = a.scale[c0 + e];
        sh[e] = a.shift[c0 + e];
        mu[e] = a.mean[c0 + e];
        is[e] = a.invstd[c0 + e];
      }
    }
    const long pbeg = (long)bid * a.chunk;
    long pend = pbeg + a.chunk;
    if (pend > a.npix) pend = a.npix;
    const T* x = reinterpret_cast<const T*>(a.x);
    if (MODE == 0) {
      // walk the chunk sample by sample: the Dropout2d keep factors are loaded once per sample and
      // no per-pixel division is needed
      const T* gp = reinterpret_cast<const T*>(a.g);
      long seg = pbeg;
      while (seg < pend) {
        const long b = seg / a.hw;
        long seg_end = (b + 1) * a.hw;
        if (seg_end > pend) seg_end = pend;
        float kp[E];
#pragma unroll
        for (int e = 0; e < E; ++e) kp[e] = a.keep ? a.keep[b * a.C + c0 + e] : 1.f;
        auto one = [&](const Vec16<T>& xv, const Vec16<T>& gv) {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const float yv = to_f32(xv.v[e]);
            float dz = to_f32(gv.v[e]) * kp[e];
            // (explicit fused multiply-adds: every instantiation of this loop, and the forward's own
            //  y*scale+shift, must round alike, whatever the compiler would contract in each)
            if (a.relu && !(fmaf(yv, sc[e], sh[e]) > 0.f)) dz = 0.f;
            s0[e] += dz;
            s1[e] = fmaf(dz, (yv - mu[e]) * is[e], s1[e]);
          }
        };
        long p = seg + pl;
        if (a.dpooled != nullptr) {
          const PoolSrc<T> src{gp, reinterpret_cast<const T*>(a.dpooled), a.argmax, a.ldg, a.H, a.W, a.C, c0};
          const unsigned hw = (unsigned)a.hw, bb = (unsigned)b;
          for (; p + lanes < seg_end; p += 2 * lanes) {
            const Vec16<T> x0 = load16(x + p * a.ldx + c0), x1 = load16(x + (p + lanes) * a.ldx + c0);
            const Vec16<T> g0 = src.at((unsigned)p, bb, hw), g1 = src.at((unsigned)(p + lanes), bb, hw);
            one(x0, g0);
            one(x1, g1);
          }
          if (p < seg_end) one(load16(x + p * a.ldx + c0), src.at((unsigned)p, bb, hw));
        } else if (a.g1_d != nullptr) {
          // the gradient of the last activation is an outer product (per-pixel scalar x per-channel
          // weight of the final 1x1 conv): formed here, rounded as the stored tensor would be
          float wv[E];
#pragma unroll
          for (int e = 0; e < E; ++e) wv[e] = a.g1_w[c0 + e];
          auto gen = [&](long pp) {
            const float d = a.g1_d[pp];
            Vec16<T> gv;
#pragma unroll
            for (int e = 0; e < E; ++e) gv.v[e] = from_f32<T>(d * wv[e]);
            return gv;
          };
          for (; p + lanes < seg_end; p += 2 * lanes) {
            const Vec16<T> x0 = load16(x + p * a.ldx + c0), x1 = load16(x + (p + lanes) * a.ldx + c0);
            one(x0, gen(p));
            one(x1, gen(p + lanes));
          }
          if (p < seg_end) one(load16(x + p * a.ldx + c0), gen(p));
        } else {
        for (; p + 3 * lanes < seg_end; p += 4 * lanes) {   // four pixels (eight loads) in flight
          const Vec16<T> x0 = load16(x + p * a.ldx + c0), g0 = load16(gp + p * a.ldg + c0);
          const Vec16<T> x1 = load16(x + (p + lanes) * a.ldx + c0), g1 = load16(gp + (p + lanes) * a.ldg + c0);
          const Vec16<T> x2 = load16(x + (p + 2 * lanes) * a.ldx + c0), g2 = load16(gp + (p + 2 * lanes) * a.ldg + c0);
          const Vec16<T> x3 = load16(x + (p + 3 * lanes) * a.ldx + c0), g3 = load16(gp + (p + 3 * lanes) * a.ldg + c0);
          one(x0, g0);
          one(x1, g1);
          one(x2, g2);
          one(x3, g3);
        }
        for (; p + lanes < seg_end; p += 2 * lanes) {
          const Vec16<T> x0 = load16(x + p * a.ldx + c0), g0 = load16(gp + p * a.ldg + c0);
          const Vec16<T> x1 = load16(x + (p + lanes) * a.ldx + c0), g1 = load16(gp + (p + lanes) * a.ldg + c0);
          one(x0, g0);
          one(x1, g1);
        }
        if (p < seg_end) one(load16(x + p * a.ldx + c0), load16(gp + p * a.ldg + c0));
        }
        seg = seg_end;
      }
    } else {
      const float* dp = reinterpret_cast<const float*>(a.g);
      auto acc1 = [&](const Vec16<T>& xv, float d) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          s0[e] += d * to_f32(xv.v[e]);
          s1[e] += d;
        }
      };
      long p = pbeg + pl;
      for (; p + 3 * lanes < pend; p += 4 * lanes) {      // four pixels in flight, added in pixel order
        const Vec16<T> x0 = load16(x + p * a.ldx + c0), x1 = load16(x + (p + lanes) * a.ldx + c0);
        const Vec16<T> x2 = load16(x + (p + 2 * lanes) * a.ldx + c0), x3 = load16(x + (p + 3 * lanes) * a.ldx + c0);
        const float d0 = dp[p], d1 = dp[p + lanes], d2 = dp[p + 2 * lanes], d3 = dp[p + 3 * lanes];
        acc1(x0, d0);
        acc1(x1, d1);
        acc1(x2, d2);
        acc1(x3, d3);
      }
      for (; p < pend; p += lanes) acc1(load16(x + p * a.ldx + c0), dp[p]);
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    red[0][tid * E + e] = s0[e];
    red[1][tid * E + e] = s1[e];
  }
  __syncthreads();
  for (int c = tid; c < a.C; c += RED_THREADS) {
    const int g = c / E, e = c - g * E;
    float t0 = 0.f, t1 = 0.f;
    for (int l = 0; l < lanes; ++l) {
      t0 += red[0][(l * CG + g) * E + e];
      t1 += red[1][(l * CG + g) * E + e];
    }
    a.partials[((long)bid * 2) * a.C + c] = t0;
    a.partials[((long)bid * 2 + 1) * a.C + c] = t1;
  }
}

// sums [nparts][2][C] partials into out0[C], out1[C] (fp32), optionally accumulating
__global__ __launch_bounds__(256) void partial_sum_kernel(const float* __restrict__ partials, int nparts,
                                                          int ldp, int C, float* out0, float* out1,
                                                          int accumulate) {
  __shared__ double red[2][64][4];
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  double s = 0.0, q = 0.0;
  if (c < C)
    for (int r = rl; r < nparts; r += 64) {
      s += (double)partials[((long)r * 2) * ldp + c];
      q += (double)partials[((long)r * 2 + 1) * ldp + c];
    }
  red[0][rl][cl] = s;
  red[1][rl][cl] = q;
  __syncthreads();
  if (threadIdx.x < 4 && blockIdx.x * 4 + threadIdx.x < C) {
    const int cc = blockIdx.x * 4 + threadIdx.x;
    double ss = 0.0, qq = 0.0;
    for (int r = 0; r < 64; ++r) {
      ss += red[0][r][threadIdx.x];
      qq += red[1][r][threadIdx.x];
    }
    if (out0) out0[cc] = accumulate ? out0[cc] + (float)ss : (float)ss;
    if (out1) out1[cc] = accumulate ? out1[cc] + (float)qq : (float)qq;
  }
}

// BatchNorm backward, everything between the reduction and the apply pass in ONE launch: sums the
// [nparts][2][C] partials (double, fixed order) into sums[0:C] = sum dz and sums[C:2C] = sum dz*xhat,
// writes (or accumulates into) dbeta / dgamma, and derives the apply pass's coefficients (see
// bn_bwd_coef_kernel) from the rounded sums exactly as the separate kernels do.
__global__ __launch_bounds__(256) void bn_bwd_sums_coef_kernel(const float* __restrict__ partials, int nparts, int C,
                                                               float* __restrict__ sums, float* dbeta, float* dgamma,
                                                               int accumulate, const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, float inv_n,
                                                               float* __restrict__ coef) {
  __shared__ double red[2][64][4];
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  double s = 0.0, q = 0.0;
  if (c < C) {
    // (this launch sits between the reduction and the apply pass of every BatchNorm backward: eight rows of a lane
    //  are in flight, added in row order as a one-by-one loop would)
    int r = rl;
    for (; r + 64 * 7 < nparts; r += 64 * 8) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[u] = partials[((long)(r + 64 * u) * 2) * C + c];
        b[u] = partials[((long)(r + 64 * u) * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s += (double)a[u];
        q += (double)b[u];
      }
    }
    for (; r < nparts; r += 64) {
      s += (double)partials[((long)r * 2) * C + c];
      q += (double)partials[((long)r * 2 + 1) * C + c];
    }
  }
  red[0][rl][cl] = s;
  red[1][rl][cl] = q;
  __syncthreads();
  if (threadIdx.x < 4 && blockIdx.x * 4 + threadIdx.x < C) {
    const int cc = blockIdx.x * 4 + threadIdx.x;
    double ss = 0.0, qq = 0.0;
    for (int r = 0; r < 64; ++r) {
      ss += red[0][r][threadIdx.x];
      qq += red[1][r][threadIdx.x];
    }
    const float sdz = (float)ss, sdzx = (float)qq;
    sums[cc] = sdz;
    sums[C + cc] = sdzx;
    if (dbeta) dbeta[cc] = accumulate ? dbeta[cc] + sdz : sdz;
    if (dgamma) dgamma[cc] = accumulate ? dgamma[cc] + sdzx : sdzx;
    const float sc = scale[cc];
    const float k1 = -sc * sdzx * invstd[cc] * inv_n;
    coef[cc] = sc;
    coef[C + cc] = shift[cc];
    coef[2 * C + cc] = k1;
    coef[3 * C + cc] = -sc * sdz * inv_n - k1 * mean[cc];
  }
}

// Per-channel coefficients of the BatchNorm backward: with dz = keep * g * [z > 0],
//   dy = scale * (dz - sum_dz/n - xhat * sum_dzxhat/n) = scale*dz + k1*y + k0,
//   k1 = -scale * sum_dzxhat * invstd / n,   k0 = -scale * sum_dz / n - k1 * mean.
// coef: [4][C] = scale, shift, k1, k0
__global__ void bn_bwd_coef_kernel(const float* __restrict__ scale, const float* __restrict__ shift,
                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                   const float* __restrict__ sums, float inv_n, float* __restrict__ coef,
                                   int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = scale[c];
  const float k1 = -sc * sums[C + c] * invstd[c] * inv_n;
  coef[c] = sc;
  coef[C + c] = shift[c];
  coef[2 * C + c] = k1;
  coef[3 * C + c] = -sc * sums[c] * inv_n - k1 * mean[c];
}

template <int E> __device__ __forceinline__ void load_params(const float* p, float (&out)[E]) {
#pragma unroll
  for (int i = 0; i < E; i += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + i);
    out[i] = v[0];
    out[i + 1] = v[1];
    out[i + 2] = v[2];
    out[i + 3] = v[3];
  }
}

// Threads are (pixel lane, channel group); each keeps its channel group's coefficients in
// registers and streams over the pixels of the workgroup's chunk.
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void bn_act_bwd_apply_kernel(
    const T* __restrict__ g, const T* __restrict__ y, T* __restrict__ dy, unsigned npix, unsigned hw, int C,
    const float* __restrict__ coef, const float* __restrict__ keep, int relu, unsigned chunk,
    const float* __restrict__ g1_d, const float* __restrict__ g1_w, const T* __restrict__ dpooled,
    const unsigned char* __restrict__ argmax, int ldg, int H, int W) {
  constexpr int E = Vec16<T>::N;
  const int CG = C / E;
  const int lanes = RED_THREADS / CG;
  const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
  if (pl >= lanes) return;
  const int c0 = cg * E;
  const PoolSrc<T> src{g, dpooled, argmax, ldg, H, W, C, c0};
  float sc[E], sh[E], k1[E], k0[E];
  load_params<E>(coef + c0, sc);
  load_params<E>(coef + C + c0, sh);
  load_params<E>(coef + 2 * C + c0, k1);
  load_params<E>(coef + 3 * C + c0, k0);
  const unsigned pbeg = blockIdx.x * chunk;
  unsigned pend = pbeg + chunk;
  if (pend > npix) pend = npix;
  float wv[E];
#pragma unroll
  for (int e = 0; e < E; ++e) wv[e] = g1_d != nullptr ? g1_w[c0 + e] : 0.f;
  auto gload = [&](unsigned p, unsigned b) {
    if (dpooled != nullptr) return src.at(p, b, hw);      // (uniform) gradient through the max-pool formed on the fly
    if (g1_d != nullptr) {      // (uniform) rank-one gradient formed on the fly
      const float d = g1_d[p];
      Vec16<T> gv;
#pragma unroll
      for (int e = 0; e < E; ++e) gv.v[e] = from_f32<T>(d * wv[e]);
      return gv;
    }
    return load16(g + (size_t)p * C + c0);
  };
  auto one = [&](unsigned p, const Vec16<T>& gv, const Vec16<T>& yv, const float (&kp)[E]) {
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float yy = to_f32(yv.v[e]);
      float dz = to_f32(gv.v[e]) * kp[e];
      if (relu && !(fmaf(yy, sc[e], sh[e]) > 0.f)) dz = 0.f;
      o.v[e] = from_f32<T>(fmaf(sc[e], dz, fmaf(k1[e], yy, k0[e])));
    }
    store16(dy + (size_t)p * C + c0, o);
  };
  // sample by sample: the Dropout2d keep factors are per (sample, channel), loaded once per segment
  unsigned seg = pbeg;
  while (seg < pend) {
    const unsigned b = seg / hw;
    unsigned seg_end = (b + 1) * hw;
    if (seg_end > pend || seg_end < seg) seg_end = pend;
    float kp[E];
    if (keep != nullptr) {
      load_params<E>(keep + (size_t)b * C + c0, kp);
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) kp[e] = 1.f;
    }
    unsigned p = seg + pl;
    for (; p + lanes < seg_end; p += 2 * lanes) {     // two pixels (four loads) in flight
      const Vec16<T> g0 = gload(p, b), y0 = load16(y + (size_t)p * C + c0);
      const Vec16<T> g1 = gload(p + lanes, b), y1 = load16(y + (size_t)(p + lanes) * C + c0);
      one(p, g0, y0, kp);
      one(p + lanes, g1, y1, kp);
    }
    if (p < seg_end) one(p, gload(p, b), load16(y + (size_t)p * C + c0), kp);
    seg = seg_end;
  }
}

// ------------------------------------------------------------------ 2x2 max-pool backward (+ skip grad)
// da[pix] = dskip[pix] + (pix is the first maximum of its quad ? dpooled[quad] : 0)
// The first maximum comes from the forward's argmax map when given, else it is recomputed from a.
template <typename T>
__global__ void pool_bwd_kernel(const T* __restrict__ a, const unsigned char* __restrict__ argmax,
                                const T* __restrict__ dpooled, const T* __restrict__ dskip, int ldskip,
                                T* __restrict__ da, int B, int H, int W, int C) {
  constexpr int E = Vec16<T>::N;
  const int CG = C / E;
  const int QH = (H + 1) >> 1, QW = (W + 1) >> 1;
  const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;     // the host checks that it fits 31 bits
  if (idx >= (unsigned)B * QH * QW * CG) return;
  const int cg = idx % CG;
  unsigned t = idx / CG;
  const int qx = t % QW;
  t /= QW;
  const int qy = t % QH;
  const int b = t / QH;
  const int c0 = cg * E;
  const bool full = (2 * qy + 1 < H) && (2 * qx + 1 < W);
  Vec16<T> av[4];
  int arg[E];
  float dp[E];
  if (full && argmax != nullptr) {
    const long q = (((long)b * (H >> 1) + qy) * (W >> 1) + qx) * C + c0;
#pragma unroll
    for (int e = 0; e < E; e += 4) {
      const unsigned int packed = *reinterpret_cast<const unsigned int*>(argmax + q + e);
      arg[e] = packed & 255;
      arg[e + 1] = (packed >> 8) & 255;
      arg[e + 2] = (packed >> 16) & 255;
      arg[e + 3] = packed >> 24;
    }
    Vec16<T> d = load16(dpooled + q);
#pragma unroll
    for (int e = 0; e < E; ++e) dp[e] = to_f32(d.v[e]);
  } else if (full) {
    float mx[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      mx[e] = -INFINITY;
      arg[e] = 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = 2 * qy + (k >> 1), xx = 2 * qx + (k & 1);
      av[k] = load16(a + (((long)b * H + yy) * W + xx) * C + c0);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float v = to_f32(av[k].v[e]);
        if (v > mx[e] || k == 0) {   // strict '>' keeps the first maximum in scan order
          mx[e] = v;
          arg[e] = k;
        }
      }
    }
    Vec16<T> d = load16(dpooled + (((long)b * (H >> 1) + qy) * (W >> 1) + qx) * C + c0);
#pragma unroll
    for (int e = 0; e < E; ++e) dp[e] = to_f32(d.v[e]);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int yy = 2 * qy + (k >> 1), xx = 2 * qx + (k & 1);
    if (yy < H && xx < W) {
      const long pix = ((long)b * H + yy) * W + xx;
      Vec16<T> o;
      Vec16<T> sk = zero16<T>();
      if (dskip) sk = load16(dskip + pix * ldskip + c0);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        float v = dskip ? to_f32(sk.v[e]) : 0.f;
        if (full && arg[e] == k) v += dp[e];
        o.v[e] = from_f32<T>(v);
      }
      store16(da + pix * C + c0, o);
    }
  }
}

// Block index for a grid of 8*k workgroups in which XCD x (= blockIdx.x % 8) owns blocks [x*k, (x+1)*k).
__device__ __forceinline__ long xcd_contiguous_block() {
  const long per = gridDim.x >> 3;
  return (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
}

// ------------------------------------------------------------------ bilinear resize (align_corners=False)
__device__ __forceinline__ void src_index(float scale, int dst, int in_size, int& i0, int& i1, float& l1) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = s - (float)i0;
}

template <typename T>
__device__ __forceinline__ Vec16<T> bilerp(const Vec16<T>& v00, const Vec16<T>& v01, const Vec16<T>& v10,
                                           const Vec16<T>& v11, float ly, float lx) {
  const float hy = 1.f - ly, hx = 1.f - lx;
  Vec16<T> o;
#pragma unroll
  for (int e = 0; e < Vec16<T>::N; ++e)
    o.v[e] = from_f32<T>(hy * (hx * to_f32(v00.v[e]) + lx * to_f32(v01.v[e])) +
                         ly * (hx * to_f32(v10.v[e]) + lx * to_f32(v11.v[e])));
  return o;
}

// One resized vector: channels [c0, c0+E) of output pixel (y,x); base = low + image offset + c0.
template <typename T>
__device__ __forceinline__ Vec16<T> resize_at(const T* __restrict__ base, int h, int w, int C1, int y, int x,
                                              float sy, float sx) {
  int y0, y1, x0, x1;
  float ly, lx;
  src_index(sy, y, h, y0, y1, ly);
  src_index(sx, x, w, x0, x1, lx);
  return bilerp(load16(base + ((long)y0 * w + x0) * C1), load16(base + ((long)y0 * w + x1) * C1),
                load16(base + ((long)y1 * w + x0) * C1), load16(base + ((long)y1 * w + x1) * C1), ly, lx);
}

// out[b,y,x, 0:C1] = resize(low)[b,y,x,:], out[b,y,x, C1:C1+C2] = skip[b,y,x,:]
template <typename T>
__global__ void upsample_concat_kernel(const T* __restrict__ low, int h, int w, int C1,
                                       const T* __restrict__ skip, int ldskip, int C2, T* __restrict__ out,
                                       int B, int H, int W, float sy, float sx) {
  constexpr int E = Vec16<T>::N;
  const int C = C1 + C2, CG1 = C1 / E;
  const int CG = skip != nullptr ? C / E : CG1;   // without a skip only channels [0,C1) are written
  const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;     // the host checks that it fits 31 bits
  if (idx >= (unsigned)B * H * W * CG) return;
  const int cg = idx % CG;
  const unsigned p = idx / CG;
  const int x = p % W;
  const int y = (p / W) % H;
  const int b = p / ((unsigned)W * H);
  Vec16<T> o;
  if (cg < CG1)
    o = resize_at(low + (long)b * h * w * C1 + cg * E, h, w, C1, y, x, sy, sx);
  else
    o = load16(skip + (size_t)p * ldskip + (cg - CG1) * E);
  store16(out + (size_t)p * C + cg * E, o);
}

// The exact 2x case (H = 2h, W = 2w): one thread produces the 2x2 output pixels that share the
// low-resolution pixel (i,j).  Away from the border they read the 3x3 neighbourhood of (i,j): nine
// loads feed four outputs where the general kernel issues sixteen.  Indices and weights still come
// from src_index and the arithmetic is the general kernel's, so both give the same bits.
template <typename T>
__global__ void upsample2x_concat_kernel(const T* __restrict__ low, int h, int w, int C1,
                                         const T* __restrict__ skip, int ldskip, int C2, T* __restrict__ out,
                                         int B) {
  constexpr int E = Vec16<T>::N;
  const int C = C1 + C2, CG1 = C1 / E;
  const int CG = skip != nullptr ? C / E : CG1;
  const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;     // the host checks that it fits 31 bits
  if (idx >= (unsigned)B * h * w * CG) return;
  const int cg = idx % CG;
  const unsigned p = idx / CG;
  const int j = p % w;
  const int i = (p / w) % h;
  const int b = p / ((unsigned)w * h);
  const int H = 2 * h, W = 2 * w;
  const size_t opix = ((size_t)b * H + 2 * i) * W + 2 * j;       // output pixel (2i, 2j)
  if (cg >= CG1) {
    const int c0 = (cg - CG1) * E;
    Vec16<T> s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = load16(skip + (opix + (size_t)(k >> 1) * W + (k & 1)) * ldskip + c0);
#pragma unroll
    for (int k = 0; k < 4; ++k) store16(out + (opix + (size_t)(k >> 1) * W + (k & 1)) * C + cg * E, s[k]);
    return;
  }
  const T* base = low + (long)b * h * w * C1 + cg * E;
  T* o = out + opix * C + cg * E;
  if (i >= 1 && i < h - 1 && j >= 1 && j < w - 1) {
    // output row 2i+a reads low rows (i-1+a, i+a), output column 2j+c reads low columns (j-1+c, j+c)
    Vec16<T> v[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) v[r][q] = load16(base + ((long)(i - 1 + r) * w + (j - 1 + q)) * C1);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      int t0, t1;
      float ly;
      src_index(0.5f, 2 * i + a, h, t0, t1, ly);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float lx;
        src_index(0.5f, 2 * j + c, w, t0, t1, lx);
        store16(o + ((size_t)a * W + c) * C, bilerp(v[a][c], v[a][c + 1], v[a + 1][c], v[a + 1][c + 1], ly, lx));
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      store16(o + ((size_t)(k >> 1) * W + (k & 1)) * C,
              resize_at(base, h, w, C1, 2 * i + (k >> 1), 2 * j + (k & 1), 0.5f, 0.5f));
  }
}


// The exact 2x case (H = 2h, W = 2w; bfloat16, C1 a multiple of 32): a workgroup owns an 8 x 8 tile of dlow times 32
// channels, stages the 18 x 18 pixels of dout it gathers from in LDS once (each element of dout would otherwise be
// loaded by four threads), and every thread sums its sixteen candidates in the order of the generic kernel below
// (rows, then columns, zero weights skipped): the same bits.  XCD x walks a contiguous eighth of the
// (tile, channel slab) list, slabs fastest: the slabs of a tile share the cache lines of its pixels.
__global__ __launch_bounds__(256) void upsample2x_bwd_tile_kernel(const bf16* __restrict__ dout, int ldout, int H, int W,
                                                                   bf16* __restrict__ dlow, int B, int h, int w, int C1,
                                                                   int tiles_x, int tiles_y, long nitems) {
  using T = bf16;
  constexpr int TP = 8, HP = 2 * TP + 2;                 // tile side in dlow, patch side in dout
  __shared__ __attribute__((aligned(16))) char patch[HP * HP * 64];
  const long item = xcd_contiguous_block();
  if (item >= nitems) return;
  const int nslab = C1 / 32;
  const int slab = (int)(item % nslab);
  long t = item / nslab;
  const int tx = (int)(t % tiles_x);
  t /= tiles_x;
  const int ty = (int)(t % tiles_y);
  const int b = (int)(t / tiles_y);
  const int i0 = ty * TP, j0 = tx * TP;
  const int Y0 = 2 * i0 - 1, X0 = 2 * j0 - 1;
  const T* base = dout + (long)b * H * W * ldout + slab * 32;
  {   // all of a thread's loads in flight before the first LDS store
    constexpr int NLD = (HP * HP * 4 + 255) / 256;
    Vec16<T> stage[NLD];
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int sidx = threadIdx.x + it * 256;
      const int px = sidx >> 2, sl = sidx & 3;
      const int py = px / HP, pxx = px - py * HP;
      const int Y = Y0 + py, X = X0 + pxx;
      stage[it] = zero16<T>();
      if (sidx < HP * HP * 4 && (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W)
        stage[it] = load16(base + ((long)Y * W + X) * ldout + sl * 8);
    }
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int sidx = threadIdx.x + it * 256;
      if (sidx < HP * HP * 4) *reinterpret_cast<Vec16<T>*>(patch + (sidx >> 2) * 64 + (sidx & 3) * 16) = stage[it];
    }
  }
  __syncthreads();
  const int cg = threadIdx.x & 3, pix = threadIdx.x >> 2;
  const int ti = pix >> 3, tj = pix & 7;
  const int i = i0 + ti, j = j0 + tj;
  if (i >= h || j >= w) return;
  const float sy = 0.5f, sx = 0.5f;
  // candidate rows / columns of dout, clipped as in the generic kernel
  int ylo = 2 * i - 1, yhi = 2 * i + 2, xlo = 2 * j - 1, xhi = 2 * j + 2;
  if (ylo < 0) ylo = 0;
  if (xlo < 0) xlo = 0;
  if (yhi > H - 1) yhi = H - 1;
  if (xhi > W - 1) xhi = W - 1;
  float wxs[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int x0, x1;
    float lx;
    src_index(sx, xlo + k, w, x0, x1, lx);
    wxs[k] = (xlo + k <= xhi) ? (x0 == j ? 1.f - lx : 0.f) + (x1 == j ? lx : 0.f) : 0.f;
  }
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int y = ylo; y <= yhi; ++y) {
    int y0, y1;
    float ly;
    src_index(sy, y, h, y0, y1, ly);
    const float wy = (y0 == i ? 1.f - ly : 0.f) + (y1 == i ? ly : 0.f);
    if (wy == 0.f) continue;
    const char* row = patch + ((y - Y0) * HP + (xlo - X0)) * 64 + cg * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (wxs[k] == 0.f) continue;
      const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(row + k * 64);
      const float wgt = wy * wxs[k];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += wgt * to_f32(v.v[e]);
    }
  }
  Vec16<T> o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o.v[e] = from_f32<T>(acc[e]);
  store16(dlow + (((long)b * h + i) * w + j) * C1 + slab * 32 + cg * 8, o);
}

// BatchNorm + ReLU (+ Dropout2d factors) of a block's last conv output y and the exact 2x up-sampling of the result
// in ONE pass (bfloat16, C1 a multiple of 32): the activation a = relu(y*scale+shift)*keep is consumed by nothing but
// the up-sampling in front of the next decoder block, so it is formed on the way into LDS (rounded to bf16 as the
// separate pass would store it) and never written.  A workgroup owns an 8 x 8 tile of y times 32 channels, stages
// its 10 x 10 neighbourhood (border pixels replicated, as the clamped source indices of the resize read them) and
// every thread writes the 2 x 2 output pixels of its y pixel with the index / weight arithmetic of
// upsample2x_concat_kernel.  XCD x walks a contiguous eighth of the (tile, channel slab) list, slabs fastest.
// BN = false: the plain resize of a stored activation (y is that activation; scale, shift and keep are not read).
template <int NSL, bool BN = true>   // NSL 32-channel slabs a workgroup takes side by side: runs of 64 * NSL bytes per pixel
__global__ __launch_bounds__(256) void bn_act_upsample2x_tile_kernel(const bf16* __restrict__ y, int h, int w, int C1,
                                                                      const float* __restrict__ scale,
                                                                      const float* __restrict__ shift,
                                                                      const float* __restrict__ keep,
                                                                      bf16* __restrict__ out, int ldo, int tiles_x,
                                                                      int tiles_y, long nitems) {
  using T = bf16;
  constexpr int TP = 8, HP = TP + 2;
  constexpr int VPP = 4 * NSL, PB = 16 * VPP;            // 16-byte vectors / bytes of a patch pixel
  __shared__ __attribute__((aligned(16))) char patch[HP * HP * PB];
  const long item = xcd_contiguous_block();
  if (item >= nitems) return;
  const int nslab = C1 / (32 * NSL);
  const int slab = (int)(item % nslab);
  long t = item / nslab;
  const int tx = (int)(t % tiles_x);
  t /= tiles_x;
  const int ty = (int)(t % tiles_y);
  const int b = (int)(t / tiles_y);
  const int i0 = ty * TP, j0 = tx * TP;
  const int cbase = slab * 32 * NSL;
  const T* base = y + (long)b * h * w * C1 + cbase;
#pragma unroll
  for (int sidx = threadIdx.x; sidx < HP * HP * VPP; sidx += 256) {
    const int px = sidx / VPP, sl = sidx - px * VPP;
    const int py = px / HP, pxx = px - py * HP;
    int Y = i0 - 1 + py, X = j0 - 1 + pxx;
    Y = Y < 0 ? 0 : (Y > h - 1 ? h - 1 : Y);
    X = X < 0 ? 0 : (X > w - 1 ? w - 1 : X);
    const Vec16<T> v = load16(base + ((long)Y * w + X) * C1 + sl * 8);
    if constexpr (!BN) {
      *reinterpret_cast<Vec16<T>*>(patch + px * PB + sl * 16) = v;
      continue;
    }
    const int c0 = cbase + sl * 8;
    // (the per-channel factors as 16-byte loads: c0 is a multiple of eight floats)
    const f32x4 sc0 = *reinterpret_cast<const f32x4*>(scale + c0), sc1 = *reinterpret_cast<const f32x4*>(scale + c0 + 4);
    const f32x4 sh0 = *reinterpret_cast<const f32x4*>(shift + c0), sh1 = *reinterpret_cast<const f32x4*>(shift + c0 + 4);
    f32x4 k0 = {1.f, 1.f, 1.f, 1.f}, k1 = k0;
    if (keep) {
      k0 = *reinterpret_cast<const f32x4*>(keep + (long)b * C1 + c0);
      k1 = *reinterpret_cast<const f32x4*>(keep + (long)b * C1 + c0 + 4);
    }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float z = fmaf(to_f32(v.v[e]), e < 4 ? sc0[e & 3] : sc1[e & 3], e < 4 ? sh0[e & 3] : sh1[e & 3]);
      z = fmaxf(z, 0.f);
      o.v[e] = from_f32<T>(z * (e < 4 ? k0[e & 3] : k1[e & 3]));
    }
    *reinterpret_cast<Vec16<T>*>(patch + px * PB + sl * 16) = o;
  }
  __syncthreads();
  const int H = 2 * h, W = 2 * w;
  for (int idx = threadIdx.x; idx < TP * TP * VPP; idx += 256) {
    const int pix = idx / VPP, vq = idx - pix * VPP;
    const int i = i0 + (pix >> 3), j = j0 + (pix & 7);
    if (i >= h || j >= w) continue;
    T* o = out + (((long)b * H + 2 * i) * W + 2 * j) * ldo + cbase + vq * 8;
    auto at = [&](int yy, int xx) {
      return *reinterpret_cast<const Vec16<T>*>(patch + ((yy - (i0 - 1)) * HP + (xx - (j0 - 1))) * PB + vq * 16);
    };
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      int y0, y1;
      float ly;
      src_index(0.5f, 2 * i + a, h, y0, y1, ly);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        int x0, x1;
        float lx;
        src_index(0.5f, 2 * j + c, w, x0, x1, lx);
        store16(o + ((long)a * W + c) * ldo, bilerp(at(y0, x0), at(y0, x1), at(y1, x0), at(y1, x1), ly, lx));
      }
    }
  }
}

// dlow[b,i,j,:] = sum over output pixels (y,x) that read (i,j) of wy*wx*dout[b,y,x,:] (gather form)
template <typename T>
__global__ void upsample_bwd_kernel(const T* __restrict__ dout, int ldout, int H, int W, T* __restrict__ dlow,
                                    int B, int h, int w, int C1, float sy, float sx) {
  constexpr int E = Vec16<T>::N;
  const int CG = C1 / E;
  // Neighbouring low-resolution rows gather from overlapping rows of dout: each XCD (workgroups
  // are dealt to them round-robin) takes a contiguous range of blocks so the overlap stays in its L2.
  const unsigned idx = (unsigned)xcd_contiguous_block() * blockDim.x + threadIdx.x;   // fits 31 bits (host check)
  if (idx >= (unsigned)B * h * w * CG) return;
  const int cg = idx % CG;
  const unsigned p = idx / CG;
  const int j = p % w;
  const int i = (p / w) % h;
  const int b = p / ((unsigned)w * h);
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = 0.f;
  int ylo = (int)floorf(((float)i - 0.5f) / sy - 0.5f) - 1;
  int yhi = (int)ceilf(((float)i + 1.5f) / sy - 0.5f) + 1;
  int xlo = (int)floorf(((float)j - 0.5f) / sx - 0.5f) - 1;
  int xhi = (int)ceilf(((float)j + 1.5f) / sx - 0.5f) + 1;
  if (ylo < 0) ylo = 0;
  if (xlo < 0) xlo = 0;
  if (yhi > H - 1) yhi = H - 1;
  if (xhi > W - 1) xhi = W - 1;
  // the column weights do not depend on the row: computed once (up to 8 candidate columns, which
  // covers every up-sampling ratio; wider windows take the generic path below)
  constexpr int MAXC = 8;
  const int ncol = xhi - xlo + 1;
  const T* base = dout + (long)b * H * W * ldout + cg * E;
  if (ncol <= MAXC) {
    float wxs[MAXC];
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      int x0, x1;
      float lx;
      src_index(sx, xlo + k, w, x0, x1, lx);
      wxs[k] = (k < ncol) ? (x0 == j ? 1.f - lx : 0.f) + (x1 == j ? lx : 0.f) : 0.f;
    }
    for (int y = ylo; y <= yhi; ++y) {
      int y0, y1;
      float ly;
      src_index(sy, y, h, y0, y1, ly);
      const float wy = (y0 == i ? 1.f - ly : 0.f) + (y1 == i ? ly : 0.f);
      if (wy == 0.f) continue;
      const T* row = base + ((long)y * W + xlo) * ldout;
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
        if (wxs[k] == 0.f) continue;
        Vec16<T> v = load16(row + (long)k * ldout);
        const float wgt = wy * wxs[k];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += wgt * to_f32(v.v[e]);
      }
    }
  } else {
    for (int y = ylo; y <= yhi; ++y) {
      int y0, y1;
      float ly;
      src_index(sy, y, h, y0, y1, ly);
      const float wy = (y0 == i ? 1.f - ly : 0.f) + (y1 == i ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int x = xlo; x <= xhi; ++x) {
        int x0, x1;
        float lx;
        src_index(sx, x, w, x0, x1, lx);
        const float wx = (x0 == j ? 1.f - lx : 0.f) + (x1 == j ? lx : 0.f);
        if (wx == 0.f) continue;
        Vec16<T> v = load16(base + ((long)y * W + x) * ldout);
        const float wgt = wy * wx;
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += wgt * to_f32(v.v[e]);
      }
    }
  }
  Vec16<T> o;
#pragma unroll
  for (int e = 0; e < E; ++e) o.v[e] = from_f32<T>(acc[e]);
  store16(dlow + (size_t)p * C1 + cg * E, o);
}

// ------------------------------------------------------------------ planar fp32 resize (model outputs)
// F.interpolate(outputs, size=labels.shape[-2:], mode='bilinear', align_corners=False) on the
// (B,1,h,w) fp32 prediction, swin_unet_model.py:812-814: one thread per output pixel.
__global__ void resize_planes_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, long total, int h,
                                         int w, int H, int W, float sy, float sx) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int x = idx % W;
  const int y = (idx / W) % H;
  const long pl = idx / ((long)W * H);
  int y0, y1, x0, x1;
  float ly, lx;
  src_index(sy, y, h, y0, y1, ly);
  src_index(sx, x, w, x0, x1, lx);
  const float* base = in + pl * h * w;
  const float hy = 1.f - ly, hx = 1.f - lx;
  out[idx] = hy * (hx * base[(long)y0 * w + x0] + lx * base[(long)y0 * w + x1]) +
             ly * (hx * base[(long)y1 * w + x0] + lx * base[(long)y1 * w + x1]);
}

// din[pl,i,j] = sum over the output pixels (y,x) that read (i,j) of wy*wx*dout[pl,y,x]: a gather in a
// fixed order, so the result does not depend on scheduling.
__global__ void resize_planes_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, long total, int h,
                                         int w, int H, int W, float sy, float sx) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int j = idx % w;
  const int i = (idx / w) % h;
  const long pl = idx / ((long)w * h);
  int ylo = (int)floorf(((float)i - 0.5f) / sy - 0.5f) - 1;
  int yhi = (int)ceilf(((float)i + 1.5f) / sy - 0.5f) + 1;
  int xlo = (int)floorf(((float)j - 0.5f) / sx - 0.5f) - 1;
  int xhi = (int)ceilf(((float)j + 1.5f) / sx - 0.5f) + 1;
  if (ylo < 0) ylo = 0;
  if (xlo < 0) xlo = 0;
  if (yhi > H - 1) yhi = H - 1;
  if (xhi > W - 1) xhi = W - 1;
  const float* base = dout + pl * H * W;
  float acc = 0.f;
  for (int y = ylo; y <= yhi; ++y) {
    int y0, y1;
    float ly;
    src_index(sy, y, h, y0, y1, ly);
    const float wy = (y0 == i ? 1.f - ly : 0.f) + (y1 == i ? ly : 0.f);
    if (wy == 0.f) continue;
    for (int x = xlo; x <= xhi; ++x) {
      int x0, x1;
      float lx;
      src_index(sx, x, w, x0, x1, lx);
      const float wx = (x0 == j ? 1.f - lx : 0.f) + (x1 == j ? lx : 0.f);
      if (wx == 0.f) continue;
      acc += wy * wx * base[(long)y * W + x];
    }
  }
  din[idx] = acc;
}

// ------------------------------------------------------------------ final 1x1 convolution (few outputs)
// out[b,o,hw] = bias[o] + sum_c x[b,hw,c] * w[o,c]      (out is NCHW fp32)
constexpr int HEAD_PIX = 4;
template <typename T>
__global__ __launch_bounds__(256) void head1x1_fwd_kernel(const T* __restrict__ x, int ldx,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ out, long npix, long hw, int C,
                                                          int Co) {
  // Threads are (pixel lane, channel group): consecutive threads read consecutive 16-byte vectors of a
  // pixel row, so the loads are coalesced; a pixel's partial dot products meet in LDS and its first
  // thread adds them in channel order.  A workgroup takes HEAD_PIX groups of pixels at once: their loads
  // are in flight together and share the two barriers.
  constexpr int E = Vec16<T>::N;
  extern __shared__ float wl[];              // [Co][C] weights, then [HEAD_PIX][256] partial sums
  float* red = wl + Co * C;
  for (int i = threadIdx.x; i < Co * C; i += blockDim.x) wl[i] = w[i];
  const int CG = C / E;
  const int lanes = 256 / CG;
  const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
  const long p0 = (long)blockIdx.x * lanes * HEAD_PIX + pl;
  Vec16<T> v[HEAD_PIX];
#pragma unroll
  for (int k = 0; k < HEAD_PIX; ++k) {
    const long p = p0 + (long)k * lanes;
    v[k] = (pl < lanes && p < npix) ? load16(x + p * ldx + cg * E) : zero16<T>();
  }
  __syncthreads();
  for (int o = 0; o < Co; ++o) {
#pragma unroll
    for (int k = 0; k < HEAD_PIX; ++k) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) s += to_f32(v[k].v[e]) * wl[o * C + cg * E + e];
      red[k * 256 + threadIdx.x] = s;
    }
    __syncthreads();
    if (pl < lanes && cg == 0) {
#pragma unroll
      for (int k = 0; k < HEAD_PIX; ++k) {
        const long p = p0 + (long)k * lanes;
        if (p >= npix) continue;
        float t = bias ? bias[o] : 0.f;
        for (int j = 0; j < CG; ++j) t += red[k * 256 + threadIdx.x + j];
        const long b = p / hw;
        out[(b * Co + o) * hw + (p - b * hw)] = t;
      }
    }
    __syncthreads();
  }
}

// dx[b,hw,c] = sum_o dout[b,o,hw] * w[o,c]
template <typename T>
__global__ void head1x1_bwd_data_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                        T* __restrict__ dx, long npix, long hw, int C, int Co) {
  constexpr int E = Vec16<T>::N;
  const int CG = C / E;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= npix * CG) return;
  const int cg = idx % CG;
  const long p = idx / CG;
  const int b = p / hw;
  const long r = p - (long)b * hw;
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = 0.f;
  for (int o = 0; o < Co; ++o) {
    const float d = dout[((long)b * Co + o) * hw + r];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] += d * w[o * C + cg * E + e];
  }
  Vec16<T> ov;
#pragma unroll
  for (int e = 0; e < E; ++e) ov.v[e] = from_f32<T>(acc[e]);
  store16(dx + p * C + cg * E, ov);
}

inline int reduce_blocks(long npix) {
  long n = (npix + 127) / 128;
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

#define CHECK_DTYPE(name, dtype) \
  UNETK_CHECK_ARG((dtype) == UNETK_F32 || (dtype) == UNETK_BF16, name ": bad dtype %d", (dtype))
#define CHECK_VEC(name, dtype, C) \
  UNETK_CHECK_ARG((C) % ((dtype) == UNETK_BF16 ? 8 : 4) == 0, name ": channel count %d not a multiple of %d", (C), ((dtype) == UNETK_BF16 ? 8 : 4))

// launch of the tiled 2x resize (with or without the BatchNorm + ReLU in front of it)
template <bool BN>
static int launch_upsample2x_tile(const void* y, int B, int h, int w, int C1, const float* scale, const float* shift,
                                  const float* keep, void* out, int ldo, hipStream_t stream) {
  const int tiles_x = cdiv(w, 8), tiles_y = cdiv(h, 8);
  // slabs a workgroup takes side by side: two where the channels allow it, else three (measured at batch 64: 128-byte
  // runs per pixel beat 64-byte ones by 10-25 %, and 192-byte ones by 5-20 %; a 96-channel tensor gains 12 % from
  // three over one)
  const int nb = C1 / 32;
  int nsl = nb % 2 == 0 ? 2 : (nb % 3 == 0 ? 3 : 1);
  if (const char* ev = std::getenv("UNETK_BN_UPSAMPLE_SLABS")) {
    const int v = atoi(ev);
    if ((v == 1 || v == 2 || v == 3) && nb % v == 0) nsl = v;
  }
  const long nitems = (long)B * tiles_x * tiles_y * (nb / nsl);
  UNETK_CHECK_ARG(nitems < (1L << 30), "upsample2x (tiled): too many tiles");
  const dim3 grid((unsigned)(8 * cdiv(nitems, 8)));
#define BN_UP(N)                                                                                                      \
  hipLaunchKernelGGL((bn_act_upsample2x_tile_kernel<N, BN>), grid, dim3(256), 0, stream, (const bf16*)y, h, w, C1, scale, \
                     shift, keep, (bf16*)out, ldo, tiles_x, tiles_y, nitems)
  if (nsl == 3) BN_UP(3);
  else if (nsl == 2) BN_UP(2);
  else BN_UP(1);
#undef BN_UP
  return UNETK_OK;
}

extern "C" {

int unetk_nchw_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int ld, int dtype,
                       hipStream_t stream) {
  UNETK_CHECK_ARG(src && dst && ld >= C, "nchw_to_nhwc: bad arguments");
  CHECK_DTYPE("nchw_to_nhwc", dtype);
  const long total = (long)B * H * W * ld;
  if (ld == (dtype == UNETK_BF16 ? 8 : 4)) {
    const long npix = (long)B * H * W;
    DISPATCH_T(dtype, hipLaunchKernelGGL(nchw_to_nhwc_vec_kernel<T>, dim3(cdiv(npix, 256)), dim3(256), 0, stream,
                                         src, (T*)dst, npix, C, (long)H * W));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(cdiv(total, 256)), dim3(256), 0, stream,
                                         src, (T*)dst, B, C, H, W, ld));
  }
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_nhwc_to_nchw(const void* src, int ld, float* dst, int B, int C, int H, int W, int dtype,
                       hipStream_t stream) {
  UNETK_CHECK_ARG(src && dst && ld >= C, "nhwc_to_nchw: bad arguments");
  CHECK_DTYPE("nhwc_to_nchw", dtype);
  const long total = (long)B * H * W * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(cdiv(total, 256)), dim3(256), 0, stream,
                                       (const T*)src, dst, B, C, H, W, ld));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_bn_finalize(float* partials, int nparts, int C, int64_t count, const float* gamma,
                      const float* beta, float eps, float momentum, float* running_mean,
                      float* running_var, float* mean, float* invstd, float* scale, float* shift,
                      hipStream_t stream) {
  UNETK_CHECK_ARG(partials && mean && invstd && scale && shift && nparts > 0 && C > 0 && count > 0,
                  "bn_finalize: bad arguments");
  int G = 0;
  if (nparts >= 1024) {   // long lists: a wide first level, in place
    G = 64;
    hipLaunchKernelGGL(bn_prereduce_kernel, dim3(cdiv(nparts, G), cdiv(2 * C, 256)), dim3(256), 0, stream,
                       partials, nparts, 2 * C, G);
    UNETK_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, stream, partials, nparts, G, C,
                     (double)count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd,
                     scale, shift);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, const float* conv_bias, float* scale,
                       float* shift, int C, hipStream_t stream) {
  UNETK_CHECK_ARG(running_mean && running_var && scale && shift && C > 0, "bn_fold_eval: bad arguments");
  hipLaunchKernelGGL(bn_fold_eval_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, gamma, beta,
                     running_mean, running_var, eps, conv_bias, scale, shift, C);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_bn_act_fwd(const void* y, int ldy, void* a, int lda, void* pooled, unsigned char* argmax, int B, int H, int W,
                     int C, const float* scale, const float* shift, const float* keep, int relu, int dtype,
                     hipStream_t stream) {
  UNETK_CHECK_ARG(y && (a || pooled), "bn_act_fwd: null pointer");
  UNETK_CHECK_ARG(argmax == nullptr || pooled != nullptr, "bn_act_fwd: argmax needs the pooled output");
  CHECK_DTYPE("bn_act_fwd", dtype);
  CHECK_VEC("bn_act_fwd", dtype, C);
  const int E = dtype == UNETK_BF16 ? 8 : 4;
  UNETK_CHECK_ARG(ldy >= C && ldy % E == 0 && ((uintptr_t)y % 16) == 0, "bn_act_fwd: input stride %d", ldy);
  UNETK_CHECK_ARG(a == nullptr || (lda >= C && lda % E == 0 && ((uintptr_t)a % 16) == 0),
                  "bn_act_fwd: output stride %d (>= C, multiple of %d, 16-byte aligned base)", lda, E);
  const long npix = (long)B * H * W;
  if (pooled == nullptr && C / E <= 256 && npix < (1L << 31)) {
    const int lanes = 256 / (C / E);
    long nb = (npix + (long)lanes * 8 - 1) / ((long)lanes * 8);
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    const unsigned chunk = (unsigned)((npix + nb - 1) / nb);
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_act_stream_kernel<T>, dim3((unsigned)cdiv(npix, chunk)), dim3(256), 0,
                                         stream, (const T*)y, ldy, (T*)a, lda, (unsigned)npix,
                                         (unsigned)((long)H * W), C, scale, shift, keep, relu, chunk));
    UNETK_CHECK_LAUNCH();
    return UNETK_OK;
  }
  const long nquads = (long)B * ((H + 1) / 2) * ((W + 1) / 2);
  UNETK_CHECK_ARG(nquads * (C / E) < (1L << 31) && npix < (1L << 31) && C / E <= 256, "bn_act_fwd: tensor too large");
  {
    const int lanes = 256 / (C / E);
    long nb = (nquads + (long)lanes * 4 - 1) / ((long)lanes * 4);
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    const unsigned chunk = (unsigned)((nquads + nb - 1) / nb);
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_act_fwd_kernel<T>, dim3((unsigned)cdiv(nquads, chunk)), dim3(256), 0, stream,
                                         (const T*)y, ldy, (T*)a, lda, (T*)pooled, argmax, B, H, W, C, scale, shift,
                                         keep, relu, chunk));
  }
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_reduce_num_partials(int64_t npix) { return reduce_blocks(npix); }


static int bn_act_bwd_impl(const void* g, const float* g1_d, const float* g1_w, const void* y, void* dy, int B, int H,
                           int W, int C, const float* scale, const float* shift, const float* mean,
                           const float* invstd, const float* keep, int relu, float* dgamma, float* dbeta,
                           int accumulate, float* partials, float* sums, int dtype, hipStream_t stream,
                           const void* dpooled = nullptr, const unsigned char* argmax = nullptr, int ldg = 0,
                           int nparts_given = 0) {
  UNETK_CHECK_ARG((g || (g1_d && g1_w)) && y && dy && scale && shift && mean && invstd && partials && sums,
                  "bn_act_bwd: null pointer");
  CHECK_DTYPE("bn_act_bwd", dtype);
  CHECK_VEC("bn_act_bwd", dtype, C);
  const int E = dtype == UNETK_BF16 ? 8 : 4;
  UNETK_CHECK_ARG(C / E <= RED_THREADS, "bn_act_bwd: too many channels (%d)", C);
  const long npix = (long)B * H * W;
  UNETK_CHECK_ARG(npix < (1L << 31), "bn_act_bwd: too many pixels");
  // (nparts_given: the caller's `partials` already hold that many rows of the two sums, from the kernel that produced g)
  const int nblk = nparts_given > 0 ? nparts_given : reduce_blocks(npix);
  if (nparts_given <= 0) {
    RedArgs r;
    r.x = y; r.g = g; r.scale = scale; r.shift = shift; r.mean = mean; r.invstd = invstd; r.keep = keep;
    r.partials = partials; r.npix = npix; r.hw = (long)H * W; r.C = C; r.ldx = C; r.ldg = dpooled ? ldg : C;
    r.chunk = cdiv(npix, nblk); r.relu = relu; r.g1_d = g1_d; r.g1_w = g1_w;
    r.dpooled = dpooled; r.argmax = argmax; r.H = H; r.W = W;
    DISPATCH_T(dtype, hipLaunchKernelGGL((channel_reduce_kernel<T, 0>), dim3(nblk), dim3(RED_THREADS), 0,
                                         stream, r));
    UNETK_CHECK_LAUNCH();
  }
  // sums[0:C] = sum dz (= dbeta), sums[C:2C] = sum dz*xhat (= dgamma); the coefficients live behind
  // the two sum rows: sums must hold 6*C floats
  float* coef = sums + 2 * C;
  hipLaunchKernelGGL(bn_bwd_sums_coef_kernel, dim3(cdiv(C, 4)), dim3(256), 0, stream, partials, nblk, C, sums, dbeta,
                     dgamma, accumulate, scale, shift, mean, invstd, 1.0f / (float)npix, coef);
  UNETK_CHECK_LAUNCH();
  {
    const int lanes = RED_THREADS / (C / E);
    long nb = (npix + (long)lanes * 8 - 1) / ((long)lanes * 8);
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    const unsigned chunk = (unsigned)((npix + nb - 1) / nb);
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_act_bwd_apply_kernel<T>, dim3((unsigned)cdiv(npix, chunk)),
                                         dim3(RED_THREADS), 0, stream, (const T*)g, (const T*)y, (T*)dy,
                                         (unsigned)npix, (unsigned)((long)H * W), C, coef, keep, relu, chunk,
                                         g1_d, g1_w, (const T*)dpooled, argmax, dpooled ? ldg : C, H, W));
  }
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_bn_act_bwd(const void* g, const void* y, void* dy, int B, int H, int W, int C,
                     const float* scale, const float* shift, const float* mean, const float* invstd,
                     const float* keep, int relu, float* dgamma, float* dbeta, int accumulate,
                     float* partials, float* sums, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(g != nullptr, "bn_act_bwd: null gradient");
  return bn_act_bwd_impl(g, nullptr, nullptr, y, dy, B, H, W, C, scale, shift, mean, invstd, keep, relu, dgamma, dbeta,
                         accumulate, partials, sums, dtype, stream);
}

int unetk_bn_act_bwd_presummed(const void* g, const void* y, void* dy, int B, int H, int W, int C, const float* scale,
                               const float* shift, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                               int accumulate, const float* partials, int nparts, float* sums, int dtype,
                               hipStream_t stream) {
  UNETK_CHECK_ARG(g != nullptr && nparts > 0, "bn_act_bwd_presummed: null gradient or no partial sums");
  return bn_act_bwd_impl(g, nullptr, nullptr, y, dy, B, H, W, C, scale, shift, mean, invstd, nullptr, 1, dgamma, dbeta,
                         accumulate, const_cast<float*>(partials), sums, dtype, stream, nullptr, nullptr, 0, nparts);
}

int unetk_bn_act_bwd_rank1(const float* gpix, const float* gchan, const void* y, void* dy, int B, int H, int W, int C,
                           const float* scale, const float* shift, const float* mean, const float* invstd,
                           const float* keep, int relu, float* dgamma, float* dbeta, int accumulate,
                           float* partials, float* sums, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(gpix && gchan, "bn_act_bwd_rank1: null gradient factors");
  return bn_act_bwd_impl(nullptr, gpix, gchan, y, dy, B, H, W, C, scale, shift, mean, invstd, keep, relu, dgamma,
                         dbeta, accumulate, partials, sums, dtype, stream);
}

int unetk_bn_act_bwd_pool(const void* dpooled, const unsigned char* argmax, const void* dskip, int ldskip,
                          const void* y, void* dy, int B, int H, int W, int C, const float* scale, const float* shift,
                          const float* mean, const float* invstd, const float* keep, int relu, float* dgamma,
                          float* dbeta, int accumulate, float* partials, float* sums, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(dpooled && argmax && dskip, "bn_act_bwd_pool: null gradient, argmax map or skip gradient");
  UNETK_CHECK_ARG(H >= 2 && W >= 2, "bn_act_bwd_pool: image smaller than one quad");
  const int E = dtype == UNETK_BF16 ? 8 : 4;
  UNETK_CHECK_ARG(ldskip >= C && ldskip % E == 0, "bn_act_bwd_pool: skip stride %d", ldskip);
  return bn_act_bwd_impl(dskip, nullptr, nullptr, y, dy, B, H, W, C, scale, shift, mean, invstd, keep, relu, dgamma,
                         dbeta, accumulate, partials, sums, dtype, stream, dpooled, argmax, ldskip);
}

int unetk_pool_bwd(const void* a, const unsigned char* argmax, const void* dpooled, const void* dskip,
                   int ldskip, void* da, int B, int H, int W, int C, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG((a || argmax) && dpooled && da, "pool_bwd: null pointer");
  CHECK_DTYPE("pool_bwd", dtype);
  CHECK_VEC("pool_bwd", dtype, C);
  const int E = dtype == UNETK_BF16 ? 8 : 4;
  UNETK_CHECK_ARG(dskip == nullptr || ldskip % E == 0, "pool_bwd: skip stride misaligned");
  const long total = (long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / E);
  UNETK_CHECK_ARG(total < (1L << 31), "pool_bwd: tensor too large");
  DISPATCH_T(dtype, hipLaunchKernelGGL(pool_bwd_kernel<T>, dim3(cdiv(total, 256)), dim3(256), 0, stream,
                                       (const T*)a, argmax, (const T*)dpooled, (const T*)dskip, ldskip,
                                       (T*)da, B, H, W, C));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_upsample_concat_fwd(const void* low, int h, int w, int C1, const void* skip, int ldskip, int C2,
                              void* out, int B, int H, int W, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(low && out, "upsample_concat_fwd: null pointer");
  CHECK_DTYPE("upsample_concat_fwd", dtype);
  CHECK_VEC("upsample_concat_fwd", dtype, C1);
  CHECK_VEC("upsample_concat_fwd", dtype, C2);
  const int E = dtype == UNETK_BF16 ? 8 : 4;
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const long total = (long)B * H * W * ((skip ? C1 + C2 : C1) / E);
  UNETK_CHECK_ARG(total < (1L << 31), "upsample_concat_fwd: tensor too large");
  if (H == 2 * h && W == 2 * w && skip == nullptr && dtype == UNETK_BF16 && C1 % 32 == 0 && (C1 + C2) % 8 == 0 &&
      ((uintptr_t)low % 16) == 0 && ((uintptr_t)out % 16) == 0 && std::getenv("UNETK_UPSAMPLE_NO_TILE") == nullptr) {
    // the LDS-tiled form (each low-resolution pixel fetched once per workgroup instead of by nine threads)
    const int rc = launch_upsample2x_tile<false>(low, B, h, w, C1, nullptr, nullptr, nullptr, out, C1 + C2, stream);
    if (rc != UNETK_OK) return rc;
    UNETK_CHECK_LAUNCH();
    return UNETK_OK;
  }
  if (H == 2 * h && W == 2 * w) {
    const long quads = total / 4;
    DISPATCH_T(dtype, hipLaunchKernelGGL(upsample2x_concat_kernel<T>, dim3(cdiv(quads, 256)), dim3(256), 0,
                                         stream, (const T*)low, h, w, C1, (const T*)skip, ldskip, C2, (T*)out, B));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL(upsample_concat_kernel<T>, dim3(cdiv(total, 256)), dim3(256), 0,
                                         stream, (const T*)low, h, w, C1, (const T*)skip, ldskip, C2, (T*)out,
                                         B, H, W, sy, sx));
  }
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_bn_act_upsample2x_supported(int h, int w, int C1, int dtype) {
  return dtype == UNETK_BF16 && C1 % 32 == 0 && h >= 1 && w >= 1 && std::getenv("UNETK_NO_BN_UPSAMPLE") == nullptr;
}

int unetk_bn_act_upsample2x(const void* y, int B, int h, int w, int C1, const float* scale, const float* shift,
                            const float* keep, void* out, int ldo, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(y && scale && shift && out, "bn_act_upsample2x: null pointer");
  UNETK_CHECK_ARG(unetk_bn_act_upsample2x_supported(h, w, C1, dtype) && B > 0,
                  "bn_act_upsample2x: bfloat16 and a multiple of 32 channels (got dtype %d, %d channels)", dtype, C1);
  UNETK_CHECK_ARG(ldo >= C1 && ldo % 8 == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)out % 16) == 0,
                  "bn_act_upsample2x: output stride %d (>= C1, multiple of 8, 16-byte aligned bases)", ldo);
  UNETK_CHECK_ARG(((uintptr_t)scale % 16) == 0 && ((uintptr_t)shift % 16) == 0 && ((uintptr_t)keep % 16) == 0,
                  "bn_act_upsample2x: scale, shift and keep must be 16-byte aligned");
  UNETK_CHECK_ARG((long)B * 4 * h * w * ldo < (1L << 40), "bn_act_upsample2x: tensor too large");
  const int rc = launch_upsample2x_tile<true>(y, B, h, w, C1, scale, shift, keep, out, ldo, stream);
  if (rc != UNETK_OK) return rc;
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_upsample_bwd(const void* dout, int ldout, int H, int W, void* dlow, int B, int h, int w, int C1,
                       int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(dout && dlow, "upsample_bwd: null pointer");
  CHECK_DTYPE("upsample_bwd", dtype);
  CHECK_VEC("upsample_bwd", dtype, C1);
  const int E = dtype == UNETK_BF16 ? 8 : 4;
  UNETK_CHECK_ARG(ldout % E == 0, "upsample_bwd: stride misaligned");
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const long total = (long)B * h * w * (C1 / E);
  UNETK_CHECK_ARG(total < (1L << 31), "upsample_bwd: tensor too large");
  if (dtype == UNETK_BF16 && H == 2 * h && W == 2 * w && C1 % 32 == 0 && h >= 8 && w >= 8 &&
      getenv("UNETK_UPSAMPLE_BWD_GENERIC") == nullptr) {
    const int tiles_x = cdiv(w, 8), tiles_y = cdiv(h, 8);
    const long nitems = (long)B * tiles_x * tiles_y * (C1 / 32);
    hipLaunchKernelGGL(upsample2x_bwd_tile_kernel, dim3((unsigned)(8 * cdiv(nitems, 8))), dim3(256), 0, stream,
                       (const bf16*)dout, ldout, H, W, (bf16*)dlow, B, h, w, C1, tiles_x, tiles_y, nitems);
    UNETK_CHECK_LAUNCH();
    return UNETK_OK;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(upsample_bwd_kernel<T>, dim3(8 * cdiv(cdiv(total, 256), 8)), dim3(256), 0, stream,
                                       (const T*)dout, ldout, H, W, (T*)dlow, B, h, w, C1, sy, sx));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_resize_planes_f32(const float* in, float* out, int64_t planes, int h, int w, int H, int W, int backward,
                            hipStream_t stream) {
  UNETK_CHECK_ARG(in && out, "resize_planes_f32: null pointer");
  UNETK_CHECK_ARG(planes > 0 && h > 0 && w > 0 && H > 0 && W > 0, "resize_planes_f32: bad shape");
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  if (!backward) {
    const long total = planes * H * W;
    UNETK_CHECK_ARG(total < (1L << 40), "resize_planes_f32: tensor too large");
    hipLaunchKernelGGL(resize_planes_fwd_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, in, out,
                       total, h, w, H, W, sy, sx);
  } else {   // in = gradient of the (planes,H,W) result, out = gradient of the (planes,h,w) source
    const long total = planes * h * w;
    UNETK_CHECK_ARG(total < (1L << 40), "resize_planes_f32: tensor too large");
    hipLaunchKernelGGL(resize_planes_bwd_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, in, out,
                       total, h, w, H, W, sy, sx);
  }
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_head1x1_fwd(const void* x, int ldx, const float* w, const float* bias, float* out, int B, int H,
                      int W, int C, int Co, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(x && w && out, "head1x1_fwd: null pointer");
  CHECK_DTYPE("head1x1_fwd", dtype);
  CHECK_VEC("head1x1_fwd", dtype, C);
  UNETK_CHECK_ARG((long)Co * C * 4 <= 48 * 1024, "head1x1_fwd: weight (%d x %d) too large", Co, C);
  const int E = dtype == UNETK_BF16 ? 8 : 4;
  UNETK_CHECK_ARG(C / E <= 256, "head1x1_fwd: too many channels (%d)", C);
  const long npix = (long)B * H * W;
  const int lanes = 256 / (C / E);           // pixels per workgroup
  DISPATCH_T(dtype, hipLaunchKernelGGL(head1x1_fwd_kernel<T>, dim3((unsigned)cdiv(npix, (long)lanes * HEAD_PIX)), dim3(256),
                                       (size_t)(Co * C + HEAD_PIX * 256) * 4, stream, (const T*)x, ldx, w, bias, out, npix,
                                       (long)H * W, C, Co));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_head1x1_bwd(const void* x, int ldx, const float* w, const float* dout, void* dx, float* dw,
                      float* db, int B, int H, int W, int C, int Co, int accumulate, float* partials,
                      int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(x && w && dout && partials, "head1x1_bwd: null pointer");
  CHECK_DTYPE("head1x1_bwd", dtype);
  CHECK_VEC("head1x1_bwd", dtype, C);
  UNETK_CHECK_ARG(!(dw || db) || B == 1 || Co == 1,
                  "head1x1_bwd: weight gradient supports Co == 1 (or B == 1) per call");
  const int E = dtype == UNETK_BF16 ? 8 : 4;
  const long npix = (long)B * H * W;
  if (dx) {
    const long total = npix * (C / E);
    DISPATCH_T(dtype, hipLaunchKernelGGL(head1x1_bwd_data_kernel<T>, dim3(cdiv(total, 256)), dim3(256), 0,
                                         stream, dout, w, (T*)dx, npix, (long)H * W, C, Co));
    UNETK_CHECK_LAUNCH();
  }
  if (dw || db) {
    const int nblk = reduce_blocks(npix);
    for (int o = 0; o < Co; ++o) {
      RedArgs r;
      r.x = x; r.g = dout + (long)o * H * W; r.scale = r.shift = r.mean = r.invstd = r.keep = nullptr;
      r.g1_d = r.g1_w = nullptr;
      r.partials = partials; r.npix = npix; r.hw = (long)H * W; r.C = C; r.ldx = ldx; r.ldg = 1;
      r.chunk = cdiv(npix, nblk); r.relu = 0;
      DISPATCH_T(dtype, hipLaunchKernelGGL((channel_reduce_kernel<T, 1>), dim3(nblk), dim3(RED_THREADS), 0,
                                           stream, r));
      UNETK_CHECK_LAUNCH();
      // out0 = sum d*x per channel -> dw[o][:]; out1 (same value in every channel) -> db[o] via channel 0
      hipLaunchKernelGGL(partial_sum_kernel, dim3(cdiv(C, 4)), dim3(256), 0, stream, partials, nblk, C, C,
                         dw ? dw + (long)o * C : nullptr, (float*)nullptr, accumulate);
      UNETK_CHECK_LAUNCH();
      if (db) {
        hipLaunchKernelGGL(partial_sum_kernel, dim3(1), dim3(256), 0, stream, partials, nblk, C, 1,
                           (float*)nullptr, db + o, accumulate);
        UNETK_CHECK_LAUNCH();
      }
    }
  }
  return UNETK_OK;
}

}  // extern "C"
