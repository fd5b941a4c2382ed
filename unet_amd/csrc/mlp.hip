// The MLP branch of a Swin block in ONE launch (gfx950, bfloat16 in, fp32 accumulate):
//
//   out = x + keep[sample] * ( fc2( GELU( fc1( LayerNorm(x) ) ) ) )
//
// reference swin_unet_model.py:372 (x = x + drop_path(mlp(norm2(x)))) with :296-303 (Linear C -> 4C, GELU(erf),
// Linear 4C -> C) and nn.LayerNorm(C).  For the narrow stages (C = 96 and 192 over 65 536 ... 262 144 tokens) the
// four separate launches (LayerNorm, fc1, GELU, fc2) are bound by the bytes of the hidden tensor they hand to
// one another through HBM: 4C values per token written by fc1, read and written by GELU, read by fc2.  Here a
// workgroup keeps its token tile and the hidden values on chip:
//
//  * A workgroup of 4 waves owns 64*TMW tokens; wave w owns tokens [16*TMW*w, 16*TMW*(w+1)) from the first load to
//    the last store, so nothing about the tokens crosses waves: the wave loads its own rows by LDS-DMA, normalises
//    them in place (the arithmetic of the stand-alone LayerNorm kernel, lane for lane: same bits), multiplies them,
//    writes its GELU tile to LDS and reads it back as the operand of fc2.  Only the weight slabs are shared.
//  * The hidden axis runs in chunks of 64 channels.  Per chunk: fc1 (K = C) into 4 accumulator tiles per 16 tokens,
//    bias, rounding to bf16 = the pre-activation the backward pass needs (stored once), GELU of the rounded value
//    (stored, optionally: the weight gradient of fc2 wants it) into an LDS tile, and at once fc2's partial product
//    over those 64 channels into the C/16 output accumulator tiles the wave keeps for the whole launch.
//  * Both weight matrices arrive in the packed layout of the token GEMM ([N/BN][K/32][BN][32], BN = 64 for fc1 and
//    96 for fc2): the slab of a hidden chunk is one contiguous run (fc1) or one run per 96-channel output block
//    (fc2).  Each has ONE LDS buffer: fc1's is refilled while fc2 of the same chunk runs and the other way round.
//  * MFMA operands are swapped as in the token GEMM (weights = A, tokens = B) with the same permuted weight row
//    order, so a lane ends up with eight consecutive channels of one token: every epilogue works on registers and
//    stores 16 bytes per lane.
//
// LDS: token tile 64*TMW*C*2 + GELU tile 64*TMW*128 + fc1 slab C*128 + fc2 slab C*128 bytes = 64 KB for
// (C = 96, TMW = 2) and 80 KB for (C = 192, TMW = 1): two workgroups per CU in both cases.  The launch moves
// per token C*2 (x) + C*2 (LayerNorm output, kept for the backward) + 4C*2 (pre-activation) [+ 4C*2 (GELU)] + C*2 (x
// again, the residual) + C*2 (out) bytes: HBM-bound by a factor of ~7 over its MFMA time.
#include "common.h"
#include "epilogue.h"
#include "mma.h"

namespace {

__device__ __attribute__((aligned(64))) unsigned int mlp_zero_page[16];

struct MlpArgs {
  const bf16* x;        // [M][C] dense
  const float* gamma;   // [C]
  const float* beta;    // [C]
  const bf16* w1;       // packed [4C/64][C/32][64][32]
  const float* b1;      // [4C]
  const bf16* w2;       // packed [C/96][4C/32][96][32]
  const float* b2;      // [C]
  const float* keep;    // per-sample DropPath factor or null
  bf16* ln;             // [M][C]   LayerNorm(x)
  float* mean;          // [M]
  float* rstd;          // [M]
  bf16* hpre;           // [M][4C]  fc1 + bias, before GELU
  bf16* hact;           // [M][4C]  GELU(hpre) or null
  bf16* out;            // [M][C]
  float eps;
  int M, tok_per_sample;
};

// GELU(erf) with the polynomial of epilogue.h, the reciprocal by the hardware approximation (1 ulp: far below the
// bf16 rounding of the result; the IEEE-exact reciprocal costs ten instructions per value, and this launch is bound
// by its vector instructions: 4C GELUs per token)
__device__ __forceinline__ float mlp_gelu(float v) {
  const float z = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float e = __expf(-z * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float h = 0.5f * poly * e;                  // 0.5 * (1 - erf|.|)
  return v * (v < 0.f ? h : 1.0f - h);
}

template <int CTRL> __device__ __forceinline__ float mlp_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row, then over G / 16 rows: the tree of the stand-alone LayerNorm kernel
template <int G> __device__ __forceinline__ float mlp_group_sum(float v) {
  v += mlp_dpp<0xB1>(v);
  v += mlp_dpp<0x4E>(v);
  v += mlp_dpp<0x141>(v);
  v += mlp_dpp<0x140>(v);
  if (G >= 32) v += __shfl_xor(v, 16, 64);
  return v;
}

template <int C, int TMW>
__global__ __launch_bounds__(256) void swin_mlp_fwd_kernel(MlpArgs a) {
  static_assert(C == 96 || C == 192, "narrow Swin stages");
  constexpr int H = 4 * C, HC = 64, NHC = H / HC;
  constexpr int BM = 64 * TMW, WT = 16 * TMW;          // tokens per workgroup / per wave
  constexpr int KC1 = C / 32;                          // k-chunks of fc1
  constexpr int NB2 = C / 96, NT2 = C / 16;            // output blocks / accumulator tiles of fc2
  constexpr int XT_BYTES = KC1 * BM * 64, HT_BYTES = 2 * BM * 64;
  constexpr int W1_BYTES = KC1 * HC * 64, W2_BYTES = NB2 * 2 * 96 * 64;
  constexpr int W1_PIECES = W1_BYTES / 1024, W2_PIECES = W2_BYTES / 1024;
  constexpr int G = (C == 96) ? 16 : 32;               // lanes per token in the LayerNorm
  constexpr int CG = C / 8;                            // 16-byte vectors per token
  __shared__ __attribute__((aligned(1024))) char lds[XT_BYTES + HT_BYTES + W1_BYTES + W2_BYTES];
  char* const xt = lds;
  char* const ht = lds + XT_BYTES;
  char* const w1b = ht + HT_BYTES;
  char* const w2b = w1b + W1_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * BM;
  const int l15 = lane & 15, q = lane >> 4;
  const bf16* zero = reinterpret_cast<const bf16*>(mlp_zero_page);

  // ---- the wave's own token rows: pieces of 16 rows x 64 bytes per k-chunk, slot swizzle on the source side
#pragma unroll
  for (int i = 0; i < TMW; ++i) {
    const int row = wave * WT + 16 * i + (lane >> 2);
    const int slot = (lane & 3) ^ swz(row);
    const bool ok = m0 + row < a.M;
    const bf16* src = a.x + (long)(m0 + row) * C + slot * 8;
#pragma unroll
    for (int kc = 0; kc < KC1; ++kc)
      glds16e(ok ? src + kc * 32 : zero, xt + kc * (BM * 64) + (wave * WT + 16 * i) * 64);
  }
  // ---- weight slabs (shared): piece j of a slab is 16 rows x 64 bytes; slot swizzle keyed on the piece's parity
  const int wrel = (lane >> 2) * 32;
  auto issue_w1 = [&](int hc) {
    const bf16* src = a.w1 + (long)hc * (KC1 * HC * 32);
    for (int j = wave; j < W1_PIECES; j += 4) {
      const int jj = j & 3;                             // piece inside its k-chunk (64 rows = 4 pieces)
      glds16e(src + j * 512 + wrel + (((lane & 3) ^ (2 * (jj & 1))) * 8), w1b + j * 1024);
    }
  };
  auto issue_w2 = [&](int hc) {
    for (int j = wave; j < W2_PIECES; j += 4) {
      const int nb = j / 12, r = j - nb * 12;           // 12 pieces per output block: 2 k-chunks x 6 pieces
      const int jj = r % 6;
      const bf16* src = a.w2 + ((long)nb * (H / 32) + 2 * hc) * (96 * 32) + r * 512;
      glds16e(src + wrel + (((lane & 3) ^ (2 * (jj & 1))) * 8), w2b + j * 1024);
    }
  };
  issue_w1(0);
  issue_w2(0);

  // ---- LayerNorm of the wave's rows, in place.  G lanes per token; lane gl < CG owns vector gl of the token.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (own rows and the first slabs; the slabs wait for the barrier)
  {
    const int gl = lane % G, tg = lane / G;
    constexpr int TPI = 64 / G;                         // tokens per pass of the wave
#pragma unroll
    for (int it = 0; it < WT / TPI; ++it) {
      const int row = wave * WT + it * TPI + tg;
      const int kc = gl >> 2, s = gl & 3;
      char* p = xt + kc * (BM * 64) + row * 64 + ((s ^ swz(row)) * 16);
      Vec16<bf16> v = zero16<bf16>();
      float sum = 0.f;
      if (gl < CG) {
        v = __builtin_bit_cast(Vec16<bf16>, *reinterpret_cast<const u32x4*>(p));
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += to_f32(v.v[e]);
      }
      const float mean = mlp_group_sum<G>(sum) / (float)C;
      float sq = 0.f;
      if (gl < CG) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = to_f32(v.v[e]) - mean;
          sq += d * d;
        }
      }
      const float rstd = 1.0f / sqrtf(mlp_group_sum<G>(sq) / (float)C + a.eps);
      const long tok = (long)m0 + row;
      if (gl < CG) {
        Vec16<bf16> o;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          o.v[e] = from_f32<bf16>((to_f32(v.v[e]) - mean) * rstd * a.gamma[gl * 8 + e] + a.beta[gl * 8 + e]);
        *reinterpret_cast<u32x4*>(p) = __builtin_bit_cast(u32x4, o);
        if (tok < a.M) store16(a.ln + tok * C + gl * 8, o);
      }
      if (gl == 0 && tok < a.M) {
        a.mean[tok] = mean;
        a.rstd[tok] = rstd;
      }
    }
  }

  // ---- fragment addresses (the token GEMM's: see gemm.hip)
  const int x_lane = (wave * WT + l15) * 64 + ((q ^ (((l15 >> 2) & 1) << 1)) * 16);
  const int w_lane = (8 * (l15 >> 2) + (l15 & 3)) * 64 + ((q ^ (2 * (l15 >> 3))) * 16);

  f32x4 acc2[TMW][NT2];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < NT2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int hc = 0; hc < NHC; ++hc) {
    // fc1 slab of this chunk has landed for every wave (and everyone's LayerNorm rows are their own business)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    f32x4 acc1[TMW][4];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KC1; ++kc) {
      bf16x8 xf[TMW], wf[4];
#pragma unroll
      for (int i = 0; i < TMW; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(xt + kc * (BM * 64) + x_lane + i * 16 * 64);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        wf[j] = *reinterpret_cast<const bf16x8*>(w1b + kc * (HC * 64) + w_lane + (32 * (j >> 1) + 4 * (j & 1)) * 64);
#pragma unroll
      for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc1[i][j], 0, 0, 0);
    }
    // everyone is done with the fc1 slab: the next chunk's goes out and lands behind GELU and fc2
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (hc + 1 < NHC) issue_w1(hc + 1);

    // bias, rounding, GELU: lane (q, token l15) holds hidden channels hc*64 + 32p + 8q .. +7 for pair p
    Vec16<bf16> pre[TMW][2], act[TMW][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int c0 = hc * HC + 32 * p + 8 * q;
      float b8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) b8[e] = a.b1[c0 + e];
#pragma unroll
      for (int i = 0; i < TMW; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pre[i][p].v[r] = from_f32<bf16>(acc1[i][2 * p][r] + b8[r]);
          pre[i][p].v[4 + r] = from_f32<bf16>(acc1[i][2 * p + 1][r] + b8[4 + r]);
        }
        // GELU of the ROUNDED pre-activation: what a separate pass over the stored tensor would compute
#pragma unroll
        for (int e = 0; e < 8; ++e) act[i][p].v[e] = from_f32<bf16>(mlp_gelu(to_f32(pre[i][p].v[e])));
        const int row = wave * WT + 16 * i + l15;
        *reinterpret_cast<u32x4*>(ht + p * (BM * 64) + row * 64 + ((q ^ swz(row)) * 16)) =
            __builtin_bit_cast(u32x4, act[i][p]);
      }
    }
    // fc2 slab of this chunk has landed; the GELU tile is the wave's own (in-order LDS queue)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // (the stores go out behind the wait, so that it does not sit on them)
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
      const long tok = (long)m0 + wave * WT + 16 * i + l15;
      if (tok < a.M) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int c0 = hc * HC + 32 * p + 8 * q;
          store16(a.hpre + tok * H + c0, pre[i][p]);
          if (a.hact != nullptr) store16(a.hact + tok * H + c0, act[i][p]);
        }
      }
    }
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      bf16x8 hf[TMW];
#pragma unroll
      for (int i = 0; i < TMW; ++i) hf[i] = *reinterpret_cast<const bf16x8*>(ht + kc * (BM * 64) + x_lane + i * 16 * 64);
#pragma unroll
      for (int nb = 0; nb < NB2; ++nb) {
        bf16x8 wf[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
          wf[j] = *reinterpret_cast<const bf16x8*>(w2b + (nb * 2 + kc) * (96 * 64) + w_lane + (32 * (j >> 1) + 4 * (j & 1)) * 64);
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j)
            acc2[i][nb * 6 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], hf[i], acc2[i][nb * 6 + j], 0, 0, 0);
      }
    }
    // everyone is done with the fc2 slab and with its GELU tile
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (hc + 1 < NHC) issue_w2(hc + 1);
  }

  // ---- out = x + keep * (fc2 + bias), the value rounded to bf16 before the add (as when it is stored and re-read)
#pragma unroll
  for (int i = 0; i < TMW; ++i) {
    const long tok = (long)m0 + wave * WT + 16 * i + l15;
    if (tok >= a.M) continue;
    const float keep = a.keep != nullptr ? a.keep[tok / a.tok_per_sample] : 1.f;
#pragma unroll
    for (int p = 0; p < NT2 / 2; ++p) {
      const int c0 = 32 * p + 8 * q;
      const Vec16<bf16> rv = load16(a.x + tok * C + c0);
      Vec16<bf16> o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v0 = acc2[i][2 * p][r] + a.b2[c0 + r];
        const float v1 = acc2[i][2 * p + 1][r] + a.b2[c0 + 4 + r];
        o.v[r] = from_f32<bf16>(to_f32(rv.v[r]) + keep * to_f32(from_f32<bf16>(v0)));
        o.v[4 + r] = from_f32<bf16>(to_f32(rv.v[4 + r]) + keep * to_f32(from_f32<bf16>(v1)));
      }
      store16(a.out + tok * C + c0, o);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// The data-gradient chain of the same branch in one launch:
//
//   dhp = (dy @ w2) * GELU'(hpre)          (gradient of the pre-activation; stored: fc1's weight gradient reads it)
//   dln = dhp @ w1                         (gradient of the LayerNorm output, rounded to bf16 as when it is stored)
//   dx  = LayerNorm'(dln; x, mean, rstd, gamma) + add      [and dx * keep[sample] as a second output]
//   partials[workgroup][0][c] = sum_tokens dln * xhat,  partials[workgroup][1][c] = sum_tokens dln
//
// (autograd of swin_unet_model.py:372 / :296-303).  dy is the gradient of the branch's output (DropPath scale
// already applied).  The structure is the forward kernel's with the roles of the matrices swapped: the wave's token
// tile holds dy, "fc1" is the product with fc2's weight in its data-gradient pack (K = C, 64 hidden channels per
// chunk), the chunk's epilogue multiplies by GELU'(hpre) instead of applying GELU, "fc2" is the product with fc1's
// weight in its data-gradient pack (K = 64 per chunk, N = C), and the last epilogue is the LayerNorm backward: the
// 96 or 192 channels of a token sit in the four lanes that share its MFMA column, so its two sums are two shuffles.
struct MlpBwdArgs {
  const bf16* dy;       // [M][C]
  const bf16* hpre;     // [M][4C]
  const bf16* w2t;      // data-gradient pack of fc2 (N = 4C, K = C), 64-channel blocks: [4C/64][C/32][64][32]
  const bf16* w1t;      // data-gradient pack of fc1 (N = C, K = 4C), 96-channel blocks: [C/96][4C/32][96][32]
  const bf16* x;        // [M][C] input of the LayerNorm
  const float* mean;    // [M]
  const float* rstd;    // [M]
  const float* gamma;   // [C]
  const bf16* add;      // [M][C] or null
  const float* keep;    // per-sample factor of the second output, or null
  bf16* dhp;            // [M][4C]
  bf16* dx;             // [M][C]
  bf16* dxs;            // [M][C] or null
  float* partials;      // [workgroups][2][C]
  int M, tok_per_sample;
};

__device__ __forceinline__ float mlp_gelu_grad(float v) {
  const float z = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float e = __expf(-z * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float h = 0.5f * poly * e;
  return (v < 0.f ? h : 1.0f - h) + v * 0.3989422804014327f * e;
}

template <int C, int TMW>
__global__ __launch_bounds__(256) void swin_mlp_bwd_kernel(MlpBwdArgs a) {
  static_assert(C == 96 || C == 192, "narrow Swin stages");
  constexpr int H = 4 * C, HC = 64, NHC = H / HC;
  constexpr int BM = 64 * TMW, WT = 16 * TMW;
  constexpr int KC1 = C / 32;
  constexpr int NB2 = C / 96, NT2 = C / 16, NP2 = NT2 / 2;
  constexpr int XT_BYTES = KC1 * BM * 64, HT_BYTES = 2 * BM * 64;
  constexpr int W1_BYTES = KC1 * HC * 64, W2_BYTES = NB2 * 2 * 96 * 64;
  constexpr int W1_PIECES = W1_BYTES / 1024, W2_PIECES = W2_BYTES / 1024;
  __shared__ __attribute__((aligned(1024))) char lds[XT_BYTES + HT_BYTES + W1_BYTES + W2_BYTES];
  char* const xt = lds;
  char* const ht = lds + XT_BYTES;
  char* const w1b = ht + HT_BYTES;
  char* const w2b = w1b + W1_BYTES;
  static_assert(4 * 2 * C * 4 <= XT_BYTES + HT_BYTES, "the per-wave channel sums reuse the token tiles");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * BM;
  const int l15 = lane & 15, q = lane >> 4;
  const bf16* zero = reinterpret_cast<const bf16*>(mlp_zero_page);

#pragma unroll
  for (int i = 0; i < TMW; ++i) {
    const int row = wave * WT + 16 * i + (lane >> 2);
    const int slot = (lane & 3) ^ swz(row);
    const bool ok = m0 + row < a.M;
    const bf16* src = a.dy + (long)(m0 + row) * C + slot * 8;
#pragma unroll
    for (int kc = 0; kc < KC1; ++kc)
      glds16e(ok ? src + kc * 32 : zero, xt + kc * (BM * 64) + (wave * WT + 16 * i) * 64);
  }
  const int wrel = (lane >> 2) * 32;
  auto issue_w1 = [&](int hc) {
    const bf16* src = a.w2t + (long)hc * (KC1 * HC * 32);
    for (int j = wave; j < W1_PIECES; j += 4) {
      const int jj = j & 3;
      glds16e(src + j * 512 + wrel + (((lane & 3) ^ (2 * (jj & 1))) * 8), w1b + j * 1024);
    }
  };
  auto issue_w2 = [&](int hc) {
    for (int j = wave; j < W2_PIECES; j += 4) {
      const int nb = j / 12, r = j - nb * 12;
      const int jj = r % 6;
      const bf16* src = a.w1t + ((long)nb * (H / 32) + 2 * hc) * (96 * 32) + r * 512;
      glds16e(src + wrel + (((lane & 3) ^ (2 * (jj & 1))) * 8), w2b + j * 1024);
    }
  };
  issue_w1(0);
  issue_w2(0);

  const int x_lane = (wave * WT + l15) * 64 + ((q ^ (((l15 >> 2) & 1) << 1)) * 16);
  const int w_lane = (8 * (l15 >> 2) + (l15 & 3)) * 64 + ((q ^ (2 * (l15 >> 3))) * 16);

  f32x4 acc2[TMW][NT2];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < NT2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int hc = 0; hc < NHC; ++hc) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // the pre-activations this chunk's epilogue multiplies by: on their way while the product runs
    Vec16<bf16> hp[TMW][2];
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
      const long tok = (long)m0 + wave * WT + 16 * i + l15;
#pragma unroll
      for (int p = 0; p < 2; ++p)
        hp[i][p] = tok < a.M ? load16(a.hpre + tok * H + hc * HC + 32 * p + 8 * q) : zero16<bf16>();
    }
    f32x4 acc1[TMW][4];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KC1; ++kc) {
      bf16x8 xf[TMW], wf[4];
#pragma unroll
      for (int i = 0; i < TMW; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(xt + kc * (BM * 64) + x_lane + i * 16 * 64);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        wf[j] = *reinterpret_cast<const bf16x8*>(w1b + kc * (HC * 64) + w_lane + (32 * (j >> 1) + 4 * (j & 1)) * 64);
#pragma unroll
      for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc1[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (hc + 1 < NHC) issue_w1(hc + 1);

    Vec16<bf16> dh[TMW][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < TMW; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dh[i][p].v[r] = from_f32<bf16>(acc1[i][2 * p][r] * mlp_gelu_grad(to_f32(hp[i][p].v[r])));
          dh[i][p].v[4 + r] = from_f32<bf16>(acc1[i][2 * p + 1][r] * mlp_gelu_grad(to_f32(hp[i][p].v[4 + r])));
        }
        const int row = wave * WT + 16 * i + l15;
        *reinterpret_cast<u32x4*>(ht + p * (BM * 64) + row * 64 + ((q ^ swz(row)) * 16)) = __builtin_bit_cast(u32x4, dh[i][p]);
      }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
      const long tok = (long)m0 + wave * WT + 16 * i + l15;
      if (tok < a.M) {
#pragma unroll
        for (int p = 0; p < 2; ++p) store16(a.dhp + tok * H + hc * HC + 32 * p + 8 * q, dh[i][p]);
      }
    }
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      bf16x8 hf[TMW];
#pragma unroll
      for (int i = 0; i < TMW; ++i) hf[i] = *reinterpret_cast<const bf16x8*>(ht + kc * (BM * 64) + x_lane + i * 16 * 64);
#pragma unroll
      for (int nb = 0; nb < NB2; ++nb) {
        bf16x8 wf[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
          wf[j] = *reinterpret_cast<const bf16x8*>(w2b + (nb * 2 + kc) * (96 * 64) + w_lane + (32 * (j >> 1) + 4 * (j & 1)) * 64);
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j)
            acc2[i][nb * 6 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], hf[i], acc2[i][nb * 6 + j], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (hc + 1 < NHC) issue_w2(hc + 1);
  }

  // ---- LayerNorm backward on the accumulators.  Lane (q, token l15) holds channels 32p + 8q .. + 7 of its token.
  float sg[NP2][8], sb[NP2][8];        // this lane's share of sum_tokens dln * xhat and sum_tokens dln
#pragma unroll
  for (int p = 0; p < NP2; ++p)
#pragma unroll
    for (int e = 0; e < 8; ++e) sg[p][e] = sb[p][e] = 0.f;
#pragma unroll
  for (int i = 0; i < TMW; ++i) {
    const long tok = (long)m0 + wave * WT + 16 * i + l15;
    const bool live = tok < a.M;
    const long tk = live ? tok : 0;
    const float mu = a.mean[tk], rs = a.rstd[tk];
    float d[NP2][8], xh[NP2][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < NP2; ++p) {
      const int c0 = 32 * p + 8 * q;
      const Vec16<bf16> xv = load16(a.x + tk * C + c0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d[p][r] = live ? to_f32(from_f32<bf16>(acc2[i][2 * p][r])) : 0.f;
        d[p][4 + r] = live ? to_f32(from_f32<bf16>(acc2[i][2 * p + 1][r])) : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xh[p][e] = (to_f32(xv.v[e]) - mu) * rs;
        const float g = d[p][e] * a.gamma[c0 + e];
        s1 += g;
        s2 += g * xh[p][e];
        sg[p][e] += d[p][e] * xh[p][e];
        sb[p][e] += d[p][e];
      }
    }
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    s1 /= (float)C;
    s2 /= (float)C;
    if (live) {
      const float kf = a.dxs != nullptr ? a.keep[tok / a.tok_per_sample] : 1.f;
#pragma unroll
      for (int p = 0; p < NP2; ++p) {
        const int c0 = 32 * p + 8 * q;
        Vec16<bf16> av = zero16<bf16>();
        if (a.add != nullptr) av = load16(a.add + tok * C + c0);
        Vec16<bf16> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float g = d[p][e] * a.gamma[c0 + e];
          o.v[e] = from_f32<bf16>(rs * (g - s1 - xh[p][e] * s2) + (a.add ? to_f32(av.v[e]) : 0.f));
        }
        store16(a.dx + tok * C + c0, o);
        if (a.dxs != nullptr) {
          Vec16<bf16> os;
#pragma unroll
          for (int e = 0; e < 8; ++e) os.v[e] = from_f32<bf16>(to_f32(o.v[e]) * kf + 0.f);
          store16(a.dxs + tok * C + c0, os);
        }
      }
    }
  }
  // channel sums: over the 16 tokens of a DPP row (every lane ends with the total), over the waves through LDS in
  // wave order; the operand tiles are dead (the barrier that ended the last chunk)
  float (*red)[2][C] = reinterpret_cast<float (*)[2][C]>(lds);
#pragma unroll
  for (int p = 0; p < NP2; ++p)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float g = sg[p][e], b = sb[p][e];
      g += mlp_dpp<0xB1>(g); g += mlp_dpp<0x4E>(g); g += mlp_dpp<0x141>(g); g += mlp_dpp<0x140>(g);
      b += mlp_dpp<0xB1>(b); b += mlp_dpp<0x4E>(b); b += mlp_dpp<0x141>(b); b += mlp_dpp<0x140>(b);
      if (l15 == 0) {
        red[wave][0][32 * p + 8 * q + e] = g;
        red[wave][1][32 * p + 8 * q + e] = b;
      }
    }
  __syncthreads();
  for (int c = tid; c < 2 * C; c += 256) {
    const int which = c / C, ch = c - which * C;
    a.partials[((long)blockIdx.x * 2 + which) * C + ch] = red[0][which][ch] + red[1][which][ch] + red[2][which][ch] + red[3][which][ch];
  }
}

}  // namespace

extern "C" {

int unetk_swin_mlp_supported(int64_t M, int C, int dtype) {
  return dtype == UNETK_BF16 && (C == 96 || C == 192) && M > 0 && M < (1LL << 31) - 256;
}

int unetk_swin_mlp_fwd(const void* x, const float* gamma, const float* beta, float eps, const void* w1_packed64,
                       const float* b1, const void* w2_packed96, const float* b2, const float* keep,
                       int64_t tok_per_sample, void* ln, float* mean, float* rstd, void* hpre, void* hact, void* out,
                       int64_t M, int C, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(x && gamma && beta && w1_packed64 && b1 && w2_packed96 && b2 && ln && mean && rstd && hpre && out,
                  "swin_mlp_fwd: null pointer");
  UNETK_CHECK_ARG(unetk_swin_mlp_supported(M, C, dtype),
                  "swin_mlp_fwd: bfloat16 tokens of 96 or 192 channels (M=%lld C=%d dtype %d)", (long long)M, C, dtype);
  UNETK_CHECK_ARG(keep == nullptr || (tok_per_sample > 0 && tok_per_sample < (1LL << 31)),
                  "swin_mlp_fwd: a DropPath scale needs the tokens per sample");
  const uintptr_t bits = (uintptr_t)x | (uintptr_t)w1_packed64 | (uintptr_t)w2_packed96 | (uintptr_t)ln | (uintptr_t)hpre |
                         (uintptr_t)hact | (uintptr_t)out;
  UNETK_CHECK_ARG((bits & 15) == 0, "swin_mlp_fwd: pointers must be 16-byte aligned");
  MlpArgs a;
  a.x = (const bf16*)x; a.gamma = gamma; a.beta = beta; a.w1 = (const bf16*)w1_packed64; a.b1 = b1;
  a.w2 = (const bf16*)w2_packed96; a.b2 = b2; a.keep = keep; a.ln = (bf16*)ln; a.mean = mean; a.rstd = rstd;
  a.hpre = (bf16*)hpre; a.hact = (bf16*)hact; a.out = (bf16*)out; a.eps = eps; a.M = (int)M;
  a.tok_per_sample = keep ? (int)tok_per_sample : 1;
  if (C == 96)
    hipLaunchKernelGGL((swin_mlp_fwd_kernel<96, 2>), dim3((unsigned)cdiv(M, 128)), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL((swin_mlp_fwd_kernel<192, 1>), dim3((unsigned)cdiv(M, 64)), dim3(256), 0, stream, a);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_swin_mlp_bwd_rows(int64_t M, int C) { return C == 96 ? cdiv(M, 128) : cdiv(M, 64); }

int unetk_swin_mlp_bwd(const void* dy, const void* hpre, const void* w2t_packed64, const void* w1t_packed96,
                       const void* x, const float* mean, const float* rstd, const float* gamma, const void* add,
                       const float* keep, int64_t tok_per_sample, void* dhp, void* dx, void* dx_scaled,
                       float* partials, int64_t M, int C, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(dy && hpre && w2t_packed64 && w1t_packed96 && x && mean && rstd && gamma && dhp && dx && partials,
                  "swin_mlp_bwd: null pointer");
  UNETK_CHECK_ARG(unetk_swin_mlp_supported(M, C, dtype),
                  "swin_mlp_bwd: bfloat16 tokens of 96 or 192 channels (M=%lld C=%d dtype %d)", (long long)M, C, dtype);
  UNETK_CHECK_ARG((dx_scaled == nullptr) == (keep == nullptr), "swin_mlp_bwd: the scaled copy and its factors go together");
  UNETK_CHECK_ARG(keep == nullptr || (tok_per_sample > 0 && tok_per_sample < (1LL << 31) && M % tok_per_sample == 0),
                  "swin_mlp_bwd: the tokens are not whole samples of %lld", (long long)tok_per_sample);
  const uintptr_t bits = (uintptr_t)dy | (uintptr_t)hpre | (uintptr_t)w2t_packed64 | (uintptr_t)w1t_packed96 | (uintptr_t)x |
                         (uintptr_t)add | (uintptr_t)dhp | (uintptr_t)dx | (uintptr_t)dx_scaled;
  UNETK_CHECK_ARG((bits & 15) == 0, "swin_mlp_bwd: pointers must be 16-byte aligned");
  MlpBwdArgs a;
  a.dy = (const bf16*)dy; a.hpre = (const bf16*)hpre; a.w2t = (const bf16*)w2t_packed64; a.w1t = (const bf16*)w1t_packed96;
  a.x = (const bf16*)x; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.add = (const bf16*)add; a.keep = keep;
  a.dhp = (bf16*)dhp; a.dx = (bf16*)dx; a.dxs = (bf16*)dx_scaled; a.partials = partials; a.M = (int)M;
  a.tok_per_sample = keep ? (int)tok_per_sample : 1;
  if (C == 96)
    hipLaunchKernelGGL((swin_mlp_bwd_kernel<96, 2>), dim3((unsigned)cdiv(M, 128)), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL((swin_mlp_bwd_kernel<192, 1>), dim3((unsigned)cdiv(M, 64)), dim3(256), 0, stream, a);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
