// Token-linear GEMM on MFMA for gfx950 (bfloat16 in, fp32 accumulate):
//
//   out[t][n] = epilogue( sum_k x[t][k] * w[n][k] )          t < M tokens, n < N, k < K
//
// nn.Linear over Swin tokens (reference swin_unet_model.py:234, 236, 298, 301, 383), its data
// gradient (the same product with the weight packed transposed), the ConvTranspose2d and 1x1
// projection GEMMs of the decoder (:501, 506).  The 3x3 convolution kernel run as a 1x1 conv did this
// job with a pipeline built for long K (18x18 halos, LDS-staged epilogue behind eight barriers,
// two workgroups per CU); token linears are short-K products on many rows, so this kernel is
// built the other way round:
//
//  * A workgroup (4 waves) owns 64*TM tokens x BN = 16*TN channels; wave w owns tokens
//    [16*TM*w, 16*TM*(w+1)) times all channels.  K runs in chunks of 32 (64 bytes per row) through a
//    ring of three LDS stages filled by LDS-DMA: two chunks are in flight behind the one being
//    multiplied, one counted wait + barrier per chunk.  A 128-token tile needs 42 KB: three
//    workgroups per CU cover each other's prologue, epilogue and barrier waits.
//  * The MFMA operands are SWAPPED: the weight tile is the A operand (rows = channels) and the token
//    tile the B operand (columns = tokens), so an accumulator register holds four CHANNELS of one
//    token.  The weight rows of a pair of 16-channel MFMA tiles are taken in the order
//    {0-3, 8-11, 16-19, 24-27} / {4-7, 12-15, 20-23, 28-31}, which makes a lane's eight values of the
//    pair eight CONSECUTIVE channels of its token: the epilogue stores them as one 16-byte vector
//    straight from registers.  No LDS staging, no barrier after the K loop.
//  * Epilogue, all optional and all on the registers: + bias[n]; GELU (erf) with the pre-activation
//    kept as a second output (the fc1 layer's backward needs it, :298-299); * GELU'(aux) for the data
//    gradient of fc2 (the gradient of the hidden activation, :299 through autograd); residual
//    + per-sample DropPath scale (:371-372).
//
// Weights arrive in the packed layout of the convolution kernels for kernel size 1,
// [N/BN][K/32][BN][32]: a stage's weight slab is one contiguous run.
#include "common.h"
#include "mma.h"

namespace {

__device__ __attribute__((aligned(64))) unsigned int gemm_zero_page[16];

__device__ __forceinline__ void glds16g(const void* src, char* lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(src, LDS_PTR(void, lds_dst), 16, 0, 0);
#endif
}

struct GemmArgs {
  const bf16* x;        // tokens [M][ldx]
  const bf16* w;        // packed [N/BN][K/32][BN][32]
  bf16* out;            // [M][ldo] (16-byte aligned rows)
  bf16* out2;           // second output [M][ldo2]: act(out) when act != 0; or null
  const float* bias;    // [N] or null
  const bf16* resid;    // [M][ldr] or null: out = resid + keep[sample] * value
  const float* keep;    // per-sample scale of the value before the residual add, or null
  const bf16* aux;      // [M][ldaux] or null: value *= GELU'(aux)
  int ldx, ldo, ldo2, ldr, ldaux;
  int M, N, K, nchunks;
  int act;              // 0 none, 2 GELU(erf) (codes of unetk.h)
  int tok_per_sample;
  int mtiles, nby, tiles_per_xcd;
};

__device__ __forceinline__ void wait_vm_barrier_g(int n) {
#define UNETK_WVB(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier" ::: "memory"); break;
  switch (n) {
    UNETK_WVB(1) UNETK_WVB(2) UNETK_WVB(3) UNETK_WVB(4) UNETK_WVB(5) UNETK_WVB(6) UNETK_WVB(7) UNETK_WVB(8)
    default: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); break;
  }
#undef UNETK_WVB
}

// GELU (erf form) and its derivative for the bf16 epilogues.  Phi(v) = 0.5 * (1 + erf(v / sqrt 2)) through
// Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the 2^-9 rounding of the bf16 result): one
// reciprocal, one exponential and five multiply-adds, against the several dozen instructions and
// branches of the library erff.  The exponential exp(-v^2 / 2) is also the Gaussian of the derivative.
struct PhiExp {
  float Phi, e;     // Phi(v), exp(-v*v/2)
};
__device__ __forceinline__ PhiExp phi_exp(float v) {
  const float z = fabsf(v) * 0.70710678118654752440f;
  const float t = __frcp_rn(1.0f + 0.3275911f * z);
  const float e = __expf(-z * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;
  PhiExp r;
  r.Phi = 0.5f * (1.0f + (v < 0.f ? -erf_abs : erf_abs));
  r.e = e;
  return r;
}
__device__ __forceinline__ float gelu_erf_g(float v) { return v * phi_exp(v).Phi; }
// d/dv [ v * Phi(v) ] = Phi(v) + v * phi(v)
__device__ __forceinline__ float gelu_grad_g(float v) {
  const PhiExp p = phi_exp(v);
  return p.Phi + v * 0.3989422804014327f * p.e;
}

template <int TM, int TN>
__global__ __launch_bounds__(256) void lin_gemm_kernel(GemmArgs a) {
  static_assert(TN % 2 == 0, "channel tiles are taken in pairs");
  constexpr int BM = 64 * TM, BN = 16 * TN;
  constexpr int APIECES = BM / 16, WPIECES = BN / 16;          // 1 KiB DMA pieces per chunk
  constexpr int STAGE = (BM + BN) * 64;
  constexpr int NSTAGE = 3;
  __shared__ __attribute__((aligned(1024))) char lds[NSTAGE * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD x (workgroups are dealt round-robin to the 8 XCDs) walks the x-th eighth of the token tiles
  // with the channel block as the fast index: the blocks that share a token tile share an L2.
  const int xcd = blockIdx.x & 7;
  const int item = blockIdx.x >> 3;
  const int nb = item % a.nby;
  const int mt = xcd * a.tiles_per_xcd + item / a.nby;
  if (item / a.nby >= a.tiles_per_xcd || mt >= a.mtiles) return;
  const int m0 = mt * BM, n0 = nb * BN;

  // ---- DMA sources.  Token piece i: rows 16i .. 16i+15, lane -> (row 16i + lane/4, 16-byte slot
  // (lane&3) ^ swz(row)); rows past M read zeros.  Weight piece j: row 16j + lane/4 of the block,
  // slot (lane&3) ^ 2*(j&1) (the swizzle that keeps the permuted fragment reads conflict-free).
  constexpr int APW = (APIECES + 3) / 4, WPW = (WPIECES + 3) / 4;
  const bf16* zero = reinterpret_cast<const bf16*>(gemm_zero_page);
  const bf16* asrc[APW];
#pragma unroll
  for (int k = 0; k < APW; ++k) {
    const int i = wave + 4 * k;
    const int row = 16 * i + (lane >> 2);
    const int slot = (lane & 3) ^ swz(row);
    asrc[k] = (i < APIECES && m0 + row < a.M) ? a.x + (long)(m0 + row) * a.ldx + slot * 8 : nullptr;
  }
  const bf16* wbase = a.w + (long)nb * a.nchunks * BN * 32;
  int woff[WPW];
#pragma unroll
  for (int k = 0; k < WPW; ++k) {
    const int j = wave + 4 * k;
    woff[k] = (16 * j + (lane >> 2)) * 32 + (((lane & 3) ^ (2 * (j & 1))) * 8);
  }
  auto issue = [&](int ch) {
    char* dst = lds + (ch % NSTAGE) * STAGE;
#pragma unroll
    for (int k = 0; k < APW; ++k) {
      const int i = wave + 4 * k;
      if (i < APIECES) glds16g(asrc[k] != nullptr ? asrc[k] + ch * 32 : zero, dst + i * 1024);
    }
    const bf16* ws = wbase + (long)ch * BN * 32;
#pragma unroll
    for (int k = 0; k < WPW; ++k) {
      const int j = wave + 4 * k;
      if (j < WPIECES) glds16g(ws + woff[k], dst + BM * 64 + j * 1024);
    }
  };
  const int my_pieces = (APIECES - wave + 3) / 4 + (WPIECES - wave + 3) / 4;

  // ---- fragment addresses (see the header comment for the weight row order)
  const int l15 = lane & 15, q = lane >> 4;
  const int a_lane = (wave * 16 * TM + l15) * 64 + ((q ^ (((l15 >> 2) & 1) << 1)) * 16);
  const int w_lane = BM * 64 + (8 * (l15 >> 2) + (l15 & 3)) * 64 + ((q ^ (2 * (l15 >> 3))) * 16);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0);
  if (a.nchunks > 1) issue(1);
#pragma unroll 1
  for (int ch = 0; ch < a.nchunks; ++ch) {
    // this chunk's pieces (issued two chunks ago) have landed; the next chunk's stay in flight
    wait_vm_barrier_g(ch + 1 < a.nchunks ? my_pieces : 0);
    if (ch + 2 < a.nchunks) issue(ch + 2);
    const char* st = lds + (ch % NSTAGE) * STAGE;
    bf16x8 xf[TM], wf[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(st + a_lane + i * 16 * 64);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      wf[j] = *reinterpret_cast<const bf16x8*>(st + w_lane + (32 * (j >> 1) + 4 * (j & 1)) * 64);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
  }

  // ---- epilogue: lane (q, token l15) holds channels n0 + 32p + 8q .. +7 of its token for pair p
  float bias8[TN / 2][8];
#pragma unroll
  for (int p = 0; p < TN / 2; ++p) {
    const int c0 = n0 + 32 * p + 8 * q;
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[p][e] = a.bias != nullptr ? a.bias[c0 + e] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int t = m0 + wave * 16 * TM + 16 * i + l15;
    if (t >= a.M) continue;
    float keep = 1.f;
    if (a.keep != nullptr) keep = a.keep[t / a.tok_per_sample];
#pragma unroll
    for (int p = 0; p < TN / 2; ++p) {
      const int c0 = n0 + 32 * p + 8 * q;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][2 * p][r] + bias8[p][r];
        v[4 + r] = acc[i][2 * p + 1][r] + bias8[p][4 + r];
      }
      if (a.aux != nullptr) {
        const Vec16<bf16> h = load16(a.aux + (long)t * a.ldaux + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= gelu_grad_g(to_f32(h.v[e]));
      }
      Vec16<bf16> o;
      if (a.act == 2 && a.out2 != nullptr) {
        // pre-activation to out, GELU of the ROUNDED pre-activation (what a separate pass would read) to out2
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = from_f32<bf16>(v[e]);
        store16(a.out + (long)t * a.ldo + c0, o);
        Vec16<bf16> g;
#pragma unroll
        for (int e = 0; e < 8; ++e) g.v[e] = from_f32<bf16>(gelu_erf_g(to_f32(o.v[e])));
        store16(a.out2 + (long)t * a.ldo2 + c0, g);
        continue;
      }
      if (a.act == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_erf_g(v[e]);
      }
      if (a.resid != nullptr) {
        // the value is rounded to the storage type before the add, as when it is stored and re-read
        const Vec16<bf16> rv = load16(a.resid + (long)t * a.ldr + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = to_f32(rv.v[e]) + keep * to_f32(from_f32<bf16>(v[e]));
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) o.v[e] = from_f32<bf16>(v[e]);
      store16(a.out + (long)t * a.ldo + c0, o);
    }
  }
}

template <int TM, int TN> int launch_gemm(GemmArgs a, hipStream_t stream) {
  constexpr int BM = 64 * TM;
  a.mtiles = cdiv(a.M, BM);
  a.tiles_per_xcd = cdiv(a.mtiles, 8);
  a.nby = a.N / (16 * TN);
  hipLaunchKernelGGL((lin_gemm_kernel<TM, TN>), dim3((unsigned)(8 * a.tiles_per_xcd * a.nby)), dim3(256), 0, stream, a);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // namespace

extern "C" {

int unetk_linear_supported(int64_t M, int K, int N, int bn, int dtype) {
  return dtype == UNETK_BF16 && M > 0 && K > 0 && K % 32 == 0 && N > 0 && (bn == 96 || bn == 64 || bn == 32) &&
         N % bn == 0 && M < (1LL << 31) - 512;
}

int unetk_linear_fwd(const void* x, int ldx, const void* wpacked, void* out, int ldo, void* out2, int ldo2,
                     int64_t M, int K, int N, int bn, int dtype, const float* bias, int act, const void* resid,
                     int ldr, const float* resid_keep, int64_t tok_per_sample, const void* gelu_grad_of, int ldaux,
                     hipStream_t stream) {
  UNETK_CHECK_ARG(x && wpacked && out, "linear_fwd: null pointer");
  UNETK_CHECK_ARG(unetk_linear_supported(M, K, N, bn, dtype),
                  "linear_fwd: unsupported shape or type (M=%lld K=%d N=%d block %d dtype %d): bfloat16, K a multiple "
                  "of 32, N a multiple of its channel block (96, 64 or 32)", (long long)M, K, N, bn, dtype);
  UNETK_CHECK_ARG(act == UNETK_ACT_NONE || act == UNETK_ACT_GELU, "linear_fwd: activation %d (none or GELU)", act);
  UNETK_CHECK_ARG(out2 == nullptr || act == UNETK_ACT_GELU, "linear_fwd: a second output goes with GELU");
  UNETK_CHECK_ARG(ldx >= K && ldo >= N && ldx % 8 == 0 && ldo % 8 == 0, "linear_fwd: bad strides (ldx=%d ldo=%d)", ldx, ldo);
  UNETK_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)wpacked % 16) == 0,
                  "linear_fwd: pointers must be 16-byte aligned");
  UNETK_CHECK_ARG(out2 == nullptr || (ldo2 >= N && ldo2 % 8 == 0 && ((uintptr_t)out2 % 16) == 0),
                  "linear_fwd: second output misaligned");
  UNETK_CHECK_ARG(resid == nullptr || (ldr >= N && ldr % 8 == 0 && ((uintptr_t)resid % 16) == 0),
                  "linear_fwd: residual misaligned");
  UNETK_CHECK_ARG(gelu_grad_of == nullptr || (ldaux >= N && ldaux % 8 == 0 && ((uintptr_t)gelu_grad_of % 16) == 0),
                  "linear_fwd: auxiliary tensor misaligned");
  UNETK_CHECK_ARG(resid_keep == nullptr || (resid != nullptr && tok_per_sample > 0 && tok_per_sample < (1LL << 31)),
                  "linear_fwd: a DropPath scale needs the residual and the tokens per sample");
  UNETK_CHECK_ARG(!(resid != nullptr && out2 != nullptr), "linear_fwd: residual and second output are exclusive");
  GemmArgs a;
  a.x = (const bf16*)x; a.w = (const bf16*)wpacked; a.out = (bf16*)out; a.out2 = (bf16*)out2;
  a.bias = bias; a.resid = (const bf16*)resid; a.keep = resid_keep; a.aux = (const bf16*)gelu_grad_of;
  a.ldx = ldx; a.ldo = ldo; a.ldo2 = ldo2; a.ldr = ldr; a.ldaux = ldaux;
  a.M = (int)M; a.N = N; a.K = K; a.nchunks = K / 32; a.act = act;
  a.tok_per_sample = resid_keep ? (int)tok_per_sample : 1;
  // 128-token tiles, three workgroups per CU.  (256-token tiles, two per CU, measured slower on every
  // Swin shape, 64x64 tokens x 96 channels to 8x8 tokens x 3072 channels: these products are bound by
  // bytes and by how many workgroups cover each other's waits, not by the MFMA : LDS-read ratio.)
  switch (bn) {
    case 96: return launch_gemm<2, 6>(a, stream);
    case 64: return launch_gemm<2, 4>(a, stream);
    default: return launch_gemm<2, 2>(a, stream);
  }
}

}  // extern "C"
