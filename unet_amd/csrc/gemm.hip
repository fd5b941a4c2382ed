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
#include "epilogue.h"
#include "mma.h"
#include <cstdlib>

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
  // ConvTranspose2d(k=2, s=2) as a GEMM with N = 4 * up: channel n = (ky, kx, co) of token (b, y, x) is channel co
  // of output pixel (b, 2y + ky, 2x + kx).  shuf_up = 0: a plain [M][ldo] result.
  int shuf_up, shuf_w, shuf_hw;
  // The data gradient of that layer is the GEMM with K = 4 * gin_up whose operand row of token (b, y, x) is gathered
  // from the gradient image: channels (ky, kx, co) = channel co of pixel (b, 2y + ky, 2x + kx) of x [B][2h][2w][ldx]
  // (a 32-channel chunk lies inside one (ky, kx): gin_up is a multiple of 32).  gin_up = 0: plain [M][ldx] rows.
  int gin_up, gin_w, gin_hw;
  int mtiles, nby, tiles_per_xcd;
};

__device__ __forceinline__ void wait_vm_barrier_g(int n) {
#define UNETK_WVB(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier" ::: "memory"); break;
  switch (n) {
    UNETK_WVB(1) UNETK_WVB(2) UNETK_WVB(3) UNETK_WVB(4) UNETK_WVB(5) UNETK_WVB(6) UNETK_WVB(7) UNETK_WVB(8)
    UNETK_WVB(9) UNETK_WVB(10) UNETK_WVB(11) UNETK_WVB(12) UNETK_WVB(13) UNETK_WVB(14) UNETK_WVB(15) UNETK_WVB(16)
    UNETK_WVB(18) UNETK_WVB(20) UNETK_WVB(21) UNETK_WVB(24)
    default: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); break;
  }
#undef UNETK_WVB
}

template <int TM, int TN, bool GIN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void lin_gemm_kernel(GemmArgs a) {
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
  // Every wave issues the same number of pieces per chunk (a wave whose index falls past the last piece
  // sends one of the existing pieces again: same bytes to the same place), so "all but the newest chunk"
  // is one compile-time vmcnt and the steady-state loop has no branch but its own.
  constexpr int APW = (APIECES + 3) / 4, WPW = (WPIECES + 3) / 4, NP = APW + WPW;
  static_assert(2 * NP <= 16, "counted waits");
  const bf16* zero = reinterpret_cast<const bf16*>(gemm_zero_page);
  const bf16* asrc[APW];
  int astep[APW], adst[APW];
#pragma unroll
  for (int k = 0; k < APW; ++k) {
    int i = wave + 4 * k;
    if (i >= APIECES) i -= 4;
    const int row = 16 * i + (lane >> 2);
    const int slot = (lane & 3) ^ swz(row);
    long first = (long)(m0 + row) * a.ldx;          // element offset of the token's row
    if constexpr (GIN) {                             // ... or of pixel (b, 2y, 2x) of the gradient image
      const int t = m0 + row, bb = t / a.gin_hw, rem = t - bb * a.gin_hw, y = rem / a.gin_w, x = rem - y * a.gin_w;
      first = (((long)bb * 4 * a.gin_hw) + (long)(2 * y) * (2 * a.gin_w) + 2 * x) * a.ldx;
    }
    const bool ok = m0 + row < a.M;
    asrc[k] = ok ? a.x + first + slot * 8 : zero;
    astep[k] = ok ? 1 : 0;                           // rows past M stay on the page of zeros
    adst[k] = i * 1024;
  }
  // element offset of chunk ch inside the token's row (plain: ch * 32)
  const int gin_cpq = GIN ? a.gin_up / 32 : 1;
  auto chunk_off = [&](int ch) -> long {
    if constexpr (!GIN) return (long)ch * 32;
    const int qd = ch / gin_cpq, c0 = (ch - qd * gin_cpq) * 32;
    return ((long)(qd >> 1) * (2 * a.gin_w) + (qd & 1)) * a.ldx + c0;
  };
  const bf16* wbase = a.w + (long)nb * a.nchunks * BN * 32;
  int woff[WPW], wdst[WPW];
#pragma unroll
  for (int k = 0; k < WPW; ++k) {
    int j = wave + 4 * k;
    while (j >= WPIECES) j -= (WPIECES >= 4 ? 4 : WPIECES);
    woff[k] = (16 * j + (lane >> 2)) * 32 + (((lane & 3) ^ (2 * (j & 1))) * 8);
    wdst[k] = BM * 64 + j * 1024;
  }
  auto issue = [&](int ch, int stage) __attribute__((always_inline)) {
    char* dst = lds + stage * STAGE;
    const long coff = chunk_off(ch);
#pragma unroll
    for (int k = 0; k < APW; ++k) glds16g(asrc[k] + coff * astep[k], dst + adst[k]);
    const bf16* ws = wbase + (long)ch * BN * 32;
#pragma unroll
    for (int k = 0; k < WPW; ++k) glds16g(ws + woff[k], dst + wdst[k]);
  };

  // ---- fragment addresses (see the header comment for the weight row order)
  const int l15 = lane & 15, q = lane >> 4;
  const int a_lane = (wave * 16 * TM + l15) * 64 + ((q ^ (((l15 >> 2) & 1) << 1)) * 16);
  const int w_lane = BM * 64 + (8 * (l15 >> 2) + (l15 & 3)) * 64 + ((q ^ (2 * (l15 >> 3))) * 16);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      // (opaque to the optimiser and held in vector registers: left alone, the accumulators went to accumulation
      // registers that the allocator rotated through the two halves of the unrolled loop, ~70 copies per 24 MFMAs)
      asm volatile("" : "+v"(acc[i][j]));
    }

  // The fragments are double-buffered in registers: while chunk ch is multiplied out of one set, the ds_reads of
  // chunk ch+1 fill the other, one read between MFMAs, so a wave's LDS round trip and its MFMAs overlap.  The
  // barrier that opens iteration ch therefore waits for chunk ch+1; by then every wave has the fragments of chunk
  // ch in registers, so stage ch % 3 is free for chunk ch+3: two chunks stay in flight behind the one being read.
  // A chunk is only 12 MFMAs per wave, so what surrounds them counts: the steady-state iteration below is
  // straight-line code (no modulo, no division, no run-time wait count); only the last four chunks take branches.
  bf16x8 xf[2][TM], wf[2][TN];
  auto load_frags = [&](auto bufc, int stage) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
    const char* st = lds + stage * STAGE;
#pragma unroll
    for (int j = 0; j < TN; ++j)
      wf[BUF][j] = *reinterpret_cast<const bf16x8*>(st + w_lane + (32 * (j >> 1) + 4 * (j & 1)) * 64);
#pragma unroll
    for (int i = 0; i < TM; ++i) xf[BUF][i] = *reinterpret_cast<const bf16x8*>(st + a_lane + i * 16 * 64);
  };
  auto multiply = [&](auto curc) __attribute__((always_inline)) {
    constexpr int CUR = decltype(curc)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[CUR][j], xf[CUR][i], acc[i][j], 0, 0, 0);
  };
  auto interleave = [&]() __attribute__((always_inline)) {      // issue order: a ds_read behind every MFMA until the reads are out
    constexpr int NRD = TM + TN, NMM = TM * TN, MPR = NMM / NRD > 0 ? NMM / NRD : 1;
#pragma unroll
    for (int k = 0; k < NRD; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    if constexpr (NMM - MPR * NRD > 0) __builtin_amdgcn_sched_group_barrier(0x008, NMM - MPR * NRD, 0);
  };
  auto next3 = [](int st) { return st == 2 ? 0 : st + 1; };
  int sr = 1, si = 0;          // stage of chunk ch+1 (read next), stage of chunk ch+3 (= that of chunk ch: refilled next)
  auto steady = [&](auto curc, int ch) __attribute__((always_inline)) {      // chunks ch+1, ch+2, ch+3 all exist
    constexpr int CUR = decltype(curc)::value;
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NP) : "memory");
    issue(ch + 3, si);
    load_frags(std::integral_constant<int, CUR ^ 1>{}, sr);
    multiply(curc);
    interleave();
    sr = next3(sr);
    si = next3(si);
  };
  auto tail = [&](auto curc, int ch) __attribute__((always_inline)) {        // one of the last three chunks
    constexpr int CUR = decltype(curc)::value;
    if (ch + 1 < a.nchunks) {          // (uniform)
      if (ch + 2 < a.nchunks) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      load_frags(std::integral_constant<int, CUR ^ 1>{}, sr);
    }
    multiply(curc);
    sr = next3(sr);
  };
  issue(0, 0);
  if (a.nchunks > 1) issue(1, 1);
  if (a.nchunks > 2) issue(2, 2);
  if (a.nchunks > 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * NP) : "memory");      // chunk 0 has landed
  else if (a.nchunks > 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NP) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  load_frags(std::integral_constant<int, 0>{}, 0);
  {
    // steady-state pairs while chunk ch+4 exists, then the last three or four chunks
    int ch = 0;
#pragma unroll 1
    for (; ch + 4 < a.nchunks; ch += 2) {
      steady(std::integral_constant<int, 0>{}, ch);
      steady(std::integral_constant<int, 1>{}, ch + 1);
    }
    if (ch + 3 < a.nchunks) {
      steady(std::integral_constant<int, 0>{}, ch);
      tail(std::integral_constant<int, 1>{}, ch + 1);
      tail(std::integral_constant<int, 0>{}, ch + 2);
      tail(std::integral_constant<int, 1>{}, ch + 3);
    } else {
      if (ch < a.nchunks) tail(std::integral_constant<int, 0>{}, ch);
      if (ch + 1 < a.nchunks) tail(std::integral_constant<int, 1>{}, ch + 1);
      if (ch + 2 < a.nchunks) tail(std::integral_constant<int, 0>{}, ch + 2);
    }
  }

  // ---- epilogue: lane (q, token l15) holds channels n0 + 32p + 8q .. +7 of its token for pair p
  float bias8[TN / 2][8];
#pragma unroll
  for (int p = 0; p < TN / 2; ++p) {
    const int c0 = n0 + 32 * p + 8 * q;
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[p][e] = a.bias != nullptr ? a.bias[a.shuf_up ? (c0 + e) % a.shuf_up : c0 + e] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int t = m0 + wave * 16 * TM + 16 * i + l15;
    if (t >= a.M) continue;
    float keep = 1.f;
    if (a.keep != nullptr) keep = a.keep[t / a.tok_per_sample];
    long shuf_row = 0;       // output pixel (b, 2y, 2x) of the token, in pixels of the (B, 2h, 2w) image
    if (a.shuf_up) {
      const int bb = t / a.shuf_hw, rem = t - bb * a.shuf_hw, y = rem / a.shuf_w, x = rem - y * a.shuf_w;
      shuf_row = ((long)bb * 4 * a.shuf_hw) + (long)(2 * y) * (2 * a.shuf_w) + 2 * x;
    }
#pragma unroll
    for (int p = 0; p < TN / 2; ++p) {
      const int c0 = n0 + 32 * p + 8 * q;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][2 * p][r] + bias8[p][r];
        v[4 + r] = acc[i][2 * p + 1][r] + bias8[p][4 + r];
      }
      if (a.shuf_up) {       // (uniform) the pixel shuffle of the transposed convolution, in the store address
        const int qd = c0 / a.shuf_up, co = c0 - qd * a.shuf_up;
        Vec16<bf16> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = from_f32<bf16>(v[e]);
        store16(a.out + (shuf_row + (qd >> 1) * (2 * a.shuf_w) + (qd & 1)) * a.ldo + co, o);
        continue;
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


// ------------------------------------------------------------------------------------------------
// Weight gradient of a token linear:  dw[n][k] = sum_t dy[t][n] * x[t][k]   (and db[n] = sum_t dy[t][n])
//
// The reduction runs over tokens, the slow axis of both operands, so the MFMA fragments (eight
// consecutive tokens of one channel per lane) are read from token-major LDS images with the
// transposing ds_read_b64_tr_b16.  A workgroup of WM x WN waves owns a 64*WM x 64*WN tile of dw (each
// wave a 64 x 64 block: 4 x 4 MFMA tiles) and a contiguous range of tokens, which it streams in
// 32-token stages through a ring of three LDS stages filled by LDS-DMA (one counted wait + barrier
// per stage, 16 MFMAs per wave between barriers).  No wave splits the reduction, so nothing is summed
// through LDS at the end.  Partial tiles of the token ranges go to fp32 slabs [range][N_pad][K_pad];
// a second kernel adds the slabs in a fixed order (bitwise reproducible, no atomics).
// LDS rows are 256 or 512 bytes (a whole number of bank rows), so a transposing read, which per
// 32-lane half touches tokens {o..o+3, o+8..o+11} at one 32-byte channel unit, would hit one unit
// eight times: the 32-byte unit index inside each 256 bytes is XORed with (token&3) | (token bit 3)<<2,
// on the DMA source address and on the read address alike.
struct WgArgs {
  const bf16* x;
  const bf16* dy;
  float* slab;          // [nsplit][N_pad][K_pad]
  float* bias_slab;     // [nsplit][N_pad] or null
  int ldx, lddy;
  int M, N, K, N_pad, K_pad;
  int nsplit, chunk;    // token ranges: [s*chunk, (s+1)*chunk)
  int gn, gk;           // output tiles along N and K
};

__device__ __forceinline__ int tok_key(int t) { return (t & 3) | (((t >> 3) & 1) << 2); }

template <int WM, int WN, int NSTAGE>
__global__ __launch_bounds__(64 * WM * WN) void lin_wgrad_kernel(WgArgs a) {
  constexpr int NW = WM * WN, NTHR = 64 * NW;
  constexpr int BNo = 64 * WM, BKo = 64 * WN;
  constexpr int RBD = 2 * BNo, RBX = 2 * BKo;            // LDS row bytes of the dy / x images
  constexpr int TT = 32;                                 // tokens per stage = one MFMA k-step
  constexpr int DPIECES = TT * RBD / 1024, XPIECES = TT * RBX / 1024;
  constexpr int DBYTES = TT * RBD, STAGE = TT * (RBD + RBX);
  constexpr int AHEAD = NSTAGE - 1;                      // stages in flight behind the one being multiplied
  // One LDS object per stage.  In the steady-state loop (unrolled six times: three slots x two register sets) every
  // stage is picked by a compile-time index and the waits go through the s_waitcnt builtin, so the compiler's own wait
  // insertion knows (a) that the DMA filling one stage cannot alias the transposing reads of another and (b) which
  // DMAs have landed.  With one array, a run-time stage and waits in inline asm it put a vmcnt(0) in front of the
  // first read of every stage (the builtin read carries no address information): the ring never had more than
  // the stage being read in flight.  The last stages of a range still take that form.
  static_assert(NSTAGE == 3, "three stages: see the loop");
  __shared__ __attribute__((aligned(1024))) char lds0[STAGE];
  __shared__ __attribute__((aligned(1024))) char lds1[STAGE];
  __shared__ __attribute__((aligned(1024))) char lds2[STAGE];
  auto slot_ptr = [&](int slot) __attribute__((always_inline)) -> char* { return slot == 0 ? lds0 : (slot == 1 ? lds1 : lds2); };
  static_assert(RBD % 256 == 0 && RBX % 256 == 0, "rows are whole bank rows");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // workgroups that stream the same token range (different tiles of dw) get ids that differ by
  // multiples of 8: one XCD, dispatched back to back, so the range is fetched from HBM once
  const int gxy = a.gn * a.gk;
  const int lin = blockIdx.x;
  const bool xcd_map = (a.nsplit & 7) == 0;
  const int cxy = xcd_map ? (lin >> 3) % gxy : lin % gxy;
  const int sp = xcd_map ? ((lin >> 3) / gxy) * 8 + (lin & 7) : lin / gxy;
  const int n0 = (cxy / a.gk) * BNo, k0 = (cxy % a.gk) * BKo;
  const int t_begin = sp * a.chunk;
  const int t_end = (t_begin + a.chunk < a.M) ? t_begin + a.chunk : a.M;
  const int nstage = t_begin < t_end ? (t_end - t_begin + TT - 1) / TT : 0;

  // ---- DMA sources: piece i is 1 KiB of the image; lane -> (token, 16-byte slot) of the image, whose
  // source slot is the swizzled one.  Channels past N / K and tokens past the range read zeros.
  constexpr int DPW = (DPIECES + NW - 1) / NW, XPW = (XPIECES + NW - 1) / NW;
  const bf16* zero = reinterpret_cast<const bf16*>(gemm_zero_page);
  int d_tok[DPW], d_off[DPW], x_tok[XPW], x_off[XPW];      // token in the stage (or 1 << 20: nothing to load)
#pragma unroll
  for (int k = 0; k < DPW; ++k) {
    const int i = wave + NW * k;
    const int o = i * 1024 + lane * 16;
    const int t = o / RBD, sl = (o % RBD) / 16;
    const int ss = (sl & ~15) | ((sl & 15) ^ (tok_key(t) << 1));
    const int c = n0 + ss * 8;
    d_tok[k] = (i < DPIECES && c < a.N) ? t : (1 << 20);
    d_off[k] = t * a.lddy + c;
  }
#pragma unroll
  for (int k = 0; k < XPW; ++k) {
    const int i = wave + NW * k;
    const int o = i * 1024 + lane * 16;
    const int t = o / RBX, sl = (o % RBX) / 16;
    const int ss = (sl & ~15) | ((sl & 15) ^ (tok_key(t) << 1));
    const int c = k0 + ss * 8;
    x_tok[k] = (i < XPIECES && c < a.K) ? t : (1 << 20);
    x_off[k] = t * a.ldx + c;
  }
  static_assert(DPIECES % NW == 0 && XPIECES % NW == 0 && NSTAGE == 3, "every wave issues the same pieces; ring of three");
  constexpr int NP = DPW + XPW;
  static_assert(2 * NP <= 16, "counted waits");
  // Full stages (all 32 tokens inside the range) are fetched through running pointers: one 64-bit add per piece and
  // no bounds test; only a last, partial stage takes the checked form.
  const int nfull = t_begin < t_end ? (t_end - t_begin) / TT : 0;
  const bf16* dptr[DPW];
  const bf16* xptr[XPW];
  long dstep[DPW], xstep[XPW];
#pragma unroll
  for (int k = 0; k < DPW; ++k) {
    const bool ok = d_tok[k] < (1 << 20);
    dptr[k] = ok ? a.dy + (long)t_begin * a.lddy + d_off[k] : zero;
    dstep[k] = ok ? (long)TT * a.lddy : 0;
  }
#pragma unroll
  for (int k = 0; k < XPW; ++k) {
    const bool ok = x_tok[k] < (1 << 20);
    xptr[k] = ok ? a.x + (long)t_begin * a.ldx + x_off[k] : zero;
    xstep[k] = ok ? (long)TT * a.ldx : 0;
  }
  auto issue_full = [&](char* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < DPW; ++k) {
      glds16g(dptr[k], dst + (wave + NW * k) * 1024);
      dptr[k] += dstep[k];
    }
#pragma unroll
    for (int k = 0; k < XPW; ++k) {
      glds16g(xptr[k], dst + DBYTES + (wave + NW * k) * 1024);
      xptr[k] += xstep[k];
    }
  };
  auto issue_any = [&](int st, int slot) __attribute__((always_inline)) {
    char* dst = slot_ptr(slot);
    if (st < nfull) {          // (uniform)
      issue_full(dst);
      return;
    }
    const int t0 = t_begin + st * TT;
    const bf16* dbase = a.dy + (long)t0 * a.lddy;
    const bf16* xbase = a.x + (long)t0 * a.ldx;
#pragma unroll
    for (int k = 0; k < DPW; ++k)
      glds16g(t0 + d_tok[k] < t_end ? dbase + d_off[k] : zero, dst + (wave + NW * k) * 1024);
#pragma unroll
    for (int k = 0; k < XPW; ++k)
      glds16g(t0 + x_tok[k] < t_end ? xbase + x_off[k] : zero, dst + DBYTES + (wave + NW * k) * 1024);
  };

  // ---- transposing-read addresses: group g = lane/16 reads tokens 8g+4h+q (h = 0, 1), lane (q, p) of the
  // group supplies channels 4p..4p+3 of the 16-channel tile
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p4 = li & 3;
  const int key = q | ((g & 1) << 2);
  int a_addr[4], b_addr[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int sd = wm * 8 + 2 * m + (p4 >> 1);
    a_addr[m] = (8 * g + q) * RBD + (((sd & ~15) | ((sd & 15) ^ (key << 1))) * 16) + (p4 & 1) * 8;
    const int sx = wn * 8 + 2 * m + (p4 >> 1);
    b_addr[m] = DBYTES + (8 * g + q) * RBX + (((sx & ~15) | ((sx & 15) ^ (key << 1))) * 16) + (p4 & 1) * 8;
  }

  f32x4 acc[4][4], accb[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    accb[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(accb[m]));
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      asm volatile("" : "+v"(acc[m][n]));      // (held in vector registers, opaque to the optimiser: see lin_gemm_kernel)
    }
  }
  const bool want_bias = a.bias_slab != nullptr && wn == 0 && k0 == 0;

  // As in lin_gemm_kernel: the fragments of stage st+1 are read (transposing reads, one between MFMAs) while stage
  // st is multiplied out of the other register set; the barrier that opens iteration st waits for stage st+1, and
  // slot st % 3, whose fragments every wave already holds, takes stage st+3.  The steady-state iteration is
  // straight-line code with a compile-time wait count; the last stages of a range take the general form.
  typedef __attribute__((ext_vector_type(8))) short short8v;
  bf16x8 af[2][4], bfr[2][4];
  auto load_frags = [&](auto bufc, const char* base) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, base + a_addr[m]));
      const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, base + a_addr[m] + 4 * RBD));
      af[BUF][m] = __builtin_bit_cast(bf16x8, short8v{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, base + b_addr[n]));
      const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, base + b_addr[n] + 4 * RBX));
      bfr[BUF][n] = __builtin_bit_cast(bf16x8, short8v{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
    }
  };
  auto multiply = [&](auto curc) __attribute__((always_inline)) {
    constexpr int CUR = decltype(curc)::value;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[CUR][m], bfr[CUR][n], acc[m][n], 0, 0, 0);
  };
  auto bias_mma = [&](auto curc) __attribute__((always_inline)) {
    constexpr int CUR = decltype(curc)::value;
    if (want_bias) {      // (wave-uniform) dy^T times a matrix of ones: every column = the sum over the stage's tokens
      const u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
      for (int m = 0; m < 4; ++m)
        accb[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[CUR][m], __builtin_bit_cast(bf16x8, ones), accb[m], 0, 0, 0);
    }
  };
  auto next3 = [](int st) { return st == 2 ? 0 : st + 1; };
  int sr = 1, si = 0;          // slot of stage st+1 (read next), slot of stage st+3 (= that of stage st)
  auto steady = [&](auto kc) __attribute__((always_inline)) {      // stages st+1, st+2 exist and st+3 is a full stage
    constexpr int K = decltype(kc)::value;          // stage number modulo 6
    char* const mine = K % 3 == 0 ? lds0 : (K % 3 == 1 ? lds1 : lds2);
    const char* const next = (K + 1) % 3 == 0 ? lds0 : ((K + 1) % 3 == 1 ? lds1 : lds2);
    wait_vm_then_barrier<NP, true>();
    issue_full(mine);
    load_frags(std::integral_constant<int, (K & 1) ^ 1>{}, next);
    multiply(std::integral_constant<int, K & 1>{});
#pragma unroll
    for (int k = 0; k < 16; ++k) {          // issue order: a transposing read behind every MFMA
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    bias_mma(std::integral_constant<int, K & 1>{});
  };
  auto general = [&](auto curc, int st) __attribute__((always_inline)) {
    constexpr int CUR = decltype(curc)::value;
    if (st + 1 < nstage) {          // (uniform)
      if (st + 2 < nstage) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (st + 3 < nstage) issue_any(st + 3, si);
      load_frags(std::integral_constant<int, CUR ^ 1>{}, slot_ptr(sr));
    }
    multiply(curc);
    bias_mma(curc);
    sr = next3(sr);
    si = next3(si);
  };
  if (nstage > 0) {
    issue_any(0, 0);
    if (nstage > 1) issue_any(1, 1);
    if (nstage > 2) issue_any(2, 2);
    if (nstage > 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * NP) : "memory");      // stage 0 has landed
    else if (nstage > 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    load_frags(std::integral_constant<int, 0>{}, lds0);
    int st = 0;
#pragma unroll 1
    for (; st + 5 + 4 < nfull; st += 6) static_for<6>([&](auto kc) __attribute__((always_inline)) { steady(kc); });
    static_for<5>([&](auto kc) __attribute__((always_inline)) {      // the (at most five) steady stages left over
      if (st + 4 < nfull) {          // (uniform)
        steady(kc);
        ++st;
      }
    });
    sr = (st + 1) % 3;
    si = st % 3;
    if ((st & 1) && st < nstage) {          // (uniform) stage st's fragments are in register set st & 1
      general(std::integral_constant<int, 1>{}, st);
      ++st;
    }
#pragma unroll 1
    for (; st < nstage; st += 2) {
      general(std::integral_constant<int, 0>{}, st);
      if (st + 1 < nstage) general(std::integral_constant<int, 1>{}, st + 1);
    }
  }

  // slab[sp][n][k]; C layout: column (k) = lane&15, row (n) = 4*(lane>>4) + r
  float* slab = a.slab + (long)sp * a.N_pad * a.K_pad;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = n0 + wm * 64 + m * 16 + 4 * g + r;
        const int col = k0 + wn * 64 + n * 16 + li;
        slab[(long)row * a.K_pad + col] = acc[m][n][r];
      }
  if (want_bias && li == 0) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) a.bias_slab[(long)sp * a.N_pad + n0 + wm * 64 + m * 16 + 4 * g + r] = accb[m][r];
  }
}

// dw[n][k] (+)= sum_s slab[s][n][k], db[n] (+)= sum_s bias_slab[s][n].  A workgroup owns 64 quads of dw and
// four slab lanes: lane j sums the slabs j, j+4, ... in order, then the four lane sums are added in lane
// order: the same bits every run, four times the loads in flight.
__global__ __launch_bounds__(256) void lin_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                               const float* __restrict__ bias_slab,
                                                               float* __restrict__ db, int nsplit, int N, int K,
                                                               int N_pad, int K_pad, int accumulate, int nmain_blocks) {
  __shared__ f32x4 part[4][64];
  if ((int)blockIdx.x >= nmain_blocks) {
    const int n = (blockIdx.x - nmain_blocks) * 256 + threadIdx.x;
    if (db != nullptr && n < N) {
      float s = 0.f;
      for (int i = 0; i < nsplit; ++i) s += bias_slab[(long)i * N_pad + n];
      db[n] = accumulate ? db[n] + s : s;
    }
    return;
  }
  const int k4 = K / 4;
  const long nquads = (long)N * k4;
  const int e = threadIdx.x & 63, j = threadIdx.x >> 6;
  const long idx = (long)blockIdx.x * 64 + e;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  int n = 0, kq = 0;
  if (idx < nquads) {
    n = idx / k4;
    kq = idx - (long)n * k4;
    const float* p = slab + (long)n * K_pad + kq * 4;
    const long stride = (long)N_pad * K_pad;
    int i = j;
    for (; i + 12 < nsplit; i += 16) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + i * stride);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (i + 4) * stride);
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (i + 8) * stride);
      const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (i + 12) * stride);
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; i < nsplit; i += 4) s += *reinterpret_cast<const f32x4*>(p + i * stride);
  }
  part[j][e] = s;
  __syncthreads();
  if (j == 0 && idx < nquads) {
    const f32x4 t = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
    f32x4* d = reinterpret_cast<f32x4*>(dw + (long)n * K + kq * 4);
    *d = accumulate ? (*d + t) : t;
  }
}


// The same sum for MANY slabs of a small dw (one slab per workgroup of the whole-row kernel): 16 slab lanes
// per quad of dw (or per bias element), each summing its slabs j, j+16, ... in order with four loads in
// flight, the 16 lane sums added in lane order: a dependency chain of nsplit/64 loads instead of nsplit/16,
// the same bits every run.
__global__ __launch_bounds__(256) void lin_wgrad_reduce16_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                 const float* __restrict__ bias_slab,
                                                                 float* __restrict__ db, int nsplit, int N, int K,
                                                                 int accumulate, int nmain_blocks) {
  __shared__ f32x4 part[16][16];
  const int e = threadIdx.x & 15, j = threadIdx.x >> 4;
  if ((int)blockIdx.x >= nmain_blocks) {
    const int n = (blockIdx.x - nmain_blocks) * 16 + e;
    float sum = 0.f;
    if (n < N)
      for (int i = j; i < nsplit; i += 16) sum += bias_slab[(long)i * N + n];
    part[j][e] = f32x4{sum, 0.f, 0.f, 0.f};
    __syncthreads();
    if (j == 0 && n < N) {
      float t = part[0][e][0];
#pragma unroll
      for (int k = 1; k < 16; ++k) t += part[k][e][0];
      db[n] = accumulate ? db[n] + t : t;
    }
    return;
  }
  const long nquads = (long)N * (K / 4);
  const long idx = (long)blockIdx.x * 16 + e;
  f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
  if (idx < nquads) {
    const float* p = slab + idx * 4;
    const long stride = (long)N * K;
    int i = j;
    for (; i + 48 < nsplit; i += 64) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + i * stride);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (i + 16) * stride);
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (i + 32) * stride);
      const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (i + 48) * stride);
      sum = (((sum + v0) + v1) + v2) + v3;
    }
    for (; i < nsplit; i += 16) sum += *reinterpret_cast<const f32x4*>(p + i * stride);
  }
  part[j][e] = sum;
  __syncthreads();
  if (j == 0 && idx < nquads) {
    f32x4 t = part[0][e];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += part[k][e];
    f32x4* d = reinterpret_cast<f32x4*>(dw + idx * 4);
    *d = accumulate ? (*d + t) : t;
  }
}

struct WgPlan {
  int N_pad, K_pad, gn, gk, nsplit, chunk;
};

inline WgPlan plan_lin_wgrad(long M, int K, int N) {
  WgPlan p;
  const int bn = 128, bk = 128;
  p.N_pad = cdiv(N, bn) * bn;
  p.K_pad = cdiv(K, bk) * bk;
  p.gn = p.N_pad / bn;
  p.gk = p.K_pad / bk;
  const long tiles = (long)p.gn * p.gk;
  // Token split: more ranges = more workgroups in flight and fewer stages each, but every range costs one
  // more fp32 copy of dw written and read back.  Pick the count with the lowest modelled time
  // (0.35 us per 32-token stage at two workgroups per CU, 3 TB/s for the slabs; measured on MI355X).
  const double stages = (double)((M + 31) / 32);
  const double slab_us = 2.0 * 4.0 * p.N_pad * p.K_pad / 3.0e6;
  long best_s = 1;
  double best = 1e30;
  for (long s = 1; s <= 256 && s * 256 <= M + 255; s += (s < 8 ? 1 : 8)) {
    const double rounds = (double)((tiles * s + 511) / 512);
    const double t = stages / s * 0.35 * rounds + slab_us * s;
    if (t < best) {
      best = t;
      best_s = s;
    }
  }
  p.nsplit = (int)best_s;
  p.chunk = (int)(((M + best_s - 1) / best_s + 31) / 32 * 32);
  return p;
}


// ------------------------------------------------------------------------------------------------
// Whole-row weight gradient of the NARROW token linears (stages 0 and 1 of the Swin encoder: 96 ... 768
// channels over 65 536 ... 262 144 tokens): dw is small and the reduction long, so the product is bound by
// the bytes of x and dy.  A workgroup of eight waves owns a COLUMN of dw - all of it when it fits the
// registers (<= 36 accumulator tiles per wave), else a range of 16-row / 16-column tiles - and streams its
// tokens exactly once: token rows are staged whole (the LDS images of dy and x are byte-for-byte copies of
// TT consecutive token rows, so every DMA piece reads 1 KiB of consecutive memory), waves split the
// output rows (NTW 16-row tiles each, all KTC column tiles), nothing is summed through LDS, and each
// workgroup leaves one fp32 slab for lin_wgrad_reduce_kernel.  Both fragments are read with the transposing
// ds_read_b64_tr_b16; the 32-byte unit of a tile inside the largest power-of-two block that divides the row
// pitch is XORed with token bits (on the DMA source and on the read address alike: a permutation inside a
// piece costs nothing) so that the eight token rows of a read fall on eight different bank groups.
struct RowsArgs {
  const bf16* x;
  const bf16* dy;
  float* slab;        // [nwg][N][K]
  float* bias_slab;   // [nwg][N]
  int M, nwg, ntile;
};

template <int K, int N> struct RowsShape {
  static constexpr int NTIL = N / 16, KTIL = K / 16;
  static constexpr int CK = KTIL > 24 ? KTIL / 12 : 1;                 // columns along K
  static constexpr int KTC = KTIL / CK;                                // k-tiles per column
  static constexpr int NTW_MAX = (36 / KTC) < 3 ? (36 / KTC) : 3;
  static constexpr int CN = (NTIL + 8 * NTW_MAX - 1) / (8 * NTW_MAX);  // columns along N
  static constexpr int NTC = NTIL / CN;                                // n-tiles per column
  static constexpr int NTW = (NTC + 7) / 8;                            // n-tiles per wave
  static constexpr int NCOL = CN * CK;
  static constexpr int TT = (2 * 64 * (N + K) * 2 <= 150 * 1024) ? 64 : 32;   // tokens per stage
  static constexpr int DBYTES = TT * N * 2, XBYTES = TT * K * 2, STAGE = DBYTES + XBYTES;
  static constexpr int S = (3 * STAGE <= 150 * 1024) ? 3 : 2;
  static_assert(N % 16 == 0 && K % 16 == 0 && NTIL % CN == 0 && KTIL % CK == 0 && STAGE % 1024 == 0 && DBYTES % 1024 == 0,
                "shape");
};

// swizzle block (bytes) of a row pitch and the XOR key of a token row
__host__ __device__ constexpr int rows_block(int pitch) { return pitch % 256 == 0 ? 256 : (pitch % 128 == 0 ? 128 : 64); }
template <int BS> __device__ __forceinline__ int rows_key(int t) {
  if (BS == 256) return (t & 3) | (((t >> 3) & 1) << 2);
  if (BS == 128) return ((t >> 1) & 1) | (((t >> 3) & 1) << 1);
  return (t >> 3) & 1;
}

template <int K, int N>
__global__ __launch_bounds__(512) void lin_wgrad_rows_kernel(RowsArgs a) {
  using SH = RowsShape<K, N>;
  constexpr int PD = N * 2, PX = K * 2;                      // row pitches
  constexpr int BD = rows_block(PD), BX = rows_block(PX);
  constexpr int TT = SH::TT, S = SH::S, STAGE = SH::STAGE, DBYTES = SH::DBYTES;
  constexpr int NINSTR = STAGE / 1024, NPW = (NINSTR + 7) / 8, DINSTR = DBYTES / 1024;
  constexpr int NTW = SH::NTW, KTC = SH::KTC, KB = KTC > 12 ? 12 : KTC;     // k-tiles per register batch
  static_assert(KTC % KB == 0 && 2 * NPW < 64, "batches");
  __shared__ __attribute__((aligned(1024))) char buf0[STAGE];
  __shared__ __attribute__((aligned(1024))) char buf1[STAGE];
  __shared__ __attribute__((aligned(1024))) char buf2[S == 3 ? STAGE : 1024];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bf16* zero = reinterpret_cast<const bf16*>(gemm_zero_page);
  // column and token-tile list of this workgroup: the columns of one list get ids a multiple of 8 apart (one XCD)
  constexpr int NCOL = SH::NCOL;
  const int col = NCOL > 1 ? (blockIdx.x >> 3) % NCOL : 0;
  const int pos = NCOL > 1 ? ((blockIdx.x >> 3) / NCOL) * 8 + (blockIdx.x & 7) : blockIdx.x;
  const int cn = col % SH::CN, ck = col / SH::CN;
  const int n = pos < a.ntile ? (a.ntile - pos + a.nwg - 1) / a.nwg : 0;

  // ---- DMA table: piece i = wave + 8k is bytes [1024 i, 1024 i + 1024) of the stage; the source is the same
  // byte of the token block with the swizzled 32-byte unit; rows past M read zeros
  int p_off[NPW], p_tok[NPW];
#pragma unroll
  for (int k = 0; k < NPW; ++k) {
    int i = wave + 8 * k;
    if (i >= NINSTR) i = NINSTR - 1;
    const bool isd = i < DINSTR;
    const int o = (isd ? i : i - DINSTR) * 1024 + lane * 16;
    const int pitch = isd ? PD : PX, bs = isd ? BD : BX;
    const int t = o / pitch, cb = o - t * pitch;
    const int key = isd ? rows_key<BD>(t) : rows_key<BX>(t);
    const int cbs = (cb & ~(bs - 1)) | ((cb & (bs - 1)) ^ (key * 32));
    p_off[k] = (t * pitch + cbs) >> 1;
    p_tok[k] = t;
  }
  auto issue_tile = [&](int tile, char* dst) {
    const int t0 = tile * TT;
    const bf16* dbase = a.dy + (long)t0 * N;
    const bf16* xbase = a.x + (long)t0 * K;
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
      int i = wave + 8 * k;
      if (i >= NINSTR) i = NINSTR - 1;
      const bool ok = t0 + p_tok[k] < a.M;
      const bf16* src = ok ? (i < DINSTR ? dbase : xbase) + p_off[k] : zero;
      glds16g(src, dst + i * 1024);
    }
  };

  // ---- fragment addresses: lane (g, q, p4) reads tokens 8g + q (+4) at the 8 bytes (p4) of a 16-channel tile
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p4 = li & 3;
  const int tl = 8 * g + q;
  const int keyd = rows_key<BD>(tl), keyx = rows_key<BX>(tl);
  constexpr int UD = BD / 32, UX = BX / 32;
  const int nt0 = cn * SH::NTC + wave * NTW;                 // this wave's first n-tile (tiles past the column: idle)
  const bool active = wave * NTW < SH::NTC;
  int a_off[NTW], b_off[KTC];
#pragma unroll
  for (int m = 0; m < NTW; ++m) {
    int nt = nt0 + m;
    if (nt >= (cn + 1) * SH::NTC) nt = (cn + 1) * SH::NTC - 1;
    const int nts = (nt & ~(UD - 1)) | ((nt ^ keyd) & (UD - 1));
    a_off[m] = tl * PD + nts * 32 + (p4 >> 1) * 16 + (p4 & 1) * 8;
  }
#pragma unroll
  for (int j = 0; j < KTC; ++j) {
    const int kt = ck * KTC + j;
    const int kts = (kt & ~(UX - 1)) | ((kt ^ keyx) & (UX - 1));
    b_off[j] = DBYTES + tl * PX + kts * 32 + (p4 >> 1) * 16 + (p4 & 1) * 8;
  }

  f32x4 acc[NTW][KTC], accb[NTW];
#pragma unroll
  for (int m = 0; m < NTW; ++m) {
    accb[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(accb[m]));
#pragma unroll
    for (int j = 0; j < KTC; ++j) {
      acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      asm volatile("" : "+v"(acc[m][j]));      // (opaque zeros: see conv_wgrad_kernel)
    }
  }
  const bool want_bias = a.bias_slab != nullptr && ck == 0;

  auto tr8 = [&](const char* p, int pitch) __attribute__((always_inline)) {
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, p));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, p + 4 * pitch));
    typedef __attribute__((ext_vector_type(8))) short short8v;
    return __builtin_bit_cast(bf16x8, short8v{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
  };
  auto multiply = [&](const char* lds) {
    if (!active) return;
#pragma unroll
    for (int ks = 0; ks < TT / 32; ++ks) {
      bf16x8 af[NTW];
#pragma unroll
      for (int m = 0; m < NTW; ++m) af[m] = tr8(lds + a_off[m] + ks * 32 * PD, PD);
      if (want_bias) {      // (uniform) dy^T times a matrix of ones: every column = the sum over the step's tokens
        const u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
        for (int m = 0; m < NTW; ++m)
          accb[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], __builtin_bit_cast(bf16x8, ones), accb[m], 0, 0, 0);
      }
#pragma unroll
      for (int jb = 0; jb < KTC; jb += KB) {
        bf16x8 bfr[KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) bfr[j] = tr8(lds + b_off[jb + j] + ks * 32 * PX, PX);
#pragma unroll
        for (int j = 0; j < KB; ++j)
#pragma unroll
          for (int m = 0; m < NTW; ++m)
            acc[m][jb + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr[j], acc[m][jb + j], 0, 0, 0);
      }
    }
  };

  auto tile_at = [&](int j) { return pos + j * a.nwg; };
  if constexpr (S == 2) {
    if (n > 0) issue_tile(tile_at(0), buf0);
#pragma unroll 1
    for (int j = 0; j < n; j += 2) {
      __syncthreads();
      if (j + 1 < n) issue_tile(tile_at(j + 1), buf1);
      multiply(buf0);
      if (j + 1 >= n) break;
      __syncthreads();
      if (j + 2 < n) issue_tile(tile_at(j + 2), buf0);
      multiply(buf1);
    }
  } else {
    auto sync_tile = [&](bool newer_in_flight) {
      if (newer_in_flight) wait_vm_then_barrier<NPW, false>();      // (the builtin form: see common.h)
      else wait_vm_then_barrier<0, false>();
    };
    if (n > 0) issue_tile(tile_at(0), buf0);
    if (n > 1) issue_tile(tile_at(1), buf1);
#pragma unroll 1
    for (int j = 0; j < n; j += 3) {
      sync_tile(j + 1 < n);
      if (j + 2 < n) issue_tile(tile_at(j + 2), buf2);
      multiply(buf0);
      if (j + 1 >= n) break;
      sync_tile(j + 2 < n);
      if (j + 3 < n) issue_tile(tile_at(j + 3), buf0);
      multiply(buf1);
      if (j + 2 >= n) break;
      sync_tile(j + 3 < n);
      if (j + 4 < n) issue_tile(tile_at(j + 4), buf1);
      multiply(buf2);
    }
  }

  // slab[pos][n][k]; C layout: column (k) = lane&15, row (n) = 4*(lane>>4) + r
  if (!active) return;
  float* slab = a.slab + (long)pos * N * K;
#pragma unroll
  for (int m = 0; m < NTW; ++m) {
    const int nt = nt0 + m;
    if (nt >= (cn + 1) * SH::NTC) continue;
#pragma unroll
    for (int j = 0; j < KTC; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        slab[(long)(nt * 16 + 4 * g + r) * K + (ck * KTC + j) * 16 + li] = acc[m][j][r];
    if (want_bias && li == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a.bias_slab[(long)pos * N + nt * 16 + 4 * g + r] = accb[m][r];
    }
  }
}

template <int K, int N> int rows_nwg(long M) {
  using SH = RowsShape<K, N>;
  const long ntile = (M + SH::TT - 1) / SH::TT;
  int nwg = (int)((ntile / 8) & ~7L);           // at least eight tiles per list
  const int most = (256 / SH::NCOL) & ~7;
  if (nwg > most) nwg = most;
  return nwg;
}
template <int K, int N>
int launch_rows(const void* x, const void* dy, float* dw, float* db, long M, void* ws, long ws_bytes, int accumulate,
                hipStream_t stream) {
  using SH = RowsShape<K, N>;
  RowsArgs a;
  a.x = (const bf16*)x; a.dy = (const bf16*)dy; a.M = (int)M;
  a.ntile = (int)((M + SH::TT - 1) / SH::TT);
  a.nwg = rows_nwg<K, N>(M);
  const long main_bytes = (long)a.nwg * N * K * 4;
  UNETK_CHECK_ARG(ws_bytes >= main_bytes + (long)a.nwg * N * 4, "linear_wgrad: workspace %ld < %ld bytes", ws_bytes,
                  main_bytes + (long)a.nwg * N * 4);
  a.slab = (float*)ws;
  a.bias_slab = db != nullptr ? (float*)((char*)ws + main_bytes) : nullptr;
  hipLaunchKernelGGL((lin_wgrad_rows_kernel<K, N>), dim3((unsigned)(a.nwg * SH::NCOL)), dim3(512), 0, stream, a);
  UNETK_CHECK_LAUNCH();
  const int nmain = cdiv((long)N * (K / 4), 16);
  hipLaunchKernelGGL(lin_wgrad_reduce16_kernel, dim3((unsigned)(nmain + (db != nullptr ? cdiv(N, 16) : 0))), dim3(256), 0,
                     stream, a.slab, dw, a.bias_slab, db, a.nwg, N, K, accumulate, nmain);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

#define UNETK_ROWS_SHAPES(X) X(96, 288) X(96, 96) X(96, 384) X(384, 96) X(192, 576) X(192, 192) X(192, 768) X(768, 192)

template <int TM, int TN, bool GIN = false> int launch_gemm(GemmArgs a, hipStream_t stream) {
  constexpr int BM = 64 * TM;
  a.mtiles = cdiv(a.M, BM);
  a.tiles_per_xcd = cdiv(a.mtiles, 8);
  a.nby = a.N / (16 * TN);
  hipLaunchKernelGGL((lin_gemm_kernel<TM, TN, GIN>), dim3((unsigned)(8 * a.tiles_per_xcd * a.nby)), dim3(256), 0, stream, a);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // namespace

// Hooks for unetk_linear_wgrad (conv_wgrad.hip): the whole-row kernel takes dense bfloat16 operands of the
// shapes listed above with at least 2048 tokens per workgroup list.
int64_t lin_wgrad_rows_workspace(int64_t M, int K, int N) {
#define X(K_, N_) if (K == K_ && N == N_) { const int nwg = rows_nwg<K_, N_>(M); return nwg >= 8 ? (int64_t)nwg * N_ * (K_ + 1) * 4 : 0; }
  UNETK_ROWS_SHAPES(X)
#undef X
  return 0;
}
int lin_wgrad_rows_try(const void* x, int ldx, const void* dy, int lddy, float* dw, float* db, int64_t M, int K, int N,
                       void* ws, int64_t ws_bytes, int accumulate, hipStream_t stream, int* rc) {
  if (ldx != K || lddy != N || M >= (1LL << 31) - 64 || getenv("UNETK_LINEAR_WGRAD_NO_ROWS") != nullptr) return 0;
  if (lin_wgrad_rows_workspace(M, K, N) == 0) return 0;
#define X(K_, N_) if (K == K_ && N == N_) { *rc = launch_rows<K_, N_>(x, dy, dw, db, M, ws, ws_bytes, accumulate, stream); return 1; }
  UNETK_ROWS_SHAPES(X)
#undef X
  return 0;
}

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
  a.shuf_up = a.shuf_w = a.shuf_hw = 0;
  a.gin_up = a.gin_w = a.gin_hw = 0;
  // 128-token tiles, three workgroups per CU.  (256-token tiles, two per CU, measured slower on every
  // Swin shape, 64x64 tokens x 96 channels to 8x8 tokens x 3072 channels: these products are bound by
  // bytes and by how many workgroups cover each other's waits, not by the MFMA : LDS-read ratio.)
  switch (bn) {
    case 96: return launch_gemm<2, 6>(a, stream);
    case 64: return launch_gemm<2, 4>(a, stream);
    default: return launch_gemm<2, 2>(a, stream);
  }
}

int unetk_convt2x2_fwd(const void* x, int ldx, const void* wpacked, void* out, int ldo, int B, int h, int w, int K,
                       int up, int bn, int dtype, const float* bias, hipStream_t stream) {
  UNETK_CHECK_ARG(x && wpacked && out && B > 0 && h > 0 && w > 0 && up > 0, "convt2x2_fwd: bad arguments");
  const int64_t M = (int64_t)B * h * w;
  const int N = 4 * up;
  UNETK_CHECK_ARG(unetk_linear_supported(M, K, N, bn, dtype) && up % 8 == 0 && 4 * M < (1LL << 31),
                  "convt2x2_fwd: unsupported shape or type (B=%d h=%d w=%d K=%d up=%d block %d dtype %d): bfloat16, K a "
                  "multiple of 32, 4*up a multiple of its channel block, up a multiple of 8", B, h, w, K, up, bn, dtype);
  UNETK_CHECK_ARG(ldx >= K && ldo >= up && ldx % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)x % 16) == 0 &&
                  ((uintptr_t)out % 16) == 0 && ((uintptr_t)wpacked % 16) == 0, "convt2x2_fwd: bad strides or alignment");
  GemmArgs a;
  a.x = (const bf16*)x; a.w = (const bf16*)wpacked; a.out = (bf16*)out; a.out2 = nullptr;
  a.bias = bias; a.resid = nullptr; a.keep = nullptr; a.aux = nullptr;
  a.ldx = ldx; a.ldo = ldo; a.ldo2 = 0; a.ldr = 0; a.ldaux = 0;
  a.M = (int)M; a.N = N; a.K = K; a.nchunks = K / 32; a.act = UNETK_ACT_NONE; a.tok_per_sample = 1;
  a.shuf_up = up; a.shuf_w = w; a.shuf_hw = h * w;
  a.gin_up = a.gin_w = a.gin_hw = 0;
  switch (bn) {
    case 96: return launch_gemm<2, 6>(a, stream);
    case 64: return launch_gemm<2, 4>(a, stream);
    default: return launch_gemm<2, 2>(a, stream);
  }
}

int unetk_convt2x2_bwd_data(const void* dout, int lddo, const void* wpackedT, void* dx, int ldo, int B, int h, int w,
                            int up, int N, int bn, int dtype, hipStream_t stream) {
  UNETK_CHECK_ARG(dout && wpackedT && dx && B > 0 && h > 0 && w > 0 && up > 0 && N > 0, "convt2x2_bwd_data: bad arguments");
  const int64_t M = (int64_t)B * h * w;
  const int K = 4 * up;
  UNETK_CHECK_ARG(unetk_linear_supported(M, K, N, bn, dtype) && up % 32 == 0 && 4 * M < (1LL << 31),
                  "convt2x2_bwd_data: unsupported shape or type (B=%d h=%d w=%d up=%d N=%d block %d dtype %d): bfloat16, "
                  "up a multiple of 32, N a multiple of its channel block", B, h, w, up, N, bn, dtype);
  UNETK_CHECK_ARG(lddo >= up && ldo >= N && lddo % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)dout % 16) == 0 &&
                  ((uintptr_t)dx % 16) == 0 && ((uintptr_t)wpackedT % 16) == 0 && 4 * M * (int64_t)lddo < (1LL << 40),
                  "convt2x2_bwd_data: bad strides or alignment");
  GemmArgs a;
  a.x = (const bf16*)dout; a.w = (const bf16*)wpackedT; a.out = (bf16*)dx; a.out2 = nullptr;
  a.bias = nullptr; a.resid = nullptr; a.keep = nullptr; a.aux = nullptr;
  a.ldx = lddo; a.ldo = ldo; a.ldo2 = 0; a.ldr = 0; a.ldaux = 0;
  a.M = (int)M; a.N = N; a.K = K; a.nchunks = K / 32; a.act = UNETK_ACT_NONE; a.tok_per_sample = 1;
  a.shuf_up = a.shuf_w = a.shuf_hw = 0;
  a.gin_up = up; a.gin_w = w; a.gin_hw = h * w;
  switch (bn) {
    case 96: return launch_gemm<2, 6, true>(a, stream);
    case 64: return launch_gemm<2, 4, true>(a, stream);
    default: return launch_gemm<2, 2, true>(a, stream);
  }
}

int unetk_token_wgrad_supported(int64_t M, int K, int N, int dtype) {
  return dtype == UNETK_BF16 && M > 0 && M < (1LL << 31) - 64 && K > 0 && N > 0 && K % 8 == 0 && N % 8 == 0;
}

int64_t unetk_token_wgrad_workspace_bytes(int64_t M, int K, int N) {
  const WgPlan p = plan_lin_wgrad(M, K, N);
  const int64_t tiles = (int64_t)p.nsplit * p.N_pad * (p.K_pad + 1) * 4;
  const int64_t rows = lin_wgrad_rows_workspace(M, K, N);
  return tiles > rows ? tiles : rows;
}

int unetk_token_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, float* dbias, int64_t M, int K,
                      int N, int dtype, void* workspace, int64_t workspace_bytes, int accumulate,
                      hipStream_t stream) {
  UNETK_CHECK_ARG(x && dy && dw && workspace, "token_wgrad: null pointer");
  UNETK_CHECK_ARG(unetk_token_wgrad_supported(M, K, N, dtype),
                  "token_wgrad: unsupported shape or type (M=%lld K=%d N=%d dtype %d): bfloat16, channel counts "
                  "multiples of 8", (long long)M, K, N, dtype);
  UNETK_CHECK_ARG(ldx >= K && lddy >= N && ldx % 8 == 0 && lddy % 8 == 0, "token_wgrad: bad strides");
  UNETK_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dw % 16) == 0,
                  "token_wgrad: pointers must be 16-byte aligned");
  UNETK_CHECK_ARG((int64_t)M * (ldx > lddy ? ldx : lddy) < (1LL << 31) * 32,
                  "token_wgrad: tensor too large for the 32-bit stage offsets");
  UNETK_CHECK_ARG(K % 4 == 0, "token_wgrad: K must be a multiple of 4");
  {
    // narrow layers over many tokens: the whole-row kernel (every token streamed once by one column of workgroups)
    int rc = UNETK_OK;
    if (lin_wgrad_rows_try(x, ldx, dy, lddy, dw, dbias, M, K, N, workspace, workspace_bytes, accumulate, stream, &rc))
      return rc;
  }
  const WgPlan p = plan_lin_wgrad(M, K, N);
  const int64_t main_bytes = (int64_t)p.nsplit * p.N_pad * p.K_pad * 4;
  const int64_t need = main_bytes + (int64_t)p.nsplit * p.N_pad * 4;
  UNETK_CHECK_ARG(workspace_bytes >= need, "token_wgrad: workspace %lld < %lld bytes", (long long)workspace_bytes,
                  (long long)need);
  WgArgs a;
  a.x = (const bf16*)x; a.dy = (const bf16*)dy; a.slab = (float*)workspace;
  a.bias_slab = dbias ? (float*)((char*)workspace + main_bytes) : nullptr;
  a.ldx = ldx; a.lddy = lddy; a.M = (int)M; a.N = N; a.K = K; a.N_pad = p.N_pad; a.K_pad = p.K_pad;
  a.nsplit = p.nsplit; a.chunk = p.chunk; a.gn = p.gn; a.gk = p.gk;
  const unsigned grid = (unsigned)(p.gn * p.gk * p.nsplit);
  // (256 x 128 tiles with eight waves and a five-stage ring measured slower on every Swin shape)
  hipLaunchKernelGGL((lin_wgrad_kernel<2, 2, 3>), dim3(grid), dim3(256), 0, stream, a);
  UNETK_CHECK_LAUNCH();
  const int nmain = cdiv((long)N * (K / 4), 64);
  hipLaunchKernelGGL(lin_wgrad_reduce_kernel, dim3((unsigned)(nmain + (dbias ? cdiv(N, 256) : 0))), dim3(256), 0, stream,
                     a.slab, dw, a.bias_slab, dbias, p.nsplit, N, K, p.N_pad, p.K_pad, accumulate, nmain);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
