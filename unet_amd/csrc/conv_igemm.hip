// Implicit-GEMM convolution (3x3 pad 1, or 1x1) on MFMA for gfx950, NHWC activations.
//
//   out[b,y,x,co] = sum_{ky,kx,ci} in[b,y+ky-1,x+kx-1,ci] * w[co,ci,ky,kx]
//
// GEMM view: M = pixels, N = Cout, K = taps*Cin.  A workgroup (4 waves) owns a 16x16
// pixel tile (3x3) or 256 consecutive pixels (1x1) and BN = NT*16 output channels; wave
// w owns tile rows 4w..4w+3 (four 16-pixel MFMA row tiles) times all NT column tiles.
// The K loop runs over channel chunks of KC (32 bf16 / 16 f32 = 64 bytes per pixel):
// the 18x18 input halo of the chunk is staged in LDS once and re-read by all nine
// taps as shifted windows, so global->LDS traffic is ~1.27x the tile instead of 9x.
// Weights come pre-packed as [n-block][chunk][tap][BN][KC] so a kernel row's three
// taps are one contiguous slab.
//
// Epilogue (all optional): per-channel partial sums / sums of squares of the fp32
// accumulators over the tile's valid pixels (BatchNorm batch statistics, reduced later
// in a deterministic order), y*scale+shift (folded BatchNorm or bias), ReLU or GELU,
// residual add with a per-sample scale.  The result goes through a per-wave LDS
// staging tile so that global stores are 16-byte vectors along the channel axis.
//
// Used for: forward conv (reference training.py:523-529), data-gradient conv (same
// kernel, weights packed transposed and flipped), Swin linear layers as 1x1.
#include "common.h"
#include "mma.h"
#include <type_traits>

namespace {

// what out-of-image pixels and channel padding load
__device__ __attribute__((aligned(64))) unsigned int conv_zero_page[16];

// LDS-DMA: 16 bytes per lane from `src` to (wave-uniform) `lds_dst` + lane*16.  The builtin exists
// only in the device pass; the host pass just needs the kernel's signature.
__device__ __forceinline__ void glds16(const void* src, char* lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(src, LDS_PTR(void, lds_dst), 16, 0, 0);
#endif
}

struct ConvArgs {
  const void* in;
  const void* w;
  void* out;
  float* stats;            // [tiles][2][Co] or null
  const float* ep_scale;   // [Co] or null (=1)
  const float* ep_shift;   // [Co] or null (=0)
  void* out2;              // output channels >= split go here (ld = ldout2), renumbered from 0; or null
  void* pooled;            // 2x2 max-pool of the (post-activation) output, dense [B][H/2][W/2][Co]; or null
  const float* head_w;     // fused trailing 1x1 conv: weights [head_co][Co] ...
  const float* head_b;     // ... bias [head_co] or null ...
  float* head_out;         // ... and its NCHW fp32 output [B][head_co][H][W]; or null
  int head_co;
  const void* resid;       // residual (same layout as out, ld = ldres) or null
  const float* resid_keep; // per-sample scale applied to the conv result before the add, or null
  // BatchNorm-backward sums of the layer whose input gradient this launch produces (BST instantiations, data
  // gradient of the second conv of a block): with y that layer's conv output and z = y*scale+shift,
  // partials[tile][0][c] = sum over the tile's pixels of dz, [1][c] = sum of dz*(y-mean)*invstd, dz = (z > 0 ? out : 0)
  // on the values as they are stored: what channel_reduce_kernel's first mode forms from the stored tensors.
  const void* bs_y;
  const float* bs_scale;
  const float* bs_shift;
  const float* bs_mean;
  const float* bs_invstd;
  float* bs_partials;
  int ldbs;
  const void* in2;         // input channels >= Ci1 come from here (ld = ldin2, renumbered from 0), or null:
  int ldin2, Ci1;          // the operand is the channel concatenation [in | in2] without ever being stored
  int ldin, ldout, ldres, ldout2, split;
  int B, H, W, Ci, Co;
  int nchunks;
  int act;                 // 0 none, 1 relu, 2 gelu(erf)
  int tiles_x, tiles_y;
  int ntiles, tiles_per_xcd;   // workgroup order: see the kernel
  int nby;                     // channel blocks
};

constexpr int ROWB = 64;   // LDS bytes per staged row (unpadded, slots swizzled: see swz())



// Workgroup barrier that first waits until at most n of this wave's vector-memory operations are
// still outstanding.  Loads complete in issue order, so with n = the number of DMA instructions
// the wave issued after the ones it needs now, the needed ones have landed while the newer ones
// stay in flight.  (__syncthreads() would drain everything.)
__device__ __forceinline__ void wait_vm_barrier(int n) {
#define UNETK_WVB(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier" ::: "memory"); break;
  switch (n) {
    UNETK_WVB(1) UNETK_WVB(2) UNETK_WVB(3) UNETK_WVB(4) UNETK_WVB(5) UNETK_WVB(6) UNETK_WVB(7) UNETK_WVB(8)
    UNETK_WVB(9) UNETK_WVB(10) UNETK_WVB(11) UNETK_WVB(12)
    default: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); break;
  }
#undef UNETK_WVB
}

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

template <typename T, int NT, int KS, bool HDB, bool FUSE, bool TWO, bool BST = false>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a) {
  using M = MMA<T>;
  constexpr int KC = M::KC;
  constexpr int EPP = 16 / (int)sizeof(T);     // elements per 16-byte part
  constexpr int BN = NT * 16;
  constexpr int HALO = (KS == 3) ? 18 : 16;
  constexpr int HPIX = HALO * HALO;
  constexpr int TAPS = KS * KS;
  constexpr int WROWS = KS * BN;
  // Operand staging is LDS-DMA (global_load_lds, 16 B per lane, 1 KiB per wave instruction) into
  // double buffers: while phase p (one kernel row of one channel chunk) is being multiplied, the
  // weights of phase p+1 (and, at a chunk boundary, the next halo) are already in flight.  One
  // barrier per phase.  The LDS image is lane-linear, so the slot swizzle is applied to the SOURCE
  // address of each lane; out-of-image pixels and channel padding read a page of zeros.
  constexpr int HINSTR = (HPIX * 4 + 63) / 64;          // wave instructions per halo
  constexpr int WINSTR = (WROWS * 4) / 64;              // per weight slab (BN multiple of 16: exact)
  static_assert((WROWS * 4) % 64 == 0, "weight slab must be whole wave instructions");
  constexpr int HBUF = HINSTR * 1024, WBUF = WINSTR * 1024;
  // 1x1 (plain GEMM) phases are a third as long as 3x3 ones: they get rings of three stages, two
  // phases of operands in flight behind the one in use, behind counted waits
  constexpr bool RING3 = HDB && KS == 1;
  constexpr int NHB = RING3 ? 3 : (HDB ? 2 : 1);        // halo buffers
  constexpr int NWB = RING3 ? 3 : 2;                    // weight buffers
  constexpr int MAIN_BYTES = NHB * HBUF + NWB * WBUF;
  constexpr int STAGE_BYTES = 4 * 16 * BN * (int)sizeof(T);
  constexpr int RED_BYTES = 4 * 2 * BN * (int)sizeof(float);
  // (BST) the tile of y behind them: 4 waves x 4 rows of 16 pixels x BN channels, whole 1 KiB DMA pieces a row
  static_assert(!BST || KS == 3, "the BatchNorm-backward sums are built for the 3x3 data gradient");
  constexpr int YPP = BN / EPP;                       // 16-byte parts per pixel
  constexpr int YNIT = (16 * YPP + 63) / 64;          // DMA pieces per tile row
  constexpr int YST_OFF = (STAGE_BYTES + RED_BYTES + 1023) / 1024 * 1024;
  constexpr int YST_BYTES = BST ? 4 * 4 * YNIT * 1024 : 0;
  constexpr int EPI_BYTES = BST ? YST_OFF + YST_BYTES : STAGE_BYTES + RED_BYTES;
  constexpr int LDS_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
  // per-wave statistics scratch aliases the operand area (behind the epilogue staging tiles)
  float (*red)[2][BN] = reinterpret_cast<float (*)[2][BN]>(lds + STAGE_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  XCD x takes the x-th
  // contiguous eighth of the tile list and walks it with the channel block as the fast index, so
  // the channel blocks of a tile (same input halo) and neighbouring tiles (shared halo rows and
  // columns) are resident together under one L2.
  const int xcd = blockIdx.x & 7;
  const int item = blockIdx.x >> 3;
  const int nb = item % a.nby;
  const int tile = xcd * a.tiles_per_xcd + item / a.nby;
  if (item / a.nby >= a.tiles_per_xcd || tile >= a.ntiles) return;
  const int n0 = nb * BN;
  const T* in = reinterpret_cast<const T*>(a.in);
  const T* wp = reinterpret_cast<const T*>(a.w);
  T* out = reinterpret_cast<T*>(a.out);

  // tile coordinates
  int b = 0, y0 = 0, x0 = 0;
  long p0 = 0;                       // first flat pixel (1x1 mode)
  const long npix = (long)a.B * a.H * a.W;
  if (KS == 3) {
    int t = tile;
    int tx = t % a.tiles_x;
    t /= a.tiles_x;
    int ty = t % a.tiles_y;
    b = t / a.tiles_y;
    y0 = ty * 16;
    x0 = tx * 16;
  } else {
    p0 = (long)tile * 256;
  }

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const T* zero = reinterpret_cast<const T*>(conv_zero_page);
  // DMA source addresses.  What depends on the lane and the piece but not on the chunk is computed
  // once, one register a piece: the element offset of the lane's 16 bytes from the halo's first
  // pixel, plus (in flags the compiler keeps as lane masks) whether the pixel lies inside the
  // image.  The lane's channel slot is the same in every piece: pieces are 16 pixels apart and
  // swz() looks at pixel bit 2 only.  Issuing a piece then costs two 32-bit operations and a
  // select instead of divisions and multiplies between the MFMAs.
  constexpr int HPW = (HINSTR + 3) / 4;                 // halo pieces per wave (at most)
  // (the fp32 four-workgroups-per-CU variant has no registers to spare for the table: it recomputes)
  constexpr bool HTAB = HDB || sizeof(T) == 2;
  const int slot0 = ((lane & 3) ^ swz(lane >> 2)) * EPP;
  const unsigned rows_left = (KS == 3) ? 0u : (unsigned)((npix - p0 < 256) ? (npix - p0) : 256);
  // (two tables when the operand is a concatenation of two tensors with different pixel strides: the
  // second one costs registers only in the instantiations launched with it)
  int hp_off[HPW], hp_off2[TWO ? HPW : 1];
  bool hp_in[HPW];
#pragma unroll
  for (int k = 0; k < (HTAB ? HPW : 0); ++k) {
    const int pix = (wave + 4 * k) * 16 + (lane >> 2);
    if (KS == 3) {
      const int hy = pix / HALO, hx = pix - hy * HALO;
      hp_off[k] = (hy * a.W + hx) * a.ldin + slot0;
      if constexpr (TWO) hp_off2[k] = (hy * a.W + hx) * a.ldin2 + slot0;
      hp_in[k] = pix < HPIX && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
    } else {
      hp_off[k] = pix * a.ldin + slot0;
      if constexpr (TWO) hp_off2[k] = pix * a.ldin2 + slot0;
      hp_in[k] = (unsigned)pix < rows_left;
    }
  }
  // first halo pixel of the tile (may lie outside the image: only valid lanes add to it)
  const T* halo_base = (KS == 3) ? in + (((long)b * a.H + (y0 - 1)) * a.W + (x0 - 1)) * a.ldin : in + p0 * a.ldin;
  const T* in2 = reinterpret_cast<const T*>(a.in2);
  const T* halo_base2 = !TWO ? nullptr
                        : (KS == 3) ? in2 + (((long)b * a.H + (y0 - 1)) * a.W + (x0 - 1)) * a.ldin2 : in2 + p0 * a.ldin2;
  auto issue_halo = [&](int ch) {
    char* dst = lds + (RING3 ? ch % 3 : (HDB ? (ch & 1) : 0)) * HBUF;
    int c0 = ch * KC;
    bool second = false;
    if constexpr (TWO) second = c0 >= a.Ci1;              // (uniform; Ci1 is a multiple of the chunk)
    if (second) c0 -= a.Ci1;
    const T* base = (second ? halo_base2 : halo_base) + c0;
    const int climit = (TWO ? (second ? a.Ci - a.Ci1 : a.Ci1) : a.Ci) - c0;   // channel slots below this exist
    if constexpr (!HTAB) {
      for (int i = wave; i < HINSTR; i += 4) {
        const int part = i * 64 + lane;
        const int pix = part >> 2;
        const int cs = ((part & 3) ^ swz(pix)) * EPP;
        const int c = c0 + cs;
        const T* from = second ? in2 : in;
        const int ld = second ? a.ldin2 : a.ldin;
        const T* src = zero;
        if (KS == 3) {
          const int hy = pix / HALO, hx = pix - hy * HALO;
          const int gy = y0 + hy - 1, gx = x0 + hx - 1;
          if (pix < HPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && cs < climit)
            src = from + (((long)b * a.H + gy) * a.W + gx) * ld + c;
        } else {
          const long p = p0 + pix;
          if (p < npix && cs < climit) src = from + p * ld + c;
        }
        glds16(src, dst + i * 1024);
      }
      return;
    }
    // (the chunk's offset is added to the lane's in 32 bits: the offsets stay one register each)
    const bool chan_ok = slot0 < climit;
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
      const int i = wave + 4 * k;
      if (i < HINSTR) {
        const int off = (TWO && second) ? hp_off2[k] : hp_off[k];
        const T* src = (hp_in[k] && chan_ok) ? (second ? halo_base2 : halo_base) + (unsigned)(off + c0) : zero;
        glds16(src, dst + i * 1024);
      }
    }
  };
  // weights: piece i of a slab is 1 KiB further on; the lane's offset inside a piece is fixed
  const int w_rel = (lane >> 2) * KC + ((lane & 3) ^ swz(lane >> 2)) * EPP;
  auto issue_w = [&](int phase) {
    const int ch = phase / KS, ky = phase - ch * KS;
    char* dst = lds + NHB * HBUF + (RING3 ? phase % 3 : (phase & 1)) * WBUF;
    const T* wsrc = wp + ((((long)nb * a.nchunks + ch) * TAPS + ky * KS) * BN) * KC + w_rel;
    // (unrolled: the trip count depends on the wave only through the last piece)
    constexpr int WPWV = (WINSTR + 3) / 4;
#pragma unroll
    for (int k = 0; k < WPWV; ++k) {
      const int i = wave + 4 * k;
      if (4 * k + 3 < WINSTR || i < WINSTR) glds16(wsrc + i * (1024 / (int)sizeof(T)), dst + i * 1024);
    }
  };

  const int nphase = a.nchunks * KS;
  issue_halo(0);
  issue_w(0);
  if (RING3 && nphase > 1) {
    issue_halo(1);
    issue_w(1);
  }
  // DMA pieces this wave issues per halo / weight slab, and whether a halo issued one phase ago is in flight
  const int my_halo = (HINSTR - wave + 3) / 4, my_w = (WINSTR - wave + 3) / 4;
  bool halo_flying = false;

  // bf16 fragment addresses without per-read arithmetic.  A fragment read is row (R + l15), 16-byte
  // slot (q ^ swz(row)); R is wave-uniform and, as 18*4*wave and BN are multiples of 8, R mod 8 is a
  // compile-time constant c of the unrolled loops, and swz() only looks at bit 2 of the row.  So
  // the per-lane part of the address takes 8 values (one per c) for the halo and one for the
  // weights; everything else is an immediate offset of the ds_read.
  const int l15 = lane & 15, q = lane >> 4;
  int a_lane[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
    a_lane[c] = (4 * wave * (KS == 3 ? HALO : 16) + l15) * ROWB + ((q ^ swz(c + l15)) * 16);
  const int w_lane = l15 * ROWB + ((q ^ swz(l15)) * 16);

  // The MFMA-bound 3x3 bf16 variant keeps two register sets of fragments: while step (ky, kx)
  // multiplies from one, the ds_reads of the next step fill the other, one read every few MFMAs,
  // so no MFMA ever waits on an LDS round trip.  For the reads to run ahead across a phase
  // boundary, the barrier that opens phase q (and the DMA issue that follows it) sits before the
  // LAST step of phase q-1: by then every wave has its fragments of that last step in registers
  // (lgkmcnt(0) before the barrier), so nobody reads the buffers the DMA refills any more.
  constexpr bool PIPE = HDB && KS == 3 && sizeof(T) == 2;
  if constexpr (PIPE) {
    using frag = typename M::frag;
    frag fa[2][4], fb[2][NT];
    auto load_frags = [&](auto bufc, auto tc, auto parc) {
      // fragments of step t of a chunk with parity par (halo buffer par, weight buffer (par + ky) & 1)
      constexpr int BUF = decltype(bufc)::value, t = decltype(tc)::value % TAPS;
      constexpr int par = (decltype(parc)::value + decltype(tc)::value / TAPS) & 1;
      constexpr int ky = t / KS, kx = t % KS;
      const char* halo = lds + par * HBUF;
      const char* wbuf = lds + NHB * HBUF + ((par + ky) & 1) * WBUF;
      // (weights first: the next step's first MFMAs need all of them but only the first halo row)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        fb[BUF][j] = *reinterpret_cast<const frag*>(wbuf + w_lane + (kx * BN + j * 16) * ROWB);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = (2 * (i + ky) + kx) & 7;
        fa[BUF][i] = *reinterpret_cast<const frag*>(halo + a_lane[c] + ((i + ky) * HALO + kx) * ROWB);
      }
    };
    bool halo_fly = false;
    auto open_phase = [&](int q, int vm_keep) {
      // q: the phase whose operands must have landed; afterwards the DMA of phase q+1 goes out.  vm_keep is 0 or
      // this wave's halo pieces, of which there are only two possible counts: two compares instead of a switch
      constexpr int HP_HI = (HINSTR + 3) / 4, HP_LO = HINSTR / 4;
      if (vm_keep == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else if (vm_keep == HP_HI) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(HP_HI) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(HP_LO > 0 ? HP_LO : 1) : "memory");
      if (q + 1 < nphase) issue_w(q + 1);
    };
    auto chunk = [&](auto parc, int ch) {
      constexpr int PAR = decltype(parc)::value;
      auto step = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int ky = t / KS, kx = t % KS;
        constexpr int cur = (t + PAR) & 1;
        if constexpr (kx == KS - 1) {
          const int q = ch * KS + ky + 1;
          if (q < nphase) {
            if constexpr (ky == KS - 1) {
              // q opens chunk ch+1: its halo was drained at the previous barrier
              open_phase(q, 0);
              halo_fly = ch + 2 < a.nchunks;
              if (halo_fly) issue_halo(ch + 2);
            } else {
              open_phase(q, (ky == 0 && halo_fly) ? my_halo : 0);
            }
          }
        }
        // (after a chunk's last step this reads the other halo buffer: the next chunk, or, at the
        // very end, stale data nobody uses)
        load_frags(std::integral_constant<int, cur ^ 1>{}, std::integral_constant<int, t + 1>{}, parc);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = M::mma(fa[cur][i], fb[cur][j], acc[i][j]);
        constexpr int NRD = 4 + NT, MPR = (4 * NT) / NRD;
#pragma unroll
        for (int k = 0; k < NRD; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);   // MPR MFMAs ...
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // ... then one ds_read
        }
        if constexpr (4 * NT - MPR * NRD > 0) __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT - MPR * NRD, 0);
      };
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
      step(std::integral_constant<int, 4>{});
      step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{});
      step(std::integral_constant<int, 7>{});
      step(std::integral_constant<int, 8>{});
    };
    // phase 0: everything issued so far has to land; then the first fragments (the one exposed read)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (nphase > 1) issue_w(1);
    halo_fly = a.nchunks > 1;
    if (halo_fly) issue_halo(1);
    load_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    int ch = 0;
#pragma unroll 1
    for (; ch + 1 < a.nchunks; ch += 2) {
      chunk(std::integral_constant<int, 0>{}, ch);
      chunk(std::integral_constant<int, 1>{}, ch + 1);
    }
    if (ch < a.nchunks) chunk(std::integral_constant<int, 0>{}, ch);
  } else {
#pragma unroll 1
  for (int ch = 0; ch < a.nchunks; ++ch) {
    const char* lds_halo = lds + (RING3 ? ch % 3 : (HDB ? (ch & 1) : 0)) * HBUF;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      const int phase = ch * KS + ky;
      if constexpr (RING3) {
        // phase == chunk.  Needed now: halo and weights of this phase (issued two phases ago);
        // the next phase's pieces are younger and stay in flight.
        wait_vm_barrier(phase + 1 < nphase ? my_halo + my_w : 0);
        if (phase + 2 < nphase) {
          issue_halo(phase + 2);
          issue_w(phase + 2);
        }
      } else {
      // The weights of this phase (and at ky == 0 its halo) have landed for every wave, and everyone
      // is done with phase-1's buffers.  With two halo buffers the next chunk's halo is issued in the
      // chunk's first phase, after that phase's weight prefetch: at the next barrier it is the newest
      // thing in flight and may stay there; the barrier after that drains it (two phases of cover).
      if (HDB) wait_vm_barrier((KS == 3 && ky == 1 && halo_flying) ? my_halo : 0);
      else __syncthreads();
      if (!HDB && ky == 0 && ch > 0) {
        // single halo buffer (more workgroups per CU instead): the chunk is fetched only now that
        // every wave is done with the previous one; other workgroups on the CU cover the wait
        issue_halo(ch);
        __syncthreads();
      }
      if (phase + 1 < nphase) issue_w(phase + 1);
      if (HDB && ky == 0) {
        halo_flying = ch + 1 < a.nchunks;
        if (halo_flying) issue_halo(ch + 1);
      }
      }
      const char* lds_w = lds + NHB * HBUF + (RING3 ? phase % 3 : (phase & 1)) * WBUF;
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
        for (int ks = 0; ks < M::KSTEPS; ++ks) {
          typename M::frag af[4], bfr[NT];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if constexpr (sizeof(T) == 2) {
              const int c = (KS == 3) ? ((2 * (i + ky) + kx) & 7) : 0;
              const int imm = (KS == 3) ? ((i + ky) * HALO + kx) * ROWB : (i * 16) * ROWB;
              af[i] = *reinterpret_cast<const typename M::frag*>(lds_halo + a_lane[c] + imm);
            } else {
              const int prow = (KS == 3) ? ((4 * wave + i + ky) * HALO + kx + (lane & 15))
                                         : ((4 * wave + i) * 16 + (lane & 15));
              af[i] = M::load_swz(lds_halo + prow * ROWB, ks, lane, swz(prow));
            }
          }
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            if constexpr (sizeof(T) == 2) {
              bfr[j] = *reinterpret_cast<const typename M::frag*>(lds_w + w_lane + (kx * BN + j * 16) * ROWB);
            } else {
              const int wrow = kx * BN + j * 16 + (lane & 15);
              bfr[j] = M::load_swz(lds_w + wrow * ROWB, ks, lane, swz(wrow));
            }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = M::mma(af[i], bfr[j], acc[i][j]);
        }
      }
    }
  }
  }

  // ---- epilogue.  C layout: column (channel) = lane&15, row (pixel) = 4*(lane>>4)+r.
  const int cl = lane & 15, pg = lane >> 4;
  bool rowvalid[4];
  bool colvalid[4];
  long pixbase[4];   // flat pixel index of (tile row i, tile col 0)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (KS == 3) {
      const int gy = y0 + 4 * wave + i;
      rowvalid[i] = gy < a.H;
      pixbase[i] = ((long)b * a.H + gy) * a.W + x0;
    } else {
      pixbase[i] = p0 + (4 * wave + i) * 16;
      rowvalid[i] = true;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (KS == 3) colvalid[r] = (x0 + 4 * pg + r) < a.W;
    else colvalid[r] = true;
  }
  auto pixel_valid = [&](int i, int r) -> bool {
    if (KS == 3) return rowvalid[i] && colvalid[r];
    return (pixbase[i] + 4 * pg + r) < npix;
  };

  const bool tile_full = (KS == 3) ? (y0 + 16 <= a.H && x0 + 16 <= a.W) : (p0 + 256 <= npix);
  __syncthreads();   // every wave is done reading the operand tiles: LDS is reused from here on
  char* const yst = lds + YST_OFF + wave * (4 * YNIT * 1024);
  if constexpr (BST) {
    // the wave's four rows of y (16 pixels x BN channels each) by LDS-DMA, in flight under the store pass below
    const T* ysrc = reinterpret_cast<const T*>(a.bs_y);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < YNIT; ++k) {
        const int part = lane + 64 * k;
        const int pc = part / YPP, cp = part - pc * YPP;
        const bool ok = part < 16 * YPP && rowvalid[i] && (KS == 3 ? (x0 + pc) < a.W : true) && n0 + cp * EPP < a.Co;
        glds16(ok ? ysrc + (pixbase[i] + pc) * a.ldbs + n0 + cp * EPP : zero, yst + (i * YNIT + k) * 1024);
      }
  }
  if (a.stats != nullptr) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float s = 0.f, q = 0.f;
      if (tile_full) {               // (uniform) no per-value masks away from the image border
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[i][j][r];
            s += v;
            q += v * v;
          }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (pixel_valid(i, r)) {
              const float v = acc[i][j][r];
              s += v;
              q += v * v;
            }
      }
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      if (pg == 0) {
        red[wave][0][j * 16 + cl] = s;
        red[wave][1][j * 16 + cl] = q;
      }
    }
  }
  __syncthreads();   // red[] visible
  if (a.stats != nullptr && tid < BN && n0 + tid < a.Co) {
    const float s = red[0][0][tid] + red[1][0][tid] + red[2][0][tid] + red[3][0][tid];
    const float q = red[0][1][tid] + red[1][1][tid] + red[2][1][tid] + red[3][1][tid];
    float* st = a.stats + (long)tile * 2 * a.Co;
    st[n0 + tid] = s;
    st[a.Co + n0 + tid] = q;
  }

  float esc[NT], esh[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int c = n0 + j * 16 + cl;
    esc[j] = (a.ep_scale != nullptr && c < a.Co) ? a.ep_scale[c] : 1.f;
    esh[j] = (a.ep_shift != nullptr && c < a.Co) ? a.ep_shift[c] : 0.f;
  }

  if constexpr (KS == 3 && FUSE) {
    // Inference-time fusions on the values as they are stored (activation applied, rounded to T):
    // a lane holds pixels (row 4*wave+i, column 4*pg+r), i.e. whole 2x2 quads, for its channels.
    // (scale, shift and the activation are applied in place first, under one uniform branch; the
    // host admits only none / ReLU here)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = acc[i][j][r] * esc[j] + esh[j];
    if (a.act == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaxf(acc[i][j][r], 0.f);
    }
    auto val = [&](int i, int j, int r) -> float { return to_f32(from_f32<T>(acc[i][j][r])); };
    if (a.pooled != nullptr) {
      T* pl = reinterpret_cast<T*>(a.pooled);
      const int PH = a.H >> 1, PW = a.W >> 1;
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          const int py = (y0 + 4 * wave + 2 * i2) >> 1, px = (x0 + 4 * pg + 2 * r2) >> 1;
          if (py < PH && px < PW) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              const float m = fmaxf(fmaxf(val(2 * i2, j, 2 * r2), val(2 * i2, j, 2 * r2 + 1)),
                                    fmaxf(val(2 * i2 + 1, j, 2 * r2), val(2 * i2 + 1, j, 2 * r2 + 1)));
              const int c = n0 + j * 16 + cl;
              if (c < a.Co) pl[(((long)b * PH + py) * PW + px) * a.Co + c] = from_f32<T>(m);
            }
          }
        }
    }
    if (a.head_out != nullptr) {
      // final 1x1 conv (few outputs): per-lane partial dot products over its channels, then a
      // butterfly over the 16 lanes that share a pixel group
      for (int o = 0; o < a.head_co; ++o) {
        float hw[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int c = n0 + j * 16 + cl;
          hw[j] = c < a.Co ? a.head_w[(long)o * a.Co + c] : 0.f;
        }
        const float hb = a.head_b ? a.head_b[o] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float sdot = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) sdot += val(i, j, r) * hw[j];
            sdot += __shfl_xor(sdot, 1, 64);
            sdot += __shfl_xor(sdot, 2, 64);
            sdot += __shfl_xor(sdot, 4, 64);
            sdot += __shfl_xor(sdot, 8, 64);
            if (cl == 0 && pixel_valid(i, r)) {
              const int gy = y0 + 4 * wave + i, gx = x0 + 4 * pg + r;
              a.head_out[(((long)b * a.head_co + o) * a.H + gy) * a.W + gx] = sdot + hb;
            }
          }
      }
    }
  }
  if (a.out == nullptr) return;      // (uniform) only the fused outputs were wanted

  T* stage = reinterpret_cast<T*>(lds) + wave * 16 * BN;
  constexpr int PPP = BN / EPP;            // 16-byte parts per staged pixel
  // Store pass: the staged 16 x BN tile leaves as 16-byte parts, lane -> (pixel pc, part cp).  What
  // does not depend on the row tile is computed once: the part's place in the staging tile, its
  // element offset from the row tile's first pixel in the destination, and whether it exists.
  constexpr int NIT = (16 * PPP + 63) / 64;
  T* obase = out;
  int ldo = a.ldout, oc0 = n0;
  if (a.out2 != nullptr && n0 >= a.split) {
    obase = reinterpret_cast<T*>(a.out2);
    ldo = a.ldout2;
    oc0 = n0 - a.split;
  }
  int st_off[NIT], g_off[NIT], st_pc[NIT];
  bool st_ok[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int part = lane + 64 * k;
    const int pc = part / PPP, cp = part - pc * PPP;
    st_pc[k] = pc;
    st_off[k] = pc * BN + cp * EPP;
    g_off[k] = pc * ldo + oc0 + cp * EPP;
    st_ok[k] = part < 16 * PPP && n0 + cp * EPP < a.Co && (KS == 3 ? (x0 + pc) < a.W : true);
  }
  // The activation is uniform: one branch per row tile, not one per value (GELU: per 4 values, fenced,
  // so that the compiler does not interleave dozens of erf expansions and their temporaries).
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i > 0) __syncthreads();
    if constexpr (!(KS == 3 && FUSE)) {        // (the fused variant has applied them above)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = acc[i][j][r] * esc[j] + esh[j];
      if (a.act == 1) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaxf(acc[i][j][r], 0.f);
      } else if (a.act == 2) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = gelu_erf(acc[i][j][r]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) stage[(4 * pg + r) * BN + j * 16 + cl] = from_f32<T>(acc[i][j][r]);
    __syncthreads();
    T* orow = obase + pixbase[i] * ldo;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const bool ok = st_ok[k] && (KS == 3 ? rowvalid[i] : (unsigned)((4 * wave + i) * 16 + st_pc[k]) < rows_left);
      if (ok) {
        Vec16<T> v = load16(stage + st_off[k]);
        if (a.resid != nullptr) {
          const long p = pixbase[i] + st_pc[k];
          const int c = n0 + st_off[k] - st_pc[k] * BN;
          const T* rs = reinterpret_cast<const T*>(a.resid) + p * a.ldres + c;
          Vec16<T> rv = load16(rs);
          float keep = 1.f;
          if (a.resid_keep != nullptr) keep = a.resid_keep[p / ((long)a.H * a.W)];
#pragma unroll
          for (int e = 0; e < EPP; ++e)
            v.v[e] = from_f32<T>(to_f32(rv.v[e]) + keep * to_f32(v.v[e]));
        }
        store16(orow + g_off[k], v);
      }
    }
  }
  if constexpr (BST) {
    // (the accumulators still hold the output: a data-gradient launch has no scale, shift or activation)
    __builtin_amdgcn_s_waitcnt(0x0070);        // vmcnt(0): this wave's rows of y have landed (each wave reads its own)
    __syncthreads();                           // ... and every wave is done with red[] and the staging tiles
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int c = n0 + j * 16 + cl;
      const float bsc = c < a.Co ? a.bs_scale[c] : 0.f, bsh = c < a.Co ? a.bs_shift[c] : 0.f;
      const float bmu = c < a.Co ? a.bs_mean[c] : 0.f, bis = c < a.Co ? a.bs_invstd[c] : 0.f;
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // the tile row's image is lane-linear in 16-byte parts: pixel pc, part cp at byte (pc*YPP + cp)*16
          const T yraw = *reinterpret_cast<const T*>(yst + i * YNIT * 1024 + ((4 * pg + r) * BN + j * 16 + cl) * (int)sizeof(T));
          const float yv = to_f32(yraw);
          float dz = to_f32(from_f32<T>(acc[i][j][r]));
          if (!(fmaf(yv, bsc, bsh) > 0.f) || !pixel_valid(i, r)) dz = 0.f;
          s += dz;
          q = fmaf(dz, (yv - bmu) * bis, q);
        }
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      if (pg == 0) {
        red[wave][0][j * 16 + cl] = s;
        red[wave][1][j * 16 + cl] = q;
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < a.Co) {
      float* st = a.bs_partials + (long)tile * 2 * a.Co;
      st[n0 + tid] = red[0][0][tid] + red[1][0][tid] + red[2][0][tid] + red[3][0][tid];
      st[a.Co + n0 + tid] = red[0][1][tid] + red[1][1][tid] + red[2][1][tid] + red[3][1][tid];
    }
  }
}

template <typename T, int NT, int KS, bool FUSE = false, bool TWO = false, bool BST = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_igemm_kernel(ConvArgs a) {
  conv_igemm_body<T, NT, KS, true, FUSE, TWO, BST>(a);
}

// Narrow layers move more bytes per flop than the MFMA-bound wide ones: they trade the second halo
// buffer and registers for four workgroups per CU, so more loads are in flight.
template <typename T, int NT, int KS, bool FUSE = false, bool TWO = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_igemm_occ4_kernel(
    ConvArgs a) {
  conv_igemm_body<T, NT, KS, false, FUSE, TWO>(a);
}

// ------------------------------------------------------------------------------------------------
// 3x3 conv, up to 48 input and 48 output channels (bf16): the full-resolution layers of the UNet.
// Measured on those shapes, the generic kernel is bound by the CU's vector-memory address path, not
// by MFMA or HBM: an LDS-DMA piece (1 KiB per wave instruction) whose lanes gather 64-byte runs - one
// 32-channel chunk of 16 pixels that are 96 bytes apart - takes ~100 cycles of that path, a piece
// that reads 1 KiB of consecutive memory ~35.  So here a pixel is staged whole (96 bytes, all 48
// channels): a halo row is one run of 18 x 96 consecutive bytes, the LDS image is a byte-for-byte copy
// of those runs (96-byte pixel pitch: fragment reads of 16 neighbouring pixels are bank-conflict free
// as they stand), and a piece is a linear 1 KiB slice of it.
//
// One persistent workgroup of eight waves per CU walks a contiguous range of 16-column x 32-row
// tiles.  All weights live in LDS for the whole launch; the 18x34 halo of the next tile lands in the
// second halo buffer while the current one is multiplied (one barrier per tile, 168 MFMAs per wave
// between barriers).  K = 9 taps x 48 channels is cut into 14 steps of 32: nine steps take channels
// 0..31 of one tap, five take channels 32..47 of TWO taps (lanes with k < 16 read one tap's pixel,
// lanes with k >= 16 the other's: a per-lane address, nothing else) - no multiply is spent on padding
// except half of the last step.  The MFMA operands are swapped - weights are the A operand, pixels
// the B operand - so a lane ends up with 12 output channels of one pixel: the weight rows of the
// three 16-channel MFMA tiles are interleaved on their way into LDS (tiles 0 and 1, row 4q+r: channel
// 8q+4t+r; tile 2: channel 32+4q+r), a lane's values are 8 + 4 consecutive channels and leave as one
// 16-byte and one 8-byte store straight from registers: no staging tile, no extra barriers.  Wave w
// owns tile rows 4w..4w+3; in the nine full steps a pixel fragment read for halo row r and column
// shift kx serves the three kernel rows (output rows r, r-1, r-2).
// The address path is shared by the CU: eight waves issuing DMA at the same moment all wait for each
// other while the matrix pipe idles.  So the next tile's DMA and the previous tile's stores are cut
// into four parts, one per block of MFMAs, and the two waves that share a SIMD (w and w+4) issue
// theirs at different points of the block: one of them is always multiplying.
constexpr int P48_PIXB = 96;                         // LDS bytes per staged pixel (48 channels)
constexpr int P48_ROWB = 18 * P48_PIXB;              // one halo row
constexpr int P48_HBYTES = 34 * P48_ROWB;            // 58752
constexpr int P48_HINSTR = (P48_HBYTES + 1023) / 1024;   // 58 pieces, the last one partial
constexpr int P48_HBUF = (P48_HBYTES + 255) / 256 * 256;
constexpr int P48_KSTEPS = 14;
constexpr int P48_WINSTR = P48_KSTEPS * 3;           // 48 rows x 64 B per step
constexpr int P48_WBUF = P48_WINSTR * 1024;

template <int CTRL> __device__ __forceinline__ float p48_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes that share lane>>4 (pairs, quads, half mirror, row mirror)
__device__ __forceinline__ float p48_row_sum(float v) {
  v += p48_dpp<0xB1>(v);
  v += p48_dpp<0x4E>(v);
  v += p48_dpp<0x141>(v);
  v += p48_dpp<0x140>(v);
  return v;
}
__device__ __forceinline__ uint32_t p48_pack2(float lo, float hi) {
  const bf16 l = (bf16)lo, h = (bf16)hi;
  return (uint32_t)__builtin_bit_cast(unsigned short, l) | ((uint32_t)__builtin_bit_cast(unsigned short, h) << 16);
}

struct P48Item {
  int b, ty, tx;
};

template <bool FUSE>
__global__ __launch_bounds__(512) void conv3x3_p48_kernel(ConvArgs a) {
  using T = bf16;
  using frag = bf16x8;
  // Separate LDS objects (not one array indexed by the tile's parity): the compiler then knows that
  // the DMA filling one halo buffer cannot alias the fragment reads of the other and leaves it in
  // flight during the multiply.  Together they are exactly the CU's 160 KiB.
  __shared__ __attribute__((aligned(256))) char halo0[P48_HBUF];
  __shared__ __attribute__((aligned(256))) char halo1[P48_HBUF];
  __shared__ __attribute__((aligned(256))) char wlds[P48_WBUF];
  __shared__ float red[8][2][48];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const bool early = wave < 4;                 // waves w and w+4 share a SIMD: they take turns at the address path
  const T* in = reinterpret_cast<const T*>(a.in);
  const T* wp = reinterpret_cast<const T*>(a.w);
  const T* zero = reinterpret_cast<const T*>(conv_zero_page);

  // this workgroup's contiguous range of tiles; each XCD (workgroups are dealt to them round-robin)
  // takes a contiguous eighth, so neighbouring halos share an L2
  const int tiles_y32 = (a.H + 31) >> 5;
  const int nitems = a.B * tiles_y32 * a.tiles_x;
  const int nwg = gridDim.x;
  const int pos = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int per = (nitems + nwg - 1) / nwg;
  const int first = pos * per;
  int mine = nitems - first;
  if (mine > per) mine = per;
  if (mine <= 0) return;
  auto advance = [&](P48Item& it) __attribute__((always_inline)) {
    if (++it.tx == a.tiles_x) {
      it.tx = 0;
      if (++it.ty == tiles_y32) {
        it.ty = 0;
        ++it.b;
      }
    }
  };
  P48Item nxt;                               // the tile whose DMA goes out next
  nxt.tx = first % a.tiles_x;
  nxt.ty = (first / a.tiles_x) % tiles_y32;
  nxt.b = first / a.tiles_x / tiles_y32;
  P48Item cur = nxt;                         // the tile being multiplied

  // ---- weights, once: 14 K-steps x 48 rows x 64 B.  Piece j: step j/3, MFMA tile j%3.  Steps 0..8 are
  // channels 0..31 of tap s; step 9+p holds channels 32..47 of tap 2p (k < 16) and of tap 2p+1 (k >= 16).
  {
    const int m = lane >> 2, slot = (lane & 3) ^ swz(m);
    const long chunk1 = 9L * 48 * 32;
#pragma unroll
    for (int k = 0; k < (P48_WINSTR + 7) / 8; ++k) {
      const int j = wave + 8 * k;
      if (j < P48_WINSTR) {
        const int st = j / 3, t = j - 3 * st;
        const int chan = t < 2 ? 8 * (m >> 2) + 4 * t + (m & 3) : 32 + m;
        const T* src;
        if (st < 9) {
          src = wp + ((long)st * 48 + chan) * 32 + slot * 8;
        } else {
          const int tap = 2 * (st - 9) + (slot >> 1);
          src = (tap < 9 && a.nchunks > 1) ? wp + chunk1 + ((long)tap * 48 + chan) * 32 + (slot & 1) * 8 : zero;
        }
        glds16(src, wlds + j * 1024);
      }
    }
  }

  // ---- halo DMA source table: piece i = wave + 8k is bytes [1024 i, 1024 i + 1024) of the halo image
  constexpr int HPW = (P48_HINSTR + 7) / 8;
  int hp_off[HPW], hp_yx[HPW];
#pragma unroll
  for (int k = 0; k < HPW; ++k) {
    const int o = (wave + 8 * k) * 1024 + lane * 16;
    const int r = o / P48_ROWB, rem = o - r * P48_ROWB;
    const int px = rem / P48_PIXB, cb = rem - px * P48_PIXB;
    hp_off[k] = (r * a.W + px) * a.ldin + (cb >> 1);
    // bit 31: nothing to write (past the image of the halo); a row no image has: read zeros
    hp_yx[k] = o >= P48_HBYTES ? (int)0x80000000 : ((cb >> 1) < a.Ci ? ((r << 16) | px) : (0x4000 << 16));
  }
  const T* hbase = nullptr;
  int iy0 = 0, ix0 = 0;
  auto issue_begin = [&]() __attribute__((always_inline)) {
    iy0 = nxt.ty * 32 - 1;
    ix0 = nxt.tx * 16 - 1;
    hbase = in + (((long)nxt.b * a.H + iy0) * a.W + ix0) * a.ldin;
    advance(nxt);
  };
  auto issue_piece = [&](auto parc, auto kc) __attribute__((always_inline)) {       // piece wave + 8K of the halo image
    constexpr int P = decltype(parc)::value, K = decltype(kc)::value;
    char* hdst = P ? halo1 : halo0;
    if constexpr (K < HPW) {
      const int i = wave + 8 * K;
      if (i < P48_HINSTR && hp_yx[K] >= 0) {
        const int hy = hp_yx[K] >> 16, hx = hp_yx[K] & 0xffff;
        const bool ok = (unsigned)(iy0 + hy) < (unsigned)a.H && (unsigned)(ix0 + hx) < (unsigned)a.W;
        glds16(ok ? hbase + hp_off[K] : zero, hdst + i * 1024);
      }
    }
  };

  // ---- fragment addresses: a lane register plus immediates
  const int p_lane = (4 * wave * 18 + l15) * P48_PIXB + 16 * q;       // full steps: + ((r*18 + kx) * 96)
  int t_lane[5];                                                       // paired steps: + (i*18*96)
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    int tap = 2 * p + (q >> 1);
    if (tap > 8) tap = 8;                  // (its weights are zeros; any finite pixel will do)
    t_lane[p] = (4 * wave * 18 + l15 + (tap / 3) * 18 + tap % 3) * P48_PIXB + 64 + 16 * (q & 1);
  }
  const int w_lane = l15 * ROWB + ((q ^ swz(l15)) * 16);

  f32x4 acc[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // halo rows [R0, R1) of kernel column kx (steps kx, 3+kx, 6+kx): 18 of its 36 MFMAs for (0,3) and (3,6)
  frag wf[3][3];
  auto load_wcol = [&](auto kxc) __attribute__((always_inline)) {
    constexpr int kx = decltype(kxc)::value;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        wf[ky][t] = *reinterpret_cast<const frag*>(wlds + w_lane + (((ky * 3 + kx) * 48 + t * 16) * ROWB));
  };
  auto col_rows = [&](auto parc, auto kxc, auto r0c, auto r1c) __attribute__((always_inline)) {
    constexpr int P = decltype(parc)::value, kx = decltype(kxc)::value, R0 = decltype(r0c)::value, R1 = decltype(r1c)::value;
    const char* halo = P ? halo1 : halo0;
    frag pf[R1 - R0];
#pragma unroll
    for (int r = R0; r < R1; ++r) pf[r - R0] = *reinterpret_cast<const frag*>(halo + p_lane + (r * 18 + kx) * P48_PIXB);
#pragma unroll
    for (int r = R0; r < R1; ++r) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int i = r - ky;
        if (i >= 0 && i < 4) {
#pragma unroll
          for (int t = 0; t < 3; ++t)
            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky][t], pf[r - R0], acc[i][t], 0, 0, 0);
        }
      }
    }
  };
  // paired steps [P0, P1): channels 32..47 of taps 2p and 2p+1
  auto tails = [&](auto parc, auto p0c, auto p1c) __attribute__((always_inline)) {
    constexpr int P = decltype(parc)::value, P0 = decltype(p0c)::value, P1 = decltype(p1c)::value;
    const char* halo = P ? halo1 : halo0;
#pragma unroll
    for (int p = P0; p < P1; ++p) {
      frag wt[3], pf[4];
#pragma unroll
      for (int t = 0; t < 3; ++t)
        wt[t] = *reinterpret_cast<const frag*>(wlds + w_lane + (((9 + p) * 48 + t * 16) * ROWB));
#pragma unroll
      for (int i = 0; i < 4; ++i) pf[i] = *reinterpret_cast<const frag*>(halo + t_lane[p] + i * 18 * P48_PIXB);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[t], pf[i], acc[i][t], 0, 0, 0);
    }
  };

  // Per-channel epilogue constants of the lane's 12 channels.  Slot j = 4t+r: tiles 0/1 -> channel 8q+j,
  // tile 2 -> 32+4q+r.
  auto chan_of = [&](int j) __attribute__((always_inline)) { return j < 8 ? 8 * q + j : 32 + 4 * q + (j - 8); };
  const bool affine = a.ep_scale != nullptr || a.ep_shift != nullptr;
  float esc[12], esh[12], hw0[FUSE ? 12 : 1];
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    const int c = chan_of(j);
    const int cc = c < a.Co ? c : 0;
    const float sc = a.ep_scale != nullptr ? a.ep_scale[cc] : 1.f;
    const float sh = a.ep_shift != nullptr ? a.ep_shift[cc] : 0.f;
    esc[j] = c < a.Co ? sc : 1.f;
    esh[j] = c < a.Co ? sh : 0.f;
    if constexpr (FUSE) {
      const float h = a.head_out != nullptr ? a.head_w[cc] : 0.f;
      hw0[j] = c < a.Co ? h : 0.f;
    }
  }

  // ---- the finished tile: results packed to bf16 in registers (pk), stored during the next tile
  uint32_t pk[4][6];
  P48Item done = cur;
  bool pending = false;
  float ssum[12], ssq[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) ssum[j] = ssq[j] = 0.f;
  auto finish_item = [&]() __attribute__((always_inline)) {
    done = cur;
    const int gx = cur.tx * 16 + l15;
    const bool colok = gx < a.W;
    const int gy0 = cur.ty * 32 + 4 * wave;
    if (a.stats != nullptr) {
      // batch statistics: per-lane running sums over all tiles of the workgroup (reduced across lanes
      // and waves once, at the end of the kernel)
      const bool full = cur.ty * 32 + 32 <= a.H && cur.tx * 16 + 16 <= a.W;
      if (full) {
#pragma unroll
        for (int j = 0; j < 12; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = acc[i][j >> 2][j & 3];
            ssum[j] += v;
            ssq[j] += v * v;
          }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float m = (colok && gy0 + i < a.H) ? 1.f : 0.f;
#pragma unroll
          for (int j = 0; j < 12; ++j) {
            const float v = acc[i][j >> 2][j & 3] * m;
            ssum[j] += v;
            ssq[j] += v * v;
          }
        }
      }
    }
    // folded BatchNorm / bias and the activation; from here on the values as they are stored
    if (affine) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][t][r] = acc[i][t][r] * esc[4 * t + r] + esh[4 * t + r];
    }
    if (a.act == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][t][r] = fmaxf(acc[i][t][r], 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        pk[i][2 * t] = p48_pack2(acc[i][t][0], acc[i][t][1]);
        pk[i][2 * t + 1] = p48_pack2(acc[i][t][2], acc[i][t][3]);
      }
    if constexpr (FUSE) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][t][r] = to_f32(from_f32<T>(acc[i][t][r]));
      if (a.pooled != nullptr) {
        T* pl = reinterpret_cast<T*>(a.pooled);
        const int PH = a.H >> 1, PW = a.W >> 1;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          const int py = (gy0 + 2 * i2) >> 1, px = gx >> 1;
          uint32_t o[6];
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            float m[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              m[r] = fmaxf(acc[2 * i2][t][r], acc[2 * i2 + 1][t][r]);
              m[r] = fmaxf(m[r], p48_dpp<0xB1>(m[r]));      // the neighbouring column
            }
            o[2 * t] = p48_pack2(m[0], m[1]);
            o[2 * t + 1] = p48_pack2(m[2], m[3]);
          }
          if ((l15 & 1) == 0 && py < PH && px < PW) {
            T* dst = pl + (((long)cur.b * PH + py) * PW + px) * a.Co;
            if (8 * q < a.Co) *reinterpret_cast<u32x4*>(dst + 8 * q) = u32x4{o[0], o[1], o[2], o[3]};
            if (32 + 4 * q < a.Co) *reinterpret_cast<uint2*>(dst + 32 + 4 * q) = uint2{o[4], o[5]};
          }
        }
      }
      if (a.head_out != nullptr) {
        // trailing 1x1 conv: the lane's partial dot product over its 12 channels, then the four
        // lanes (q = 0..3) of a pixel are added.  Output 0 uses the weights held in registers.
        for (int o = 0; o < a.head_co; ++o) {
          float hw[12];
#pragma unroll
          for (int j = 0; j < 12; ++j) {
            const int c = chan_of(j);
            hw[j] = o == 0 ? hw0[j] : (c < a.Co ? a.head_w[(long)o * a.Co + c] : 0.f);
          }
          const float hb = a.head_b ? a.head_b[o] : 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float sdot = 0.f;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) sdot += acc[i][t][r] * hw[4 * t + r];
            sdot += __shfl_xor(sdot, 16, 64);
            sdot += __shfl_xor(sdot, 32, 64);
            if (q == 0 && colok && gy0 + i < a.H)
              a.head_out[(((long)cur.b * a.head_co + o) * a.H + gy0 + i) * a.W + gx] = sdot + hb;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    pending = true;
  };
  // row G of the finished tile's result: its 16-byte part (HALF = 0) or its 8-byte part (HALF = 1)
  auto store_piece = [&](auto gc, auto hc) __attribute__((always_inline)) {
    constexpr int G = decltype(gc)::value, HALF = decltype(hc)::value;
    if (a.out == nullptr) return;
    const int gy = done.ty * 32 + 4 * wave + G;
    const int gx = done.tx * 16 + l15;
    if (gy < a.H && gx < a.W) {
      T* dst = reinterpret_cast<T*>(a.out) + (((long)done.b * a.H + gy) * a.W + gx) * a.ldout;
      if constexpr (HALF == 0) {
        if (8 * q < a.Co) *reinterpret_cast<u32x4*>(dst + 8 * q) = u32x4{pk[G][0], pk[G][1], pk[G][2], pk[G][3]};
      } else {
        if (32 + 4 * q < a.Co) *reinterpret_cast<uint2*>(dst + 32 + 4 * q) = uint2{pk[G][4], pk[G][5]};
      }
    }
  };

#ifdef P48_PROF
  long prof[6] = {0, 0, 0, 0, 0, 0};
  const long tstart = __builtin_readcyclecounter();
#endif
  issue_begin();
  static_for<HPW>([&](auto kc) { issue_piece(std::integral_constant<int, 0>{}, kc); });
  auto tile_step = [&](auto parc, int s) __attribute__((always_inline)) {
    constexpr int P = decltype(parc)::value;
    using PN = std::integral_constant<int, P ^ 1>;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>;
    // Tile s has landed (this wave's pieces; after the barrier everyone's), every wave is done with
    // the halo buffer tile s+1 goes into, and the statistics partials of the finished tile are visible.
    // (The wait is the builtin, not inline asm: the compiler then knows that no DMA is in flight
    // here and does not put a vmcnt(0) of its own in front of the first fragment read.)
#ifdef P48_PROF
#define STAMP(k) { const long t_ = __builtin_readcyclecounter(); prof[k] += t_ - tl; tl = t_; }
    long tl = __builtin_readcyclecounter();
#else
#define STAMP(k)
#endif
    __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0)
    STAMP(0)
    asm volatile("s_barrier" ::: "memory");
    STAMP(1)
    const bool more = s + 1 < mine;
    const bool st = pending;
    pending = false;
    if (more) issue_begin();
    // Sixteen side operations per tile - the eight DMA pieces of the next tile's halo, then the eight store
    // instructions of the previous tile's result - one in front of (waves 0..3) or behind (waves 4..7) each
    // group of 9 or 12 MFMAs: the address path sees a steady trickle instead of bursts that every wave queues in.
    auto side_op = [&](auto nc) __attribute__((always_inline)) {
      constexpr int N = decltype(nc)::value;
      STAMP(4)
      if constexpr (N < 8) {
        if (more) issue_piece(PN{}, nc);
        STAMP(2)
      } else {
        if (st) store_piece(std::integral_constant<int, (N - 8) / 2>{}, std::integral_constant<int, (N - 8) % 2>{});
        STAMP(3)
      }
    };
#define P48_GROUP(N, ...)                                          \
    if (early) side_op(std::integral_constant<int, N>{});          \
    __VA_ARGS__;                                                   \
    if (!early) side_op(std::integral_constant<int, N>{});
    using I4 = std::integral_constant<int, 4>;
    load_wcol(I0{});
    P48_GROUP(0, col_rows(parc, I0{}, I0{}, I2{}))
    P48_GROUP(1, col_rows(parc, I0{}, I2{}, I3{}))
    P48_GROUP(2, col_rows(parc, I0{}, I3{}, I4{}))
    P48_GROUP(3, col_rows(parc, I0{}, I4{}, I6{}))
    load_wcol(I1{});
    P48_GROUP(4, col_rows(parc, I1{}, I0{}, I2{}))
    P48_GROUP(5, col_rows(parc, I1{}, I2{}, I3{}))
    P48_GROUP(6, col_rows(parc, I1{}, I3{}, I4{}))
    P48_GROUP(7, col_rows(parc, I1{}, I4{}, I6{}))
    load_wcol(I2{});
    P48_GROUP(8, col_rows(parc, I2{}, I0{}, I2{}))
    P48_GROUP(9, col_rows(parc, I2{}, I2{}, I3{}))
    P48_GROUP(10, col_rows(parc, I2{}, I3{}, I4{}))
    P48_GROUP(11, col_rows(parc, I2{}, I4{}, I6{}))
    P48_GROUP(12, tails(parc, I0{}, I1{}))
    P48_GROUP(13, tails(parc, I1{}, I2{}))
    P48_GROUP(14, tails(parc, I2{}, I3{}))
    P48_GROUP(15, tails(parc, I3{}, I4{}))
    tails(parc, I4{}, I5{});
#undef P48_GROUP
    STAMP(4)
    finish_item();
    advance(cur);
    STAMP(5)
  };
#pragma unroll 1
  for (int s = 0; s < mine; s += 2) {
    tile_step(std::integral_constant<int, 0>{}, s);
    if (s + 1 < mine) tile_step(std::integral_constant<int, 1>{}, s + 1);
  }
  static_for<4>([&](auto gc) {
    store_piece(gc, std::integral_constant<int, 0>{});
    store_piece(gc, std::integral_constant<int, 1>{});
  });
  if (a.stats != nullptr) {
    // one row of partial sums per workgroup: lanes of a pixel row by DPP, the eight waves through LDS
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const float sv = p48_row_sum(ssum[j]), qv = p48_row_sum(ssq[j]);
      if (l15 == 0) {
        red[wave][0][chan_of(j)] = sv;
        red[wave][1][chan_of(j)] = qv;
      }
    }
    __syncthreads();
    if (tid < 96) {
      const int which = tid / 48, c = tid % 48;
      if (c < a.Co) {
        float v = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) v += red[w8][which][c];
        a.stats[(long)pos * 2 * a.Co + (long)which * a.Co + c] = v;
      }
    }
  }
#ifdef P48_PROF
  if (blockIdx.x == 8 && lane == 0 && (wave == 0 || wave == 5))
    printf("p48 wave %d tiles %d: vmwait %ld barrier %ld issue %ld stores %ld mfma %ld finish %ld total %ld (cycles per tile)\n",
           wave, mine, prof[0] / mine, prof[1] / mine, prof[2] / mine, prof[3] / mine, prof[4] / mine, prof[5] / mine,
           (long)(__builtin_readcyclecounter() - tstart) / mine);
#endif
}

// The same structure for THREE 48-channel output blocks over a 33..48-channel input: the data gradient of
// the 144 -> 48 layer.  A tile's halo serves its three blocks; what changes from block to block is the
// weight set (42 KB), and LDS has room for one next to the two halo buffers.  So the blocks of a tile run in
// the order 0,1,2 on even tiles and 2,1,0 on odd ones: the last block's weights stay for the next tile's
// first, the middle block's weights are prefetched into the halo buffer that is idle during the tile's
// first block, the last block's into the weight buffer during the middle one, and the next tile's halo
// during the last one - every load is issued one block ahead, nothing waits.
__global__ __launch_bounds__(512) void conv3x3_p48x3_kernel(ConvArgs a) {
  using T = bf16;
  using frag = bf16x8;
  __shared__ __attribute__((aligned(256))) char halo0[P48_HBUF];
  __shared__ __attribute__((aligned(256))) char halo1[P48_HBUF];
  __shared__ __attribute__((aligned(256))) char wlds[P48_WBUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const bool early = wave < 4;
  const T* in = reinterpret_cast<const T*>(a.in);
  const T* wp = reinterpret_cast<const T*>(a.w);
  const T* zero = reinterpret_cast<const T*>(conv_zero_page);
  const long wblock = (long)a.nchunks * 9 * 48 * 32;        // packed elements per output block

  const int tiles_y32 = (a.H + 31) >> 5;
  const int nitems = a.B * tiles_y32 * a.tiles_x;
  const int nwg = gridDim.x;
  const int pos = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int per = (nitems + nwg - 1) / nwg;
  const int first = pos * per;
  int mine = nitems - first;
  if (mine > per) mine = per;
  if (mine <= 0) return;
  auto advance = [&](P48Item& it) __attribute__((always_inline)) {
    if (++it.tx == a.tiles_x) {
      it.tx = 0;
      if (++it.ty == tiles_y32) {
        it.ty = 0;
        ++it.b;
      }
    }
  };
  P48Item nxt;
  nxt.tx = first % a.tiles_x;
  nxt.ty = (first / a.tiles_x) % tiles_y32;
  nxt.b = first / a.tiles_x / tiles_y32;
  P48Item cur = nxt;

  // weight DMA: piece j = wave + 8k (step j/3, MFMA tile j%3); element offset inside a block's pack, or -1: zeros
  constexpr int WPW = (P48_WINSTR + 7) / 8;
  int w_off[WPW];
  {
    const int m = lane >> 2, slot = (lane & 3) ^ swz(m);
#pragma unroll
    for (int k = 0; k < WPW; ++k) {
      const int j = wave + 8 * k;
      const int st = j / 3, t = j - 3 * st;
      const int chan = t < 2 ? 8 * (m >> 2) + 4 * t + (m & 3) : 32 + m;
      if (st < 9) {
        w_off[k] = (st * 48 + chan) * 32 + slot * 8;
      } else {
        const int tap = 2 * (st - 9) + (slot >> 1);
        w_off[k] = (tap < 9 && a.nchunks > 1) ? 9 * 48 * 32 + (tap * 48 + chan) * 32 + (slot & 1) * 8 : -1;
      }
    }
  }
  auto issue_w_part = [&](char* dst, int nb, auto gc) __attribute__((always_inline)) {    // pieces 2G, 2G+1 of this wave
    constexpr int G = decltype(gc)::value;
    const T* base = wp + nb * wblock;
#pragma unroll
    for (int k = 2 * G; k < 2 * G + 2 && k < WPW; ++k) {
      const int j = wave + 8 * k;
      if (j < P48_WINSTR) glds16(w_off[k] >= 0 ? base + w_off[k] : zero, dst + j * 1024);
    }
  };

  constexpr int HPW = (P48_HINSTR + 7) / 8;
  int hp_off[HPW], hp_yx[HPW];
#pragma unroll
  for (int k = 0; k < HPW; ++k) {
    const int o = (wave + 8 * k) * 1024 + lane * 16;
    const int r = o / P48_ROWB, rem = o - r * P48_ROWB;
    const int px = rem / P48_PIXB, cb = rem - px * P48_PIXB;
    hp_off[k] = (r * a.W + px) * a.ldin + (cb >> 1);
    hp_yx[k] = o >= P48_HBYTES ? (int)0x80000000 : ((cb >> 1) < a.Ci ? ((r << 16) | px) : (0x4000 << 16));
  }
  const T* hbase = nullptr;
  int iy0 = 0, ix0 = 0;
  auto issue_h_begin = [&]() __attribute__((always_inline)) {
    iy0 = nxt.ty * 32 - 1;
    ix0 = nxt.tx * 16 - 1;
    hbase = in + (((long)nxt.b * a.H + iy0) * a.W + ix0) * a.ldin;
    advance(nxt);
  };
  auto issue_h_part = [&](char* hdst, auto gc) __attribute__((always_inline)) {
    constexpr int G = decltype(gc)::value;
#pragma unroll
    for (int k = 2 * G; k < 2 * G + 2 && k < HPW; ++k) {
      const int i = wave + 8 * k;
      if (i < P48_HINSTR && hp_yx[k] >= 0) {
        const int hy = hp_yx[k] >> 16, hx = hp_yx[k] & 0xffff;
        const bool ok = (unsigned)(iy0 + hy) < (unsigned)a.H && (unsigned)(ix0 + hx) < (unsigned)a.W;
        glds16(ok ? hbase + hp_off[k] : zero, hdst + i * 1024);
      }
    }
  };

  const int p_lane = (4 * wave * 18 + l15) * P48_PIXB + 16 * q;
  int t_lane[5];
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    int tap = 2 * p + (q >> 1);
    if (tap > 8) tap = 8;
    t_lane[p] = (4 * wave * 18 + l15 + (tap / 3) * 18 + tap % 3) * P48_PIXB + 64 + 16 * (q & 1);
  }
  const int w_lane = l15 * ROWB + ((q ^ swz(l15)) * 16);

  f32x4 acc[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  frag wf[3][3];
  auto load_wcol = [&](const char* wbuf, auto kxc) __attribute__((always_inline)) {
    constexpr int kx = decltype(kxc)::value;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        wf[ky][t] = *reinterpret_cast<const frag*>(wbuf + w_lane + (((ky * 3 + kx) * 48 + t * 16) * ROWB));
  };
  auto col_rows = [&](const char* halo, auto kxc, auto r0c, auto r1c) __attribute__((always_inline)) {
    constexpr int kx = decltype(kxc)::value, R0 = decltype(r0c)::value, R1 = decltype(r1c)::value;
    frag pf[R1 - R0];
#pragma unroll
    for (int r = R0; r < R1; ++r) pf[r - R0] = *reinterpret_cast<const frag*>(halo + p_lane + (r * 18 + kx) * P48_PIXB);
#pragma unroll
    for (int r = R0; r < R1; ++r) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int i = r - ky;
        if (i >= 0 && i < 4) {
#pragma unroll
          for (int t = 0; t < 3; ++t)
            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky][t], pf[r - R0], acc[i][t], 0, 0, 0);
        }
      }
    }
  };
  auto tails = [&](const char* halo, const char* wbuf, auto p0c, auto p1c) __attribute__((always_inline)) {
    constexpr int P0 = decltype(p0c)::value, P1 = decltype(p1c)::value;
#pragma unroll
    for (int p = P0; p < P1; ++p) {
      frag wt[3], pf[4];
#pragma unroll
      for (int t = 0; t < 3; ++t)
        wt[t] = *reinterpret_cast<const frag*>(wbuf + w_lane + (((9 + p) * 48 + t * 16) * ROWB));
#pragma unroll
      for (int i = 0; i < 4; ++i) pf[i] = *reinterpret_cast<const frag*>(halo + t_lane[p] + i * 18 * P48_PIXB);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[t], pf[i], acc[i][t], 0, 0, 0);
    }
  };

  // the finished block: results packed to bf16 (pk), stored during the next block
  uint32_t pk[4][6];
  P48Item done = cur;
  int done_nb = 0;
  bool pending = false;
  const bool relu = a.act == 1;
  auto finish_block = [&](int nb) __attribute__((always_inline)) {
    done = cur;
    done_nb = nb;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        f32x4 v = acc[i][t];
        if (relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        pk[i][2 * t] = p48_pack2(v[0], v[1]);
        pk[i][2 * t + 1] = p48_pack2(v[2], v[3]);
        acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    pending = true;
  };
  auto store_part = [&](auto gc) __attribute__((always_inline)) {
    constexpr int G = decltype(gc)::value;
    const int n0 = done_nb * 48;
    T* obase = reinterpret_cast<T*>(a.out);
    int ldo = a.ldout, oc = n0;
    if (a.out2 != nullptr && n0 >= a.split) {
      obase = reinterpret_cast<T*>(a.out2);
      ldo = a.ldout2;
      oc = n0 - a.split;
    }
    const int gx = done.tx * 16 + l15;
    const int gy = done.ty * 32 + 4 * wave + G;
    if (gy < a.H && gx < a.W) {
      T* dst = obase + (((long)done.b * a.H + gy) * a.W + gx) * ldo + oc;
      if (n0 + 8 * q < a.Co) *reinterpret_cast<u32x4*>(dst + 8 * q) = u32x4{pk[G][0], pk[G][1], pk[G][2], pk[G][3]};
      if (n0 + 32 + 4 * q < a.Co) *reinterpret_cast<uint2*>(dst + 32 + 4 * q) = uint2{pk[G][4], pk[G][5]};
    }
  };

  // one block of one tile.  P: the halo buffer of the tile; K: 0 first, 1 middle, 2 last block of the tile
  // (weights in wlds / the other halo buffer / wlds); the side job loads what the NEXT block needs.
  auto block = [&](auto parc, auto kc, int nb, int nb_next, bool more_tiles) __attribute__((always_inline)) {
    constexpr int P = decltype(parc)::value, K = decltype(kc)::value;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>;
    const char* halo = P ? halo1 : halo0;
    char* other = P ? halo0 : halo1;
    const char* wbuf = K == 1 ? other : wlds;
    // everything issued during the previous block has landed, and every wave is done multiplying it
    __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0)
    asm volatile("s_barrier" ::: "memory");
    const bool st = pending;
    pending = false;
    if (K == 2 && more_tiles) issue_h_begin();
    auto side = [&](auto gc) __attribute__((always_inline)) {
      if constexpr (K == 0) issue_w_part(other, nb_next, gc);
      if constexpr (K == 1) issue_w_part(wlds, nb_next, gc);
      if constexpr (K == 2) {
        if (more_tiles) issue_h_part(other, gc);
      }
      if (st) store_part(gc);
    };
    load_wcol(wbuf, I0{});
    if (early) side(I0{});
    col_rows(halo, I0{}, I0{}, I3{});
    if (!early) side(I0{});
    col_rows(halo, I0{}, I3{}, I6{});
    load_wcol(wbuf, I1{});
    if (early) side(I1{});
    col_rows(halo, I1{}, I0{}, I3{});
    if (!early) side(I1{});
    col_rows(halo, I1{}, I3{}, I6{});
    load_wcol(wbuf, I2{});
    if (early) side(I2{});
    col_rows(halo, I2{}, I0{}, I3{});
    if (!early) side(I2{});
    col_rows(halo, I2{}, I3{}, I6{});
    if (early) side(I3{});
    tails(halo, wbuf, I0{}, I2{});
    if (!early) side(I3{});
    tails(halo, wbuf, I2{}, I5{});
    finish_block(nb);
  };
  auto tile = [&](auto parc, int s) __attribute__((always_inline)) {
    const bool fwd = (s & 1) == 0;
    const int b0 = fwd ? 0 : 2, b2 = fwd ? 2 : 0;
    block(parc, std::integral_constant<int, 0>{}, b0, 1, true);
    block(parc, std::integral_constant<int, 1>{}, 1, b2, true);
    block(parc, std::integral_constant<int, 2>{}, b2, b2, s + 1 < mine);
    advance(cur);
  };

  // prologue: the first tile's halo and its first block's weights
  issue_h_begin();
  issue_h_part(halo0, std::integral_constant<int, 0>{});
  issue_h_part(halo0, std::integral_constant<int, 1>{});
  issue_h_part(halo0, std::integral_constant<int, 2>{});
  issue_h_part(halo0, std::integral_constant<int, 3>{});
  issue_w_part(wlds, 0, std::integral_constant<int, 0>{});
  issue_w_part(wlds, 0, std::integral_constant<int, 1>{});
  issue_w_part(wlds, 0, std::integral_constant<int, 2>{});
#pragma unroll 1
  for (int s = 0; s < mine; s += 2) {
    tile(std::integral_constant<int, 0>{}, s);
    if (s + 1 < mine) tile(std::integral_constant<int, 1>{}, s + 1);
  }
  store_part(std::integral_constant<int, 0>{});
  store_part(std::integral_constant<int, 1>{});
  store_part(std::integral_constant<int, 2>{});
  store_part(std::integral_constant<int, 3>{});
}

// ---- weight packing: [Co][Ci][KS][KS] f32 -> [nblk][nchunk][tap][BN][KC] T
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ dst, int Co, int Ci,
                                   int ks, int bn, int kc, int nchunks, int transpose, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int taps = ks * ks;
  long t = idx;
  const int k = t % kc;
  t /= kc;
  const int row = t % bn;
  t /= bn;
  const int tap = t % taps;
  t /= taps;
  const int ch = t % nchunks;
  const int nb = t / nchunks;
  const int n = nb * bn + row;   // output channel of the packed conv
  const int c = ch * kc + k;     // input channel of the packed conv
  float v = 0.f;
  if (!transpose) {
    if (n < Co && c < Ci) v = w[((long)n * Ci + c) * taps + tap];
  } else {
    // data-gradient conv: out channel n = original ci, in channel c = original co, taps flipped
    if (n < Ci && c < Co) v = w[((long)c * Ci + n) * taps + (taps - 1 - tap)];
  }
  dst[idx] = from_f32<T>(v);
}

// Many weights in one launch (a training step re-packs every conv / linear weight, forward and
// data-gradient form: dozens of tiny launches otherwise).  jobs: 9 int64 per job = source pointer,
// destination offset (elements from dst_base), Co, Ci, ks, bn, chunks, transpose, packed elements;
// blocks: per workgroup (job, group inside the job).  A group is RG packed rows (16 for 3x3, the whole
// bn-row block for 1x1) x one channel chunk x all taps: its source is RG runs of kc*taps consecutive
// floats (forward form) or kc runs of RG*taps consecutive floats (data-gradient form), read once and
// coalesced into LDS, and its destination is one run of RG*kc elements per tap.  (The one-thread-per-
// destination-element form above gathers 4-byte words 36 bytes apart for a 3x3 weight and a whole source
// row apart for a data-gradient form: 3-6x slower on the weights of a training step.)
inline int pack_group_rows(int ks, int bn) { return ks == 1 ? bn : 16; }
constexpr int PACK_LDS_FLOATS = 32 * (16 * 9 + 1);      // the data-gradient form of a 3x3 bf16 group, padded rows
static_assert(PACK_LDS_FLOATS >= 32 * (96 + 1) && PACK_LDS_FLOATS >= 16 * 32 * 9, "every group form fits");

// floor(e / d) for 0 <= e < 2^13, 0 < d < 2^11 by one multiplication (inv = 1 / d): (e + 0.5) / d is at least
// 0.5 / d away from an integer, far more than the rounding of the product
__device__ __forceinline__ int small_div(int e, float inv) { return (int)(((float)e + 0.5f) * inv); }

template <typename T, int TAPS>
__device__ __forceinline__ void pack_group(const float* __restrict__ w, T* __restrict__ dst, float* buf, int Co, int Ci,
                                           int bn, int nchunks, int transpose, int convt, int g) {
  // convt (1x1 jobs only): the source is a ConvTranspose2d(k=2, s=2) weight [Ci][up][2][2] read as the matrix
  // W[(ky*2+kx)*up + co][ci] of the GEMM that computes the four output pixels of an input pixel (Co = 4*up)
  const int up = Co >> 2;
  auto at = [&](int a, int b) -> float {       // element (row a, column b) of the Co x Ci source matrix
    return convt ? w[((long)b * up + (a % up)) * 4 + a / up] : w[(long)a * Ci + b];
  };
  constexpr int KC = MMA<T>::KC;
  const int RG = TAPS == 1 ? bn : 16, gpb = bn / RG;
  const int rg = g % gpb, ch = (g / gpb) % nchunks, nb = g / (gpb * nchunks);
  const int n0 = nb * bn + rg * RG;          // first packed row (output channel of the packed conv)
  const int c0 = ch * KC;                    // first packed input channel
  const int tid = threadIdx.x;
  // 1x1 weights whose rows start on 16-byte boundaries move as float4 (four times fewer load instructions: the
  // loads, not the bytes, bound this kernel); the 36-byte taps of a 3x3 weight keep no such alignment
  const bool vec = TAPS == 1 && !convt && (Ci & 3) == 0 && ((uintptr_t)w & 15) == 0;
  if (!transpose) {
    constexpr int RUN = KC * TAPS;           // floats of one source row that belong to the chunk
    if (vec) {
      for (int e = tid * 4; e < RG * RUN; e += 1024) {
        const int r = e / RUN, q = e % RUN;
        const int n = n0 + r, c = c0 + q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < Co && c < Ci) v = *reinterpret_cast<const float4*>(w + (long)n * Ci + c);     // (Ci % 4 == 0: all four inside)
        *reinterpret_cast<float4*>(buf + e) = v;
      }
    } else {
      for (int e = tid; e < RG * RUN; e += 256) {
        const int r = e / RUN, q = e % RUN;
        const int n = n0 + r, c = c0 + q / TAPS;
        buf[e] = (n < Co && c < Ci) ? (TAPS == 1 ? at(n, c) : w[((long)n * Ci + c0) * TAPS + q]) : 0.f;
      }
    }
  } else {
    // packed row n = source input channel, packed channel c = source output channel, taps flipped
    const int run = RG * TAPS, pitch = run + 1;
    const float inv = 1.f / (float)run;
    if (vec) {                               // (RG = bn is a multiple of 16: a run is whole float4s)
      for (int e = tid * 4; e < KC * run; e += 1024) {
        const int k = small_div(e, inv), q = e - k * run;
        const int c = c0 + k, n = n0 + q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < Co && n < Ci) v = *reinterpret_cast<const float4*>(w + (long)c * Ci + n);
        float* b = buf + k * pitch + q;
        b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
      }
    } else {
      for (int e = tid; e < KC * run; e += 256) {
        const int k = small_div(e, inv), q = e - k * run;
        const int c = c0 + k, n = n0 + q / TAPS;
        buf[k * pitch + q] = (c < Co && n < Ci) ? (TAPS == 1 ? at(c, n) : w[((long)c * Ci + n0) * TAPS + q]) : 0.f;
      }
    }
  }
  __syncthreads();
  constexpr int E = Vec16<T>::N;             // destination elements per 16-byte store (KC is a multiple)
  const int per_tap = RG * KC;
  const float inv_pt = 1.f / (float)per_tap;
  const long tap_stride = (long)bn * KC;
  T* __restrict__ d0 = dst + ((long)(nb * nchunks + ch) * TAPS * bn + rg * RG) * KC;
  for (int i = tid * E; i < per_tap * TAPS; i += 256 * E) {
    const int tap = TAPS == 1 ? 0 : small_div(i, inv_pt), rem = i - tap * per_tap;
    const int r = rem / KC, k = rem % KC;
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < E; ++e)
      o.v[e] = from_f32<T>(transpose ? buf[(k + e) * (RG * TAPS + 1) + r * TAPS + (TAPS - 1 - tap)]
                                     : buf[(r * KC + k + e) * TAPS + tap]);
    store16(d0 + tap * tap_stride + rem, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const long* __restrict__ jobs,
                                                                   const int* __restrict__ blocks,
                                                                   T* __restrict__ dst_base) {
  __shared__ __attribute__((aligned(16))) float buf[PACK_LDS_FLOATS];
  const int job = blocks[2 * blockIdx.x], g = blocks[2 * blockIdx.x + 1];
  const long* j = jobs + 9L * job;
  const float* __restrict__ w = reinterpret_cast<const float*>(j[0]);
  T* __restrict__ dst = dst_base + j[1];
  const int Co = (int)j[2], Ci = (int)j[3], ks = (int)j[4], bn = (int)j[5], nchunks = (int)j[6];
  const int transpose = (int)j[7] & 1, convt = ((int)j[7] >> 1) & 1;
  if (ks == 1) pack_group<T, 1>(w, dst, buf, Co, Ci, bn, nchunks, transpose, convt, g);
  else         pack_group<T, 9>(w, dst, buf, Co, Ci, bn, nchunks, transpose, 0, g);
}

// up to 48 -> 48 channels, bf16, 3x3 (the full-resolution layers): the persistent whole-pixel kernel (it has
// no residual, GELU or split-output epilogue)
inline bool use_p48(const ConvArgs& a, int bn) {
  static const bool off = getenv("UNETK_NO_P48") != nullptr;
  return bn == 48 && a.Co <= 48 && a.Ci > 32 && a.Ci <= 48 && !off && a.resid == nullptr && a.out2 == nullptr && a.in2 == nullptr &&
         a.act <= 1 && (int64_t)a.W * a.ldin * 36 < (1LL << 31);
}
inline int p48_items(const ConvArgs& a) { return a.B * cdiv(a.H, 32) * cdiv(a.W, 16); }
inline int p48_workgroups(const ConvArgs& a) {
  const int n = p48_items(a);
  return n < 256 ? 8 * cdiv(n, 8) : 256;
}
// workgroups that have tiles = rows of batch statistics written
inline int p48_stats_rows(const ConvArgs& a) {
  const int n = p48_items(a), per = cdiv(n, p48_workgroups(a));
  return cdiv(n, per);
}

template <typename T, int KS>
int launch_conv(ConvArgs a, int bn, hipStream_t stream) {
  const long npix = (long)a.B * a.H * a.W;
  a.ntiles = (KS == 3) ? a.B * a.tiles_x * a.tiles_y : (int)cdiv(npix, 256);
  a.tiles_per_xcd = cdiv(a.ntiles, 8);
  a.nby = cdiv(a.Co, bn);
  dim3 grid((unsigned)(8 * a.tiles_per_xcd * a.nby));
  if constexpr (KS == 3 && sizeof(T) == 2) {
    if (a.bs_partials != nullptr) {    // data gradient + the next BatchNorm backward's sums (arguments checked by the entry point)
      if (bn == 96) hipLaunchKernelGGL((conv_igemm_kernel<T, 6, KS, false, false, true>), grid, dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((conv_igemm_kernel<T, 4, KS, false, false, true>), grid, dim3(256), 0, stream, a);
      UNETK_CHECK_LAUNCH();
      return UNETK_OK;
    }
  }
  if constexpr (KS == 3 && sizeof(T) == 2) {
    if (a.nby == 3 && a.stats == nullptr && a.ep_scale == nullptr && a.ep_shift == nullptr && a.pooled == nullptr &&
        a.head_out == nullptr && (a.out2 == nullptr || a.split % 48 == 0)) {
      // three 48-channel output blocks over a narrow input (the data gradient of the 144 -> 48 layer)
      ConvArgs b = a;
      b.Co = 48;
      b.out2 = nullptr;
      if (use_p48(b, bn)) {
        hipLaunchKernelGGL(conv3x3_p48x3_kernel, dim3(p48_workgroups(b)), dim3(512), 0, stream, a);
        UNETK_CHECK_LAUNCH();
        return UNETK_OK;
      }
    }
    if (use_p48(a, bn)) {
      const int nwg = p48_workgroups(a);
      if (a.pooled != nullptr || a.head_out != nullptr)
        hipLaunchKernelGGL((conv3x3_p48_kernel<true>), dim3(nwg), dim3(512), 0, stream, a);
      else
        hipLaunchKernelGGL((conv3x3_p48_kernel<false>), dim3(nwg), dim3(512), 0, stream, a);
      UNETK_CHECK_LAUNCH();
      return UNETK_OK;
    }
  }
  if constexpr (KS == 3) {
    if (a.pooled != nullptr || a.head_out != nullptr) {      // inference-time fused epilogue
      switch (bn) {
        case 32: hipLaunchKernelGGL((conv_igemm_occ4_kernel<T, 2, KS, true>), grid, dim3(256), 0, stream, a); break;
        case 48: hipLaunchKernelGGL((conv_igemm_occ4_kernel<T, 3, KS, true>), grid, dim3(256), 0, stream, a); break;
        case 64: hipLaunchKernelGGL((conv_igemm_kernel<T, 4, KS, true>), grid, dim3(256), 0, stream, a); break;
        case 96: hipLaunchKernelGGL((conv_igemm_kernel<T, 6, KS, true>), grid, dim3(256), 0, stream, a); break;
        default:
          unetk_set_error("conv: unsupported channel block %d (32, 48, 64 or 96)", bn);
          return UNETK_EINVAL;
      }
      UNETK_CHECK_LAUNCH();
      return UNETK_OK;
    }
  }
  if constexpr (KS == 3) {
    if (a.in2 != nullptr) {        // the operand is [in | in2]
      switch (bn) {
        case 32: hipLaunchKernelGGL((conv_igemm_occ4_kernel<T, 2, KS, false, true>), grid, dim3(256), 0, stream, a); break;
        case 48: hipLaunchKernelGGL((conv_igemm_occ4_kernel<T, 3, KS, false, true>), grid, dim3(256), 0, stream, a); break;
        case 64: hipLaunchKernelGGL((conv_igemm_kernel<T, 4, KS, false, true>), grid, dim3(256), 0, stream, a); break;
        case 96: hipLaunchKernelGGL((conv_igemm_kernel<T, 6, KS, false, true>), grid, dim3(256), 0, stream, a); break;
        default:
          unetk_set_error("conv: unsupported channel block %d (32, 48, 64 or 96)", bn);
          return UNETK_EINVAL;
      }
      UNETK_CHECK_LAUNCH();
      return UNETK_OK;
    }
  }
  if (KS == 3 && bn <= 48) {
    if (bn == 32) hipLaunchKernelGGL((conv_igemm_occ4_kernel<T, 2, KS>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv_igemm_occ4_kernel<T, 3, KS>), grid, dim3(256), 0, stream, a);
    UNETK_CHECK_LAUNCH();
    return UNETK_OK;
  }
  switch (bn) {
    case 32: hipLaunchKernelGGL((conv_igemm_kernel<T, 2, KS>), grid, dim3(256), 0, stream, a); break;
    case 48: hipLaunchKernelGGL((conv_igemm_kernel<T, 3, KS>), grid, dim3(256), 0, stream, a); break;
    case 64: hipLaunchKernelGGL((conv_igemm_kernel<T, 4, KS>), grid, dim3(256), 0, stream, a); break;
    case 96: hipLaunchKernelGGL((conv_igemm_kernel<T, 6, KS>), grid, dim3(256), 0, stream, a); break;
    default:
      unetk_set_error("conv: unsupported channel block %d (32, 48, 64 or 96)", bn);
      return UNETK_EINVAL;
  }
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

inline int kc_of(int dtype) { return dtype == UNETK_BF16 ? 32 : 16; }
inline int epp_of(int dtype) { return dtype == UNETK_BF16 ? 8 : 4; }

}  // namespace

extern "C" {

int unetk_conv_pick_bn(int Co) {
  if (getenv("UNETK_FORCE_BN48") && Co % 48 == 0) return 48;
  if (Co % 96 == 0) return 96;
  if (Co % 64 == 0) return 64;
  if (Co % 48 == 0) return 48;
  if (Co <= 32) return 32;
  if (Co <= 48) return 48;
  if (Co <= 64) return 64;
  return (Co % 32 == 0) ? 32 : 96;
}

int64_t unetk_conv_packed_elems(int Co, int Ci, int ks, int bn, int dtype) {
  const int kc = kc_of(dtype);
  return (int64_t)cdiv(Co, bn) * cdiv(Ci, kc) * ks * ks * bn * kc;
}

int unetk_conv_pack_weight(const float* w, void* packed, int Co, int Ci, int ks, int bn, int dtype,
                           int transpose, hipStream_t stream) {
  UNETK_CHECK_ARG(w && packed, "conv_pack_weight: null pointer");
  UNETK_CHECK_ARG(ks == 1 || ks == 3, "conv_pack_weight: kernel size %d (1 or 3)", ks);
  UNETK_CHECK_ARG(dtype == UNETK_F32 || dtype == UNETK_BF16, "conv_pack_weight: bad dtype %d", dtype);
  const int pCo = transpose ? Ci : Co, pCi = transpose ? Co : Ci;
  const int kc = kc_of(dtype);
  const int nchunks = cdiv(pCi, kc);
  const long total = unetk_conv_packed_elems(pCo, pCi, ks, bn, dtype);
  const int grid = cdiv(total, 256);
  if (dtype == UNETK_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<bf16>, dim3(grid), dim3(256), 0, stream, w, (bf16*)packed, Co,
                       Ci, ks, bn, kc, nchunks, transpose, total);
  else
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(grid), dim3(256), 0, stream, w, (float*)packed,
                       Co, Ci, ks, bn, kc, nchunks, transpose, total);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_conv_pack_groups(int Co, int Ci, int ks, int bn, int dtype) {
  if (ks != 1 && ks != 3) return 0;
  return cdiv(Co, bn) * cdiv(Ci, kc_of(dtype)) * (bn / pack_group_rows(ks, bn));
}

int unetk_conv_pack_weights(const int64_t* jobs, const int* blocks, int nblocks, void* dst_base, int dtype,
                            hipStream_t stream) {
  UNETK_CHECK_ARG(jobs && blocks && nblocks > 0 && dst_base && ((uintptr_t)dst_base % 16) == 0, "conv_pack_weights: bad arguments");
  UNETK_CHECK_ARG(dtype == UNETK_F32 || dtype == UNETK_BF16, "conv_pack_weights: bad dtype %d", dtype);
  if (dtype == UNETK_BF16)
    hipLaunchKernelGGL(pack_weights_batched_kernel<bf16>, dim3(nblocks), dim3(256), 0, stream,
                       reinterpret_cast<const long*>(jobs), blocks, (bf16*)dst_base);
  else
    hipLaunchKernelGGL(pack_weights_batched_kernel<float>, dim3(nblocks), dim3(256), 0, stream,
                       reinterpret_cast<const long*>(jobs), blocks, (float*)dst_base);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_conv_num_tiles(int B, int H, int W, int ks) {
  if (ks == 3) return B * cdiv(H, 16) * cdiv(W, 16);
  return cdiv((long)B * H * W, 256);
}

int unetk_conv_stats_rows(int B, int H, int W, int Ci, int Co, int ks, int bn, int dtype, int ldin, int act) {
  if (ks == 3 && dtype == UNETK_BF16) {
    ConvArgs a = {};
    a.B = B; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.ldin = ldin; a.act = act; a.tiles_x = cdiv(W, 16);
    if (use_p48(a, bn)) return p48_stats_rows(a);
  }
  return unetk_conv_num_tiles(B, H, W, ks);
}

static int conv_fwd_impl(const void* in, int ldin, const void* in2, int ldin2, int Ci1, const void* wpacked, void* out,
                         int ldout, void* out2, int ldout2, int split, void* pooled, const float* head_w,
                         const float* head_b, float* head_out, int head_co, int B, int H, int W, int Ci, int Co, int ks,
                         int bn, int dtype, float* stats, const float* ep_scale, const float* ep_shift, int act,
                         const void* resid, int ldres, const float* resid_keep, hipStream_t stream,
                         const ConvArgs* bnsums = nullptr);

int unetk_conv_fwd_fused(const void* in, int ldin, const void* wpacked, void* out, int ldout, void* out2,
                         int ldout2, int split, void* pooled, const float* head_w, const float* head_b,
                         float* head_out, int head_co, int B, int H, int W, int Ci, int Co, int ks, int bn,
                         int dtype, float* stats, const float* ep_scale, const float* ep_shift, int act,
                         const void* resid, int ldres, const float* resid_keep, hipStream_t stream) {
  return conv_fwd_impl(in, ldin, nullptr, 0, 0, wpacked, out, ldout, out2, ldout2, split, pooled, head_w, head_b, head_out,
                       head_co, B, H, W, Ci, Co, ks, bn, dtype, stats, ep_scale, ep_shift, act, resid, ldres, resid_keep,
                       stream);
}

int unetk_conv_fwd_cat(const void* in, int ldin, int Ci1, const void* in2, int ldin2, const void* wpacked, void* out,
                       int ldout, int B, int H, int W, int Ci, int Co, int bn, int dtype, float* stats,
                       const float* ep_scale, const float* ep_shift, int act, hipStream_t stream) {
  UNETK_CHECK_ARG(in2 != nullptr, "conv_fwd_cat: null second input");
  const int epp = epp_of(dtype), kc = kc_of(dtype);
  UNETK_CHECK_ARG(Ci1 > 0 && Ci1 < Ci && Ci1 % kc == 0, "conv_fwd_cat: the first input must hold a multiple of %d channels (%d of %d)", kc, Ci1, Ci);
  UNETK_CHECK_ARG(ldin >= Ci1 && ldin2 >= Ci - Ci1 && ldin2 % epp == 0 && ((uintptr_t)in2 % 16) == 0 &&
                  (int64_t)W * ldin2 * 20 + ldin2 < (1LL << 31),
                  "conv_fwd_cat: second input misaligned, too narrow or too wide");
  return conv_fwd_impl(in, ldin, in2, ldin2, Ci1, wpacked, out, ldout, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, 0,
                       B, H, W, Ci, Co, 3, bn, dtype, stats, ep_scale, ep_shift, act, nullptr, 0, nullptr, stream);
}

static int conv_fwd_impl(const void* in, int ldin, const void* in2, int ldin2, int Ci1, const void* wpacked, void* out,
                         int ldout, void* out2, int ldout2, int split, void* pooled, const float* head_w,
                         const float* head_b, float* head_out, int head_co, int B, int H, int W, int Ci, int Co, int ks,
                         int bn, int dtype, float* stats, const float* ep_scale, const float* ep_shift, int act,
                         const void* resid, int ldres, const float* resid_keep, hipStream_t stream,
                         const ConvArgs* bnsums) {
  UNETK_CHECK_ARG(in && wpacked && (out || head_out), "conv_fwd: null pointer");
  UNETK_CHECK_ARG((pooled == nullptr && head_out == nullptr) || (ks == 3 && resid == nullptr && out2 == nullptr),
                  "conv_fwd: pooled / head outputs need a 3x3 conv without residual or split");
  UNETK_CHECK_ARG(head_out == nullptr || (head_w && head_co > 0 && Co <= bn),
                  "conv_fwd: the fused 1x1 head needs all %d channels in one block of %d", Co, bn);
  UNETK_CHECK_ARG((pooled == nullptr && head_out == nullptr) || act == 0 || act == 1,
                  "conv_fwd: pooled / head outputs go with no activation or ReLU");
  if (out == nullptr) ldout = Co;
  UNETK_CHECK_ARG(ks == 1 || ks == 3, "conv_fwd: kernel size %d (1 or 3)", ks);
  UNETK_CHECK_ARG(dtype == UNETK_F32 || dtype == UNETK_BF16, "conv_fwd: bad dtype %d", dtype);
  UNETK_CHECK_ARG(B > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0, "conv_fwd: bad shape");
  const int epp = epp_of(dtype);
  UNETK_CHECK_ARG(Ci % epp == 0 && Co % epp == 0 && ldin % epp == 0 && ldout % epp == 0,
                  "conv_fwd: channels and strides must be multiples of %d (Ci=%d Co=%d ldin=%d ldout=%d)",
                  epp, Ci, Co, ldin, ldout);
  if (out2 != nullptr) {
    UNETK_CHECK_ARG(split > 0 && split < Co && split % bn == 0, "conv_fwd: split %d must be a multiple of the channel block %d inside (0, Co)", split, bn);
    UNETK_CHECK_ARG(ldout2 % epp == 0 && ldout2 >= Co - split && ldout >= split && ((uintptr_t)out2 % 16) == 0,
                    "conv_fwd: second output misaligned or too narrow");
    UNETK_CHECK_ARG(resid == nullptr, "conv_fwd: residual and split output are exclusive");
  } else {
    UNETK_CHECK_ARG(ldout >= Co, "conv_fwd: stride smaller than channel count");
  }
  UNETK_CHECK_ARG(ldin >= (in2 ? Ci1 : Ci), "conv_fwd: stride smaller than channel count");
  UNETK_CHECK_ARG((int64_t)W * ldin * 20 + ldin < (1LL << 31) && 256LL * ldin < (1LL << 31),
                  "conv_fwd: a halo of the input spans 2^31 elements or more");
  UNETK_CHECK_ARG(((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)wpacked % 16) == 0,
                  "conv_fwd: pointers must be 16-byte aligned");
  UNETK_CHECK_ARG(resid == nullptr || (ldres % epp == 0 && ((uintptr_t)resid % 16) == 0),
                  "conv_fwd: residual misaligned");
  ConvArgs a = {};
  if (bnsums != nullptr) a = *bnsums;        // (only the bs_* fields are set in it)
  a.in = in; a.w = wpacked; a.out = out; a.stats = stats;
  a.in2 = in2; a.ldin2 = ldin2; a.Ci1 = Ci1;
  a.out2 = out2; a.ldout2 = ldout2; a.split = out2 ? split : 0;
  a.pooled = pooled; a.head_w = head_w; a.head_b = head_b; a.head_out = head_out; a.head_co = head_co;
  a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.resid = resid; a.resid_keep = resid_keep;
  a.ldin = ldin; a.ldout = ldout; a.ldres = ldres;
  a.B = B; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
  a.nchunks = cdiv(Ci, kc_of(dtype));
  a.act = act;
  a.tiles_x = cdiv(W, 16); a.tiles_y = cdiv(H, 16);
  if (dtype == UNETK_BF16)
    return ks == 3 ? launch_conv<bf16, 3>(a, bn, stream) : launch_conv<bf16, 1>(a, bn, stream);
  return ks == 3 ? launch_conv<float, 3>(a, bn, stream) : launch_conv<float, 1>(a, bn, stream);
}


int unetk_conv_bwd_data_bnsums_supported(int Co, int bn, int dtype) {
  static const bool off = getenv("UNETK_NO_CONV_BNSUMS") != nullptr;
  return !off && dtype == UNETK_BF16 && (bn == 64 || bn == 96) && Co % 8 == 0;
}

int unetk_conv_bwd_data_bnsums(const void* dy, int lddy, const void* wpackedT, void* out, int B, int H, int W, int Ci,
                               int Co, int bn, int dtype, const void* y, int ldy, const float* scale, const float* shift,
                               const float* mean, const float* invstd, float* partials, hipStream_t stream) {
  UNETK_CHECK_ARG(unetk_conv_bwd_data_bnsums_supported(Co, bn, dtype),
                  "conv_bwd_data_bnsums: bfloat16 with a channel block of 64 or 96 only (Co=%d bn=%d dtype=%d)", Co, bn, dtype);
  UNETK_CHECK_ARG(out && y && scale && shift && mean && invstd && partials, "conv_bwd_data_bnsums: null pointer");
  UNETK_CHECK_ARG(ldy >= Co && ldy % 8 == 0 && ((uintptr_t)y % 16) == 0, "conv_bwd_data_bnsums: y misaligned or too narrow (ld %d)", ldy);
  ConvArgs bs = {};
  bs.bs_y = y; bs.ldbs = ldy; bs.bs_scale = scale; bs.bs_shift = shift; bs.bs_mean = mean; bs.bs_invstd = invstd;
  bs.bs_partials = partials;
  return conv_fwd_impl(dy, lddy, nullptr, 0, 0, wpackedT, out, Co, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, B, H,
                       W, Ci, Co, 3, bn, dtype, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, stream, &bs);
}

int unetk_conv_fwd_split(const void* in, int ldin, const void* wpacked, void* out, int ldout, void* out2,
                         int ldout2, int split, int B, int H, int W, int Ci, int Co, int ks, int bn, int dtype,
                         float* stats, const float* ep_scale, const float* ep_shift, int act,
                         const void* resid, int ldres, const float* resid_keep, hipStream_t stream) {
  return unetk_conv_fwd_fused(in, ldin, wpacked, out, ldout, out2, ldout2, split, nullptr, nullptr, nullptr, nullptr,
                              0, B, H, W, Ci, Co, ks, bn, dtype, stats, ep_scale, ep_shift, act, resid, ldres,
                              resid_keep, stream);
}

int unetk_conv_fwd(const void* in, int ldin, const void* wpacked, void* out, int ldout, int B, int H,
                   int W, int Ci, int Co, int ks, int bn, int dtype, float* stats,
                   const float* ep_scale, const float* ep_shift, int act, const void* resid, int ldres,
                   const float* resid_keep, hipStream_t stream) {
  return unetk_conv_fwd_split(in, ldin, wpacked, out, ldout, nullptr, 0, 0, B, H, W, Ci, Co, ks, bn, dtype, stats,
                              ep_scale, ep_shift, act, resid, ldres, resid_keep, stream);
}

}  // extern "C"
