// Weight-gradient of the 3x3 (pad 1) / 1x1 convolution on MFMA for gfx950.
//
//   dw[co,ci,ky,kx] = sum_{b,y,x} dy[b,y,x,co] * in[b,y+ky-1,x+kx-1,ci]
//
// GEMM view: M = Cout, N = taps*Cin, K = pixels.  Both operands are stored pixel-major (NHWC),
// i.e. K is the slow axis of both, while an MFMA fragment wants K contiguous per lane.  For bf16
// the fragments are therefore read from LDS with the transposing ds_read_b64_tr_b16 (4 pixels x
// 16 channels per 16-lane group); the f32 MFMA (16x16x4) takes one element per lane and needs no
// transpose.
//
// A workgroup (8 waves) owns BM = MT*16 output channels x KCI = NWN*16 input channels x all taps
// and walks a strided list of pixel tiles (TR x TW pixels: 256 px bf16 / 128 px f32).  The dy tile
// and the input halo of a tile are brought in by LDS-DMA (global_load_lds) into a double buffer:
// tile k+1 is in flight while tile k is multiplied, one barrier per tile.  Every tap is a shifted
// window of the same halo.  Waves split the work as (16-channel slice of the input chunk) x
// (k-step residue); the residues are summed through LDS at the end and the workgroup writes one
// fp32 partial into a slab [S][Co_pad][taps][Ci_pad]; a second kernel sums the slabs in a fixed
// order (bitwise reproducible, no atomics) and writes the [Co][Ci][KS][KS] fp32 gradient.
#include "common.h"
#include <cstdlib>

// the whole-row kernel of the narrow token linears (gemm.hip)
int64_t lin_wgrad_rows_workspace(int64_t M, int K, int N);
int lin_wgrad_rows_try(const void* x, int ldx, const void* dy, int lddy, float* dw, float* db, int64_t M, int K, int N,
                       void* ws, int64_t ws_bytes, int accumulate, hipStream_t stream, int* rc);

namespace {

// page of zeros that out-of-image pixels and channel padding load
__device__ __attribute__((aligned(64))) unsigned int wgrad_zero_page[16];

__device__ __forceinline__ void glds16(const void* src, char* lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(src, LDS_PTR(void, lds_dst), 16, 0, 0);
#endif
}

struct WgradArgs {
  const void* in;
  const void* in2;          // input channels >= Ci1 come from here (ld = ldin2, renumbered from 0), or null
  int ldin2, Ci1;
  const void* dy;
  float* slab;
  float* bias_slab;         // [nsplit][Co_pad] column sums of dy (1x1 bf16 only), or null
  int ldin, lddy;
  int B, H, W, Ci, Co;
  int Ci_pad, Co_pad;
  int tiles_x, tiles_y, ntiles, nsplit;
  int gx, gy;               // workgroups along Ci and Co
  int px_ngrp1, px_ngrp2;   // whole-pixel kernel: column groups of input slices in `in` and in `in2`
};

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ bf16x8 tr_read8(const char* p0, const char* p1) {
  short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, p0));
  short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, p1));
  typedef __attribute__((ext_vector_type(8))) short short8v;
  short8v r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

// Slot swizzle of the (unpadded, lane-linear) halo image, keyed on the pixel's COLUMN in the halo.
// A transposing read touches, per 32-lane half, the pixels {o..o+3, o+8..o+11} of one halo row at
// one 32-byte column unit; XORing the unit index with column bits 1 and 3 (128-byte rows) or bit 3
// (64-byte rows) spreads those eight pixels over eight distinct 32-byte units of the 256-byte bank
// row (the halo width is even, so a pixel's row parity in the image is its column parity):
// conflict-free.  Because the key ignores the halo row, the reads of another kernel row or k-step
// are the same lane address plus a compile-time offset.  (The dy image uses it with the tile pixel
// as key: k-steps are 32 pixels apart, which leaves bits 1 and 3 alone.)
template <int ROW_BYTES> __device__ __forceinline__ int hswz(int col, int slot16) {
  if (ROW_BYTES == 64) return slot16 ^ (((col >> 3) & 1) << 1);
  if (ROW_BYTES == 128) return slot16 ^ ((((col >> 1) & 1) | (((col >> 3) & 1) << 1)) << 1);
  return slot16;
}

// LDS image of the bf16 dy tile.  A transposing read touches, per 32-lane half, tile pixels
// {o..o+3, o+8..o+11} at one 32-byte column unit.  With 96-byte rows (48 channels) pixels p and p+8
// are 768 bytes = 3 bank rows apart and would collide whatever the slot order, so the pixels of a
// 16-pixel block are stored interleaved (0,8,1,9,...): the eight rows of a read are then
// consecutive and cover all 64 banks exactly once.  64- and 128-byte rows take the slot swizzle
// of the halo instead.
template <int ROW_BYTES> __device__ __forceinline__ int dy_row(int pix) {
  return ROW_BYTES == 96 ? ((pix & ~15) | ((pix & 7) << 1) | ((pix >> 3) & 1)) : pix;
}
template <int ROW_BYTES> __device__ __forceinline__ int dy_pix(int row) {   // inverse of dy_row
  return ROW_BYTES == 96 ? ((row & ~15) | ((row >> 1) & 7) | ((row & 1) << 3)) : row;
}
template <int ROW_BYTES> __device__ __forceinline__ int dy_slot(int pix, int slot16) {
  return ROW_BYTES == 96 ? slot16 : hswz<ROW_BYTES>(pix, slot16);
}

template <typename T, int MT, int TW, int KS, int NWN>
__global__ __launch_bounds__(512) void conv_wgrad_kernel(WgradArgs a) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int EPP = 16 / (int)sizeof(T);
  constexpr int TPIX = BF ? 256 : 128;          // pixels per tile
  constexpr int TR = TPIX / TW;
  constexpr int BM = MT * 16;
  constexpr int NWK = 8 / NWN;                  // k-step residues (waves along K)
  // 1x1 (plain GEMM) with NWN == 1: there are no taps to amortise the dy fragments over, so every
  // wave takes four 16-channel slices of a 64-channel chunk and the eight waves only split K
  constexpr int NSL = (BF && KS == 1 && NWN == 1) ? 4 : 1;
  constexpr int KCI = NWN * NSL * 16;           // input channels per workgroup
  constexpr int TAPS = KS * KS;
  constexpr int NACC = TAPS * NSL;              // accumulator tiles per 16 output channels
  // 1x1 bf16 (linear layers): one more tile per row block can take sum_pixels dy (the bias gradient),
  // as dy^T times a matrix of ones, from the fragments the weight gradient loads anyway
  constexpr int NB = (BF && KS == 1) ? 1 : 0;
  constexpr int HW_ = (KS == 3) ? TW + 2 : TW;  // halo width
  constexpr int HH_ = (KS == 3) ? TR + 2 : TR;
  constexpr int HPIX = HH_ * HW_;
  constexpr int RBD = BM * (int)sizeof(T);      // dy row bytes (LDS image: see dy_row / dy_slot)
  constexpr int RBH = KCI * (int)sizeof(T);     // halo row bytes
  constexpr int PPD = RBD / 16, PPH = RBH / 16;
  constexpr int DINSTR = (TPIX * PPD + 63) / 64, HINSTR = (HPIX * PPH + 63) / 64;
  constexpr int DBUF = DINSTR * 1024, HBUF = HINSTR * 1024;
  constexpr int KSTEP_PIX = BF ? 32 : 4;
  constexpr int NKS = TPIX / KSTEP_PIX;
  // Separate LDS objects per pipeline stage (not one array indexed by a counter): the compiler then
  // knows that the DMA filling one stage cannot alias the reads of another and leaves it in flight
  // during the multiply instead of waiting for it before the first LDS read.  Three stages where
  // they fit (the narrow, bandwidth-bound shapes): two tiles in flight behind the one being used.
  constexpr int STAGE = DBUF + HBUF;
  constexpr int S = (3 * STAGE <= 160 * 1024) ? 3 : 2;
  __shared__ __attribute__((aligned(1024))) char buf0[STAGE];
  __shared__ __attribute__((aligned(1024))) char buf1[STAGE];
  __shared__ __attribute__((aligned(1024))) char buf2[S == 3 ? STAGE : 1024];
  // every wave issues the same number of DMA instructions per tile (the surplus ones repeat the
  // last piece), so "all but the newest tile" is a fixed vmcnt
  constexpr int DPW = (DINSTR + 7) / 8, HPW = (HINSTR + 7) / 8, NPW = DPW + HPW;
  static_assert(2 * NPW < 64, "vmcnt range");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % NWN, wk = wave / NWN;
  // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  The (ci chunk, co
  // block) workgroups of one pixel-tile list share every dy tile and halo, so they are given
  // linear ids that differ by multiples of 8: same XCD, dispatched back to back, and the
  // re-reads hit that XCD's L2 instead of each of the eight L2s fetching the tile for itself.
  // XCD x (workgroups lin = x, x+8, ...) walks the x-th contiguous eighth of the list ordered (pixel split, chunk,
  // block), the pixel splits in the order 0, 8, 16, ..., 1, 9, 17, ...: with a multiple of 8 splits XCD x has the
  // splits x, x+8, ..., so the tiles in work across the chip at one time are neighbours in memory; with any other
  // count a split may straddle two XCDs.  The grid is padded to a multiple of 8; the surplus workgroups leave.
  const int gxy = a.gx * a.gy;
  const int lin = blockIdx.x;
  const int pos = (lin & 7) * (int)(gridDim.x >> 3) + (lin >> 3);
  if (pos >= gxy * a.nsplit) return;
  const int cxy = pos % gxy;
  int bz;
  {
    const int sidx = pos / gxy, nfull = a.nsplit >> 3, rem = a.nsplit & 7;
    const int head = rem * (nfull + 1);          // splits in the residue classes that hold one more
    if (sidx < head) {
      bz = sidx / (nfull + 1) + 8 * (sidx % (nfull + 1));
    } else {
      const int s2 = sidx - head;                // (nfull > 0 here: head == nsplit otherwise)
      bz = rem + s2 / nfull + 8 * (s2 % nfull);
    }
  }
  // (the shorter of the two axes runs fastest: the ~32 workgroups an XCD has in flight then form the squarest patch
  // of the (chunk, block) plane, i.e. share the fewest distinct dy tiles and halos)
  const bool co_fast = a.gy < a.gx;
  const int ci0 = (co_fast ? cxy / a.gy : cxy % a.gx) * KCI;
  const int co0 = (co_fast ? cxy % a.gy : cxy / a.gx) * BM;
  // the input may be the channel concatenation [in | in2] of two tensors (never stored as one): a
  // workgroup's chunk lies in one of them (Ci1 is a multiple of the chunk)
  const T* in = reinterpret_cast<const T*>(a.in);
  int ldin = a.ldin, sci0 = ci0, sCi = a.Ci;
  if (a.in2 != nullptr) {
    if (ci0 >= a.Ci1) {
      in = reinterpret_cast<const T*>(a.in2);
      ldin = a.ldin2;
      sci0 = ci0 - a.Ci1;
      sCi = a.Ci - a.Ci1;
    } else {
      sCi = a.Ci1;
    }
  }
  const T* dy = reinterpret_cast<const T*>(a.dy);
  const T* zero = reinterpret_cast<const T*>(wgrad_zero_page);

  f32x4 acc[MT][NACC + NB];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NACC + NB; ++t) {
      acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      // (opaque to the optimiser: knowing the zeros, it peels the first tile with accumulators that are
      // not updated in place and runs out of registers in the 64-channel 3x3 form)
      asm volatile("" : "+v"(acc[m][t]));
    }
  const bool want_bias = NB > 0 && a.bias_slab != nullptr;

  const int g = lane >> 4, li = lane & 15, q = li >> 2, p4 = li & 3;

  // DMA source tables: what depends on the lane and the piece but not on the tile, two registers
  // a piece.  off = element offset from the tile's (halo's) first pixel and the workgroup's first
  // channel; yx = the pixel's row and column in the tile (halo), or a row no image has when the
  // piece's lanes lie past the tile or past the channel count.  (Kept this small on purpose: left
  // to itself the compiler keeps a 64-bit address and more per piece, which spills next to the
  // accumulators, and a spill reload beside the DMA waits for vmcnt(0).)
  int d_off[DPW], d_yx[DPW], h_off[HPW], h_yx[HPW];
#pragma unroll
  for (int k = 0; k < DPW; ++k) {                 // dy tile: [pixel][BM channels]
    int i = wave + 8 * k;
    if (i >= DINSTR) i = DINSTR - 1;
    const int part = i * 64 + lane;
    const int row = part / PPD, rslot = part - row * PPD;          // position in the LDS image
    const int pix = BF ? dy_pix<RBD>(row) : row;
    const int slot = BF ? dy_slot<RBD>(pix, rslot) : rslot;
    const int tr = pix / TW, tc = pix - tr * TW;
    d_off[k] = (tr * a.W + tc) * a.lddy + slot * EPP;
    d_yx[k] = (pix < TPIX && co0 + slot * EPP < a.Co) ? ((tr << 16) | tc) : (0x7fff << 16);
  }
#pragma unroll
  for (int k = 0; k < HPW; ++k) {                 // input halo: [(TR+2)*(TW+2) pixels][KCI channels]
    int i = wave + 8 * k;
    if (i >= HINSTR) i = HINSTR - 1;
    const int part = i * 64 + lane;
    const int pix = part / PPH, slot = part - pix * PPH;
    const int hy = pix / HW_, hx = pix - hy * HW_;
    const int cs = hswz<RBH>(hx, slot) * EPP;
    h_off[k] = (hy * a.W + hx) * ldin + cs;
    h_yx[k] = (pix < HPIX && sci0 + cs < sCi) ? ((hy << 16) | hx) : (0x7fff << 16);
  }
  auto issue_tile = [&](int tile, char* dbuf) {
    int t = tile;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int y0 = ty * TR, x0 = tx * TW;
    const int ph = (KS == 3) ? 1 : 0;             // halo origin = tile origin - ph
    char* hbuf = dbuf + DBUF;
    // element offsets of the tile's / halo's first pixel (the host checks that they fit 31 bits);
    // added to the lane's offset in 32 bits
    const int d0 = ((b * a.H + y0) * a.W + x0) * a.lddy + co0;
    const int h0 = ((b * a.H + y0 - ph) * a.W + x0 - ph) * ldin + sci0;
#pragma unroll
    for (int k = 0; k < DPW; ++k) {
      int i = wave + 8 * k;
      if (i >= DINSTR) i = DINSTR - 1;
      const bool ok = y0 + (d_yx[k] >> 16) < a.H && x0 + (d_yx[k] & 0xffff) < a.W;
      glds16(ok ? dy + (unsigned)(d0 + d_off[k]) : zero, dbuf + i * 1024);
    }
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
      int i = wave + 8 * k;
      if (i >= HINSTR) i = HINSTR - 1;
      const bool ok = (unsigned)(y0 - ph + (h_yx[k] >> 16)) < (unsigned)a.H &&
                      (unsigned)(x0 - ph + (h_yx[k] & 0xffff)) < (unsigned)a.W;
      glds16(ok ? in + (unsigned)(h0 + h_off[k]) : zero, hbuf + i * 1024);
    }
  };

  auto multiply = [&](const char* lds_dy) {
    const char* lds_h = lds_dy + DBUF;
    if constexpr (BF) {
      // k-step = 32 consecutive tile pixels; this wave takes steps wk, wk+NWK, ...: NST of them per
      // tile.  A k-step falls into sub-steps (one kernel row, or one channel slice in the sliced
      // 1x1 mode) of KS*MT MFMAs each.  Fragments are double-buffered in registers: while a
      // sub-step multiplies, the transposing reads of the next one (and, at a k-step boundary, the
      // next dy fragments) are issued between its MFMAs, so only the first sub-step of a tile
      // waits for an LDS round trip.
      constexpr int NST = NKS / NWK, SUB = KS * NSL, NU = NST * SUB, NTR = (KS == 3) ? 3 : 2;
      static_assert(NKS % NWK == 0, "k-steps divide evenly over the K waves");
      bf16x8 af[MT];
      u32x2 t[2][NTR];
      // Lane addresses of the reads of this wave's first k-step and kernel row 0; every other read
      // of the tile is one of them plus a constant.
      const int pkb = wk * 32 + 8 * g + q;                  // the pixel this lane addresses in block kk=0
      const int rb = pkb / TW, cb = pkb - rb * TW;
      int a_off[MT][2], t_off[NSL][NTR];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) {                       // block kk=1 is four pixels further on
          const int pk = pkb + 4 * h;
          a_off[m][h] = dy_row<RBD>(pk) * RBD + dy_slot<RBD>(pk, 2 * m + (p4 >> 1)) * 16 + (p4 & 1) * 8;
        }
      // A lane's eight k-slots are eight consecutive pixels of one image row, so the three
      // horizontal taps are the 8-pixel windows at offsets 0, 1, 2 of a 12-pixel run: three
      // transposing reads per kernel row, the windows cut out of them in registers.
#pragma unroll
      for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int i = 0; i < NTR; ++i) {
          const int hx = cb + 4 * i;
          t_off[sl][i] = (rb * HW_ + hx) * RBH + hswz<RBH>(hx, (wn * NSL + sl) * 2 + (p4 >> 1)) * 16 + (p4 & 1) * 8;
        }
      auto load_a = [&](int st, int m) __attribute__((always_inline)) {
        const int adv = st * NWK * 32 * RBD;                // k-steps are whole 16-pixel blocks apart
        af[m] = tr_read8(lds_dy + a_off[m][0] + adv, lds_dy + a_off[m][1] + adv);
      };
      auto load_t = [&](int buf, int u) __attribute__((always_inline)) {
        const int st = u / SUB, ky = (u % SUB) / NSL, sl = u % NSL;
        const int adv = ((st * NWK * 32 / TW + ky) * HW_) * RBH;
#pragma unroll
        for (int i = 0; i < NTR; ++i)
          t[buf][i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                    LDS_PTR(short4v, lds_h + t_off[sl][i] + adv)));
      };
#pragma unroll
      for (int m = 0; m < MT; ++m) load_a(0, m);
      load_t(0, 0);
      static_for<NU>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        constexpr int st = u / SUB, ky = (u % SUB) / NSL, sl = u % NSL, cur = u & 1;
        // While this sub-step multiplies, the window reads of the next one go out behind its first
        // MFMAs.  The dy fragments have one register set: in the k-step's last sub-step each is
        // reloaded for the next k-step right behind its last MFMA; the MFMAs run row block by row
        // block, so a reloaded fragment is first needed a sub-step's worth of MFMAs later.
        constexpr bool more = u + 1 < NU, reload = (u % SUB == SUB - 1) && st + 1 < NST;
        if constexpr (more) load_t(cur ^ 1, u + 1);
        u32x4 win[KS];
        win[0] = u32x4{t[cur][0][0], t[cur][0][1], t[cur][1][0], t[cur][1][1]};
        if constexpr (KS == 3) {
          win[1] = u32x4{__builtin_amdgcn_alignbit(t[cur][0][1], t[cur][0][0], 16),
                         __builtin_amdgcn_alignbit(t[cur][1][0], t[cur][0][1], 16),
                         __builtin_amdgcn_alignbit(t[cur][1][1], t[cur][1][0], 16),
                         __builtin_amdgcn_alignbit(t[cur][2][0], t[cur][1][1], 16)};
          win[2] = u32x4{t[cur][0][1], t[cur][1][0], t[cur][1][1], t[cur][2][0]};
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int kx = 0; kx < KS; ++kx)
            acc[m][(ky * KS + kx) * NSL + sl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                af[m], __builtin_bit_cast(bf16x8, win[kx]), acc[m][(ky * KS + kx) * NSL + sl], 0, 0, 0);
          if constexpr (NB > 0 && u % SUB == 0) {
            if (want_bias) {      // (uniform) every column of the tile = sum over the k-step's pixels
              const u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
              acc[m][NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], __builtin_bit_cast(bf16x8, ones),
                                                                     acc[m][NACC], 0, 0, 0);
            }
          }
          if constexpr (reload) load_a(st + 1, m);
        }
        // issue order: first MFMA, window reads, rest of the row block, its reload, next row block ...
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (more) __builtin_amdgcn_sched_group_barrier(0x100, NTR, 0);
        if constexpr (reload) {
          if constexpr (KS > 1) __builtin_amdgcn_sched_group_barrier(0x008, KS - 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          static_for<MT - 1>([&](auto) __attribute__((always_inline)) {
            __builtin_amdgcn_sched_group_barrier(0x008, KS, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          });
        } else {
          if constexpr (KS * MT > 1) __builtin_amdgcn_sched_group_barrier(0x008, KS * MT - 1, 0);
        }
      });
    } else {
      // k-step = 4 consecutive tile pixels; A[m=lane&15][k=lane>>4], B[k=lane>>4][n=lane&15]
#pragma unroll 1
      for (int ks = wk; ks < NKS; ks += NWK) {
        const int pk = ks * 4 + g;
        const int r0 = pk / TW, c0_ = pk - r0 * TW;
        float af[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
          af[m] = *reinterpret_cast<const float*>(lds_dy + pk * RBD + (m * 16 + li) * 4);
        const int byte = (wn * 16 + li) * 4;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            const int hr = (r0 + ky) * HW_ + c0_ + kx;
            const float bv = *reinterpret_cast<const float*>(lds_h + hr * RBH + hswz<RBH>(c0_ + kx, byte >> 4) * 16 +
                                                             (byte & 15));
#pragma unroll
            for (int m = 0; m < MT; ++m)
              acc[m][ky * KS + kx] =
                  __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bv, acc[m][ky * KS + kx], 0, 0, 0);
          }
      }
    }
  };

  const int n = bz < a.ntiles ? (a.ntiles - bz + a.nsplit - 1) / a.nsplit : 0;   // tiles of this workgroup
  auto tile_at = [&](int j) { return bz + j * a.nsplit; };
  if constexpr (S == 2) {
    // tile j+1 is in flight while tile j is multiplied; one barrier per tile
    if (n > 0) issue_tile(tile_at(0), buf0);
#pragma unroll 1
    for (int j = 0; j < n; j += 2) {
      __syncthreads();     // buf0 has landed for every wave; everyone is done reading buf1
      if (j + 1 < n) issue_tile(tile_at(j + 1), buf1);
      multiply(buf0);
      if (j + 1 >= n) break;
      __syncthreads();
      if (j + 2 < n) issue_tile(tile_at(j + 2), buf0);
      multiply(buf1);
    }
  } else {
    // Tiles j+1 and j+2 are in flight while tile j is multiplied.  Before the barrier a wave waits
    // for its own pieces of tile j only: vector-memory loads complete in order, so that is "all
    // but the NPW newest" whenever tile j+1 has been issued.  After the barrier tile j is complete
    // and every wave is done with tile j-1, whose stage the next issue overwrites.
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

  // Sum the k-residues of the workgroup through LDS (fixed order: residue w + half into w, halving),
  // one 16-channel row block at a time so the exchange area is 4 waves x 9 KiB at most.
  constexpr int NACCX = NACC + NB;
  static_assert(4 * NACCX * 4 * 64 * 4 <= STAGE, "exchange area");
  static_assert(BF || NSL == 1, "channel slices per wave: bf16 only");
  float* xch = reinterpret_cast<float*>(buf0);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int half = NWK / 2; half >= 1; half >>= 1) {
      __syncthreads();   // (first trip: every wave is done with the last tile)
      if (wk >= half && wk < 2 * half) {
        float* dst = xch + ((wk - half) * NWN + wn) * (NACCX * 4 * 64) + lane;
#pragma unroll
        for (int t = 0; t < NACCX; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[(t * 4 + r) * 64] = acc[m][t][r];
      }
      __syncthreads();
      if (wk < half) {
        const float* src = xch + (wk * NWN + wn) * (NACCX * 4 * 64) + lane;
#pragma unroll
        for (int t = 0; t < NACCX; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[m][t][r] += src[(t * 4 + r) * 64];
      }
    }
  }
  if (wk != 0) return;

  // slab[z][co][tap][ci]; C layout: col (ci) = lane&15, row (co) = 4*(lane>>4)+r
  float* slab = a.slab + (long)bz * a.Co_pad * TAPS * a.Ci_pad;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + m * 16 + 4 * g + r;
          const int ci = ci0 + (wn * NSL + sl) * 16 + li;
          slab[((long)co * TAPS + t) * a.Ci_pad + ci] = acc[m][t * NSL + sl][r];
        }
  if constexpr (NB > 0) {
    // the workgroups of the first input-channel chunk also hand over the column sums of dy
    if (want_bias && ci0 == 0 && wn == 0 && li == 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) a.bias_slab[(long)bz * a.Co_pad + co0 + m * 16 + 4 * g + r] = acc[m][NACC][r];
    }
  }
}

// dw[co][ci][tap] (+)= sum_s slab[s][co][tap][ci].  One workgroup per (block of R output channels,
// 32-channel chunk), all taps: R*taps*32 elements (R = 8 for 1x1, 1 for 3x3).  Each thread sums one
// element over the slabs in a fixed order (bitwise reproducible) reading 128-byte runs along ci, the
// tile turns in LDS, and the results leave as one contiguous run of dw per output channel.
constexpr int WR_ROWS3 = 4;                      // output channels per workgroup, 3x3
constexpr int WR_MANY_SLABS = 32;                // from this many slabs on (and a dw of at most 2^20 elements) the slab-lane reduction runs instead
constexpr int WR_MAX = WR_ROWS3 * 288;           // elements per workgroup: rows x (9 taps x 32 channels), or 8 rows x 32 channels
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab,
                                                           float* __restrict__ dw, int nslab, int Co, int Ci,
                                                           int Co_pad, int Ci_pad, int taps,
                                                           int accumulate, int nmain,
                                                           const float* __restrict__ bias_slab,
                                                           float* __restrict__ db) {
  __shared__ __attribute__((aligned(16))) float tile[WR_MAX];
  if ((int)blockIdx.x >= nmain) {
    // trailing workgroups: db[co] (+)= sum_s bias_slab[s][co], slabs in a fixed order
    const int co = (blockIdx.x - nmain) * 256 + threadIdx.x;
    if (co < Co) {
      float s = 0.f;
      for (int i = 0; i < nslab; ++i) s += bias_slab[(long)i * Co_pad + co];
      db[co] = accumulate ? db[co] + s : s;
    }
    return;
  }
  const int R = taps == 1 ? 8 : WR_ROWS3;
  const int nchunk = Ci_pad / 32;
  const int chunk = blockIdx.x % nchunk;
  const int co0 = (blockIdx.x / nchunk) * R;
  const int per_row = taps * 32, nel = R * per_row;
  const long stride = (long)Co_pad * taps * Ci_pad;
  // four consecutive input channels per thread (16-byte loads), four slabs in flight, summed in slab order
  for (int e = threadIdx.x * 4; e < nel; e += 1024) {
    const int row = e / per_row, rem = e - row * per_row;
    const int tap = rem >> 5, cl = rem & 31;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (co0 + row < Co) {
      const float* p = slab + ((long)(co0 + row) * taps + tap) * Ci_pad + chunk * 32 + cl;
      int i = 0;
      // (a narrow layer at full resolution has a handful of these workgroups and hundreds of slabs: its time is the
      //  chain of load latencies, so sixteen slabs are in flight while there are that many)
      for (; i + 16 <= nslab; i += 16) {
        f32x4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const f32x4*>(p + (i + j) * stride);
#pragma unroll
        for (int j = 0; j < 16; ++j) s = s + v[j];
      }
      for (; i + 4 <= nslab; i += 4) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + i * stride), v1 = *reinterpret_cast<const f32x4*>(p + (i + 1) * stride);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (i + 2) * stride), v3 = *reinterpret_cast<const f32x4*>(p + (i + 3) * stride);
        s = (((s + v0) + v1) + v2) + v3;
      }
      for (; i < nslab; ++i) s += *reinterpret_cast<const f32x4*>(p + i * stride);
    }
    *reinterpret_cast<f32x4*>(tile + e) = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nel; e += 256) {
    const int row = e / per_row, rem = e - row * per_row;
    const int cl = rem / taps, tap = rem - cl * taps;
    const int co = co0 + row, ci = chunk * 32 + cl;
    if (co < Co && ci < Ci) {
      float* d = dw + ((long)co * Ci + ci) * taps + tap;
      const float tot = tile[row * per_row + tap * 32 + cl];
      *d = accumulate ? (*d + tot) : tot;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Whole-pixel weight gradient (bf16, 3x3, exactly 48 output channels, 48 / 96+48 / 144 dense input
// channels): the full-resolution layers of the UNet, whose reduction runs over millions of pixels for
// a small dw.  They are bound by bytes: the generic kernel above reads dy once per 32-channel input
// chunk and gathers 64-byte runs a pixel stride apart (~100 cycles of the CU's address path per
// 1 KiB DMA piece, against ~33 for a piece whose lanes cover 1 KiB of consecutive memory in any
// order).  Here one workgroup owns ALL of dw and streams every pixel exactly once:
//  * pixels are staged whole, so a tile row of dy (32 pixels x 96 bytes) and a halo row of an input
//    (34 pixels x 96 / 192 / 288 bytes) are single runs and the LDS images are byte-for-byte copies;
//  * the eight waves do not split the reduction: a wave owns up to four UNITS = (16-channel input
//    slice, kernel row), nine accumulator tiles each (3 row blocks x 3 horizontal taps cut out of three
//    transposing reads, as above), takes every k-step of every tile, and writes its part of the fp32
//    slab itself - nothing is summed through LDS;
//  * workgroup w takes the 4-row x 32-column tiles w, w + nwg, ... in image order: at any moment the chip
//    streams one contiguous band of the image (DRAM pages and TLB entries are shared), and the tiles
//    above and below one another, which share two halo rows, run at the same time on one XCD (tile ids a
//    multiple of 8 apart when the image is 256 wide) and share its L2.
// With 96 output channels (NCO = 2: the half-resolution layers) dw no longer fits one workgroup: COLUMNS of
// workgroups = (48-channel half of the output) x (group of NS1 input slices of one operand) walk the same
// tile lists; a column stages whole dy pixels (192 bytes) and the pixels of its input group (whole when the
// group is the operand's full width, runs of NS1 * 32 bytes otherwise), and the columns of one tile list
// get workgroup ids a multiple of 8 apart: one XCD, one L2.
template <int NS1, int NS2, int NKG, int NCO = 1>
__global__ __launch_bounds__(512) void conv_wgrad_px_kernel(WgradArgs a) {
  using T = bf16;
  constexpr int MT = 3, TW = 32, TR = 4, TPIX = TW * TR, HW_ = TW + 2, HH_ = TR + 2, HPIX = HW_ * HH_;
  constexpr int RBD = NCO * MT * 32;                   // dy row bytes = the whole pixel
  static_assert(NCO == 1 || NS2 == 0, "columns take their slices from one operand");
  constexpr int P1 = NS1 * 32, P2 = NS2 * 32;          // pixel pitch of the two input images
  constexpr int DINSTR = TPIX * RBD / 1024;
  constexpr int H1INSTR = (HPIX * P1 + 1023) / 1024, H2INSTR = (HPIX * P2 + 1023) / 1024;
  constexpr int DBUF = DINSTR * 1024, H1BUF = H1INSTR * 1024, H2BUF = H2INSTR * 1024;
  constexpr int STAGE = DBUF + H1BUF + H2BUF;
  constexpr int S = (3 * STAGE <= 156 * 1024) ? 3 : 2;
  constexpr int NINSTR = DINSTR + H1INSTR + H2INSTR, NPW = (NINSTR + 7) / 8;
  // NKG groups of 8 / NKG waves split the tile's k-steps (few units: nine units on eight waves would leave
  // seven of them idle half of the time); each group writes a slab of its own
  constexpr int WPG = 8 / NKG, KPG = TR / NKG;
  constexpr int NUNITS = (NS1 + NS2) * 3, UPW = (NUNITS + WPG - 1) / WPG;
  static_assert(TPIX * RBD % 1024 == 0 && 2 * STAGE <= 159 * 1024 && 2 * NPW < 64 && TR % NKG == 0, "stage layout");
  __shared__ __attribute__((aligned(1024))) char buf0[STAGE];
  __shared__ __attribute__((aligned(1024))) char buf1[STAGE];
  __shared__ __attribute__((aligned(1024))) char buf2[S == 3 ? STAGE : 1024];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T* dy = reinterpret_cast<const T*>(a.dy);
  const T* in1 = reinterpret_cast<const T*>(a.in);
  const T* in2 = reinterpret_cast<const T*>(a.in2);
  const T* zero = reinterpret_cast<const T*>(wgrad_zero_page);

  // column (see above) and tile list of this workgroup
  const int ncol = NCO * (a.px_ngrp1 + a.px_ngrp2);
  const int nwg = gridDim.x / ncol;
  const int col = ncol > 1 ? (blockIdx.x >> 3) % ncol : 0;
  const int pos = ncol > 1 ? ((blockIdx.x >> 3) / ncol) * 8 + (blockIdx.x & 7) : blockIdx.x;
  const int half = col % NCO, grp = col / NCO;
  int ld1 = a.ldin, coff = grp * NS1 * 16, ci_base = coff;
  if (NS2 == 0 && grp >= a.px_ngrp1) {            // this column's slices lie in the second operand
    in1 = in2;
    ld1 = a.ldin2;
    coff = (grp - a.px_ngrp1) * NS1 * 16;
    ci_base = a.Ci1 + coff;
  }
  const int n = pos < a.ntiles ? (a.ntiles - pos + nwg - 1) / nwg : 0;      // tiles of this workgroup

  // ---- DMA tables: piece i = wave + 8k covers bytes [1024 i, 1024 i + 1024) of the stage (dy image, then
  // the halo images); per piece an element offset from the tile's first (halo) pixel and the pixel's
  // (row, column), or a row no image has
  int p_off[NPW], p_yx[NPW];
#pragma unroll
  for (int k = 0; k < NPW; ++k) {
    int i = wave + 8 * k;
    if (i >= NINSTR) i = NINSTR - 1;
    if (i < DINSTR) {
      const int o = i * 1024 + lane * 16;
      const int pix = o / RBD, byte = o - pix * RBD;
      const int tr = pix / TW, tc = pix - tr * TW;
      p_off[k] = (tr * a.W + tc) * a.lddy + (byte >> 1);
      p_yx[k] = (tr << 16) | tc;
    } else {
      const bool second = NS2 > 0 && i >= DINSTR + H1INSTR;
      const int o = (i - DINSTR - (second ? H1INSTR : 0)) * 1024 + lane * 16;
      const int pitch = second ? P2 : P1;
      const int hp = o / pitch, byte = o - hp * pitch;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      p_off[k] = (hy * a.W + hx) * (second ? a.ldin2 : ld1) + (second ? 0 : coff) + (byte >> 1);
      p_yx[k] = hp < HPIX ? ((hy << 16) | hx) : (0x7fff << 16);
    }
  }
  auto issue_tile = [&](int tile, char* dbuf) {
    const int tx = tile % a.tiles_x;
    const int t2 = tile / a.tiles_x;
    const int ty = t2 % a.tiles_y;
    const int b = t2 / a.tiles_y;
    const int y0 = ty * TR, x0 = tx * TW;
    const int d0 = ((b * a.H + y0) * a.W + x0) * a.lddy;
    const int h1 = ((b * a.H + y0 - 1) * a.W + x0 - 1) * ld1;
    const int h2 = NS2 > 0 ? ((b * a.H + y0 - 1) * a.W + x0 - 1) * a.ldin2 : 0;
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
      int i = wave + 8 * k;
      if (i >= NINSTR) i = NINSTR - 1;
      const int py = p_yx[k] >> 16, px = p_yx[k] & 0xffff;
      const T* src;
      if (i < DINSTR) {
        src = (y0 + py < a.H && x0 + px < a.W) ? dy + (unsigned)(d0 + p_off[k]) : zero;
      } else {
        const bool ok = (unsigned)(y0 - 1 + py) < (unsigned)a.H && (unsigned)(x0 - 1 + px) < (unsigned)a.W;
        const bool second = NS2 > 0 && i >= DINSTR + H1INSTR;
        src = ok ? (second ? in2 + (unsigned)(h2 + p_off[k]) : in1 + (unsigned)(h1 + p_off[k])) : zero;
      }
      glds16(src, dbuf + i * 1024);
    }
  };

  // ---- this wave's units and fragment addresses
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p4 = li & 3;
  int a_off[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int h = 0; h < 2; ++h)
      a_off[m][h] = ((wave / WPG) * KPG * TW + 8 * g + q + 4 * h) * RBD + half * (MT * 32) + (2 * m + (p4 >> 1)) * 16 +
                    (p4 & 1) * 8;
  int t_off[UPW], rowstep[UPW], pixstep[UPW];
  bool uvalid[UPW];
  const int kg = wave / WPG, wu = wave % WPG;
#pragma unroll
  for (int j = 0; j < UPW; ++j) {
    const int u = wu + WPG * j;
    uvalid[j] = u < NUNITS;
    const int uc = uvalid[j] ? u : NUNITS - 1;
    const int slice = uc / 3, ky = uc - 3 * slice;
    const bool second = NS2 > 0 && slice >= NS1;
    const int pitch = second ? P2 : P1;
    const int sl = second ? slice - NS1 : slice;
    t_off[j] = DBUF + (second ? H1BUF : 0) + ((kg * KPG + ky) * HW_ + 8 * g + q) * pitch + (sl * 2 + (p4 >> 1)) * 16 +
               (p4 & 1) * 8;
    rowstep[j] = HW_ * pitch;
    pixstep[j] = 4 * pitch;
  }

  f32x4 acc[UPW][MT][3];
#pragma unroll
  for (int j = 0; j < UPW; ++j)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        acc[j][m][kx] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("" : "+v"(acc[j][m][kx]));      // (see the generic kernel)
      }

  auto multiply = [&](const char* lds) {
    bf16x8 af[MT];
    u32x2 t[2][3];
    auto load_a = [&](int st, int m) __attribute__((always_inline)) {
      af[m] = tr_read8(lds + a_off[m][0] + st * TW * RBD, lds + a_off[m][1] + st * TW * RBD);
    };
    auto load_t = [&](int buf, int st, int j) __attribute__((always_inline)) {
      const char* p = lds + t_off[j] + st * rowstep[j];
#pragma unroll
      for (int i = 0; i < 3; ++i)
        t[buf][i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, p + i * pixstep[j])));
    };
#pragma unroll
    for (int m = 0; m < MT; ++m) load_a(0, m);
    load_t(0, 0, 0);
    static_for<KPG * UPW>([&](auto uc) __attribute__((always_inline)) {
      constexpr int us = decltype(uc)::value;
      constexpr int st = us / UPW, j = us % UPW, cur = us & 1;
      constexpr bool more = us + 1 < KPG * UPW;
      constexpr bool reload = (j == UPW - 1) && st + 1 < KPG;
      if constexpr (more) load_t(cur ^ 1, (us + 1) / UPW, (us + 1) % UPW);
      u32x4 win[3];
      win[0] = u32x4{t[cur][0][0], t[cur][0][1], t[cur][1][0], t[cur][1][1]};
      win[1] = u32x4{__builtin_amdgcn_alignbit(t[cur][0][1], t[cur][0][0], 16),
                     __builtin_amdgcn_alignbit(t[cur][1][0], t[cur][0][1], 16),
                     __builtin_amdgcn_alignbit(t[cur][1][1], t[cur][1][0], 16),
                     __builtin_amdgcn_alignbit(t[cur][2][0], t[cur][1][1], 16)};
      win[2] = u32x4{t[cur][0][1], t[cur][1][0], t[cur][1][1], t[cur][2][0]};
      // (a unit past the wave's last one multiplies into accumulators nobody stores: every wave runs
      // the same instruction stream, no branch around the reads in flight)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          acc[j][m][kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], __builtin_bit_cast(bf16x8, win[kx]),
                                                                  acc[j][m][kx], 0, 0, 0);
        if constexpr (reload) load_a(st + 1, m);
      }
      // issue order (as in the generic kernel): first MFMA, the next unit's window reads, the rest of the row
      // block, its dy reload, the next row block ...: no MFMA waits for a read issued in its own unit-step
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if constexpr (more) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      if constexpr (reload) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        static_for<MT - 1>([&](auto) __attribute__((always_inline)) {
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        });
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * MT - 1, 0);
      }
    });
  };

  auto tile_at = [&](int j) { return pos + j * nwg; };
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
      if (newer_in_flight) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
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

  // slab[pos][co][tap][ci]; C layout: col (ci) = lane&15, row (co) = 4*(lane>>4)+r
  float* slab = a.slab + ((long)pos * NKG + kg) * a.Co_pad * 9 * a.Ci_pad;
#pragma unroll
  for (int j = 0; j < UPW; ++j) {
    const int u = wu + WPG * j;
    if (u >= NUNITS) continue;
    const int slice = u / 3, ky = u - 3 * slice;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = half * (MT * 16) + m * 16 + 4 * g + r;
          slab[((long)co * 9 + ky * 3 + kx) * a.Ci_pad + ci_base + slice * 16 + li] = acc[j][m][kx][r];
        }
  }
}


// dw[co][ci][tap] (+)= sum_s slab[s][co][tap][ci] for MANY slabs of a small dw (the whole-pixel kernel
// leaves one slab per workgroup): a workgroup owns 16 quads of four consecutive input channels and 16
// slab lanes; lane j sums the slabs j, j+16, ... in order with four loads in flight, then the 16 lane sums
// are added in lane order: the same bits every run, and a dependency chain of nslab/64 loads
// instead of nslab/4.
__global__ __launch_bounds__(256) void wgrad_reduce_many_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                int nslab, int Co, int Ci, int Co_pad, int Ci_pad,
                                                                int taps, int accumulate) {
  __shared__ f32x4 part[16][16];
  const int e = threadIdx.x & 15, j = threadIdx.x >> 4;
  const int cq = Ci_pad / 4;
  const long nquads = (long)Co * taps * cq;
  const long idx = (long)blockIdx.x * 16 + e;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  if (idx < nquads) {
    const float* p = slab + idx * 4;                       // [co][tap][ci] is the slab's own order
    const long stride = (long)Co_pad * taps * Ci_pad;
    int i = j;
    for (; i + 48 < nslab; i += 64) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + i * stride);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (i + 16) * stride);
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (i + 32) * stride);
      const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (i + 48) * stride);
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; i < nslab; i += 16) s += *reinterpret_cast<const f32x4*>(p + i * stride);
  }
  part[j][e] = s;
  __syncthreads();
  if (j == 0 && idx < nquads) {
    f32x4 t = part[0][e];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += part[k][e];
    const int ci = (int)(idx % cq) * 4;
    const long ct = idx / cq;
    const int tap = (int)(ct % taps), co = (int)(ct / taps);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (ci + k < Ci) {
        float* d = dw + ((long)co * Ci + ci + k) * taps + tap;
        *d = accumulate ? (*d + t[k]) : t[k];
      }
  }
}

struct PxPlan {
  bool ok;
  int ns1, ns2, nkg, nco, ngrp1, ngrp2, ncol, tiles_x, tiles_y, ntiles, nwg, Ci_pad;
};

// The whole-pixel kernel takes 48 output channels and 48, 96 + 48 or 144 input channels in one workgroup, or 96
// output channels and 48, 96, 192 + 96 or 288 input channels in columns of workgroups.
inline PxPlan plan_px(int B, int H, int W, int Ci, int Co, int ks, int dtype, int Ci1) {
  PxPlan p;
  p.ok = false;
  p.nco = 1; p.ngrp1 = 1; p.ngrp2 = 0; p.ns2 = 0; p.nkg = 1;
  if (dtype != UNETK_BF16 || ks != 3) return p;
  if (Co == 48) {
    if (Ci1 == 0 && Ci == 48) { p.ns1 = 3; p.nkg = 2; }
    else if (Ci1 == 0 && Ci == 144) { p.ns1 = 9; }
    else if (Ci1 == 96 && Ci == 144) { p.ns1 = 6; p.ns2 = 3; }
    else return p;
  } else if (Co == 96 && getenv("UNETK_WGRAD_NO_PX96") == nullptr) {
    p.nco = 2;
    if (Ci1 == 0 && Ci == 48) { p.ns1 = 3; p.nkg = 2; }
    else if (Ci1 == 0 && (Ci == 96 || Ci == 288)) { p.ns1 = 6; p.ngrp1 = Ci / 96; }
    else if (Ci1 == 192 && Ci == 288) { p.ns1 = 6; p.ngrp1 = 2; p.ngrp2 = 1; }
    else return p;
  } else {
    return p;
  }
  p.ncol = p.nco * (p.ngrp1 + p.ngrp2);
  p.tiles_x = cdiv(W, 32);
  p.tiles_y = cdiv(H, 4);
  p.ntiles = B * p.tiles_x * p.tiles_y;
  if (p.ntiles < 256) return p;            // small images: the generic kernel's finer split fills the chip better
  p.nwg = (p.ntiles / 8) & ~7;             // tile lists: at least eight tiles each, one workgroup per CU at most
  const int most = (256 / p.ncol) & ~7;
  if (p.nwg > most) p.nwg = most;
  p.Ci_pad = cdiv(Ci, 32) * 32;
  p.ok = true;
  return p;
}
inline int64_t px_workspace_bytes(int B, int H, int W, int Ci, int Co, int ks, int dtype) {
  int64_t need = 0;
  for (int c1 : {0, 96, 192}) {
    const PxPlan p = plan_px(B, H, W, Ci, Co, ks, dtype, c1);
    const int64_t bytes = (int64_t)p.nwg * p.nkg * 48 * p.nco * 9 * p.Ci_pad * 4;
    if (p.ok && bytes > need) need = bytes;
  }
  return need;
}

struct WgradPlan {
  int mt, tw, nwn, nwk, tpix, tr, tiles_x, tiles_y, ntiles, nsplit, Co_pad, Ci_pad;
};

WgradPlan plan_wgrad(int B, int H, int W, int Ci, int Co, int dtype, int ks, bool chunk64 = true) {
  WgradPlan p;
  p.mt = (Co % 48 == 0) ? 3 : ((Co % 64 == 0 || Co > 48) ? 4 : (Co > 32 ? 3 : 2));
  // 3x3: 64-channel row blocks where they divide Co: fewer passes over the input halo per output,
  // and more MFMAs per fragment read (not at <= 16-pixel rows, where the wider tile measured slower)
  if (ks == 3 && Co % 64 == 0 && W > 16) p.mt = 4;
  p.tw = (W <= 16) ? 16 : 32;
  // 3x3: 32 input channels per workgroup (two waves across the channels, four along K); the
  // 64-channel form only for 1x1 shapes, where it wastes no padding or the shape is bandwidth-bound
  // (flops per operand byte = Co*Ci/(Co+Ci) small): fewer passes over dy for padded MFMA work.
  const bool low_intensity = 9.0 * Co * Ci / (Co + Ci) < 350.0;
  p.nwn = 2;
  if (ks == 1 && dtype == UNETK_BF16 &&
      (Ci % 64 == 0 || (Ci > 256 && (Ci % 64) > 32) || (low_intensity && Ci > 32)))
    p.nwn = 4;
  if (ks == 1 && dtype == UNETK_BF16 && Ci >= 64) p.nwn = 1;   // plain GEMM: 4 channel slices per wave
  // pieces per MFMA (chunk64 = false: a two-operand input whose first part is not whole 64-channel chunks)
  if (ks == 3 && dtype == UNETK_BF16 && Ci % 64 == 0 && chunk64) p.nwn = 4;
  if (ks == 1 && p.nwn == 1 && Co % 64 == 0) p.mt = 4;     // same for the sliced 1x1 form (64 x 64 outputs per workgroup)
  p.nwk = 8 / p.nwn;
  p.tpix = dtype == UNETK_BF16 ? 256 : 128;
  p.tr = p.tpix / p.tw;
  p.tiles_x = cdiv(W, p.tw);
  p.tiles_y = cdiv(H, p.tr);
  p.ntiles = B * p.tiles_x * p.tiles_y;
  p.Co_pad = cdiv(Co, p.mt * 16) * p.mt * 16;
  const int kci = p.nwn == 1 ? 64 : p.nwn * 16;
  p.Ci_pad = cdiv(Ci, kci) * kci;
  const int gxy = (p.Ci_pad / kci) * (p.Co_pad / (p.mt * 16));
  // Pixel split: one workgroup per CU at a time, so the grid should be a whole number of rounds of
  // 256 (and at least two, when there are tiles enough); every split costs one more partial slab
  // to write and read back.  Score = fill of the last round, discounted by the slab traffic
  // relative to the multiply (at ~700 TFLOP/s and ~0.9 TB/s: inside a training step the slabs are written and summed
  // beside the HBM-bound passes of the main stream - priced at the 3 TB/s they reach alone, the UNet step measured
  // 0.06...0.17 ms slower).
  const double t_mul = 2.0 * B * H * W * ks * ks * (double)p.Ci_pad * p.Co_pad / 700e12;
  const double t_slab = 8.0 * ks * ks * (double)p.Ci_pad * p.Co_pad / 0.9e12;
  int s = 1;
  double best = -1.0;
  for (int c = 1; c <= 256 && c <= p.ntiles; c += (c < 32 ? 1 : 8)) {
    const long wgs = (long)gxy * c;
    const long rounds = (wgs + 255) / 256;
    double score = (double)wgs / (256.0 * rounds) / (1.0 + c * t_slab / t_mul);
    if (rounds < 2) score *= 0.8;
    if (score > best) {
      best = score;
      s = c;
    }
  }
  p.nsplit = s;
  return p;
}

// A 1x1 weight gradient is a plain sum over pixels: any 2-D shape of the pixel axis will do, so the
// pixels are refolded into 32-pixel rows and every 256-pixel tile is full whatever the image size.
inline void refold_1x1(int& B, int& H, int& W, int ks) {
  const long npix = (long)B * H * W;
  if (ks == 1 && npix % 32 == 0 && npix / 32 < (1L << 30)) {
    B = 1;
    H = (int)(npix / 32);
    W = 32;
  }
}

template <typename T, int KS>
int launch_wgrad(const WgradArgs& a, const WgradPlan& p, hipStream_t stream) {
  dim3 grid((unsigned)(8 * cdiv(a.gx * a.gy * p.nsplit, 8)));
#define WG_CASE(MT_, TW_, NWN_)                                                                        \
  if (p.mt == MT_ && p.tw == TW_ && p.nwn == NWN_) {                                                   \
    hipLaunchKernelGGL((conv_wgrad_kernel<T, MT_, TW_, KS, NWN_>), grid, dim3(512), 0, stream, a);     \
    UNETK_CHECK_LAUNCH();                                                                              \
    return UNETK_OK;                                                                                   \
  }
  WG_CASE(2, 16, 2) WG_CASE(2, 32, 2) WG_CASE(3, 16, 2) WG_CASE(3, 32, 2) WG_CASE(4, 16, 2) WG_CASE(4, 32, 2)
  if constexpr (sizeof(T) == 2 && KS == 3) {   // 64-channel chunks
    WG_CASE(2, 32, 4) WG_CASE(3, 32, 4) WG_CASE(4, 32, 4) WG_CASE(3, 16, 4) WG_CASE(4, 16, 4)
  }
  if constexpr (sizeof(T) == 2 && KS == 1) {   // 64-channel chunks, sliced form: bf16 1x1 only
    WG_CASE(2, 16, 4) WG_CASE(2, 32, 4) WG_CASE(3, 16, 4) WG_CASE(3, 32, 4) WG_CASE(4, 16, 4) WG_CASE(4, 32, 4)
    {
      WG_CASE(2, 16, 1) WG_CASE(2, 32, 1) WG_CASE(3, 16, 1) WG_CASE(3, 32, 1) WG_CASE(4, 16, 1) WG_CASE(4, 32, 1)
    }
  }
#undef WG_CASE
  unetk_set_error("conv_wgrad: no kernel for mt=%d tw=%d nwn=%d", p.mt, p.tw, p.nwn);
  return UNETK_EINVAL;
}

}  // namespace

extern "C" {

int64_t unetk_conv_wgrad_workspace_bytes(int B, int H, int W, int Ci, int Co, int ks, int dtype) {
  refold_1x1(B, H, W, ks);
  int64_t need = px_workspace_bytes(B, H, W, Ci, Co, ks, dtype);
  for (int c64 = 0; c64 < 2; ++c64) {        // (which 3x3 form runs depends on the operands: room for either)
    const WgradPlan p = plan_wgrad(B, H, W, Ci, Co, dtype, ks, c64 != 0);
    const int64_t bytes = (int64_t)p.nsplit * p.Co_pad * ks * ks * p.Ci_pad * 4;
    if (bytes > need) need = bytes;
  }
  return need;
}

static int wgrad_impl(const void* in, int ldin, const void* dy, int lddy, float* dw, float* dbias, int B, int H, int W,
                      int Ci, int Co, int ks, int dtype, void* workspace, int64_t workspace_bytes,
                      int accumulate, hipStream_t stream, const void* in2 = nullptr, int ldin2 = 0, int Ci1 = 0) {
  UNETK_CHECK_ARG(in && dy && dw && workspace, "conv_wgrad: null pointer");
  UNETK_CHECK_ARG(ks == 1 || ks == 3, "conv_wgrad: kernel size %d (1 or 3)", ks);
  UNETK_CHECK_ARG(dtype == UNETK_F32 || dtype == UNETK_BF16, "conv_wgrad: bad dtype %d", dtype);
  const int epp = dtype == UNETK_BF16 ? 8 : 4;
  UNETK_CHECK_ARG(Ci % epp == 0 && Co % epp == 0 && ldin % epp == 0 && lddy % epp == 0,
                  "conv_wgrad: channels and strides must be multiples of %d", epp);
  UNETK_CHECK_ARG(((uintptr_t)in % 16) == 0 && ((uintptr_t)dy % 16) == 0, "conv_wgrad: misaligned pointer");
  UNETK_CHECK_ARG((int64_t)B * H * W * (ldin > lddy ? ldin : lddy) < (1LL << 31),
                  "conv_wgrad: tensors of 2^31 elements or more are not supported");
  if (dbias == nullptr && getenv("UNETK_WGRAD_NO_PX") == nullptr) {
    // full-resolution 48-channel layers with dense operands: every pixel streamed once by one workgroup
    const PxPlan pp = plan_px(B, H, W, Ci, Co, ks, dtype, in2 != nullptr ? Ci1 : 0);
    const bool dense = in2 != nullptr ? (ldin == Ci1 && ldin2 == Ci - Ci1 && ((uintptr_t)in2 % 16) == 0 &&
                                         (int64_t)B * H * W * ldin2 < (1LL << 31))
                                      : ldin == Ci;
    if (pp.ok && dense) {
      const int co_pad = 48 * pp.nco;
      const int64_t need_px = (int64_t)pp.nwg * pp.nkg * co_pad * 9 * pp.Ci_pad * 4;
      UNETK_CHECK_ARG(workspace_bytes >= need_px, "conv_wgrad: workspace %lld < %lld bytes", (long long)workspace_bytes,
                      (long long)need_px);
      WgradArgs a;
      a.in = in; a.in2 = in2; a.ldin2 = ldin2; a.Ci1 = Ci1; a.dy = dy; a.slab = (float*)workspace; a.bias_slab = nullptr;
      a.ldin = ldin; a.lddy = lddy; a.B = B; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.Ci_pad = pp.Ci_pad; a.Co_pad = co_pad;
      a.tiles_x = pp.tiles_x; a.tiles_y = pp.tiles_y; a.ntiles = pp.ntiles; a.nsplit = pp.nwg; a.gx = a.gy = 0;
      a.px_ngrp1 = pp.ngrp1; a.px_ngrp2 = pp.ngrp2;
      const dim3 grid((unsigned)(pp.nwg * pp.ncol));
      if (pp.nco == 2 && pp.ns1 == 3) hipLaunchKernelGGL((conv_wgrad_px_kernel<3, 0, 2, 2>), grid, dim3(512), 0, stream, a);
      else if (pp.nco == 2) hipLaunchKernelGGL((conv_wgrad_px_kernel<6, 0, 1, 2>), grid, dim3(512), 0, stream, a);
      else if (pp.ns1 == 3) hipLaunchKernelGGL((conv_wgrad_px_kernel<3, 0, 2>), grid, dim3(512), 0, stream, a);
      else if (pp.ns1 == 9) hipLaunchKernelGGL((conv_wgrad_px_kernel<9, 0, 1>), grid, dim3(512), 0, stream, a);
      else hipLaunchKernelGGL((conv_wgrad_px_kernel<6, 3, 1>), grid, dim3(512), 0, stream, a);
      UNETK_CHECK_LAUNCH();
      const long nblocks = cdiv((long)Co * 9 * (pp.Ci_pad / 4), 16);
      hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3((unsigned)nblocks), dim3(256), 0, stream, a.slab, dw,
                         pp.nwg * pp.nkg, Co, Ci, co_pad, pp.Ci_pad, 9, accumulate);
      UNETK_CHECK_LAUNCH();
      return UNETK_OK;
    }
  }
  refold_1x1(B, H, W, ks);
  const WgradPlan p = plan_wgrad(B, H, W, Ci, Co, dtype, ks, in2 == nullptr || Ci1 % 64 == 0);
  const int64_t main_bytes = (int64_t)p.nsplit * p.Co_pad * ks * ks * p.Ci_pad * 4;
  const int64_t need = main_bytes + (dbias ? (int64_t)p.nsplit * p.Co_pad * 4 : 0);
  UNETK_CHECK_ARG(workspace_bytes >= need, "conv_wgrad: workspace %lld < %lld bytes",
                  (long long)workspace_bytes, (long long)need);
  WgradArgs a;
  if (in2 != nullptr) {
    const int kci = p.nwn == 1 ? 64 : p.nwn * 16;
    UNETK_CHECK_ARG(ks == 3 && Ci1 > 0 && Ci1 < Ci && Ci1 % kci == 0 && ldin >= Ci1 && ldin2 >= Ci - Ci1 &&
                    ldin2 % epp == 0 && ((uintptr_t)in2 % 16) == 0 && (int64_t)B * H * W * ldin2 < (1LL << 31),
                    "conv_wgrad_cat: the first input must hold a multiple of %d channels (%d of %d), the second be aligned", kci, Ci1, Ci);
  }
  a.in2 = in2; a.ldin2 = ldin2; a.Ci1 = Ci1;
  a.in = in; a.dy = dy; a.slab = (float*)workspace; a.bias_slab = dbias ? (float*)((char*)workspace + main_bytes) : nullptr; a.ldin = ldin; a.lddy = lddy;
  a.B = B; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.Ci_pad = p.Ci_pad; a.Co_pad = p.Co_pad;
  a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.ntiles = p.ntiles; a.nsplit = p.nsplit;
  a.gx = p.Ci_pad / (p.nwn == 1 ? 64 : p.nwn * 16); a.gy = p.Co_pad / (p.mt * 16);
  int rc;
  if (dtype == UNETK_BF16)
    rc = ks == 3 ? launch_wgrad<bf16, 3>(a, p, stream) : launch_wgrad<bf16, 1>(a, p, stream);
  else
    rc = ks == 3 ? launch_wgrad<float, 3>(a, p, stream) : launch_wgrad<float, 1>(a, p, stream);
  if (rc != UNETK_OK) return rc;
  if (dbias == nullptr && p.nsplit >= WR_MANY_SLABS && (long)Co * Ci * ks * ks <= (1L << 20)) {
    // a small dw under many slabs (a narrow layer at full resolution): the slab-lane reduction, whose chain of
    // dependent loads is nslab/64 long instead of nslab/4 (same bits every run; the choice depends on the shape only)
    const long nb = cdiv((long)Co * ks * ks * (p.Ci_pad / 4), 16);
    hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a.slab, dw, p.nsplit, Co, Ci,
                       p.Co_pad, p.Ci_pad, ks * ks, accumulate);
    UNETK_CHECK_LAUNCH();
    return UNETK_OK;
  }
  const long nblocks = (long)cdiv(Co, ks == 1 ? 8 : WR_ROWS3) * (p.Ci_pad / 32);
  const long nbias = dbias ? cdiv(Co, 256) : 0;      // trailing workgroups sum the bias slabs
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(nblocks + nbias)), dim3(256), 0, stream, a.slab, dw,
                     p.nsplit, Co, Ci, p.Co_pad, p.Ci_pad, ks * ks, accumulate, (int)nblocks, a.bias_slab, dbias);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_conv_wgrad(const void* in, int ldin, const void* dy, int lddy, float* dw, int B, int H, int W,
                     int Ci, int Co, int ks, int dtype, void* workspace, int64_t workspace_bytes,
                     int accumulate, hipStream_t stream) {
  return wgrad_impl(in, ldin, dy, lddy, dw, nullptr, B, H, W, Ci, Co, ks, dtype, workspace, workspace_bytes,
                    accumulate, stream);
}

int unetk_conv_wgrad_cat(const void* in, int ldin, int Ci1, const void* in2, int ldin2, const void* dy, int lddy,
                         float* dw, int B, int H, int W, int Ci, int Co, int dtype, void* workspace,
                         int64_t workspace_bytes, int accumulate, hipStream_t stream) {
  UNETK_CHECK_ARG(in2 != nullptr, "conv_wgrad_cat: null second input");
  return wgrad_impl(in, ldin, dy, lddy, dw, nullptr, B, H, W, Ci, Co, 3, dtype, workspace, workspace_bytes, accumulate,
                    stream, in2, ldin2, Ci1);
}

int64_t unetk_linear_wgrad_workspace_bytes(int64_t npix, int Ci, int Co, int dtype) {
  int B = 1, H = 1, W = (int)npix;
  if (npix >= (1LL << 31)) return -1;
  refold_1x1(B, H, W, 1);
  const WgradPlan p = plan_wgrad(B, H, W, Ci, Co, dtype, 1);
  const int64_t generic = (int64_t)p.nsplit * p.Co_pad * p.Ci_pad * 4 + (int64_t)p.nsplit * p.Co_pad * 4;
  const int64_t rows = dtype == UNETK_BF16 ? lin_wgrad_rows_workspace(npix, Ci, Co) : 0;
  return generic > rows ? generic : rows;
}

int unetk_linear_wgrad(const void* in, int ldin, const void* dy, int lddy, float* dw, float* dbias, int64_t npix,
                       int Ci, int Co, int dtype, void* workspace, int64_t workspace_bytes, int accumulate,
                       hipStream_t stream) {
  UNETK_CHECK_ARG(dbias != nullptr, "linear_wgrad: null bias gradient (use conv_wgrad)");
  UNETK_CHECK_ARG(dtype == UNETK_BF16, "linear_wgrad: bfloat16 only (the bias sums ride on the bf16 MFMA path)");
  UNETK_CHECK_ARG(npix > 0 && npix < (1LL << 31), "linear_wgrad: bad token count");
  UNETK_CHECK_ARG(in && dy && dw && workspace && ((uintptr_t)in % 16) == 0 && ((uintptr_t)dy % 16) == 0 &&
                  ((uintptr_t)dw % 16) == 0, "linear_wgrad: null or misaligned pointer");
  int rc = UNETK_OK;
  if (lin_wgrad_rows_try(in, ldin, dy, lddy, dw, dbias, npix, Ci, Co, workspace, workspace_bytes, accumulate, stream, &rc))
    return rc;
  return wgrad_impl(in, ldin, dy, lddy, dw, dbias, 1, 1, (int)npix, Ci, Co, 1, dtype, workspace, workspace_bytes,
                    accumulate, stream);
}

}  // extern "C"
