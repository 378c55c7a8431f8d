// The row-gather implicit-GEMM tile of conv_mfma.hip as templates: the device code (conv_igemm_tile, its kernels, the grouped forms,
// split-K finalize) and the launch templates.  Two translation units instantiate them: conv_mfma.hip the tiles whose K loop the
// compiler schedules (and all host-side planning), conv_mfma4w.hip the four-wave tiles with the assembly K loop of kloop4w.inc -
// built side by side, the conv kernels are two thirds of the library's build time.
#pragma once
#include <algorithm>

#include "conv_device.h"
#include "kloop4w.inc"

namespace ron {
namespace detail {

// all of this wave's LDS reads retired (the stage about to be refilled is no longer being read) and all but
// its N youngest LDS-DMA transfers landed
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// Issue order of the k-steps s, s+1, ... of one stage (see the K loop): per k-step its MFMAs, the next k-step's RD fragment
// reads one per MFMA gap, its share of the stage's LPT LDS-DMA pieces spaced evenly between them.  Everything is a
// compile-time constant (the builtin wants immediates); the split-precision traits issue 1 MFMA per pair in k-step 0, 2 in 1.
template <class Tr, int MR, int NR, int KS, int LPT, int SPREAD, int s>
__device__ __forceinline__ void pin_ksteps() {
  if constexpr (s < KS) {
    constexpr int RD = MR + NR, MM = MR * NR * mfma_in_step<Tr>(s);
    constexpr int first = SPREAD == 2 ? 0 : (s * LPT + KS - 1) / KS;                       // pieces [first, last) go out in k-step s
    constexpr int last = SPREAD == 2 ? (s == 0 ? LPT : 0) : ((s + 1) * LPT + KS - 1) / KS;
    constexpr int ps = last - first;
#pragma unroll
    for (int q2 = 0; q2 < MM; ++q2) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (s < KS - 1 && q2 < RD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (((q2 + 1) * ps) / MM > (q2 * ps) / MM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    if constexpr (s < KS - 1 && RD > MM) __builtin_amdgcn_sched_group_barrier(0x100, RD - MM, 0);
    if constexpr (ps > MM) {                 // more pieces than MFMA gaps in this k-step (never with the shipped tiles): the rest in a row
#pragma unroll
      for (int x = 0; x < ps - MM; ++x) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    pin_ksteps<Tr, MR, NR, KS, LPT, SPREAD, s + 1>();
  }
}

// Reads the accumulators of the assembly K loop where it left them (a[0:255], block (i, j) in a[4 * (8 * i + j) : +3]): one
// v_accvgpr_read per value at the point of use, eight values (one output row of the lane) at a time, in asm statements (volatile:
// they stay behind the loop's).  The loop declares the accumulators as clobbers; nothing tells the compiler that they stay in use
// through the epilogue - it has no reason to touch them (the epilogue needs < 100 vector registers, accumulation registers are only
// ever its spill space), and tools/check_dma_counts.py verifies in the emitted ISA that it does not.
struct AccAgpr4w {
  static constexpr bool kSpecialise = true;      // conv_epilogue_r: row loops specialised per switch combination
  __device__ __forceinline__ void row(int i, int e, float (&v)[8]) const {
    switch (i * 4 + e) { RON_ACC4W_CASES }
  }
};

// ... of the 256 x 128 tile: block (i, j), j < 4, in a[4 * (4 * i + j) : +3]
struct AccAgpr4wN128 {
  static constexpr bool kSpecialise = true;
  __device__ __forceinline__ void row(int i, int e, float (&v)[4]) const {
    switch (i * 4 + e) { RON_ACC4W_N128_CASES }
  }
};

// Everything after the K loop of conv_igemm_tile: raw fp32 slab store of a split-K slice, or the conv epilogue.
template <class Tr, int MR, int NR, int MT, int EPA, int TM, int TN, class Reader>
__device__ __forceinline__ void igemm_finish(const ConvArgs& p, const Reader& rd, const int* s_out_off, const int* s_out2_off, int zsplit,
                                             int m0, int n0, int wm, int wn, int fr, int fh) {
  // C/D layout of the 16x16 MFMA: column = lane % 16, row = e + 4 * (lane / 16), e < 4
  int tap_off = 0, n_base = n0;
  if (p.up > 0) {
    const int tap = n0 / p.up_cout;                       // BN divides up_cout: uniform per tile
    tap_off = ((tap / p.up) * p.out_Wp + (tap % p.up)) * p.out_cstride;
    n_base = n0 - tap * p.up_cout;
  }
  const int nloc = wn * TN + fr * NR;                     // lane -> NR adjacent output channels
  if (p.splitk > 1) {
    float* slab = p.partial + (size_t)zsplit * p.M * p.Npad;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
#pragma unroll
      for (int e = 0; e < EPA; ++e) {
        float v[NR];
        rd.row(i, e, v);
        const int rt = wm * TM + i * MT + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (s_out_off[rt] < 0) continue;
        store_f32_vec<NR>(slab + (size_t)(m0 + rt) * p.Npad + n0 + nloc, v);
      }
    }
    return;
  }
  conv_epilogue_r<Tr, MR, NR, MT, EPA>(p, rd, s_out_off, wm * TM, fh, n0 + nloc, n_base + nloc, tap_off, s_out2_off);
}

// Tile configuration: BM x BN block tile, WM x WN waves (each wave owns (BM/WM) x (BN/WN)), S LDS stages.
// SPREAD 1: the LDS-DMA pieces of a tile are shared out over the k-steps of the stage; 2: all go out during k-step 0.
// The K order (tap-major, or chunk-major with the taps innermost: ConvArgs::taps_inner, see below) is a run-time property of the
// launch: wave-uniform bookkeeping of a few scalar instructions per step.
// One tile of launch `p`: workgroup `bid` of the `nwg` that launch consists of (a launch of its own, or a range of the
// workgroups of a grouped launch).
// ConvArgs::m_fastest >= 2 = the width of a panel in column tiles (four-wave tiles only)
// tile index inside a run of `cols` column tiles x `tiles_m` row tiles -> (row tile, column tile), panels of `P` columns walked row by
// row: the 32 tiles an XCD runs at a time are 32 / P rows x P columns
__device__ __forceinline__ void panel_tile(unsigned t, unsigned tiles_m, unsigned cols, unsigned P, int* tile_m, int* tile_n) {
  const unsigned per_panel = tiles_m * P, panel = t / per_panel, r = t - panel * per_panel;
  const unsigned width = min(P, cols - panel * P);
  *tile_m = (int)(r / width);
  *tile_n = (int)(panel * P + r - (r / width) * width);
}

template <class Tr, int BM, int BN, int WM, int WN, int S, int SPREAD>
__device__ __forceinline__ void conv_igemm_tile(const ConvArgs& p, const unsigned bid, const unsigned nwg, char* smem) {
  constexpr int kLanesPerRow = kRowBytes / 16;       // 16-byte chunks per row
  constexpr int MT = Tr::kMT;                        // MFMA output tile (16)
  constexpr int kGroups = 64 / MT;                   // 16-byte K groups one instruction consumes per row
  constexpr int KS = kLanesPerRow / kGroups;         // MFMA k-steps per stage (one u32x4 fragment per lane and step)
  constexpr int EPA = MT * MT / 64;                  // accumulator registers per MFMA tile
  constexpr int kThreads = WM * WN * 64;
  constexpr int TM = BM / WM, TN = BN / WN;          // wave tile
  constexpr int MR = TM / MT, NR = TN / MT;          // MT x MT accumulators per wave: MR x NR
  constexpr int kRowsPerIt = kThreads / kLanesPerRow; // tile rows one LDS-DMA pass of the block covers
  constexpr int A_IT = BM / kRowsPerIt, B_IT = BN / kRowsPerIt;
  constexpr int LPT = A_IT + B_IT;                   // LDS-DMA instructions per thread and K step
  constexpr int kABytes = BM * kRowBytes, kBBytes = BN * kRowBytes;
  constexpr int kRingBytes = S * (kABytes + kBBytes);
  constexpr int kChunkElems = kRowBytes / Tr::kEsz;
  static_assert(BM % kRowsPerIt == 0 && BN % kRowsPerIt == 0 && kRowsPerIt % 16 == 0, "tile / thread-count mismatch");
  static_assert(TM % 32 == 0 && TN % 32 == 0 && S >= 2 && S <= 5, "bad wave tile / stage count");
  static_assert(NR <= 8, "vector epilogue: at most 8 channels per lane");
  static_assert(SPREAD == 1 || SPREAD == 2, "SPREAD");
  // The 256 x 256 tile on FOUR waves (128 x 128 per wave, one wave per SIMD, 256 accumulator + 256 vector registers) runs its K loop
  // as the assembly of kloop4w.inc (tools/gen_kloop4w.py): two tiles of LDS-DMA in flight over two LDS stages, three barriers per
  // K step.  bf16, f16 and the split-precision form (three MFMAs per block); fp32 stays on the eight-wave loop.
  constexpr bool kAsmLoop = AsmLoop<Tr>::value && BM == 256 && (BN == 256 || BN == 128) && WM == 2 && WN == 2 && S == 2;
  // layout: [A stage 0 .. S-1][B stage 0 .. S-1][in_off: BM ints][out_off: BM ints][out2_off: BM ints][step table (kAsmLoop)]
  char* s_a = smem;
  char* s_b = smem + S * kABytes;
  int* s_in_off = reinterpret_cast<int*>(smem + kRingBytes);
  int* s_out_off = s_in_off + BM;
  int* s_out2_off = s_out_off + BM;                   // fused pool with the un-pooled map as a second output (ConvArgs::out2)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  if constexpr (IsSplit<Tr>::value) split_mode_on();      // the epilogue's fp32 -> f16 conversions saturate (conv_device.h)

  // XCD-aware tile order: workgroups that share an XCD (blockIdx % 8) take consecutive tiles,
  // so the N-tiles that re-read one A tile hit the same L2.
  const unsigned xcd = bid & 7u, q = nwg >> 3, r8 = nwg & 7u;
  const unsigned wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
  int zsplit = (int)(wgid / (unsigned)p.tiles_total);
  const unsigned tile = wgid - (unsigned)zsplit * (unsigned)p.tiles_total;
  const unsigned tiles_m = (unsigned)p.tiles_total / (unsigned)p.tiles_n;
  int tile_n = (int)(p.m_fastest ? tile / tiles_m : tile % (unsigned)p.tiles_n);
  int tile_m = (int)(p.m_fastest ? tile % tiles_m : tile / (unsigned)p.tiles_n);
  if constexpr (kAsmLoop) if (p.m_fastest >= 2) {
    // (four-wave tiles only - the launches wide and tall enough are theirs; compiled into every tile the two divisions cost the
    // small-batch launches 1-2 us each through the other tiles' register allocation: batch 4 1.24 -> 1.29 ms, measured round 5)
    // panels of m_fastest column tiles, walked row by row: with 8 the 32 tiles an XCD runs at a time are 4 rows x 8 columns (12
    // operand streams instead of 1 + 32 on a wide launch), with 2 they are 16 x 2 (the weights of two column tiles stream per round)
    panel_tile(tile, tiles_m, (unsigned)p.tiles_n, (unsigned)p.m_fastest, &tile_m, &tile_n);
  }
  if (p.center_from_n > 0 && p.splitk == 1) {
    // Column tiles that run the centre tap only (a 1x1 branch beside 3x3 ones) are short: a ninth of the K steps plus a whole
    // tile's set-up and stores.  Interleaved with the long ones they cost more than they save (measured: 857 -> 736 us where
    // 440 + 63 us as two launches); dispatched AFTER every long tile they run on the CUs the last round of long tiles leaves idle.
    // Workgroups are dispatched in blockIdx order: the first n_long ids take the long tiles (XCD-major among themselves), the rest
    // the short ones.
    const unsigned cols_long = (unsigned)(p.center_from_n / BN), cols_short = (unsigned)p.tiles_n - cols_long;
    const unsigned n_long = tiles_m * cols_long;
    if (bid < n_long) {
      const unsigned xq = n_long >> 3, xr = n_long & 7u;
      const unsigned t = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
      tile_m = (int)(t / cols_long); tile_n = (int)(t % cols_long);
      if constexpr (kAsmLoop) if (p.m_fastest >= 2) panel_tile(t, tiles_m, cols_long, (unsigned)p.m_fastest, &tile_m, &tile_n);
    } else {
      const unsigned b2 = bid - n_long, n_short = nwg - n_long;
      const unsigned x2 = b2 & 7u, xq = n_short >> 3, xr = n_short & 7u;
      const unsigned t = (x2 < xr ? x2 * (xq + 1) : xr * (xq + 1) + (x2 - xr) * xq) + (b2 >> 3);
      tile_m = (int)(t / cols_short); tile_n = (int)(cols_long + t % cols_short);
    }
    zsplit = 0;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  int kt0 = zsplit * p.kt_split;
  int kt1 = min(p.KT, kt0 + p.kt_split);
  const bool center_only = p.center_from_n > 0 && n0 >= p.center_from_n;
  // Position-major rows: the tile runs the filter rows [ky_lo, ky_hi] that touch the image for at least one of its output rows, and
  // its K steps are numbered locally, [0, rows * kw * steps_per_tap).  The rows are walked from the filter's centre row up to ky_hi,
  // then from ky_lo up to the centre: every tile of a launch starts on the same weights, and tiles whose row sets differ still meet
  // on the rows they share at about the same time - the workgroups of an XCD stream a weight column through its L2 together
  // instead of each at its own filter row (fc6, 7x7 rate 3 on 10x10: 13 row tiles with row sets 3..6, 2..5, 1..4, 0..3).
  const bool pm = p.pos_major && !center_only;
  int ky_lo = 0, ky_hi = 0x7fffff, ky_first = 0;
  if (p.pos_major || center_only) {
    // split-K slices share the tile's steps evenly (a slice may be empty: it stores zeros)
    const int steps_per_tap = p.Cin / (kRowBytes / Tr::kEsz);
    int lo, hi;
    if (center_only) {
      // a 1x1 branch in the centre tap of the filter: that tap alone (tap-major numbering of the launch)
      lo = ((p.kh >> 1) * p.kw + (p.kw >> 1)) * steps_per_tap;
      hi = lo + steps_per_tap;
    } else {
      const int oy_lo = m0 / (p.n_img * p.Wo), oy_hi = (min(p.M, m0 + BM) - 1) / (p.n_img * p.Wo);
      ky_hi = p.kh - 1;
      while (ky_lo < ky_hi && oy_hi * p.stride - p.cpad + ky_lo * p.dil < 0) ++ky_lo;
      while (ky_hi > ky_lo && oy_lo * p.stride - p.cpad + ky_hi * p.dil > p.in_H - 1) --ky_hi;
      ky_first = min(max(p.cpad / p.dil, ky_lo), ky_hi);
      lo = 0;
      hi = (ky_hi - ky_lo + 1) * p.kw * steps_per_tap;
    }
    const int per = (hi - lo + p.splitk - 1) / p.splitk;
    kt0 = min(hi, lo + zsplit * per);
    kt1 = min(hi, kt0 + per);
  }

  // per-row addressing, once per tile
  for (int r = tid; r < BM; r += kThreads) {
    int img, oy, ox, off;
    bool valid;
    if (p.pool) {
      const int pw = p.Wo >> 1, ph = p.Ho >> 1;
      int P = (m0 >> 2) + (r >> 2);                 // pooled pixel of this window
      valid = P < (p.M >> 2);
      P = valid ? P : (p.M >> 2) - 1;
      img = P / (ph * pw);
      const int rem = P - img * (ph * pw);
      const int py = rem / pw, px = rem - (rem / pw) * pw;
      oy = 2 * py + ((r >> 1) & 1);
      ox = 2 * px + (r & 1);
      off = ((img * p.out_Hp + py + p.out_pad) * p.out_Wp + px + p.out_pad) * p.out_cstride + p.out_coff;
      if (p.out2 != nullptr)
        s_out2_off[r] = valid ? ((img * p.out2_Hp + oy + p.out2_pad) * p.out2_Wp + ox + p.out2_pad) * p.out2_cstride + p.out2_coff : -1;
    } else {
      int m = m0 + r;
      valid = m < p.M;
      m = valid ? m : p.M - 1;
      const int hw = p.Ho * p.Wo;
      int rem;
      if (p.pos_major) {                       // rows ordered (oy, img, ox): ConvArgs::pos_major
        const int rw = p.n_img * p.Wo;
        oy = m / rw;
        rem = m - oy * rw;
        img = rem / p.Wo;
        ox = rem - img * p.Wo;
      } else {
        img = m / hw;
        rem = m - img * hw;
        oy = rem / p.Wo;
        ox = rem - oy * p.Wo;
      }
      const int os = p.up > 0 ? p.up : 1;
      off = ((img * p.out_Hp + oy * os + p.out_pad) * p.out_Wp + ox * os + p.out_pad) * p.out_cstride + p.out_coff;
    }
    const int iy = oy * p.stride + p.in_org, ix = ox * p.stride + p.in_org;
    s_in_off[r] = (int)((((unsigned)(img * p.in_Hp + iy) * p.in_Wp + ix) * p.in_cstride + p.in_coff) * Tr::kEsz);
    s_out_off[r] = valid ? off : -1;
  }
  __syncthreads();

  // LDS-DMA source offsets (bytes): thread -> (row = it*kRowsPerIt + tid/8, slot = tid%8), source chunk = slot ^ key(row)
  const int ld_row = tid / kLanesPerRow;
  const int ld_chunk = (tid & 7) ^ ((tid >> 4) & 7);
  // fixed-size arrays on purpose: with a template-dependent bound the LDS-DMA builtin's voffset becomes a
  // type-dependent expression and hipcc (ROCm 7.2) silently drops the kernel's host stub.
  int a_voff[8], b_voff[8];
  static_assert(A_IT <= 8 && B_IT <= 8, "tile too large");
#pragma unroll
  for (int it = 0; it < A_IT; ++it) a_voff[it] = s_in_off[it * kRowsPerIt + ld_row] + ld_chunk * 16;
  // B rows are permuted on the way in: LDS row (j*MT + r) of a wave's TN-wide group holds weight row (r*NR + j), so
  // that MFMA column r of the wave's j-th MT-column tile is output channel r*NR + j: a lane's NR accumulators are NR
  // adjacent channels and the epilogue stores them as one contiguous NR-element vector (full 128-B lines per row).
  // Weight row n, K step kt sits at ((n / 64) * KT + kt) * 8 KB + (n % 64) * 128 B (pack.h, block_rows): every 8-row piece
  // of a wave's DMA instruction reads rows that are 128 B apart inside one 8-KB block, not K*esz bytes apart.
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int lrow = it * kRowsPerIt + ld_row;
    const int grp = lrow / TN, loc = lrow % TN;
    const int nrow = n0 + grp * TN + (loc % MT) * NR + (loc / MT);
    b_voff[it] = (int)((unsigned)(nrow >> 6) * (unsigned)p.KT * (unsigned)kWeightBlockBytes + (unsigned)(nrow & 63) * kRowBytes + ld_chunk * 16);
  }

  // K-step bookkeeping (wave-uniform): tap (ky, kx) and channel chunk cc of the NEXT tile to stage
  const int chunks_per_tap = p.Cin / kChunkElems;
  // TI ("taps innermost"): K runs chunk-major, step q = chunk * taps + tap (the weights of tap t, chunk c are block
  // t * chunks + c whatever the order).  Consecutive steps then re-read almost the same input lines, one pixel over, instead of
  // coming back to them a whole sweep of the channels later: the re-reads hit L2 without another workgroup's help.  Worth
  // 1-6 % on layers with one or two column tiles (nobody else on the XCD stages the same input rows at the same time), nothing
  // or -1 % on the wide ones (profiles/r02/sweep_conv_exp_v4_tapsinner.txt); conv_pick_igemm_cfg selects it accordingly.
  // A column tile that runs the centre tap only (ConvArgs::center_from_n) walks that tap's chunks in the tap-major numbering whatever
  // the launch's order: its [kt0, kt1) is a contiguous range there.
  const bool ti = p.taps_inner != 0 && !center_only;
  const int n_taps = p.KT / chunks_per_tap;
  typename Tr::acc_t acc[MR][NR];
  // fragment read offsets: lane -> row r = lane % MT, K group h = lane / MT; step s reads chunk kGroups*s + h
  const int fr = lane & (MT - 1), fh = lane / MT;
  int rd_off[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) rd_off[s] = fr * kRowBytes + (((kGroups * s + fh) ^ ((fr >> 1) & 7)) << 4);
  const int a_base = wm * TM * kRowBytes;
  const int b_base = wn * TN * kRowBytes;
  if constexpr (kAsmLoop) {
    static_assert(KS == 2 && MR == 8 && (NR == 8 || NR == 4) && A_IT == 8 && B_IT == NR, "kloop4w.inc is written for these tiles");
    // Step table: entry q = {soffset of the A pieces, soffset of the B pieces} of the tile's K step kt0 + q, in the launch's K order
    // (tap-major / taps innermost / position-major walk, see the loop of the other tiles below); two more entries than steps: the
    // loop stages two tiles ahead, past the end with zero-record descriptors.
    int* s_tab = s_out2_off + BM;
    const int nsteps = max(kt1 - kt0, 0);
    for (int q = tid; q < nsteps + 2; q += kThreads) {
      const int g = kt0 + q;
      int ky_, kx_, cq, wb;
      if (pm) {
        const int per_row = p.kw * chunks_per_tap;
        const int i0 = g / per_row, r0 = g - i0 * per_row;
        ky_ = ky_first + i0;
        if (ky_ > ky_hi) ky_ -= ky_hi - ky_lo + 1;
        cq = r0 / p.kw;
        kx_ = r0 - cq * p.kw;
        wb = (ky_ * p.kw + kx_) * chunks_per_tap + cq;
      } else if (ti) {
        cq = g / n_taps;
        const int tap = g - cq * n_taps;
        ky_ = tap / p.kw;
        kx_ = tap - ky_ * p.kw;
        wb = tap * chunks_per_tap + cq;
      } else {
        const int tap = g / chunks_per_tap;
        cq = g - tap * chunks_per_tap;
        ky_ = tap / p.kw;
        kx_ = tap - ky_ * p.kw;
        wb = g;
      }
      s_tab[2 * q] = ((ky_ * p.dil * p.in_Wp + kx_ * p.dil) * p.in_cstride + cq * kChunkElems) * Tr::kEsz;
      s_tab[2 * q + 1] = wb * kWeightBlockBytes;
    }
    __syncthreads();
    u32x8 av, bv;
#pragma unroll
    for (int it = 0; it < 8; ++it) { av[it] = (unsigned)a_voff[it]; bv[it] = it < B_IT ? (unsigned)b_voff[it < B_IT ? it : 0] : 0u; }
    const unsigned lds_a = (unsigned)(uintptr_t)(lds_void*)s_a, lds_b = (unsigned)(uintptr_t)(lds_void*)s_b;
    const unsigned dst_a = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_a + wave * 1024));
    const unsigned dst_b = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_b + wave * 1024));
    const unsigned rd_a0 = lds_a + a_base + rd_off[0], rd_a1 = lds_a + a_base + rd_off[1];
    const unsigned rd_b0 = lds_b + b_base + rd_off[0], rd_b1 = lds_b + b_base + rd_off[1];
    const unsigned tab = (unsigned)(uintptr_t)(lds_void*)s_tab;
    const unsigned long long in_ptr = (unsigned long long)(uintptr_t)p.in, wgt_ptr = (unsigned long long)(uintptr_t)p.wgt;
    const unsigned in_bytes = p.in_bytes, wgt_bytes = p.wgt_bytes;
    const unsigned ns = (unsigned)__builtin_amdgcn_readfirstlane(nsteps);
#define RON_KLOOP4W_OPERANDS                                                                                                        \
        :                                                                                                                           \
        : "{s[36:37]}"(in_ptr), "{s[38:39]}"(wgt_ptr), "{s40}"(in_bytes), "{s41}"(wgt_bytes), "{s42}"(ns), "{s43}"(dst_a),          \
          "{s44}"(dst_b), "{v[100:107]}"(av), "{v[108:115]}"(bv), "{v116}"(rd_a0), "{v117}"(rd_a1), "{v118}"(rd_b0),                \
          "{v119}"(rd_b1), "{v120}"(tab)                                                                                            \
        : RON_KLOOP4W_CLOBBERS
    if constexpr (NR == 8) {
      if constexpr (IsSplit<Tr>::value) asm volatile(RON_KLOOP4W_F16X3 RON_KLOOP4W_OPERANDS);
      else if constexpr (Tr::kIsBf16) asm volatile(RON_KLOOP4W_BF16 RON_KLOOP4W_OPERANDS);
      else asm volatile(RON_KLOOP4W_F16 RON_KLOOP4W_OPERANDS);
      igemm_finish<Tr, MR, NR, MT, EPA, TM, TN>(p, AccAgpr4w{}, s_out_off, s_out2_off, zsplit, m0, n0, wm, wn, fr, fh);
    } else {
      if constexpr (IsSplit<Tr>::value) asm volatile(RON_KLOOP4W_N128_F16X3 RON_KLOOP4W_OPERANDS);
      else if constexpr (Tr::kIsBf16) asm volatile(RON_KLOOP4W_N128_BF16 RON_KLOOP4W_OPERANDS);
      else asm volatile(RON_KLOOP4W_N128_F16 RON_KLOOP4W_OPERANDS);
      igemm_finish<Tr, MR, NR, MT, EPA, TM, TN>(p, AccAgpr4wN128{}, s_out_off, s_out2_off, zsplit, m0, n0, wm, wn, fr, fh);
    }
#undef RON_KLOOP4W_OPERANDS
    return;
  } else {
  const int tap0 = ti ? kt0 % n_taps : kt0 / chunks_per_tap;
  int ky = tap0 / p.kw, kx = tap0 - (tap0 / p.kw) * p.kw;
  int cc = (ti ? kt0 / n_taps : kt0 - tap0 * chunks_per_tap) * kChunkElems;
  if (pm) {
    // Inside a filter row the K steps run chunk-major (step = (row, chunk, kx)): the kw taps of a chunk re-read the same 128-byte
    // pieces of the same pixels, shifted, so an XCD's tiles keep ~0.5 MB of activations live per chunk instead of sweeping every
    // pixel's whole channel vector once per tap (fc6: 3.3 MB per tap and XCD, which the 4 MB L2 does not hold beside the weights:
    // 0.7 GB of activation re-reads per launch).  Local step kt0 -> (row in walking order, chunk, kx).
    const int per_row = p.kw * chunks_per_tap;
    const int i0 = kt0 / per_row, r0 = kt0 - i0 * per_row;
    ky = ky_first + i0;
    if (ky > ky_hi) ky -= ky_hi - ky_lo + 1;
    kx = r0 % p.kw;
    cc = (r0 / p.kw) * kChunkElems;
  }
  // weight block of the next tile to stage (tap-major orders): the K step itself unless the filter rows are walked from the centre
  int wblk = pm ? (ky * p.kw + kx) * chunks_per_tap + cc / kChunkElems : kt0;
  int tb = ti ? kt0 % n_taps : 0, cb = ti ? kt0 / n_taps : 0;          // weight ring: tap / chunk of its next tile
  // One K step issues LPT LDS-DMA pieces per thread: the B_IT weight pieces of tile kt+S-1 first, then its A_IT
  // activation pieces.  RON_STAGE_BEGIN computes the wave-uniform part once per step, RON_STAGE_PIECE issues piece j
  // (compile-time), RON_STAGE_END advances the tap of the activation ring.
#define RON_STAGE_BEGIN(kt_)                                                                                         \
    /* past the last tile: zero-record descriptors, the DMA moves nothing but keeps the vmcnt bookkeeping uniform */  \
    const int ktb_ = (kt_) + S - 1;                                                                                  \
    const __amdgpu_buffer_rsrc_t rs_a =                                                                              \
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, ktb_ < kt1 ? p.in_bytes : 0u, 0x00020000);     \
    const __amdgpu_buffer_rsrc_t rs_b =                                                                              \
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, ktb_ < kt1 ? p.wgt_bytes : 0u, 0x00020000);   \
    const int a_soff = ((ky * p.dil * p.in_Wp + kx * p.dil) * p.in_cstride + cc) * Tr::kEsz;                         \
    const int b_soff = (ti ? tb * chunks_per_tap + cb : wblk) * kWeightBlockBytes;                                   \
    char* dst_a = s_a + ((ktb_ - kt0) % S) * kABytes + wave * 1024;                                                  \
    char* dst_b = s_b + ((ktb_ - kt0) % S) * kBBytes + wave * 1024;
#define RON_STAGE_PIECE_B(i_)                                                                                        \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(dst_b + (i_) * kRowsPerIt * kRowBytes), 16, b_voff[i_], b_soff, 0, 0)
#define RON_STAGE_PIECE_A(i_)                                                                                        \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(dst_a + (i_) * kRowsPerIt * kRowBytes), 16, a_voff[i_], a_soff, 0, 0)
#define RON_STAGE_PIECE(j_)                                                                                          \
    do {                                                                                                             \
      if ((j_) < B_IT) RON_STAGE_PIECE_B((j_) < B_IT ? (j_) : 0);                                                    \
      else RON_STAGE_PIECE_A((j_) >= B_IT ? (j_) - B_IT : 0);                                                        \
    } while (0)
#define RON_STAGE_END()                                                                                              \
    do {                                                                                                             \
      if (ti) {                                                                                                      \
        if (++kx == p.kw) { kx = 0; if (++ky * p.kw >= n_taps) { ky = 0; cc += kChunkElems; } }                       \
        if (++tb == n_taps) { tb = 0; ++cb; }                                                                        \
        break;                                                                                                       \
      }                                                                                                              \
      if (pm) {                                                                                                      \
        wblk += chunks_per_tap;                                                                                      \
        if (++kx == p.kw) {                                                                                          \
          kx = 0;                                                                                                    \
          cc += kChunkElems;                                                                                         \
          if (cc >= p.Cin) { cc = 0; if (++ky > ky_hi) ky = ky_lo; }                                                 \
          wblk = ky * p.kw * chunks_per_tap + cc / kChunkElems;                                                      \
        }                                                                                                            \
        break;                                                                                                       \
      }                                                                                                              \
      cc += kChunkElems;                                                                                             \
      ++wblk;                                                                                                        \
      if (cc >= p.Cin) {                                                                                             \
        cc = 0;                                                                                                      \
        if (++kx == p.kw) { kx = 0; ++ky; }                                                                          \
      }                                                                                                              \
    } while (0)

#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < EPA; ++e) acc[i][j][e] = 0.f;

  // prologue: the steps before the first one, in the order the loop issues them: S-1 tiles in flight
#pragma unroll
  for (int t = -(S - 1); t < 0; ++t) {
    RON_STAGE_BEGIN(kt0 + t)
#pragma unroll
    for (int i = 0; i < B_IT; ++i) RON_STAGE_PIECE_B(i);
#pragma unroll
    for (int i = 0; i < A_IT; ++i) RON_STAGE_PIECE_A(i);
    RON_STAGE_END();
  }

  for (int kt = kt0; kt < kt1; ++kt) {
    // this wave's share of tile kt has landed; the youngest S-2 groups of LPT pieces may still be in flight.  The count relies on
    // the compiler emitting exactly LPT LDS-DMA instructions per step (prologue included): tests/test_isa_protocol.py checks the ISA
    wait_vmcnt<(S - 2) * LPT>();
    __builtin_amdgcn_s_barrier();           // ... everyone's has, and everyone is done reading tile kt-1
    // refill the stages tile kt-1 occupied; the LPT pieces go out between the MFMAs below
    RON_STAGE_BEGIN(kt)
    const char* sbuf_a = s_a + ((kt - kt0) % S) * kABytes;
    const char* sbuf_b = s_b + ((kt - kt0) % S) * kBBytes;
    // fragments of k-step s+1 are read while the MFMAs of k-step s run (two register sets)
    u32x4 fa[2][MR], fb[2][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i) fa[0][i] = *reinterpret_cast<const u32x4*>(sbuf_a + a_base + i * MT * kRowBytes + rd_off[0]);
#pragma unroll
    for (int j = 0; j < NR; ++j) fb[0][j] = *reinterpret_cast<const u32x4*>(sbuf_b + b_base + j * MT * kRowBytes + rd_off[0]);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s < KS - 1) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
          fa[(s + 1) & 1][i] = *reinterpret_cast<const u32x4*>(sbuf_a + a_base + i * MT * kRowBytes + rd_off[(s + 1) % KS]);
#pragma unroll
        for (int j = 0; j < NR; ++j)
          fb[(s + 1) & 1][j] = *reinterpret_cast<const u32x4*>(sbuf_b + b_base + j * MT * kRowBytes + rd_off[(s + 1) % KS]);
      }
#pragma unroll
      for (int i = 0; i < LPT; ++i)
        if ((SPREAD == 2 ? 0 : (i * KS) / LPT) == s) RON_STAGE_PIECE(i);
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) mma_step<Tr, MR, NR>(s, fa, fb, i, j, acc[i][j]);
    }
    // Pin the issue order (hipcc otherwise sinks the next k-step's fragment reads below the MFMAs to save
    // registers): R0 | per k-step: MFMAs with the next k-step's reads one per MFMA gap and this k-step's LDS-DMA pieces
    // spaced evenly between them | MFMAs of the last k-step.
    __builtin_amdgcn_sched_group_barrier(0x100, MR + NR, 0);
    pin_ksteps<Tr, MR, NR, KS, LPT, SPREAD, 0>();
    RON_STAGE_END();
  }
#undef RON_STAGE_BEGIN
#undef RON_STAGE_PIECE
#undef RON_STAGE_PIECE_A
#undef RON_STAGE_PIECE_B
#undef RON_STAGE_END
  }   // !kAsmLoop
  igemm_finish<Tr, MR, NR, MT, EPA, TM, TN>(p, AccArray<Tr, MR, NR>{acc}, s_out_off, s_out2_off, zsplit, m0, n0, wm, wn, fr, fh);
}


// K steps one workgroup of the assembly K loop can take (its step table has two more entries): 16 KB of LDS
constexpr int kAsmLoopMaxSteps = 2046;
constexpr int igemm_lds_bytes(int BM, int BN, int S, bool asm_loop = false) {
  return S * (BM + BN) * kRowBytes + 3 * BM * (int)sizeof(int) + (asm_loop ? (kAsmLoopMaxSteps + 2) * 2 * (int)sizeof(int) : 0);
}

template <class Tr, int BM, int BN, int WM, int WN, int S, int SPREAD>
__global__ __launch_bounds__(WM * WN * 64, (igemm_lds_bytes(BM, BN, S) > 80 * 1024) ? (WM * WN + 3) / 4 : 2) void conv_igemm_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  conv_igemm_tile<Tr, BM, BN, WM, WN, S, SPREAD>(p, blockIdx.x, gridDim.x, smem);
}

// Several independent small convolutions in ONE launch (the per-scale head layers of the coarse scales, each of which
// alone leaves most of the chip idle): workgroups [first[k], first[k+1]) run conv k exactly as its own launch would.
// The descriptions travel in the kernel-argument segment; a workgroup copies its own with scalar loads.
constexpr int kMaxGroup = kMaxConvGroup;
// The launch is a list of ENTRIES: entry e = workgroups [first[e], first[e+1]) of the grid = workgroups [ebid0[e], ...) of member
// eop[e].  A member is one entry, or two when it has centre-tap-only column tiles: its long tiles and its short ones (a ninth of the K
// steps) are separate entries, so that the host can order ALL long work of the launch before ANY short work - workgroups are
// dispatched in blockIdx order, and 400 short tiles in front of another member's long ones delayed those by a third of the launch.
constexpr int kMaxEntries = 2 * kMaxGroup;
struct ConvGroupArgs {
  ConvArgs op[kMaxGroup];
  int first[kMaxEntries + 1];
  int eop[kMaxEntries];       // entry -> member
  int ebid0[kMaxEntries];     // the entry's first workgroup, counted inside the member
  int enwg[kMaxEntries];      // workgroups of the member as a whole (the tile order is computed from it)
  int n;                      // members
  int ne;                     // entries
  unsigned narrow;            // kGroupMixed: bit k set = member k runs on 128 x 64 tiles, else on 128 x 128
};
struct GroupPick { int k; unsigned bid, nwg; };
__device__ __forceinline__ GroupPick pick_group_entry(const ConvGroupArgs& g, int b) {
  int k = g.eop[0], f = 0, bid0 = g.ebid0[0], nwg = g.enwg[0];
#pragma unroll
  for (int j = 1; j < kMaxEntries; ++j)
    if (j < g.ne && b >= g.first[j]) { k = g.eop[j]; f = g.first[j]; bid0 = g.ebid0[j]; nwg = g.enwg[j]; }
  return GroupPick{k, (unsigned)(b - f + bid0), (unsigned)nwg};
}
// op[k] of the ConvGroupArgs this kernel was launched with (its only argument), read from the kernel-argument segment
__device__ __forceinline__ ConvArgs load_group_op(int k) {
  static_assert(sizeof(ConvArgs) % 4 == 0, "dword copy");
  typedef __attribute__((address_space(4))) const unsigned* KernargWords;
  const KernargWords src = (KernargWords)__builtin_amdgcn_kernarg_segment_ptr() + k * (int)(sizeof(ConvArgs) / 4);
  ConvArgs p;
  unsigned* dst = reinterpret_cast<unsigned*>(&p);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(ConvArgs) / 4); ++i) dst[i] = src[i];
  return p;
}

template <class Tr, int BM, int BN, int WM, int WN, int S, int SPREAD>
__global__ __launch_bounds__(WM * WN * 64, (igemm_lds_bytes(BM, BN, S) > 80 * 1024) ? (WM * WN + 3) / 4 : 2) void conv_igemm_group_kernel(ConvGroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GroupPick e = pick_group_entry(g, (int)blockIdx.x);
  const ConvArgs p = load_group_op(e.k);
  conv_igemm_tile<Tr, BM, BN, WM, WN, S, SPREAD>(p, e.bid, e.nwg, smem);
}

// The same with the tile WIDTH chosen per member (kGroupMixed): both 128-row tiles run on 4 waves and fit two workgroups per CU, so
// skinny members (Npad = 64) and wide ones share a launch - a dependency level of the heads is then ONE launch whatever its mix.
template <class Tr>
__global__ __launch_bounds__(256, 2) void conv_igemm_group_mixed_kernel(ConvGroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GroupPick e = pick_group_entry(g, (int)blockIdx.x);
  const ConvArgs p = load_group_op(e.k);
  if ((g.narrow >> e.k) & 1u) conv_igemm_tile<Tr, 128, 64, 2, 2, 2, 2>(p, e.bid, e.nwg, smem);
  else conv_igemm_tile<Tr, 128, 128, 2, 2, 2, 2>(p, e.bid, e.nwg, smem);
}

template <class Tr>
__global__ void splitk_finalize_group_kernel(ConvGroupArgs g);

// Adds the split-K slabs and applies the conv epilogue (bias, ReLU, relu(x + residual), dtype / fp32 store).
template <class Tr>
__device__ __forceinline__ void splitk_finalize_body(const ConvArgs& p) {
  if constexpr (IsSplit<Tr>::value) split_mode_on();
  const int groups = p.Npad / 4;
  const long long total = (long long)p.M * groups;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % groups);
    const int m = (int)(idx / groups);
    const int n = g * 4;
    if (n >= p.Cout) continue;
    // slabs added in slice order (the sum's bits do not depend on how the loads are batched): eight loads in flight at a time -
    // one dependent HBM / L2 round trip per slab made this pass 5-8 us at split factors of 9 and more
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    const float* src = p.partial + (size_t)m * p.Npad + n;
    const size_t slab = (size_t)p.M * p.Npad;
    int z = 0;
    for (; z + 8 <= p.splitk; z += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(z + u) * slab);
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v[u];
    }
    if (z + 4 <= p.splitk) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(z + u) * slab);
#pragma unroll
      for (int u = 0; u < 4; ++u) sum += v[u];
      z += 4;
    }
    for (; z < p.splitk; ++z) sum += *reinterpret_cast<const f32x4*>(src + (size_t)z * slab);
    const int hw = p.Ho * p.Wo;
    int img, oy, ox;
    if (p.pos_major) {                         // rows ordered (oy, img, ox): ConvArgs::pos_major
      const int rw = p.n_img * p.Wo;
      oy = m / rw;
      img = (m - oy * rw) / p.Wo;
      ox = m - oy * rw - img * p.Wo;
    } else {
      img = m / hw;
      const int rem = m - img * hw;
      oy = rem / p.Wo;
      ox = rem - oy * p.Wo;
    }
    int o = ((img * p.out_Hp + oy + p.out_pad) * p.out_Wp + ox + p.out_pad) * p.out_cstride + p.out_coff + n;
    int n_end = p.Cout;
    void* dst = p.out;
    if (p.split_n > 0) {                       // two fp32 outputs (ConvArgs::split_n)
      if (n >= p.split_n) { o = ((img * p.out2_Hp + oy) * p.out2_Wp + ox) * p.out2_cstride + n - p.split_n; dst = p.out2; }
      else n_end = p.split_first;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (n + j >= n_end) break;
      float v = fmaf(sum[j], p.oscale, p.bias[n + j]);
      if (p.relu) v = fmaxf(v, 0.f);
      if (p.res != nullptr) v = fmaxf(v + Tr::load(p.res, o + j), 0.f);
      if (p.out_f32) reinterpret_cast<float*>(dst)[o + j] = v;
      else Tr::store(p.out, o + j, v);
    }
  }
}

template <class Tr>
__global__ void splitk_finalize_kernel(ConvArgs p) { splitk_finalize_body<Tr>(p); }

// blockIdx.y = conv of the group; the ones that did not split K have nothing to add
template <class Tr>
__global__ void splitk_finalize_group_kernel(ConvGroupArgs g) {
  const ConvArgs p = load_group_op((int)blockIdx.y);
  if (p.splitk > 1) splitk_finalize_body<Tr>(p);
}

template <class Tr, int BM, int BN, int WM, int WN, int S, int SPREAD>
int launch_t(const ConvArgs& a, hipStream_t s) {
  const size_t lds = (size_t)igemm_lds_bytes(BM, BN, S, AsmLoop<Tr>::value && BM == 256 && WM == 2);
  static PerDeviceOnce once;
  RON_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void*>(&conv_igemm_kernel<Tr, BM, BN, WM, WN, S, SPREAD>), (int)lds));
  RON_LAUNCH((conv_igemm_kernel<Tr, BM, BN, WM, WN, S, SPREAD>), dim3(a.tiles_total * a.splitk), dim3(WM * WN * 64), lds, s, a);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}



template <class Tr>
int launch_group_finalize(const ConvGroupArgs& g, hipStream_t s) {
  long long most = 0;
  for (int k = 0; k < g.n; ++k)
    if (g.op[k].splitk > 1) most = std::max(most, (long long)g.op[k].M * (g.op[k].Npad / 4));
  const int grid = (int)std::min<long long>((most + 255) / 256, 512);
  RON_LAUNCH(splitk_finalize_group_kernel<Tr>, dim3(grid, g.n), dim3(256), 0, s, g);
  return RON_OK;
}

template <class Tr, int BM, int BN, int WM, int WN, int S, int SPREAD>
int launch_group_t(const ConvGroupArgs& g, bool any_split, hipStream_t s) {
  const size_t lds = (size_t)igemm_lds_bytes(BM, BN, S, AsmLoop<Tr>::value && BM == 256 && WM == 2);
  static PerDeviceOnce once;
  RON_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void*>(&conv_igemm_group_kernel<Tr, BM, BN, WM, WN, S, SPREAD>), (int)lds));
  RON_LAUNCH((conv_igemm_group_kernel<Tr, BM, BN, WM, WN, S, SPREAD>), dim3(g.first[g.ne]), dim3(WM * WN * 64), lds, s, g);
  if (any_split) launch_group_finalize<Tr>(g, s);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

template <class Tr>
int launch_group_mixed(const ConvGroupArgs& g, bool any_split, hipStream_t s) {
  const size_t lds = (size_t)igemm_lds_bytes(128, 128, 2);
  static PerDeviceOnce once;
  RON_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void*>(&conv_igemm_group_mixed_kernel<Tr>), (int)lds));
  RON_LAUNCH((conv_igemm_group_mixed_kernel<Tr>), dim3(g.first[g.ne]), dim3(256), lds, s, g);
  if (any_split) launch_group_finalize<Tr>(g, s);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

// ron_dtype of a traits class
template <class Tr> struct DtypeOf;
template <> struct DtypeOf<TraitsBF16S> { static constexpr int value = RON_DTYPE_BF16; };
template <> struct DtypeOf<TraitsF16S> { static constexpr int value = RON_DTYPE_F16; };
template <> struct DtypeOf<TraitsF32S> { static constexpr int value = RON_DTYPE_F32; };
template <> struct DtypeOf<TraitsF16X3S> { static constexpr int value = RON_DTYPE_F16X3; };

// The four-wave tiles with the assembly K loop (bf16 / f16 / f16x3), instantiated in conv_mfma4w.hip: the 256 x bn tile of one
// convolution (bn = 256 or 128), and the grouped form of the 256 x 256 tile.
int launch_igemm4w(int dtype, int bn, const ConvArgs& a, hipStream_t s);
int launch_igemm4w_group(int dtype, const ConvGroupArgs& g, bool any_split, hipStream_t s);

}  // namespace detail
}  // namespace ron
