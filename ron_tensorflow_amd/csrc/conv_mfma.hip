// Implicit-GEMM convolution for gfx950 (CDNA4):   out[m, n] = sum_{tap, c} in[pix(m) + tap, c] * w[n][tap, c]
//
//   M = N_img * Ho * Wo output pixels, N = Cout, K = kh*kw*Cin, NHWC activations with a zero halo (no bounds checks in
//   the loop), weights pre-packed in 64-row x 128-byte blocks (pack.h, block_rows).
//
// One workgroup (WM x WN waves) computes a BM x BN tile; each wave owns a (BM/WM) x (BN/WN) sub-tile of 16x16 MFMA
// accumulators (v_mfma_f32_16x16x32_{bf16,f16}; fp32 mode: v_mfma_f32_16x16x4_f32, exact fp32 fma chains = the parity
// mode).  Per K step both operands advance by one 128-byte row chunk (64 bf16 / 32 fp32 of one filter tap), staged
// global -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane) into an S-stage ring: one counted vmcnt + one raw
// s_barrier per K step, the DMA of tile kt+S-1 is issued after the barrier into the stage whose reads just retired.
// LDS rows are 128 B; the 16-byte chunk c of row r lives in slot c ^ ((r >> 1) & 7): the DMA writes linearly, so the
// permutation is applied on the per-lane *source* address and again on the ds_read_b128 side (conflict-free for the
// 16-lane groups of ds_read_b128).  The A operand is a row gather: row r of the tile is the Cin-chunk of input pixel
// pix(m0 + r) shifted by the tap, so the per-lane voffset is fixed for the whole K loop and the tap / chunk advance is
// a wave-uniform soffset.
// Epilogue: + bias, ReLU, optional relu(x + residual) (reverse-connection sum), optional pixel-shuffle addressing
// (2x2 stride-2 transposed conv), optional fused 2x2 max-pool, store as dtype or fp32.
//
// This file holds the configurations conv_pick_cfg() can select, nothing else.  The ablation / stamp / experimental forms of
// this kernel that rounds 1-3 measured (HISTORY.md) are not in the tree: tools/experiments/README.md says where they are.
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
// v_accvgpr_read per value at the point of use, eight values (one output row of the lane) at a time.  The eight 32-register tuples
// are the loop's outputs, so the compiler knows they are live; extracting single elements from such a tuple makes hipcc (ROCm 7.2)
// copy all 256 values into vector registers first (spills), hence the reads by hand.
struct AccAgpr4w {
  const f32x32 &c0, &c1, &c2, &c3, &c4, &c5, &c6, &c7;
  __device__ __forceinline__ void row(int i, int e, float (&v)[8]) const {
    switch (i * 4 + e) { RON_ACC4W_CASES }
  }
};

// ... of the 256 x 128 tile: block (i, j), j < 4, in a[4 * (4 * i + j) : +3], two row tiles per tuple
struct AccAgpr4wN128 {
  const f32x32 &c0, &c1, &c2, &c3;
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
constexpr int kPanelCols = 8;      // ConvArgs::m_fastest == 2

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
  if (p.m_fastest == 2) {
    // panels of kPanelCols column tiles, walked row by row: the 32 tiles an XCD runs at a time are 4 rows x 8 columns (12 operand
    // streams instead of 1 + 32 on a wide launch)
    const unsigned per_panel = tiles_m * kPanelCols, panel = tile / per_panel, r = tile - panel * per_panel;
    const unsigned width = min((unsigned)kPanelCols, (unsigned)p.tiles_n - panel * kPanelCols);
    tile_m = (int)(r / width);
    tile_n = (int)(panel * kPanelCols + r - (r / width) * width);
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
#define RON_KLOOP4W_INPUTS                                                                                                          \
        : "{s[36:37]}"(in_ptr), "{s[38:39]}"(wgt_ptr), "{s40}"(in_bytes), "{s41}"(wgt_bytes), "{s42}"(ns), "{s43}"(dst_a),          \
          "{s44}"(dst_b), "{v[100:107]}"(av), "{v[108:115]}"(bv), "{v116}"(rd_a0), "{v117}"(rd_a1), "{v118}"(rd_b0),                \
          "{v119}"(rd_b1), "{v120}"(tab)                                                                                            \
        : RON_KLOOP4W_CLOBBERS
    if constexpr (NR == 8) {
      f32x32 c0, c1, c2, c3, c4, c5, c6, c7;
#define RON_KLOOP4W_OPERANDS                                                                                                        \
        : "={a[0:31]}"(c0), "={a[32:63]}"(c1), "={a[64:95]}"(c2), "={a[96:127]}"(c3), "={a[128:159]}"(c4), "={a[160:191]}"(c5),      \
          "={a[192:223]}"(c6), "={a[224:255]}"(c7)                                                                                  \
        RON_KLOOP4W_INPUTS
      if constexpr (IsSplit<Tr>::value) asm volatile(RON_KLOOP4W_F16X3 RON_KLOOP4W_OPERANDS);
      else if constexpr (Tr::kIsBf16) asm volatile(RON_KLOOP4W_BF16 RON_KLOOP4W_OPERANDS);
      else asm volatile(RON_KLOOP4W_F16 RON_KLOOP4W_OPERANDS);
#undef RON_KLOOP4W_OPERANDS
      igemm_finish<Tr, MR, NR, MT, EPA, TM, TN>(p, AccAgpr4w{c0, c1, c2, c3, c4, c5, c6, c7}, s_out_off, s_out2_off, zsplit, m0, n0, wm, wn, fr, fh);
    } else {
      f32x32 c0, c1, c2, c3;
#define RON_KLOOP4W_OPERANDS                                                                                                        \
        : "={a[0:31]}"(c0), "={a[32:63]}"(c1), "={a[64:95]}"(c2), "={a[96:127]}"(c3)                                                \
        RON_KLOOP4W_INPUTS
      if constexpr (IsSplit<Tr>::value) asm volatile(RON_KLOOP4W_N128_F16X3 RON_KLOOP4W_OPERANDS);
      else if constexpr (Tr::kIsBf16) asm volatile(RON_KLOOP4W_N128_BF16 RON_KLOOP4W_OPERANDS);
      else asm volatile(RON_KLOOP4W_N128_F16 RON_KLOOP4W_OPERANDS);
#undef RON_KLOOP4W_OPERANDS
      igemm_finish<Tr, MR, NR, MT, EPA, TM, TN>(p, AccAgpr4wN128{c0, c1, c2, c3}, s_out_off, s_out2_off, zsplit, m0, n0, wm, wn, fr, fh);
    }
#undef RON_KLOOP4W_INPUTS
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
    const int o = ((img * p.out_Hp + oy + p.out_pad) * p.out_Wp + ox + p.out_pad) * p.out_cstride + p.out_coff + n;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (n + j >= p.Cout) break;
      float v = fmaf(sum[j], p.oscale, p.bias[n + j]);
      if (p.relu) v = fmaxf(v, 0.f);
      if (p.res != nullptr) v = fmaxf(v + Tr::load(p.res, o + j), 0.f);
      if (p.out_f32) reinterpret_cast<float*>(p.out)[o + j] = v;
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


// The four-wave assembly K loop takes the 256 x 256 launches of bf16 / f16 whose workgroups' K chains fit its step table
// (RON_IGEMM256_V1=1: the eight-wave loop instead, for side-by-side timing).
static bool asm_loop_ok(int steps_per_wg) {
  static const bool v1 = getenv("RON_IGEMM256_V1") != nullptr;
  return !v1 && steps_per_wg <= kAsmLoopMaxSteps;
}

template <class Tr>
int launch_cfg(int cfg, const ConvArgs& a, hipStream_t s) {
  switch (cfg) {
    case kCfgIgemm256: case kCfgIgemm256TapsInner:                                                      // (the K order: ConvArgs::taps_inner)
      if constexpr (AsmLoop<Tr>::value) {
        if (asm_loop_ok(a.kt_split)) return launch_t<Tr, 256, 256, 2, 2, 2, 1>(a, s);                    // four waves, assembly K loop
      }
      return launch_t<Tr, 256, 256, 4, 2, 2, 1>(a, s);
    case kCfgIgemm128: return launch_t<Tr, 128, 128, 2, 2, 2, 1>(a, s);
    case kCfgIgemm128Early: case kCfgIgemm128EarlyTapsInner: return launch_t<Tr, 128, 128, 2, 2, 2, 2>(a, s);
    case kCfgIgemm128x64: return launch_t<Tr, 128, 64, 2, 2, 2, 2>(a, s);
    case kCfgIgemm256x128:
      if constexpr (AsmLoop<Tr>::value) {
        if (asm_loop_ok(a.kt_split)) return launch_t<Tr, 256, 128, 2, 2, 2, 1>(a, s);
      }
      ron::set_error("conv: the 256 x 128 tile is the four-wave assembly loop's (bf16 / f16 / f16x3, <= %d K steps per workgroup)", kAsmLoopMaxSteps);
      return RON_ERR_INVALID;
  }
  ron::set_error("conv: unknown tile config %d", cfg);
  return RON_ERR_INVALID;
}

template <class Tr>
int launch_finalize(const ConvArgs& a, hipStream_t s) {
  const long long total = (long long)a.M * (a.Npad / 4);
  const int grid = (int)std::min<long long>((total + 255) / 256, 2048);
  RON_LAUNCH(splitk_finalize_kernel<Tr>, dim3(grid), dim3(256), 0, s, a);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

}  // namespace detail
using namespace detail;

size_t dtype_size(int dtype) { return dtype_is_half(dtype) ? 2 : 4; }     // F16X3: a hi and a lo f16 per element
int conv_k_chunk(int dtype) { return kRowBytes / (int)dtype_size(dtype); }
int conv_n_tile(int cout) { return cout <= 64 ? 64 : 128; }
int conv_num_cfgs() { return kNumCfgs; }

static bool igemm_is256(int cfg) { return cfg == kCfgIgemm256 || cfg == kCfgIgemm256TapsInner; }
static int igemm_bm(int cfg) { return (igemm_is256(cfg) || cfg == kCfgIgemm256x128) ? 256 : 128; }
static int igemm_bn(int cfg) { return igemm_is256(cfg) ? 256 : (cfg == kCfgIgemm128x64 ? 64 : 128); }
// workgroups of a configuration the chip holds at once (256 CUs; 64 KB of LDS lets two share a CU)
static int igemm_slots(int cfg) { return (igemm_is256(cfg) || cfg == kCfgIgemm256x128) ? 256 : 512; }

// Split-K factor for grids that leave most CUs idle: such launches are a serial chain of KT dependent
// HBM round trips per workgroup, so the K loop is spread over enough workgroups to fill the chip (>= 8 steps each).
int conv_pick_splitk(int tiles, int KT, int slots) {
  if (tiles < 1 || tiles * 2 > slots || KT < 16) return 1;
  int sk = slots / tiles;
  int min_steps = 8;
  if (sk > KT / min_steps) sk = KT / min_steps;
  return sk < 1 ? 1 : sk;
}

constexpr int kAsmLoopMaxStepsHost = detail::kAsmLoopMaxSteps;

// Default tile of the row-gather kernel, from tools/sweep_conv.py on MI355X at batch 32 (profiles/r01/sweep_*.txt).
// The 256x256 tile wins once it yields >= ~160 workgroups; below that the grid is the problem and the 128x128 /
// 2-workgroups-per-CU form keeps more CUs busy.
constexpr int kTapsInnerMaxN = 2048;       // widest run of long columns the taps-innermost K order is chosen for
int conv_pick_igemm_cfg(int M, int Npad, int taps, int K, bool may_split) {
  const int tm256 = (M + 255) / 256;
  if (Npad % 128 != 0) return kCfgIgemm128x64;
  if (Npad % 256 == 0) {
    const int t256 = tm256 * (Npad / 256);
    // taps innermost: worth 1-6 % of the run time up to two column tiles, time-neutral on wider layers (trio3: six long column tiles,
    // 669 vs 660 us) where it still cuts the fabric reads by a fifth (profiles/r04/sweep_taps_inner_all.txt); conv_pick_cfg takes it
    // back where skipping halo filter rows is worth more (fc6: 514 us tap-major with skipping, 567 us taps-innermost)
    if (t256 >= 160) return (taps > 1 && Npad <= kTapsInnerMaxN) ? kCfgIgemm256TapsInner : kCfgIgemm256;
    // few fat tiles with a long K (in elements: the split-precision and fp32 forms take twice the steps for the same K): split-K
    // fills the chip with them too, at twice the arithmetic per staged byte of the 128 x 128 tiles (tools/sweep_conv.py, round 3:
    // block6_conv_left 26 tiles x 36 864 132 -> 126 us, fc6 at batch 8 214 -> 191, cls_pred / inception2 shapes of 50-100 tiles x 9 216
    // +3-10 %; with K = 4 608 or a handful of tiles the 128 x 128 form wins, in f16x3 as well: conv5_1 188 vs 202 us)
    if (may_split && t256 <= 128 && ((t256 >= 48 && K >= 9216) || (t256 >= 24 && K >= 32768))) return kCfgIgemm256;     // <= 128 tiles: K gets split
  }
  // the 128 x 128 tile with a stage's LDS-DMA pieces issued during its first k-step: level or 3-8 % ahead of the spread-out issue on
  // every layer and batch of the round-3 sweep (profiles/r03/sweep_conv_128_early.txt), not only on conv2_x's many rounds
  return kCfgIgemm128Early;
}

// Order of the tiles inside the run of workgroups that shares an XCD's L2 (1/8 of the launch): N fastest re-reads few activation
// tiles and every weight slice of those columns, M fastest the other way round.  Estimated bytes an XCD has to fetch once:
// activation tiles it touches x their size + weight slices it touches x theirs; M fastest when that is less (fc6: 13 row tiles x
// 16 column tiles of 12.8 MB of weights each - N fastest makes every XCD stream all 205 MB: 1.66 GB fetched per launch, 0.65 GB
// M fastest, 566 -> 541 us; fc7 113 -> 102 us).  Forced on every launch it costs the activation-heavy layers 2-4 %; walking the
// column tiles of an XCD's rows in blocks of 1-3 instead changed nothing (both measured, not kept).
static int pick_m_fastest(const ConvLaunch& c, int BM, int BN, int tiles_m, int tiles_n, int splitk) {
  // split-K: the workgroups of one K slice are consecutive, an XCD's run then covers (nearly) every tile of its slices either way
  if (tiles_m < 2 || tiles_n < 2 || splitk > 1) return 0;
  const double esz = (double)dtype_size(c.dtype);
  const int taps = c.kh * c.kw;
  const double a_tile = (double)BM * c.in.C * esz * (c.stride > 1 ? taps : (taps > 1 ? 2 : 1));   // unique input bytes of a row tile
  const double b_tile = (double)BN * taps * c.in.C * esz;
  const int per_xcd = std::max(1, tiles_m * tiles_n / 8);
  const double cost_n = (per_xcd / tiles_n + 1) * a_tile + std::min(tiles_n, per_xcd) * b_tile;
  const double cost_m = std::min(tiles_m, per_xcd) * a_tile + (per_xcd / tiles_m + 1) * b_tile;
  // 2: panels of kPanelCols column tiles (wide AND tall launches: plain GEMM shapes, fc7 at large batches)
  if (tiles_n >= 2 * kPanelCols && tiles_m >= 8 && c.center_from == 0) {
    const double cost_p = std::min(tiles_m, per_xcd / kPanelCols + 1) * a_tile +
                          std::min(tiles_n, kPanelCols * (per_xcd / (tiles_m * kPanelCols) + 1)) * b_tile;
    if (cost_p < 0.9 * std::min(cost_n, cost_m)) return 2;
  }
  return cost_m < 0.95 * cost_n ? 1 : 0;
}

// Position-major rows + per-tile skipping of filter rows that only see the zero halo (ConvArgs::pos_major): where the K steps a
// launch executes drop by >= 5 % (fc6 7x7 on 10 x 10: 91 -> 77 filter rows over its 13 row tiles; conv6 of SSD-512, rate 6).
// Only the tap-major K order (a skipped filter row is a contiguous K range), no fused pool / transposed conv; split-K slices
// share what is left of a tile's K range.
static int pick_pos_major(const ConvLaunch& c, int cfg, int BM) {
  if (!c.halo_skip || c.pool || c.up > 0 || c.kh < 2 || conv_cfg_taps_inner(cfg)) return 0;
  // at most (kh - 1) * dil of a map's H output rows can skip anything: below 5 % there is nothing to decide (and no per-tile walk
  // over the thousands of tiles of a large map on the launch path)
  if ((c.kh - 1) * c.dil * 20 < c.in.H) return 0;
  const int M = c.in.N * c.Ho * c.Wo, tiles_m = (M + BM - 1) / BM;
  long long rows_all = 0, rows_kept = 0;
  for (int t = 0; t < tiles_m; ++t) {
    const int oy_lo = (t * BM / c.in.N) / c.Wo, oy_hi = ((std::min(M, (t + 1) * BM) - 1) / c.in.N) / c.Wo;
    int lo = 0, hi = c.kh - 1;
    while (lo < hi && oy_hi * c.stride - c.cpad + lo * c.dil < 0) ++lo;
    while (hi > lo && oy_lo * c.stride - c.cpad + hi * c.dil > c.in.H - 1) --hi;
    rows_all += c.kh;
    rows_kept += hi - lo + 1;
  }
  return rows_kept * 100 <= rows_all * 95 ? 1 : 0;
}

// The halo-patch kernel (conv_patch.hip) where it applies and wins (conv_patch_pick), else the row-gather kernel.
int conv_pick_cfg(const ConvLaunch& c) {
  const int M = c.in.N * c.Ho * c.Wo;
  if (conv_c64_applicable(c)) return kCfgC64Resident;
  if (c.out2.base == nullptr && conv_patch_applicable(c)) {
    const int cfg = conv_patch_pick(c);
    if (cfg >= 0) return cfg;
  }
  // centre-tap-only columns are a ninth of a column's work: the grid that has to fill the chip is the long columns'
  // (split K: not with a fused pool / transposed conv, and a launch with centre-tap-only columns counts those tiles too)
  const bool may_split = c.center_from == 0 && !c.pool && c.up == 0 && c.splitk < 0;      // (the same answer when the scratch is being sized)
  const int KT = c.kh * c.kw * c.in.C / conv_k_chunk(c.dtype);
  // Cout = 128 (conv2_x) on maps large enough to fill the chip with 256 x 128 tiles: the four-wave assembly loop at 48 KB staged per
  // K step instead of two 128 x 128 workgroups per CU at 64 KB (sweep: profiles/r05/sweep_256x128.txt)
  if (c.dtype != RON_DTYPE_F32 && c.Npad % 256 == 128 && c.center_from == 0 && c.up == 0 && c.out2.base == nullptr &&
      ((M + 255) / 256) * (c.Npad / 128) >= 256 && KT <= kAsmLoopMaxStepsHost)
    return kCfgIgemm256x128;
  const int cfg = conv_pick_igemm_cfg(M, c.center_from > 0 ? c.center_from : c.Npad, c.kh * c.kw, c.kh * c.kw * c.in.C, may_split);
  // taps innermost: only for the stride-1 convolutions it has been measured and tested on (the stride-2 3x3 convolutions of SSD-512's
  // extra blocks keep the tap-major order).  Centre-tap-only column tiles of such a launch keep the tap-major walk of their one tap.
  if (cfg == kCfgIgemm256TapsInner && (c.stride != 1 || pick_pos_major(c, kCfgIgemm256, 256))) return kCfgIgemm256;
  if (cfg == kCfgIgemm128Early && c.kh * c.kw > 1 && c.up == 0 && c.stride == 1) {
    // taps innermost for the 128 x 128 tile too (consecutive steps re-read almost the same input lines): 3-6 % on launches that do
    // not split K (with split-K it loses: conv5_1 at batch 4 +10 %), and where no filter rows can be skipped instead
    const int tiles = ((M + 127) / 128) * (c.Npad / 128);
    const bool splits = may_split && conv_pick_splitk(tiles, KT, igemm_slots(kCfgIgemm128Early)) > 1;
    if (!splits && !pick_pos_major(c, kCfgIgemm128Early, 128)) return kCfgIgemm128EarlyTapsInner;
  }
  return cfg;
}

int launch_conv(const ConvLaunch& c, hipStream_t stream) {
  int cfg = c.cfg >= 0 ? c.cfg : conv_pick_cfg(c);
  RON_REQUIRE(cfg >= 0 && cfg < kNumCfgs, "conv: tile config %d out of range [0, %d)", cfg, kNumCfgs);
  if (conv_cfg_is_patch(cfg)) return launch_conv_patch(c, cfg, stream);
  if (cfg == kCfgC64Resident) return launch_conv_c64(c, stream);
  RON_REQUIRE(!conv_cfg_taps_inner(cfg) || c.up == 0, "conv: the taps-innermost order is for plain convolutions");
  const int esz = (int)dtype_size(c.dtype);
  const int chunk = conv_k_chunk(c.dtype);
  RON_REQUIRE(c.in.C % chunk == 0, "conv: Cin %d is not a multiple of the K chunk %d", c.in.C, chunk);
  RON_REQUIRE(c.in.pad >= c.cpad, "conv: input halo %d < conv padding %d", c.in.pad, c.cpad);
  RON_REQUIRE(c.in.bytes > 0 && c.in.bytes < (int64_t)1 << 32, "conv: input allocation must be < 4 GiB for buffer addressing");
  RON_REQUIRE(c.wgt_bytes > 0 && c.wgt_bytes < (int64_t)1 << 32, "conv: weight allocation must be < 4 GiB");
  RON_REQUIRE(c.out.pixels() * c.out.cstride < (int64_t)1 << 31, "conv: output too large for 32-bit offsets");
  const int K = c.kh * c.kw * c.in.C;
  const int M = c.in.N * c.Ho * c.Wo;
  const int BN = igemm_bn(cfg), BM = igemm_bm(cfg);
  RON_REQUIRE(c.Npad % BN == 0, "conv: Npad %d not a multiple of the N tile %d", c.Npad, BN);
  if (c.up > 0) RON_REQUIRE(c.up_cout % BN == 0, "transposed conv: channels per tap %d not a multiple of %d", c.up_cout, BN);
  ConvArgs a;
  fill_conv_args(c, &a);
  a.tiles_n = c.Npad / BN;
  a.tiles_total = ((M + BM - 1) / BM) * a.tiles_n;
  a.splitk = 1; a.kt_split = a.KT; a.partial = nullptr;
  if (c.pool) {
    RON_REQUIRE(c.up == 0 && c.res == nullptr && !c.out_f32 && c.Ho % 2 == 0 && c.Wo % 2 == 0 && c.stride == 1,
                "conv + fused pool: plain stride-1 conv on an even map only");
    RON_REQUIRE(c.out.H == c.Ho / 2 && c.out.W == c.Wo / 2, "conv + fused pool: output view must be the pooled map");
  }
  if (c.out2.base != nullptr) {
    RON_REQUIRE(c.pool && c.out2.H == c.Ho && c.out2.W == c.Wo && c.out2.C >= c.Cout && c.out2.N == c.in.N,
                "conv: a second output is the un-pooled map of a fused-pool launch");
    RON_REQUIRE(c.out2.pixels() * c.out2.cstride < (int64_t)1 << 31, "conv: second output too large for 32-bit offsets");
  }
  const int sk = c.splitk >= 0 ? c.splitk : conv_pick_splitk(a.tiles_total, a.KT, igemm_slots(cfg));
  if (sk > 1 && c.up == 0 && !c.pool && c.scratch != nullptr && (int64_t)sk * M * c.Npad * 4 <= c.scratch_bytes) {
    a.kt_split = (a.KT + sk - 1) / sk;
    a.splitk = (a.KT + a.kt_split - 1) / a.kt_split;      // no empty split
    a.partial = (float*)c.scratch;
    if (a.splitk == 1) { a.kt_split = a.KT; a.partial = nullptr; }
  }
  a.m_fastest = pick_m_fastest(c, BM, BN, a.tiles_total / a.tiles_n, a.tiles_n, a.splitk);
  a.pos_major = pick_pos_major(c, cfg, BM);
  a.taps_inner = conv_cfg_taps_inner(cfg) ? 1 : 0;
  // the 256 x 128 tile: taps innermost by the rule of the other tiles (stride-1 filters that neither split K nor skip filter rows)
  if (cfg == kCfgIgemm256x128 && c.kh * c.kw > 1 && c.stride == 1 && a.splitk == 1 && !a.pos_major) a.taps_inner = 1;
  if (c.center_from > 0)
    RON_REQUIRE(c.center_from % BN == 0 && (c.kh & 1) && (c.kw & 1) && c.up == 0,
                "conv: centre-tap-only columns need an odd filter and a boundary on the N tile (%d)", BN);
  RON_REQUIRE((int64_t)c.Npad * K * esz == c.wgt_bytes, "conv: packed weight size mismatch");
  int rc;
  if (c.dtype == RON_DTYPE_BF16) rc = launch_cfg<TraitsBF16S>(cfg, a, stream);
  else if (c.dtype == RON_DTYPE_F16) rc = launch_cfg<TraitsF16S>(cfg, a, stream);
  else if (c.dtype == RON_DTYPE_F32) rc = launch_cfg<TraitsF32S>(cfg, a, stream);
  else if (c.dtype == RON_DTYPE_F16X3) rc = launch_cfg<TraitsF16X3S>(cfg, a, stream);
  else { ron::set_error("conv: unknown dtype %d", c.dtype); return RON_ERR_INVALID; }
  if (rc != RON_OK || a.splitk == 1) return rc;
  if (c.dtype == RON_DTYPE_BF16) return launch_finalize<TraitsBF16S>(a, stream);
  if (c.dtype == RON_DTYPE_F16) return launch_finalize<TraitsF16S>(a, stream);
  if (c.dtype == RON_DTYPE_F16X3) return launch_finalize<TraitsF16X3S>(a, stream);
  return launch_finalize<TraitsF32S>(a, stream);
}

// ---- grouped launches ---------------------------------------------------------------------------------------------
namespace detail {

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

template <class Tr>
int launch_group_cfg(int cfg, const ConvGroupArgs& g, bool any_split, hipStream_t s) {
  if (cfg == kGroupMixed) return launch_group_mixed<Tr>(g, any_split, s);
  if (cfg == kCfgIgemm128x64) return launch_group_t<Tr, 128, 64, 2, 2, 2, 2>(g, any_split, s);
  if (cfg == kCfgIgemm128) return launch_group_t<Tr, 128, 128, 2, 2, 2, 2>(g, any_split, s);      // pieces issued early, as kCfgIgemm128Early
  if (cfg == kCfgIgemm256) {                                                                       // two launches of < 1 round each as one
    if constexpr (AsmLoop<Tr>::value) {
      int longest = 0;
      for (int k = 0; k < g.n; ++k) longest = std::max(longest, g.op[k].kt_split);
      if (asm_loop_ok(longest)) return launch_group_t<Tr, 256, 256, 2, 2, 2, 1>(g, any_split, s);
    }
    return launch_group_t<Tr, 256, 256, 4, 2, 2, 1>(g, any_split, s);
  }
  ron::set_error("conv group: tile config %d has no grouped form", cfg);
  return RON_ERR_INVALID;
}

// Split-K inside a group: the group as a whole fills the chip, so K is split only to bound the serial chain of K steps
// of one workgroup (the latency of the launch; <= ~24 steps each, >= 8) and only for convolutions with few tiles: the
// fp32 slabs of a many-tile convolution cost more HBM traffic than the shorter chain saves.
int group_pick_splitk(int KT, int tiles) {
  if (KT < 16 || tiles < 1) return 1;
  int sk = (KT + 23) / 24;
  if (sk > 512 / tiles) sk = 512 / tiles;
  if (sk > KT / 8) sk = KT / 8;
  return sk < 1 ? 1 : sk;
}

}  // namespace detail

// Split-K factors of the members of a group.  A member's own bound (group_pick_splitk) assumes the group fills the chip; when
// the whole group is short of that (small batches: every member is a few tiles), K is cut further so that the group's
// workgroups come to about the chip's slots, each with >= 8 K steps - what conv_pick_splitk does for a launch of its own.
// tile configuration member `c` of a group launched as `group_cfg` runs on
static int group_member_cfg(int group_cfg, const ConvLaunch& c) {
  return group_cfg == kGroupMixed ? (c.Npad % 128 == 0 ? kCfgIgemm128 : kCfgIgemm128x64) : group_cfg;
}

// Split-K factors of a mixed-width group (one dependency level of the heads: a large "carrier" member and latency-bound small
// ones) from a model of how the launch runs: workgroups are dispatched in blockIdx order onto 512 slots (two per CU for both
// 128-row tiles), a workgroup takes as long as its K steps.  For a common chunk length T (K steps per workgroup) every member
// is cut into ceil(KT / T) slices (>= 8 steps each; transposed convs, fused pools and forced factors stay as they are); the
// makespan of the resulting list schedule plus what the fp32 slabs cost (written and read once, in K-step units) is evaluated for
// a ladder of T, the cheapest wins.  What the fixed "<= 24 steps per workgroup" rule missed: {block7_conv_left, block6_conv_left}
// came to 808 workgroups of 116 / 24 steps = 1.6 rounds of the long ones; four slices of 144 steps are one round.
static long long group_makespan(const int* len, const int* cnt, int n, int slots) {
  // list scheduling in dispatch order (members as given, longest first is the caller's job); slot finish times in a min-heap
  std::vector<long long> heap(slots, 0);
  auto sift = [&](size_t i) {
    const size_t N = heap.size();
    for (;;) {
      size_t l = 2 * i + 1, r = l + 1, m = i;
      if (l < N && heap[l] < heap[m]) m = l;
      if (r < N && heap[r] < heap[m]) m = r;
      if (m == i) return;
      std::swap(heap[i], heap[m]);
      i = m;
    }
  };
  long long end = 0;
  for (int k = 0; k < n; ++k)
    for (int w = 0; w < cnt[k]; ++w) {
      heap[0] += len[k];
      end = std::max(end, heap[0]);
      sift(0);
    }
  return end;
}

static void group_splitks_scheduled(const ConvLaunch* ls, int n, int cfg, int* sk) {
  // workgroup slots of the chip and what a K step of a workgroup costs with the slots full (measured, tools/_tune in HISTORY.md:
  // 256 x 256: one per CU, 1.4 us; 128 x 128: two per CU, 1.15 us each; 128 x 64: 0.75 us); slab bytes the memory side moves per us
  const int slots = cfg == kCfgIgemm256 ? 256 : 512;
  const double kSlabBytesPerUs = 2.0e6;
  int tiles[kMaxConvGroup], KT[kMaxConvGroup];
  double step_us[kMaxConvGroup];
  bool fixed[kMaxConvGroup];
  for (int k = 0; k < n; ++k) {
    const ConvLaunch& c = ls[k];
    const int M = c.in.N * c.Ho * c.Wo;
    const int mcfg = group_member_cfg(cfg, c), BM = igemm_bm(mcfg);
    step_us[k] = mcfg == kCfgIgemm256 ? 1.4 : (mcfg == kCfgIgemm128 ? 1.15 : 0.75);
    KT[k] = c.kh * c.kw * c.in.C / conv_k_chunk(c.dtype);
    // centre-tap-only column tiles are a ninth of a tile: count them as that
    const int cols = c.Npad / igemm_bn(mcfg), cols_long = c.center_from > 0 ? c.center_from / igemm_bn(mcfg) : cols;
    tiles[k] = ((M + BM - 1) / BM) * cols_long + ((M + BM - 1) / BM) * (cols - cols_long) / (c.kh * c.kw);
    fixed[k] = c.up > 0 || c.pool || c.splitk >= 0 || KT[k] < 16;
    sk[k] = c.splitk >= 0 ? std::max(c.splitk, 1) : 1;
  }
  static const int ladder[] = {8, 12, 16, 24, 32, 36, 48, 64, 72, 96, 128, 144, 192, 256, 288, 384, 576, 1 << 20};
  double best = 1e30;
  int best_sk[kMaxConvGroup];
  for (int T : ladder) {
    int cand[kMaxConvGroup], len[kMaxConvGroup], cnt[kMaxConvGroup], order[kMaxConvGroup];
    double slab_us = 0;
    bool any = false;
    for (int k = 0; k < n; ++k) {
      cand[k] = fixed[k] ? sk[k] : std::max(1, std::min((KT[k] + T - 1) / T, KT[k] / 8));
      len[k] = (int)(((KT[k] + cand[k] - 1) / cand[k]) * step_us[k] * 20.0 + 0.5);      // workgroup duration in ticks of 0.05 us
      cnt[k] = tiles[k] * cand[k];
      order[k] = k;
      if (cand[k] > 1) {
        slab_us += 2.0 * cand[k] * (double)ls[k].in.N * ls[k].Ho * ls[k].Wo * ls[k].Npad * 4 / kSlabBytesPerUs;
        any = true;
      }
    }
    std::sort(order, order + n, [&](int a, int b) { return len[a] > len[b]; });
    int l2[kMaxConvGroup], c2[kMaxConvGroup];
    for (int k = 0; k < n; ++k) { l2[k] = len[order[k]]; c2[k] = cnt[order[k]]; }
    const double us = group_makespan(l2, c2, n, slots) * 0.05 + slab_us + (any ? 6.0 : 0.0);      // + the finalize launch
    if (us < best) { best = us; for (int k = 0; k < n; ++k) best_sk[k] = cand[k]; }
  }
  for (int k = 0; k < n; ++k) sk[k] = best_sk[k];
}

static void group_splitks(const ConvLaunch* ls, int n, int cfg, int* sk) {
  if (cfg == kGroupMixed || cfg == kCfgIgemm256) {
    // the schedule model assumes a launch that can fill the chip; below that (small batches: every member is a few tiles) the
    // per-member rule that follows measured better (batch 1, six levels: 271 vs 291 us)
    long long wg1 = 0;
    for (int k = 0; k < n; ++k) {
      const int mcfg = group_member_cfg(cfg, ls[k]);
      wg1 += (long long)((ls[k].in.N * ls[k].Ho * ls[k].Wo + igemm_bm(mcfg) - 1) / igemm_bm(mcfg)) * (ls[k].Npad / igemm_bn(mcfg));
    }
    if (wg1 * 2 >= (cfg == kCfgIgemm256 ? 256 : 512)) return group_splitks_scheduled(ls, n, cfg, sk);
  }
  const int slots = cfg == kCfgIgemm256 ? 256 : 512;     // workgroups the chip holds (256 x 256: one per CU, the 128-row tiles: two)
  int tiles[kMaxConvGroup], KT[kMaxConvGroup];
  long long steps = 0, wgs = 0;
  for (int k = 0; k < n; ++k) {
    const ConvLaunch& c = ls[k];
    const int M = c.in.N * c.Ho * c.Wo;
    KT[k] = c.kh * c.kw * c.in.C / conv_k_chunk(c.dtype);
    const int BM = igemm_bm(group_member_cfg(cfg, c)), BN = igemm_bn(group_member_cfg(cfg, c));
    tiles[k] = ((M + BM - 1) / BM) * (c.Npad / BN);
    sk[k] = (c.up > 0 || c.pool) ? 1 : (c.splitk >= 0 ? std::max(c.splitk, 1) : group_pick_splitk(KT[k], tiles[k]));
    steps += (long long)tiles[k] * KT[k];
    wgs += (long long)tiles[k] * sk[k];
  }
  if (wgs * 4 > slots * 3) return;                       // the group (nearly) fills the chip as it is
  const int per_wg = (int)std::max<long long>(8, (steps + slots - 1) / slots);      // K steps per workgroup to aim for
  for (int k = 0; k < n; ++k) {
    const ConvLaunch& c = ls[k];
    if (c.up > 0 || c.pool || c.splitk >= 0 || KT[k] < 16) continue;
    const int want = std::min(KT[k] / 8, std::max(1, KT[k] / per_wg));
    if (want > sk[k]) sk[k] = want;
  }
}

static int64_t group_slab_bytes(const ConvLaunch& c, int sk) {
  return sk > 1 ? ron::align_up((int64_t)sk * c.in.N * c.Ho * c.Wo * c.Npad * 4, 256) : 0;
}

int64_t conv_group_scratch_bytes(const ConvLaunch* ls, int n, int cfg, const int* sk_plan) {
  int64_t total = 0;
  if (cfg == kCfgPatch64) {                      // a pair of the patch kernel, or each launch on its own
    for (int k = 0; k < n; ++k) total = std::max(total, conv_scratch_bytes(ls[k]));
    return total;
  }
  if (n < 1 || n > kMaxConvGroup) return 0;
  int sk[kMaxConvGroup];
  if (sk_plan != nullptr) for (int k = 0; k < n; ++k) sk[k] = sk_plan[k];
  else group_splitks(ls, n, cfg, sk);
  for (int k = 0; k < n; ++k) total += group_slab_bytes(ls[k], sk[k]);
  return total;
}

// `n` mutually independent convolutions as one launch of tile configuration `cfg` (kCfgIgemm128x64, kCfgIgemm128, or kGroupMixed:
// each member on the 128-row tile of its own width);
// `scratch`: conv_group_scratch_bytes() for the split-K slabs (each conv gets its own part).
void conv_group_plan(const ConvLaunch* ls, int n, int cfg, int* sk) {
  if (cfg == kCfgPatch64 || n < 1 || n > kMaxConvGroup) { for (int k = 0; k < n && k < kMaxConvGroup; ++k) sk[k] = 1; return; }
  group_splitks(ls, n, cfg, sk);
}

int launch_conv_group(const ConvLaunch* ls_in, int n, int cfg, void* scratch, int64_t scratch_bytes, hipStream_t stream, const int* sk_plan) {
  RON_REQUIRE(n >= 1 && n <= kMaxGroup, "conv group: %d launches (1..%d)", n, kMaxGroup);
  if (cfg == kCfgPatch64) {
    // the two skinny heads of a scale: one launch of the patch kernel where it is the choice for both (dtype, map, batch),
    // otherwise each as the launch it would be on its own
    RON_REQUIRE(n == 2, "conv group: the patch-kernel form takes a pair");
    if (conv_patch_pair_applicable(ls_in[0], ls_in[1])) return launch_conv_patch_pair(ls_in[0], ls_in[1], stream);
    for (int k = 0; k < n; ++k) {
      const int rc = launch_conv(ls_in[k], stream);
      if (rc) return rc;
    }
    return RON_OK;
  }
  RON_REQUIRE(cfg == kCfgIgemm128x64 || cfg == kCfgIgemm128 || cfg == kGroupMixed || cfg == kCfgIgemm256,
              "conv group: tile config %d has no grouped form", cfg);
  ConvGroupArgs g = ConvGroupArgs();
  g.n = n;
  int64_t used = 0;
  bool any_split = false;
  int sks_in[kMaxConvGroup], sks[kMaxConvGroup], order[kMaxConvGroup];
  if (sk_plan != nullptr) for (int k = 0; k < n; ++k) sks_in[k] = sk_plan[k];
  else group_splitks(ls_in, n, cfg, sks_in);
  // workgroups are dispatched in blockIdx order: the members with the longest K chains per workgroup go first, the short ones fill
  // the slots the long ones leave (members are independent: their order is free)
  for (int k = 0; k < n; ++k) order[k] = k;
  if (cfg == kGroupMixed || cfg == kCfgIgemm256) {
    auto chain = [&](int k) {
      const ConvLaunch& c = ls_in[k];
      const int KT = c.kh * c.kw * c.in.C / conv_k_chunk(c.dtype);
      return (KT + sks_in[k] - 1) / sks_in[k];
    };
    std::stable_sort(order, order + n, [&](int a, int b) { return chain(a) > chain(b); });
  }
  ConvLaunch ls[kMaxConvGroup];
  for (int k = 0; k < n; ++k) { ls[k] = ls_in[order[k]]; sks[k] = sks_in[order[k]]; }
  for (int k = 0; k < n; ++k) {
    const ConvLaunch& c = ls[k];
    const int esz = (int)dtype_size(c.dtype), chunk = conv_k_chunk(c.dtype);
    const int mcfg = group_member_cfg(cfg, c), BM = igemm_bm(mcfg), BN = igemm_bn(mcfg);
    if (mcfg == kCfgIgemm128x64) g.narrow |= 1u << k;
    RON_REQUIRE(c.dtype == ls[0].dtype, "conv group: mixed dtypes");
    RON_REQUIRE(c.in.C % chunk == 0 && c.in.pad >= c.cpad, "conv group: bad input (Cin %d, halo %d < %d)", c.in.C, c.in.pad, c.cpad);
    RON_REQUIRE(c.in.bytes > 0 && c.in.bytes < (int64_t)1 << 32 && c.wgt_bytes > 0 && c.wgt_bytes < (int64_t)1 << 32,
                "conv group: allocations must be < 4 GiB");
    RON_REQUIRE(c.out.pixels() * c.out.cstride < (int64_t)1 << 31, "conv group: output too large for 32-bit offsets");
    RON_REQUIRE(c.Npad % BN == 0 && !c.pool, "conv group: Npad %d not a multiple of the N tile %d, or a fused pool", c.Npad, BN);
    if (c.up > 0) RON_REQUIRE(c.up_cout % BN == 0, "transposed conv: channels per tap %d not a multiple of %d", c.up_cout, BN);
    ConvArgs& a = g.op[k];
    fill_conv_args(c, &a);
    RON_REQUIRE((int64_t)c.Npad * a.K * esz == c.wgt_bytes, "conv group: packed weight size mismatch");
    a.tiles_n = c.Npad / BN;
    a.tiles_total = ((a.M + BM - 1) / BM) * a.tiles_n;
    const int64_t need = group_slab_bytes(c, sks[k]);
    if (need > 0 && scratch != nullptr && used + need <= scratch_bytes) {
      const int sk = sks[k];
      a.kt_split = (a.KT + sk - 1) / sk;
      a.splitk = (a.KT + a.kt_split - 1) / a.kt_split;
      a.partial = reinterpret_cast<float*>(static_cast<char*>(scratch) + used);
      if (a.splitk == 1) { a.kt_split = a.KT; a.partial = nullptr; }
      else { used += need; any_split = true; }
    }
    a.m_fastest = pick_m_fastest(c, BM, BN, a.tiles_total / a.tiles_n, a.tiles_n, a.splitk);
    a.pos_major = pick_pos_major(c, mcfg, BM);
    // K order of the member, by the rules of a launch of its own (conv_pick_cfg): taps innermost for stride-1 filters that neither
    // split K nor skip filter rows - on the 256 x 256 tile where the long columns are at most two tiles wide, on 128 x 128 always
    const int n_long = c.center_from > 0 ? c.center_from : c.Npad;
    a.taps_inner = (c.kh * c.kw > 1 && c.up == 0 && c.stride == 1 && a.splitk == 1 && !a.pos_major &&
                    (mcfg == kCfgIgemm128 || (mcfg == kCfgIgemm256 && n_long <= kTapsInnerMaxN))) ? 1 : 0;
    if (c.center_from > 0) RON_REQUIRE(c.center_from % BN == 0 && (c.kh & 1) && (c.kw & 1), "conv group: bad centre-tap-only columns");
  }
  // entries, longest K chain first (see ConvGroupArgs)
  struct Ent { int op, bid0, cnt, chain; };
  Ent ents[kMaxEntries];
  int ne = 0;
  for (int k = 0; k < n; ++k) {
    const ConvArgs& a = g.op[k];
    const int total = a.tiles_total * a.splitk;
    if (a.center_from_n > 0 && a.splitk == 1) {
      const int BN = igemm_bn(group_member_cfg(cfg, ls[k]));
      const int n_long = (a.tiles_total / a.tiles_n) * (a.center_from_n / BN);
      ents[ne++] = Ent{k, 0, n_long, a.KT};
      ents[ne++] = Ent{k, n_long, total - n_long, a.Cin / conv_k_chunk(ls[k].dtype)};
    } else {
      ents[ne++] = Ent{k, 0, total, a.kt_split};
    }
  }
  std::stable_sort(ents, ents + ne, [](const Ent& x, const Ent& y) { return x.chain > y.chain; });
  g.ne = ne;
  for (int e = 0; e < ne; ++e) {
    g.eop[e] = ents[e].op; g.ebid0[e] = ents[e].bid0; g.enwg[e] = g.op[ents[e].op].tiles_total * g.op[ents[e].op].splitk;
    g.first[e + 1] = g.first[e] + ents[e].cnt;
  }
  if (ls[0].dtype == RON_DTYPE_BF16) return launch_group_cfg<TraitsBF16S>(cfg, g, any_split, stream);
  if (ls[0].dtype == RON_DTYPE_F16) return launch_group_cfg<TraitsF16S>(cfg, g, any_split, stream);
  if (ls[0].dtype == RON_DTYPE_F32) return launch_group_cfg<TraitsF32S>(cfg, g, any_split, stream);
  if (ls[0].dtype == RON_DTYPE_F16X3) return launch_group_cfg<TraitsF16X3S>(cfg, g, any_split, stream);
  ron::set_error("conv group: unknown dtype %d", ls[0].dtype);
  return RON_ERR_INVALID;
}

int64_t conv_scratch_bytes(const ConvLaunch& c) {
  const int cfg = c.cfg >= 0 ? c.cfg : conv_pick_cfg(c);
  if (conv_cfg_is_patch(cfg) || c.up > 0 || c.pool) return 0;      // the halo-patch kernel never splits K
  const int M = c.in.N * c.Ho * c.Wo;
  const int KT = c.kh * c.kw * c.in.C / conv_k_chunk(c.dtype);
  const int tiles = ((M + igemm_bm(cfg) - 1) / igemm_bm(cfg)) * (c.Npad / igemm_bn(cfg));
  const int sk = c.splitk >= 0 ? c.splitk : conv_pick_splitk(tiles, KT, igemm_slots(cfg));
  return sk > 1 ? (int64_t)sk * M * c.Npad * 4 : 0;
}

}  // namespace ron
