// DIAGNOSTIC BUILD ONLY (make DIAG=1 -> libron_hip_diag.so, used by tools/sweep_conv.py --diag): the round-1 form of the
// row-gather kernel with every ablation / stamp / experimental switch of DESIGN.md 3.1 (64 tile configurations; ABL != 0
// builds compute wrong results on purpose).  Never linked into libron_hip.so.  Weights are row-major [Npad][K] here
// (pack.h, RON_DIAG), not blocked.
//
// Implicit-GEMM convolution for gfx950 (CDNA4):   out[m, n] = sum_{tap, c} in[pix(m) + tap, c] * w[n][tap, c]
//
//   M = N_img * Ho * Wo output pixels, N = Cout, K = kh*kw*Cin, NHWC activations with a zero halo
//   (no bounds checks in the loop), weights pre-packed [Cout][K] with K contiguous ("B transposed").
//
// One 256-thread workgroup (4 waves, 2x2) computes a BM x BN tile; each wave owns a (BM/2) x (BN/2)
// sub-tile as 32x32 MFMA accumulators.  Per K step both operands advance by one 128-byte row chunk
// (64 bf16/f16 or 32 f32 of one filter tap) which is staged global -> LDS by LDS-DMA
// (buffer_load ... lds, 16 B per lane), double buffered.  LDS rows are 128 B; the 16-byte chunk c of
// row r lives in slot c ^ ((r >> 1) & 7): the DMA writes linearly, so the permutation is applied on
// the per-lane *source* address and again on the ds_read_b128 side (conflict-free for the
// 16-lane groups of ds_read_b128).  The A operand is a row gather: row r of the tile is the
// Cin-chunk of input pixel pix(m0 + r) shifted by the tap, so the per-lane voffset is fixed for the
// whole K loop and the tap/chunk advance is a wave-uniform soffset.
//
// dtype variants share everything except the MFMA: bf16/f16 use v_mfma_f32_32x32x16_{bf16,f16}
// (one per 16 k), f32 uses v_mfma_f32_32x32x2_f32 (exact fp32 fma chain; the parity mode).
// Epilogue: + bias, ReLU, optional  relu(x + residual)  (reverse-connection sum), optional
// pixel-shuffle addressing (2x2 stride-2 transposed conv), store as dtype or fp32.
#include "conv_device.h"

namespace ron {
namespace detail {


// all of this wave's LDS reads retired (the stage about to be refilled is no longer being read) and all but
// its N youngest LDS-DMA transfers landed
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// Tile configuration: BM x BN block tile, WM x WN waves (each wave owns (BM/WM) x (BN/WN)), S LDS stages.
// The K loop keeps S-1 tiles in flight: one counted vmcnt + one raw s_barrier per K step, the LDS-DMA of
// tile kt+S-1 is issued right after the barrier into the stage whose reads finished before it.
// ABL (diagnostic builds only, outputs are wrong): 1 = no LDS-DMA (compute structure alone), 2 = no LDS reads / MFMA
// (data movement alone).
// RB = bytes of one LDS row = K extent of one stage (128: 64 bf16; 64: 32 bf16).  The shorter row halves the stage, so
// the same LDS holds twice the stages and the LDS-DMA of a tile gets S-1 compute periods of lead instead of one:
// bytes in flight per CU, not L2 bandwidth, is what bounds the staging stream ("bytes in flight").
template <class Tr, int BM, int BN, int WM, int WN, int S, int SPREAD, int ABL = 0, int RB = 128, int ROT = 0>
__global__ __launch_bounds__(WM * WN * 64, (S * (BM + BN) * RB > 80 * 1024) ? (WM * WN) / 4 : 2) void conv_igemm_kernel(ConvArgs p) {
  constexpr int kRowBytes = RB;                      // shadows the 128-byte default of conv_device.h
  constexpr int kLanesPerRow = RB / 16;              // 16-byte chunks per row
  constexpr int MT = Tr::kMT;                        // MFMA output tile: 32 (32x32x16) or 16 (16x16x32)
  constexpr int kGroups = 64 / MT;                   // 16-byte K groups one instruction consumes per row (2 or 4)
  constexpr int KS = kLanesPerRow / kGroups;         // MFMA k-steps per stage (one u32x4 fragment per lane and step)
  constexpr int EPA = MT * MT / 64;                  // accumulator registers per MFMA tile
  constexpr int kThreads = WM * WN * 64;
  constexpr int TM = BM / WM, TN = BN / WN;          // wave tile
  constexpr int MR = TM / MT, NR = TN / MT;          // MT x MT accumulators per wave: MR x NR
  constexpr int kRowsPerIt = kThreads / kLanesPerRow; // tile rows one LDS-DMA pass of the block covers
  constexpr int A_IT = BM / kRowsPerIt, B_IT = BN / kRowsPerIt;
  constexpr int LPT = A_IT + B_IT;                   // LDS-DMA instructions per thread and K step
  constexpr int kABytes = BM * kRowBytes, kBBytes = BN * kRowBytes;
  constexpr int kStage = kABytes + kBBytes;
  constexpr int kChunkElems = kRowBytes / Tr::kEsz;
  static_assert(BM % kRowsPerIt == 0 && BN % kRowsPerIt == 0 && kRowsPerIt % 16 == 0, "tile / thread-count mismatch");
  static_assert(TM % 32 == 0 && TN % 32 == 0 && S >= 2 && S <= 5, "bad wave tile / stage count");
  static_assert(NR <= 8, "vector epilogue: at most 8 channels per lane");
  static_assert(RB == 128 || RB == 64, "row bytes");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: [stage 0 .. S-1: A | B][in_off: BM ints][out_off: BM ints]
  int* s_in_off = reinterpret_cast<int*>(smem + S * kStage);
  int* s_out_off = s_in_off + BM;
  // ROT == 2 ("PF"): one more LDS-DMA per wave and K step that touches (4 bytes per lane) the 128-byte lines of the tile AFTER
  // the one being staged.  Its bytes land in a sink; what it buys is that the fabric round trip of lines that miss L2 happens a
  // whole stage period earlier, so the real pieces - which have one stage period to land, all LDS leaves room for - find them in L2.
  constexpr int PF = ROT == 2 ? 1 : 0;
  char* s_sink = reinterpret_cast<char*>(s_out_off + BM);
  (void)s_sink;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware tile order: workgroups that share an XCD (blockIdx % 8) take consecutive tiles,
  // so the N-tiles that re-read one A tile hit the same L2.
  const unsigned nwg = gridDim.x;
  const unsigned bid = blockIdx.x;
  const unsigned xcd = bid & 7u, q = nwg >> 3, r8 = nwg & 7u;
  const unsigned wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
  const int zsplit = (int)(wgid / (unsigned)p.tiles_total);
  const unsigned tile = wgid - (unsigned)zsplit * (unsigned)p.tiles_total;
  const int tile_n = (int)(tile % (unsigned)p.tiles_n);
  const int tile_m = (int)(tile / (unsigned)p.tiles_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kt0 = zsplit * p.kt_split;
  const int kt1 = min(p.KT, kt0 + p.kt_split);

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
    } else {
      int m = m0 + r;
      valid = m < p.M;
      m = valid ? m : p.M - 1;
      const int hw = p.Ho * p.Wo;
      img = m / hw;
      const int rem = m - img * hw;
      oy = rem / p.Wo;
      ox = rem - oy * p.Wo;
      const int os = p.up > 0 ? p.up : 1;
      off = ((img * p.out_Hp + oy * os + p.out_pad) * p.out_Wp + ox * os + p.out_pad) * p.out_cstride + p.out_coff;
    }
    const int iy = oy * p.stride + p.in_org, ix = ox * p.stride + p.in_org;
    s_in_off[r] = (int)((((unsigned)(img * p.in_Hp + iy) * p.in_Wp + ix) * p.in_cstride + p.in_coff) * Tr::kEsz);
    s_out_off[r] = valid ? off : -1;
  }
  __syncthreads();

  // LDS-DMA source offsets (bytes): thread -> (row = it*kRowsPerIt + tid/8, slot = tid%8), source chunk = slot ^ key(row)
  // swizzle key of row r: (r >> 1) & 7 for 128-byte rows, (r >> 2) & 3 for 64-byte rows (both conflict-free for the
  // 16-lane groups of ds_read_b128: a group's rows differ in (r & 1 | r & 3) or in the key)
  const int ld_row = tid / kLanesPerRow;
  const int ld_chunk = RB == 128 ? (tid & 7) ^ ((tid >> 4) & 7) : (tid & 3) ^ ((tid >> 4) & 3);
  // fixed-size arrays on purpose: with a template-dependent bound the LDS-DMA builtin's voffset becomes a
  // type-dependent expression and hipcc (ROCm 7.2) silently drops the kernel's host stub.
  int a_voff[8], b_voff[8];
  const int hot_voff = (tid & 63) * 16;
  (void)hot_voff;
  static_assert(A_IT <= 8 && B_IT <= 8, "tile too large");
#pragma unroll
  for (int it = 0; it < A_IT; ++it) a_voff[it] = s_in_off[it * kRowsPerIt + ld_row] + ld_chunk * 16;
  // B rows are permuted on the way in: LDS row (j*MT + r) of a wave's TN-wide group holds weight row (r*NR + j), so
  // that MFMA column r of the wave's j-th MT-column tile is output channel r*NR + j: a lane's NR accumulators are NR
  // adjacent channels and the epilogue stores them as one contiguous NR-element vector (full 128-B lines per row).
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int lrow = it * kRowsPerIt + ld_row;
    const int grp = lrow / TN, loc = lrow % TN;
    const int nrow = grp * TN + (loc % MT) * NR + (loc / MT);
    b_voff[it] = (n0 + nrow) * p.K * Tr::kEsz + ld_chunk * 16;
  }

  // prefetch line of this lane: tile row `tid` of A (waves below BM / 64) or of B (the next BN / 64 waves)
  const bool pf_a = wave * 64 < BM, pf_any = wave * 64 < BM + BN;
  int pf_voff = 0;
  if (PF) pf_voff = pf_a ? s_in_off[min(tid, BM - 1)] : (n0 + min(tid - BM, BN - 1)) * p.K * Tr::kEsz;
  // K-step bookkeeping (wave-uniform): tap (ky, kx) and channel chunk cc of the NEXT tile to stage
  const int chunks_per_tap = p.Cin / kChunkElems;
  const int tap0 = kt0 / chunks_per_tap;
  int ky = tap0 / p.kw, kx = tap0 - (tap0 / p.kw) * p.kw, cc = (kt0 - tap0 * chunks_per_tap) * kChunkElems;
  // One tile = LPT LDS-DMA pieces per thread (A pieces first).  RON_STAGE_BEGIN computes the wave-uniform
  // part once per tile, RON_STAGE_PIECE issues piece i (compile-time), RON_STAGE_END advances the tap.
#define RON_STAGE_BEGIN(kt_)                                                                                         \
    /* past the last tile: zero-record descriptors, the DMA moves nothing but keeps the vmcnt bookkeeping uniform */  \
    const bool live_ = ABL != 4 && (kt_) < kt1;                                                                               \
    const __amdgpu_buffer_rsrc_t rs_a =                                                                              \
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (live_ && ABL != 6) ? p.in_bytes : 0u, 0x00020000); \
    const __amdgpu_buffer_rsrc_t rs_b =                                                                              \
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, (live_ && ABL != 7) ? p.wgt_bytes : 0u, 0x00020000); \
    const int a_soff = ((ky * p.dil * p.in_Wp + kx * p.dil) * p.in_cstride + cc) * Tr::kEsz;                         \
    const int b_soff = (kt_) * kRowBytes;                                                                            \
    char* dst = smem + (((kt_) - kt0) % S) * kStage + wave * 1024;
#define RON_STAGE_PIECE(i_)                                                                                          \
    do {                                                                                                             \
      if (ABL == 1 || ABL == 3) break;                                                                                        \
      if (ABL == 9) { /* timing-only: every piece re-reads the same 1 KB (L1 hits): the LDS side of the stream alone */ \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(dst + (i_) * kRowsPerIt * kRowBytes), 16, hot_voff, 0, 0, 0); \
        break;                                                                                                       \
      }                                                                                                              \
      if ((i_) < A_IT)                                                                                               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(dst + (i_) * kRowsPerIt * kRowBytes), 16,         \
                                                 a_voff[(i_) < A_IT ? (i_) : 0], a_soff, 0, ABL == 11 ? 2 : 0);      \
      else                                                                                                           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(dst + kABytes + ((i_) - A_IT) * kRowsPerIt * kRowBytes), 16, \
                                                 b_voff[(i_) >= A_IT ? (i_) - A_IT : 0], b_soff, 0,                  \
                                                 (ABL == 10 || ABL == 11 || ABL == 12) ? 2 : 0);                     \
    } while (0)
#define RON_STAGE_END()                                                                                              \
    do {                                                                                                             \
      if (ABL == 8 || ABL == 12) { /* timing-only: taps innermost, channel chunk outermost (weights are not in that order) */     \
        if (++kx == p.kw) { kx = 0; if (++ky * p.kw >= p.KT * kChunkElems / p.Cin) { ky = 0; cc += kChunkElems; } }  \
        break;                                                                                                       \
      }                                                                                                              \
      cc += kChunkElems;                                                                                             \
      if (cc >= p.Cin) {                                                                                             \
        cc = 0;                                                                                                      \
        if (++kx == p.kw) { kx = 0; ++ky; }                                                                          \
      }                                                                                                              \
    } while (0)

  // after RON_STAGE_END the state (ky, kx, cc) describes tile tpf_ = the one after the tile just staged
#define RON_PREFETCH(tpf_)                                                                                           \
    do {                                                                                                             \
      if (!PF) break;                                                                                                \
      const bool lv_ = (tpf_) < kt1 && pf_any;                                                                       \
      const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                                          \
          const_cast<void*>(pf_a ? p.in : p.wgt), 0, lv_ ? (pf_a ? p.in_bytes : p.wgt_bytes) : 0u, 0x00020000);      \
      const int so_ = pf_a ? ((ky * p.dil * p.in_Wp + kx * p.dil) * p.in_cstride + cc) * Tr::kEsz : (tpf_) * kRowBytes; \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_void*)(s_sink + wave * 256), 4, pf_voff, so_, 0, 0);        \
    } while (0)

  typename Tr::acc_t acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < EPA; ++e) acc[i][j][e] = 0.f;

  // fragment read offsets: lane -> row r = lane % MT, K group h = lane / MT; step s reads chunk kGroups*s + h
  const int fr = lane & (MT - 1), fh = lane / MT;
  int rd_off[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) rd_off[s] = fr * kRowBytes + (((kGroups * s + fh) ^ (RB == 128 ? (fr >> 1) & 7 : (fr >> 2) & 3)) << 4);
  const int a_base = wm * TM * kRowBytes;
  const int b_base = kABytes + wn * TN * kRowBytes;

  // prologue: S-1 tiles in flight
#pragma unroll
  for (int t = 0; t < S - 1; ++t) {
    RON_STAGE_BEGIN(kt0 + t)
#pragma unroll
    for (int i = 0; i < LPT; ++i) RON_STAGE_PIECE(i);
    RON_STAGE_END();
    RON_PREFETCH(kt0 + t + 1);
  }

  // ABL 5: s_memtime stamps around the wait, the barrier and the rest of the K step (shares, not run time)
  unsigned long long t_wait = 0, t_bar = 0, t_comp = 0, t_a = 0, t_b = 0, t_c = 0, t_d = 0;
#define RON_STAMP(t_)                                                                      \
    do {                                                                                   \
      if (ABL == 5) {                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
        __builtin_amdgcn_sched_barrier(0);                                                 \
      }                                                                                    \
    } while (0)
  if (ROT == 1) {
    // Rotated K loop: the wait + barrier of tile kt+1 sit in front of the LAST k-step of tile kt, whose fragments are
    // already in registers.  After the barrier the MFMAs of that k-step restart at once, and in their shadow go (a) the
    // first fragment reads of tile kt+1 (the un-rotated loop pays them as an LDS burst with idle matrix cores at the top
    // of every tile) and (b) the LDS-DMA of tile kt+S into the stage tile kt just vacated.  All S stages hold tiles.
    {
      RON_STAGE_BEGIN(kt0 + S - 1)
#pragma unroll
      for (int i = 0; i < LPT; ++i) RON_STAGE_PIECE(i);
      RON_STAGE_END();
    }
    wait_vmcnt<(S - 1) * LPT>();
    __builtin_amdgcn_s_barrier();
    u32x4 fa[2][MR], fb[2][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i) fa[0][i] = *reinterpret_cast<const u32x4*>(smem + a_base + i * MT * kRowBytes + rd_off[0]);
#pragma unroll
    for (int j = 0; j < NR; ++j) fb[0][j] = *reinterpret_cast<const u32x4*>(smem + b_base + j * MT * kRowBytes + rd_off[0]);
    constexpr int RD = MR + NR, MM = MR * NR * Tr::kMfmaPerMma;
    for (int kt = kt0; kt < kt1; ++kt) {
      const char* sbuf = smem + ((kt - kt0) % S) * kStage;
      const char* snext = smem + ((kt + 1 - kt0) % S) * kStage;
#pragma unroll
      for (int s = 0; s < KS - 1; ++s) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
          fa[(s + 1) & 1][i] = *reinterpret_cast<const u32x4*>(sbuf + a_base + i * MT * kRowBytes + rd_off[s + 1]);
#pragma unroll
        for (int j = 0; j < NR; ++j)
          fb[(s + 1) & 1][j] = *reinterpret_cast<const u32x4*>(sbuf + b_base + j * MT * kRowBytes + rd_off[s + 1]);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < NR; ++j) Tr::mma(fa[s & 1][i], fb[s & 1][j], acc[i][j]);
      }
#pragma unroll
      for (int s = 0; s < KS - 1; ++s) {
#pragma unroll
        for (int q = 0; q < MM; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (q < RD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (RD > MM) __builtin_amdgcn_sched_group_barrier(0x100, RD - MM, 0);
      }
      RON_STAMP(t_a);
      wait_vmcnt<(S - 2) * LPT>();          // tile kt+1 has landed; this wave's reads of tile kt are in registers
      RON_STAMP(t_b);
      __builtin_amdgcn_s_barrier();
      RON_STAMP(t_c);
      RON_STAGE_BEGIN(kt + S)
#pragma unroll
      for (int i = 0; i < MR; ++i) fa[KS & 1][i] = *reinterpret_cast<const u32x4*>(snext + a_base + i * MT * kRowBytes + rd_off[0]);
#pragma unroll
      for (int j = 0; j < NR; ++j) fb[KS & 1][j] = *reinterpret_cast<const u32x4*>(snext + b_base + j * MT * kRowBytes + rd_off[0]);
#pragma unroll
      for (int i = 0; i < LPT; ++i) RON_STAGE_PIECE(i);
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) Tr::mma(fa[(KS - 1) & 1][i], fb[(KS - 1) & 1][j], acc[i][j]);
#pragma unroll
      for (int q = 0; q < MM; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (q < RD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (((q + 1) * LPT) / MM > (q * LPT) / MM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      if (RD > MM) __builtin_amdgcn_sched_group_barrier(0x100, RD - MM, 0);
#pragma unroll
      for (int x = 0; x < 16; ++x)
        if (x < LPT - MM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      RON_STAGE_END();
      RON_STAMP(t_d);
      if (ABL == 5) { t_wait += t_b - t_a; t_bar += t_c - t_b; t_comp += t_d - t_c; }
    }
  } else
  for (int kt = kt0; kt < kt1; ++kt) {
    RON_STAMP(t_a);
    if (ABL != 3) {
    wait_vmcnt<(S - 2) * (LPT + PF) + PF>();   // this wave's share of tile kt has landed (a younger prefetch may be out)
    RON_STAMP(t_b);
    __builtin_amdgcn_s_barrier();           // ... everyone's has, and everyone is done reading tile kt-1
    RON_STAMP(t_c);
    }
    // refill the stage tile kt-1 occupied; the LPT pieces are spread over the four k-steps below so that
    // their issue slots fall into the MFMA shadow instead of ahead of it (SPREAD) or are issued up front
    RON_STAGE_BEGIN(kt + S - 1)
    if (!SPREAD) {
#pragma unroll
      for (int i = 0; i < LPT; ++i) RON_STAGE_PIECE(i);
    }
    const char* sbuf = smem + ((kt - kt0) % S) * kStage;
    // fragments of k-step s+1 are read while the MFMAs of k-step s run (two register sets)
    u32x4 fa[2][MR], fb[2][NR];
    if (ABL == 2) {
      if (SPREAD) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) RON_STAGE_PIECE(i);
      }
      RON_STAGE_END();
      continue;
    }
#pragma unroll
    for (int i = 0; i < MR; ++i) fa[0][i] = *reinterpret_cast<const u32x4*>(sbuf + a_base + i * MT * kRowBytes + rd_off[0]);
#pragma unroll
    for (int j = 0; j < NR; ++j) fb[0][j] = *reinterpret_cast<const u32x4*>(sbuf + b_base + j * MT * kRowBytes + rd_off[0]);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s < KS - 1) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
          fa[(s + 1) & 1][i] = *reinterpret_cast<const u32x4*>(sbuf + a_base + i * MT * kRowBytes + rd_off[(s + 1) % KS]);
#pragma unroll
        for (int j = 0; j < NR; ++j)
          fb[(s + 1) & 1][j] = *reinterpret_cast<const u32x4*>(sbuf + b_base + j * MT * kRowBytes + rd_off[(s + 1) % KS]);
      }
      if (SPREAD) {
#pragma unroll
        for (int i = 0; i < LPT; ++i)
          if ((SPREAD == 2 ? 0 : (i * KS) / LPT) == s) RON_STAGE_PIECE(i);
      }
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) Tr::mma(fa[s & 1][i], fb[s & 1][j], acc[i][j]);
    }
    // Pin the issue order (hipcc otherwise sinks the next k-step's fragment reads below the MFMAs to save
    // registers): R0 | per k-step: MFMAs with the next k-step's reads one per MFMA gap and this k-step's LDS-DMA pieces
    // spaced evenly between them (SPREAD 1: a tile's pieces are shared out over the k-steps; SPREAD 2: all of them go
    // out during k-step 0, so the last one has most of a stage period to land) | MFMAs of the last k-step.
    {
      constexpr int RD = MR + NR, MM = MR * NR * Tr::kMfmaPerMma;
      __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
      if (SPREAD == 0) __builtin_amdgcn_sched_group_barrier(0x020, LPT, 0);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int first = SPREAD == 2 ? 0 : (s * LPT + KS - 1) / KS;                       // pieces [first, last) go out in k-step s
        const int last = SPREAD == 2 ? (s == 0 ? LPT : 0) : ((s + 1) * LPT + KS - 1) / KS;
        const int ps = SPREAD ? last - first : 0;
#pragma unroll
        for (int q = 0; q < MM; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (s < KS - 1 && q < RD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (((q + 1) * ps) / MM > (q * ps) / MM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        if (s < KS - 1 && RD > MM) __builtin_amdgcn_sched_group_barrier(0x100, RD - MM, 0);
#pragma unroll
        for (int x = 0; x < 16; ++x)
          if (x < ps - MM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    RON_STAGE_END();
    RON_PREFETCH(kt + S);
    RON_STAMP(t_d);
    if (ABL == 5) { t_wait += t_b - t_a; t_bar += t_c - t_b; t_comp += t_d - t_c; }
  }
  if (ABL == 5 && p.dbg != nullptr && lane == 0) {
    unsigned long long* d = p.dbg + ((size_t)blockIdx.x * (kThreads / 64) + wave) * 4;
    d[0] = t_wait; d[1] = t_bar; d[2] = t_comp; d[3] = (unsigned long long)(kt1 - kt0);
  }
#undef RON_STAMP
#undef RON_PREFETCH
#undef RON_STAGE_BEGIN
#undef RON_STAGE_PIECE
#undef RON_STAGE_END

  // epilogue.  C/D layout of the MFMA: column = lane % MT, row = (e & 3) + 8 * (e >> 2) + 4 * (lane / MT)
  // (32x32: e < 16; 16x16: e < 4, the e >> 2 term vanishes)
  int tap_off = 0, n_base = n0;
  if (p.up > 0) {
    const int tap = n0 / p.up_cout;                       // BN divides up_cout: uniform per tile
    tap_off = ((tap / p.up) * p.out_Wp + (tap % p.up)) * p.out_cstride;
    n_base = n0 - tap * p.up_cout;
  }
  // lane -> NR adjacent output channels starting at wn*TN + fr*NR
  const int nloc = wn * TN + fr * NR;
  if (p.splitk > 1) {
    float* slab = p.partial + (size_t)zsplit * p.M * p.Npad;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
#pragma unroll
      for (int e = 0; e < EPA; ++e) {
        const int rt = wm * TM + i * MT + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (s_out_off[rt] < 0) continue;
        float v[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) v[j] = acc[i][j][e];
        store_f32_vec<NR>(slab + (size_t)(m0 + rt) * p.Npad + n0 + nloc, v);
      }
    }
    return;
  }
  float bias_v[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) bias_v[j] = p.bias[n0 + nloc + j];
  const int n_valid = p.Cout - (n0 + nloc);            // channels of this lane's group that exist (may be <= 0)
  const int ncol0 = n_base + nloc;
  if (p.pool) {
    // max over the 2x2 window = max over registers 4t..4t+3; relu(max(x) + b) == max(relu(x + b))
#pragma unroll
    for (int i = 0; i < MR; ++i) {
#pragma unroll
      for (int t = 0; t < EPA / 4; ++t) {
        const int ooff = s_out_off[wm * TM + i * MT + 8 * t + 4 * fh];
        if (ooff < 0 || n_valid <= 0) continue;
        float v[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const float mx = fmaxf(fmaxf(acc[i][j][4 * t], acc[i][j][4 * t + 1]), fmaxf(acc[i][j][4 * t + 2], acc[i][j][4 * t + 3]));
          v[j] = mx + bias_v[j];
          if (p.relu) v[j] = fmaxf(v[j], 0.f);
        }
        const int o = ooff + ncol0;
        if (n_valid >= NR) {
          Tr::template store_vec<NR>(p.out, o, v);
        } else {
#pragma unroll
          for (int j = 0; j < NR; ++j) if (j < n_valid) Tr::store(p.out, o + j, v[j]);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < MR; ++i) {
#pragma unroll
    for (int e = 0; e < EPA; ++e) {
      const int rt = wm * TM + i * MT + (e & 3) + 8 * (e >> 2) + 4 * fh;
      const int ooff = s_out_off[rt];
      if (ooff < 0 || n_valid <= 0) continue;
      const int o = ooff + tap_off + ncol0;
      float v[NR];
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        v[j] = acc[i][j][e] + bias_v[j];
        if (p.relu) v[j] = fmaxf(v[j], 0.f);
      }
      if (n_valid >= NR) {
        if (p.res != nullptr) {
          float rv[NR];
          Tr::template load_vec<NR>(p.res, o, rv);
#pragma unroll
          for (int j = 0; j < NR; ++j) v[j] = fmaxf(v[j] + rv[j], 0.f);
        }
        if (p.out_f32) store_f32_vec<NR>(reinterpret_cast<float*>(p.out) + o, v);
        else Tr::template store_vec<NR>(p.out, o, v);
      } else {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          if (j >= n_valid) break;
          float x = v[j];
          if (p.res != nullptr) x = fmaxf(x + Tr::load(p.res, o + j), 0.f);
          if (p.out_f32) reinterpret_cast<float*>(p.out)[o + j] = x;
          else Tr::store(p.out, o + j, x);
        }
      }
    }
  }
}

// Adds the split-K slabs and applies the conv epilogue (bias, ReLU, relu(x + residual), dtype / fp32 store).
template <class Tr>
__global__ void splitk_finalize_kernel(ConvArgs p) {
  const int groups = p.Npad / 4;
  const long long total = (long long)p.M * groups;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % groups);
    const int m = (int)(idx / groups);
    const int n = g * 4;
    if (n >= p.Cout) continue;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < p.splitk; ++z)
      sum += *reinterpret_cast<const f32x4*>(p.partial + ((size_t)z * p.M + m) * p.Npad + n);
    const int hw = p.Ho * p.Wo;
    const int img = m / hw;
    const int rem = m - img * hw;
    const int oy = rem / p.Wo, ox = rem - (rem / p.Wo) * p.Wo;
    const int o = ((img * p.out_Hp + oy + p.out_pad) * p.out_Wp + ox + p.out_pad) * p.out_cstride + p.out_coff + n;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (n + j >= p.Cout) break;
      float v = sum[j] + p.bias[n + j];
      if (p.relu) v = fmaxf(v, 0.f);
      if (p.res != nullptr) v = fmaxf(v + Tr::load(p.res, o + j), 0.f);
      if (p.out_f32) reinterpret_cast<float*>(p.out)[o + j] = v;
      else Tr::store(p.out, o + j, v);
    }
  }
}

struct TileCfg { int bm, bn, wm, wn, stages, spread, rb = 128, mt = 32; };
// index = ConvLaunch.cfg
constexpr TileCfg kCfgs[] = {
    {128, 128, 2, 2, 2, 0},   // 0: 64 KB LDS, 2 workgroups / CU (round-1 baseline structure)
    {128, 64, 2, 2, 2, 0},    // 1
    {256, 128, 4, 2, 3, 0},   // 2: 144 KB, 8 waves, 2 tiles in flight
    {256, 64, 4, 2, 3, 0},    // 3
    {128, 128, 2, 2, 2, 1},   // 4: as 0, DMA issue spread over the k-steps
    {128, 64, 2, 2, 2, 1},    // 5
    {256, 128, 4, 2, 3, 1},   // 6
    {256, 64, 4, 2, 3, 1},    // 7
    {256, 256, 2, 4, 2, 0},   // 8: 128 KB, 8 waves, wave tile 128 x 64 (the CDNA4 guide's 256^2 geometry)
    {256, 256, 4, 2, 2, 0},   // 9: wave tile 64 x 128
    {256, 256, 2, 4, 2, 1},   // 10
    {256, 256, 4, 2, 2, 1},   // 11
    {256, 256, 4, 2, 2, 1},   // 12: diagnostic, 11 without LDS-DMA
    {256, 256, 4, 2, 2, 1},   // 13: diagnostic, 11 without LDS reads / MFMA
    {128, 128, 2, 2, 2, 1},   // 14: diagnostic, 4 without LDS-DMA
    {128, 128, 2, 2, 2, 1},   // 15: diagnostic, 4 without LDS reads / MFMA
    {256, 256, 2, 2, 2, 1},   // 16: 4 waves, one per SIMD, wave tile 128 x 128 (256 accumulator registers)
    {256, 256, 2, 2, 2, 1},   // 17: diagnostic, 16 without LDS-DMA
    {256, 256, 4, 2, 2, 1},   // 18: diagnostic, 11 without LDS-DMA, waits and barriers (free-running ds_read + MFMA)
    {256, 256, 4, 2, 2, 1},   // 19: diagnostic, 11 with zero-record descriptors (DMA instructions issue, nothing moves)
    {256, 256, 4, 2, 4, 1, 64},   // 20: 11 with 64-byte rows: 4 stages of 32 KB, three tiles (96 KB) in flight
    {256, 256, 2, 4, 4, 1, 64},   // 21: 10 likewise
    {256, 128, 4, 2, 5, 1, 64},   // 22: 6 likewise: 5 stages of 24 KB
    {128, 128, 2, 2, 4, 1, 64},   // 23: 4 likewise: 4 stages of 16 KB, 2 workgroups / CU
    {128, 64, 2, 2, 4, 1, 64},    // 24: 5 likewise
    {256, 256, 4, 2, 4, 1, 64},   // 25: diagnostic, 20 without LDS reads / MFMA (staging stream alone)
    {128, 128, 2, 2, 4, 0, 64},   // 26: 23 with the pieces issued up front
    {256, 256, 4, 2, 2, 1, 128, 16},   // 27: diagnostic, 30 with s_memtime stamps (ConvLaunch.dbg)
    {128, 128, 2, 2, 2, 1, 128, 16},   // 28: diagnostic, 33 with stamps
    {256, 128, 4, 2, 3, 1},       // 29: diagnostic, 6 with stamps
    {256, 256, 4, 2, 2, 1, 128, 16},   // 30: 11 on 16x16x32 MFMAs
    {256, 256, 2, 4, 2, 1, 128, 16},   // 31: 10 likewise
    {256, 128, 4, 2, 3, 1, 128, 16},   // 32: 6 likewise
    {128, 128, 2, 2, 2, 1, 128, 16},   // 33: 4 likewise
    {128, 64, 2, 2, 2, 1, 128, 16},    // 34: 5 likewise
    {256, 256, 4, 2, 2, 2, 128, 16},   // 35: 30 with all LDS-DMA pieces of a tile issued during k-step 0
    {128, 128, 2, 2, 2, 2, 128, 16},   // 36: 33 likewise
    {128, 64, 2, 2, 2, 2, 128, 16},    // 37: 34 likewise
    {256, 256, 2, 4, 2, 2, 128, 16},   // 38: 31 likewise
    {256, 256, 4, 2, 2, 2, 128, 16},   // 39: 35 with the rotated K loop
    {128, 128, 2, 2, 2, 2, 128, 16},   // 40: 36 likewise
    {128, 64, 2, 2, 2, 2, 128, 16},    // 41: 37 likewise
    {256, 256, 2, 4, 2, 2, 128, 16},   // 42: 38 likewise
    {256, 256, 4, 2, 2, 2, 128, 32},   // 43: rotated loop on 32x32x16 MFMAs (four k-steps per stage)
    {256, 256, 4, 2, 2, 2, 128, 16},   // 44: diagnostic, 39 with stamps
    {256, 256, 4, 2, 2, 1, 128, 16},   // 45: diagnostic, 30 without LDS-DMA
    {256, 256, 4, 2, 2, 1, 128, 16},   // 46: diagnostic, 30 without LDS-DMA, waits and barriers
    {256, 256, 4, 2, 2, 1, 128, 16},   // 47: diagnostic, 30 without LDS reads / MFMA
    {256, 256, 4, 2, 2, 1, 128, 16},   // 48: diagnostic, 30 with zero-record descriptors
    {256, 256, 4, 2, 2, 1, 128, 16},   // 49: diagnostic, 30 with a zero-record A descriptor (only the weights move)
    {256, 256, 4, 2, 2, 1, 128, 16},   // 50: diagnostic, 30 with a zero-record B descriptor (only the activations move)
    {256, 64, 4, 2, 3, 2, 128, 16},    // 51: 256 x 64, three stages, 16x16 MFMAs (Cout 64)
    {256, 64, 8, 1, 3, 2, 128, 16},    // 52: likewise, wave tile 32 x 64
    {256, 128, 4, 2, 2, 2, 128, 16},   // 53: 256 x 128, two stages (96 KB)
    {256, 256, 4, 2, 2, 1, 128, 16},   // 54: diagnostic (timing only), 30 with the taps innermost in the K order
    {256, 256, 4, 2, 2, 1, 128, 16},   // 55: diagnostic (timing only), 30 with every LDS-DMA piece reading the same 1 KB
    {256, 256, 4, 2, 2, 1, 128, 16},   // 56: 30 + L2 prefetch of the tile after next (ROT 2)
    {128, 128, 2, 2, 2, 1, 128, 16},   // 57: 33 likewise
    {128, 128, 2, 2, 2, 2, 128, 16},   // 58: 36 likewise
    {128, 64, 2, 2, 2, 2, 128, 16},    // 59: 37 likewise
    {256, 256, 4, 2, 2, 2, 128, 16},   // 60: 35 likewise
    {256, 256, 4, 2, 2, 1, 128, 16},   // 61: 30 with non-temporal (aux 2) weight loads            (results valid)
    {256, 256, 4, 2, 2, 1, 128, 16},   // 62: 30 with non-temporal weight AND activation loads    (results valid)
    {256, 256, 4, 2, 2, 1, 128, 16},   // 63: diagnostic (timing only): taps innermost + non-temporal weight loads
};
constexpr int kNumDiagCfgs = (int)(sizeof(kCfgs) / sizeof(kCfgs[0]));
constexpr int kCfgPatchDiag = 100;      // the round-1 halo-patch kernel (diag/conv_patch_diag.hip)
// workgroups of configuration i the chip holds at once (256 CUs; 64 KB of LDS lets two share a CU)
inline int cfg_slots(int i) { return kCfgs[i].stages * (kCfgs[i].bm + kCfgs[i].bn) * kCfgs[i].rb <= 80 * 1024 ? 512 : 256; }

template <class Tr, int BM, int BN, int WM, int WN, int S, int SPREAD, int ABL = 0, int RB = 128, int ROT = 0>
int launch_t(const ConvArgs& a, hipStream_t s) {
  const size_t lds = (size_t)S * (BM + BN) * RB + 2 * BM * sizeof(int) + (ROT == 2 ? WM * WN * 256 : 0);
  static PerDeviceOnce once;
  RON_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void*>(&conv_igemm_kernel<Tr, BM, BN, WM, WN, S, SPREAD, ABL, RB, ROT>), (int)lds));
  hipLaunchKernelGGL((conv_igemm_kernel<Tr, BM, BN, WM, WN, S, SPREAD, ABL, RB, ROT>), dim3(a.tiles_total * a.splitk), dim3(WM * WN * 64), lds, s, a);
  RON_HIP_CHECK(hipGetLastError());
  return RON_OK;
}

template <class Tr>
int launch_cfg(int cfg, const ConvArgs& a, hipStream_t s) {
  switch (cfg) {
    case 0: return launch_t<Tr, 128, 128, 2, 2, 2, false>(a, s);
    case 1: return launch_t<Tr, 128, 64, 2, 2, 2, false>(a, s);
    case 2: return launch_t<Tr, 256, 128, 4, 2, 3, false>(a, s);
    case 3: return launch_t<Tr, 256, 64, 4, 2, 3, false>(a, s);
    case 4: return launch_t<Tr, 128, 128, 2, 2, 2, true>(a, s);
    case 5: return launch_t<Tr, 128, 64, 2, 2, 2, true>(a, s);
    case 6: return launch_t<Tr, 256, 128, 4, 2, 3, true>(a, s);
    case 7: return launch_t<Tr, 256, 64, 4, 2, 3, true>(a, s);
    case 8: return launch_t<Tr, 256, 256, 2, 4, 2, false>(a, s);
    case 9: return launch_t<Tr, 256, 256, 4, 2, 2, false>(a, s);
    case 10: return launch_t<Tr, 256, 256, 2, 4, 2, true>(a, s);
    case 11: return launch_t<Tr, 256, 256, 4, 2, 2, true>(a, s);
    case 12: return launch_t<Tr, 256, 256, 4, 2, 2, true, 1>(a, s);
    case 13: return launch_t<Tr, 256, 256, 4, 2, 2, true, 2>(a, s);
    case 14: return launch_t<Tr, 128, 128, 2, 2, 2, true, 1>(a, s);
    case 15: return launch_t<Tr, 128, 128, 2, 2, 2, true, 2>(a, s);
    case 16: return launch_t<Tr, 256, 256, 2, 2, 2, true>(a, s);
    case 17: return launch_t<Tr, 256, 256, 2, 2, 2, true, 1>(a, s);
    case 18: return launch_t<Tr, 256, 256, 4, 2, 2, true, 3>(a, s);
    case 19: return launch_t<Tr, 256, 256, 4, 2, 2, true, 4>(a, s);
    case 20: return launch_t<Tr, 256, 256, 4, 2, 4, true, 0, 64>(a, s);
    case 21: return launch_t<Tr, 256, 256, 2, 4, 4, true, 0, 64>(a, s);
    case 22: return launch_t<Tr, 256, 128, 4, 2, 5, true, 0, 64>(a, s);
    case 23: return launch_t<Tr, 128, 128, 2, 2, 4, true, 0, 64>(a, s);
    case 24: return launch_t<Tr, 128, 64, 2, 2, 4, true, 0, 64>(a, s);
    case 25: return launch_t<Tr, 256, 256, 4, 2, 4, true, 2, 64>(a, s);
    case 26: return launch_t<Tr, 128, 128, 2, 2, 4, false, 0, 64>(a, s);
    case 27: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, true, 5>(a, s);
    case 28: return launch_t<typename SmallShape<Tr>::type, 128, 128, 2, 2, 2, true, 5>(a, s);
    case 29: return launch_t<Tr, 256, 128, 4, 2, 3, true, 5>(a, s);
    case 30: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, true>(a, s);
    case 31: return launch_t<typename SmallShape<Tr>::type, 256, 256, 2, 4, 2, true>(a, s);
    case 32: return launch_t<typename SmallShape<Tr>::type, 256, 128, 4, 2, 3, true>(a, s);
    case 33: return launch_t<typename SmallShape<Tr>::type, 128, 128, 2, 2, 2, true>(a, s);
    case 34: return launch_t<typename SmallShape<Tr>::type, 128, 64, 2, 2, 2, true>(a, s);
    case 35: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, 2>(a, s);
    case 36: return launch_t<typename SmallShape<Tr>::type, 128, 128, 2, 2, 2, 2>(a, s);
    case 37: return launch_t<typename SmallShape<Tr>::type, 128, 64, 2, 2, 2, 2>(a, s);
    case 38: return launch_t<typename SmallShape<Tr>::type, 256, 256, 2, 4, 2, 2>(a, s);
    case 39: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, 2, 0, 128, 1>(a, s);
    case 40: return launch_t<typename SmallShape<Tr>::type, 128, 128, 2, 2, 2, 2, 0, 128, 1>(a, s);
    case 41: return launch_t<typename SmallShape<Tr>::type, 128, 64, 2, 2, 2, 2, 0, 128, 1>(a, s);
    case 42: return launch_t<typename SmallShape<Tr>::type, 256, 256, 2, 4, 2, 2, 0, 128, 1>(a, s);
    case 43: return launch_t<Tr, 256, 256, 4, 2, 2, 2, 0, 128, 1>(a, s);
    case 44: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, 2, 5, 128, 1>(a, s);
    case 45: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, 1, 1>(a, s);
    case 46: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, 1, 3>(a, s);
    case 47: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, 1, 2>(a, s);
    case 48: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, 1, 4>(a, s);
    case 49: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, 1, 6>(a, s);
    case 50: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, 1, 7>(a, s);
    case 51: return launch_t<typename SmallShape<Tr>::type, 256, 64, 4, 2, 3, 2>(a, s);
    case 52: return launch_t<typename SmallShape<Tr>::type, 256, 64, 8, 1, 3, 2>(a, s);
    case 53: return launch_t<typename SmallShape<Tr>::type, 256, 128, 4, 2, 2, 2>(a, s);
    case 54: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, 1, 8>(a, s);
    case 55: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, 1, 9>(a, s);
    case 56: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, 1, 0, 128, 2>(a, s);
    case 57: return launch_t<typename SmallShape<Tr>::type, 128, 128, 2, 2, 2, 1, 0, 128, 2>(a, s);
    case 58: return launch_t<typename SmallShape<Tr>::type, 128, 128, 2, 2, 2, 2, 0, 128, 2>(a, s);
    case 59: return launch_t<typename SmallShape<Tr>::type, 128, 64, 2, 2, 2, 2, 0, 128, 2>(a, s);
    case 60: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, 2, 0, 128, 2>(a, s);
    case 61: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, 1, 10>(a, s);
    case 62: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, 1, 11>(a, s);
    case 63: return launch_t<typename SmallShape<Tr>::type, 256, 256, 4, 2, 2, 1, 12>(a, s);
  }
  ron::set_error("conv: unknown tile config %d", cfg);
  return RON_ERR_INVALID;
}

template <class Tr>
int launch_finalize(const ConvArgs& a, hipStream_t s) {
  const long long total = (long long)a.M * (a.Npad / 4);
  const int grid = (int)std::min<long long>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(splitk_finalize_kernel<Tr>, dim3(grid), dim3(256), 0, s, a);
  RON_HIP_CHECK(hipGetLastError());
  return RON_OK;
}

}  // namespace detail
using namespace detail;

size_t dtype_size(int dtype) { return dtype == RON_DTYPE_F32 ? 4 : 2; }
int conv_k_chunk(int dtype) { return kRowBytes / (int)dtype_size(dtype); }
int conv_n_tile(int cout) { return cout <= 64 ? 64 : 128; }
int conv_num_cfgs() { return kNumDiagCfgs; }

// Split-K factor for grids that leave most CUs idle: such launches are a serial chain of KT dependent
// HBM round trips per workgroup, so the K loop is spread over enough workgroups to fill the chip (>= 8 steps each).
int conv_pick_splitk(int tiles, int KT, int slots) {
  if (tiles * 2 > slots || KT < 16) return 1;
  int sk = slots / tiles;
  if (sk > KT / 8) sk = KT / 8;
  return sk < 1 ? 1 : sk;
}

// Default tile choice, from tools/sweep_conv.py on MI355X at batch 32 (profiles/r01/sweep_*.txt).  All defaults are
// the 16x16x32-MFMA forms (5-19 % over the 32x32x16 forms of the same tile: the chip holds a higher clock on them).
// The 256x256 tile wins once it yields >= ~160 workgroups; below that the grid is the problem and the 128x128 /
// 2-workgroups-per-CU form keeps more CUs busy (it also beats the 3-stage 256x128 tile wherever that used to win).
static int conv_pick_cfg_shape(int M, int Npad) {
  const int tm256 = (M + 255) / 256;
  if (Npad % 128 != 0) return 37;                                   // N tile 64
  if (Npad % 256 == 0 && tm256 * (Npad / 256) >= 160) return 30;
  // many rounds of small tiles (conv2_x): issuing a tile's LDS-DMA pieces early in the stage wins a few per cent
  return ((M + 127) / 128) * (Npad / 128) >= 2048 ? 36 : 33;
}

int conv_pick_cfg(const ConvLaunch& c) { return conv_pick_cfg_shape(c.in.N * c.Ho * c.Wo, c.Npad); }
int conv_patch_pick(const ConvLaunch&) { return -1; }

int launch_conv(const ConvLaunch& c, hipStream_t stream) {
  if (c.cfg == kCfgPatchDiag || c.cfg == kCfgPatchDiag + 1) return launch_conv_patch(c, c.cfg, stream);
  const int esz = (int)dtype_size(c.dtype);
  int chunk = conv_k_chunk(c.dtype);
  RON_REQUIRE(c.in.C % chunk == 0, "conv: Cin %d is not a multiple of the K chunk %d", c.in.C, chunk);
  RON_REQUIRE(c.in.pad >= c.cpad, "conv: input halo %d < conv padding %d", c.in.pad, c.cpad);
  RON_REQUIRE(c.in.bytes > 0 && c.in.bytes < (int64_t)1 << 32, "conv: input allocation must be < 4 GiB for buffer addressing");
  RON_REQUIRE(c.wgt_bytes > 0 && c.wgt_bytes < (int64_t)1 << 32, "conv: weight allocation must be < 4 GiB");
  RON_REQUIRE(c.out.pixels() * c.out.cstride < (int64_t)1 << 31, "conv: output too large for 32-bit offsets");
  const int K = c.kh * c.kw * c.in.C;
  const int M = c.in.N * c.Ho * c.Wo;
  const int cfg = c.cfg >= 0 ? c.cfg : conv_pick_cfg_shape(M, c.Npad);
  RON_REQUIRE(cfg >= 0 && cfg < kNumDiagCfgs, "conv: tile config %d out of range", cfg);
  const int BN = kCfgs[cfg].bn;
  const int kt_heur = K / chunk;                    // the split-K heuristic counts 128-byte K steps
  chunk = kCfgs[cfg].rb / esz;
  RON_REQUIRE(c.Npad % BN == 0, "conv: Npad %d not a multiple of the N tile %d", c.Npad, BN);
  if (c.up > 0) RON_REQUIRE(c.up_cout % BN == 0, "transposed conv: channels per tap %d not a multiple of %d", c.up_cout, BN);
  ConvArgs a;
  a.in = c.in.base; a.in_bytes = (unsigned)c.in.bytes;
  a.wgt = c.wgt; a.wgt_bytes = (unsigned)c.wgt_bytes;
  a.bias = c.bias; a.out = c.out.base; a.res = c.res;
  a.Ho = c.Ho; a.Wo = c.Wo; a.M = M;
  a.in_Hp = c.in.Hp(); a.in_Wp = c.in.Wp(); a.in_cstride = c.in.cstride; a.in_org = c.in.pad - c.cpad;
  a.in_coff = c.in.coff;
  a.Cin = c.in.C; a.kw = c.kw; a.K = K; a.KT = a.K / chunk;
  a.stride = c.stride; a.dil = c.dil;
  a.Cout = c.Cout;
  a.out_Hp = c.out.Hp(); a.out_Wp = c.out.Wp(); a.out_cstride = c.out.cstride; a.out_pad = c.out.pad;
  a.out_coff = c.out.coff;
  a.up = c.up; a.up_cout = c.up_cout;
  a.relu = c.relu; a.out_f32 = c.out_f32;
  a.tiles_n = c.Npad / BN;
  a.tiles_total = ((M + kCfgs[cfg].bm - 1) / kCfgs[cfg].bm) * a.tiles_n;
  a.Npad = c.Npad;
  a.splitk = 1; a.kt_split = a.KT; a.partial = nullptr;
  a.pool = c.pool;
  a.dbg = c.dbg;
  if (c.pool) {
    RON_REQUIRE(c.up == 0 && c.res == nullptr && !c.out_f32 && c.Ho % 2 == 0 && c.Wo % 2 == 0 && c.stride == 1,
                "conv + fused pool: plain stride-1 conv on an even map only");
    RON_REQUIRE(c.out.H == c.Ho / 2 && c.out.W == c.Wo / 2, "conv + fused pool: output view must be the pooled map");
  }
  const int sk = c.splitk >= 0 ? c.splitk : conv_pick_splitk(a.tiles_total, kt_heur, cfg_slots(cfg));
  if (sk > 1 && c.up == 0 && !c.pool && c.scratch != nullptr && (int64_t)sk * M * c.Npad * 4 <= c.scratch_bytes) {
    a.kt_split = (a.KT + sk - 1) / sk;
    a.splitk = (a.KT + a.kt_split - 1) / a.kt_split;      // no empty split
    a.partial = (float*)c.scratch;
    if (a.splitk == 1) { a.kt_split = a.KT; a.partial = nullptr; }
  }
  RON_REQUIRE((int64_t)c.Npad * a.K * esz == c.wgt_bytes, "conv: packed weight size mismatch");
  int rc;
  if (c.dtype == RON_DTYPE_BF16) rc = launch_cfg<TraitsBF16>(cfg, a, stream);
  else if (c.dtype == RON_DTYPE_F16) rc = launch_cfg<TraitsF16>(cfg, a, stream);
  else if (c.dtype == RON_DTYPE_F32) rc = launch_cfg<TraitsF32>(cfg, a, stream);
  else { ron::set_error("conv: unknown dtype %d", c.dtype); return RON_ERR_INVALID; }
  if (rc != RON_OK || a.splitk == 1) return rc;
  if (c.dtype == RON_DTYPE_BF16) return launch_finalize<TraitsBF16>(a, stream);
  if (c.dtype == RON_DTYPE_F16) return launch_finalize<TraitsF16>(a, stream);
  return launch_finalize<TraitsF32>(a, stream);
}

// the round-1 kernels know no grouped launches (graph.cpp plans none under RON_DIAG)
int launch_conv_group(const ConvLaunch*, int, int, void*, int64_t, hipStream_t) {
  ron::set_error("the diagnostic library has no grouped launches");
  return RON_ERR_UNSUPPORTED;
}
int64_t conv_group_scratch_bytes(const ConvLaunch*, int, int) { return 0; }

int64_t conv_scratch_bytes(const ConvLaunch& l) {
  const int cfg = l.cfg;
  if (cfg == kCfgPatchDiag || cfg == kCfgPatchDiag + 1 || l.up > 0 || l.pool) return 0;   // the halo-patch kernel never splits K
  const int M = l.in.N * l.Ho * l.Wo;
  const int KT = l.kh * l.kw * l.in.C / conv_k_chunk(l.dtype);
  const int c = cfg >= 0 ? cfg : conv_pick_cfg_shape(M, l.Npad);
  const int tiles = ((M + kCfgs[c].bm - 1) / kCfgs[c].bm) * (l.Npad / kCfgs[c].bn);
  const int sk = l.splitk >= 0 ? l.splitk : conv_pick_splitk(tiles, KT, cfg_slots(c));
  return sk > 1 ? (int64_t)sk * M * l.Npad * 4 : 0;
}

}  // namespace ron
