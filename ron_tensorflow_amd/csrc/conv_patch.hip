// 3x3 / stride-1 / pad-1 convolution with an LDS-staged input halo patch (gfx950).
//
// The row-gather kernel (conv_mfma.hip) re-stages the A operand for every filter tap: the same input pixels travel
// L2 -> LDS nine times, and with the K loop ordered tap-major the nine uses of a line are ~16 MB of other traffic apart,
// so they come from beyond the XCD's L2 (round 1: 9.0 M fabric read requests for a launch whose input is 57 MB).  The
// memory side, not the matrix cores, bounded that kernel.
//
// Here a workgroup owns 256 output positions and BN output channels.  Per 128-byte channel chunk it stages the input
// patch those positions need ONCE (LDS-DMA, double buffered across chunks, one piece per tap step) and runs the nine taps
// against it: the A fragment of tile row r for tap (ky, kx) is patch row pp(r) + ky*PW + kx, an LDS address shift.  Only
// the weights (BN x 128 B per step) are staged per tap.  Two tilings, chosen per map on the host (PatchGeom):
//   * pixel tile TH x TW (16 x 16, or 32 x 8) on maps these divide: no tile row is wasted, patch = (TH+2) x (TW+2) rows;
//   * FLAT run of 256 consecutive positions of the input tensor's memory order, for maps <= 42 pixels wide.  Activations
//     carry SHARED halos (conv_mfma.h, TensorView): consecutive positions are consecutive pixels with only `pad` halo
//     positions per row between them, and a run may cross rows and images.  Patch = 256 + 2*(Wp+1) rows of the same
//     order, tap shift = ky*Wp + kx.  A 40 x 40 map wastes 4.8 % of the rows (the halo positions), where 6 x 40-pixel
//     tiles wasted 15 %.
// Patch rows are 128 B; the 16-B chunk c of patch row i sits in slot c ^ key(i), key(i) = ((i >> 1) & 3) << 1: with the
// 16-row fragments of the 16x16 MFMA this is conflict-free for ds_read_b128 at EVERY start row (a tap shift moves a
// fragment to an arbitrary start), as long as the fragment's rows are consecutive patch rows.
// MFMA, fragment double buffering, pinned issue order, weight row permutation and the epilogue are conv_mfma.hip's.
#include "conv_device.h"

namespace ron {
namespace detail {

constexpr int kPatchPieces = 6;                  // LDS-DMA pieces per thread and chunk (512 threads x 16 B = 64 rows each)
constexpr int kPatchRows = 344;                  // rows a patch buffer holds (43 KB): 18 x 18, 34 x 10, 256 + 2 * 44

__device__ __forceinline__ int patch_key(int i) { return ((i >> 1) & 3) << 1; }

// Issue order of the NQ k-steps between two barriers, everything a compile-time constant (the builtin wants immediates): per k-step
// its MFMAs (split precision: one per fragment pair in a hi step, two in a lo step), the next k-step's RD fragment reads one per MFMA
// gap, and - in k-step 0 only - the step's PS0 LDS-DMA instructions spaced evenly between the MFMAs.
template <class Tr, int MR, int NR, int NQ, int PS0, int q = 0>
__device__ __forceinline__ void pin_patch_ksteps() {
  if constexpr (q < NQ) {
    constexpr int RD = q < NQ - 1 ? MR + NR : 0;
    constexpr int MM = MR * NR * (IsSplit<Tr>::value ? 1 + (q & 1) : Tr::kMfmaPerMma);
    constexpr int PS = q == 0 ? PS0 : 0;
#pragma unroll
    for (int m = 0; m < MM; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (m < RD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (((m + 1) * PS) / MM > (m * PS) / MM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    if constexpr (RD > MM) __builtin_amdgcn_sched_group_barrier(0x100, RD - MM, 0);
#pragma unroll
    for (int x = 0; x < 16; ++x)
      if (x < PS - MM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    pin_patch_ksteps<Tr, MR, NR, NQ, PS0, q + 1>();
  }
}

struct PatchArgs {
  ConvArgs c;
  int flat;                   // 1: runs of 256 positions; 0: TH x TW pixel tiles
  int TH, TW, PW;             // pixel tile; PW = patch rows per tap row (TW + 2, flat: the tensor's row pitch Wp)
  int tiles_x, tiles_y;       // pixel tiles per image
  int H, W, pad;              // map (conv output == input size) and the input tensor's halo
  int chunks;                 // Cin / chunk elements
  int P0;                     // flat: first position (pixel index in the input tensor) of tile 0
  int n_img;
  unsigned max_pixel;         // last pixel index of the input allocation (patch addresses are clamped to it)
};

// SB weight stages: SB-1 steps of lead for the per-tap weight tiles.
// One tile of convolution `p` (geometry `pa`): workgroup `bid` of the `nwg` that convolution's launch -- or its share of a pair
// launch -- consists of.
// TPS: filter taps per step (= per barrier).  1: the form above.  3: a step is a whole filter ROW -- three taps' weights staged
// together, 3 x KS k-steps of MFMAs between two barriers.  For N <= 64 a tap is only MR * NR * KS = 16 MFMAs per wave, too few to
// hide a step's fixed cost (counted wait + barrier, LDS-DMA issue, first-fragment latency: ~1500 cycles per step whatever is in
// flight, profiles/r03/skinny_heads_pair.txt); three taps per barrier triple the work behind it.
template <class Tr, int BN, int WN, int SB, int TPS = 1>
__device__ __forceinline__ void conv3x3_patch_tile(const PatchArgs& pa, const ConvArgs& p, const unsigned bid, const unsigned nwg) {
  constexpr int BM = 256, WM = 8 / WN, kThreads = 512;
  constexpr int MT = Tr::kMT, kGroups = 64 / MT, KS = 8 / kGroups, EPA = MT * MT / 64;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MR = TM / MT, NR = TN / MT;
  constexpr int B_IT = BN / 64;                   // B pieces per thread and step
  constexpr int G = B_IT + 1;                     // LDS-DMA instructions per thread and step (the weights + one patch piece)
  constexpr int kPatchBytes = kPatchRows * kRowBytes;
  constexpr int kTapBytes = BN * kRowBytes;       // one tap's weights of this N tile
  constexpr int kBBytes = TPS * kTapBytes;        // one weight stage
  static_assert(TPS == 1 || TPS == 3, "taps per step");
  static_assert(MT == 16 && TM % MT == 0 && TN % MT == 0 && NR <= 8 && SB >= 2, "bad wave tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: [patch 0][patch 1][B stage 0 .. SB-1][pp: BM ints][out_off: BM ints][sink 1 KB]
  char* s_b = smem + 2 * kPatchBytes;
  int* s_pp = reinterpret_cast<int*>(s_b + SB * kBBytes);
  int* s_out_off = s_pp + BM;
  // a zero-record LDS-DMA still writes (zeros): the placeholder pieces that keep the vmcnt groups uniform land here
  char* s_sink = reinterpret_cast<char*>(s_out_off + BM);

  // `wave` is wave-uniform; said explicitly, or hipcc wraps every LDS-DMA whose descriptor / LDS address depends on it (the
  // "piece lies inside the patch buffer" tests below) in a waterfall loop
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if constexpr (IsSplit<Tr>::value) split_mode_on();      // the epilogue's fp32 -> f16 conversions saturate (conv_device.h)
  const int wm = wave / WN, wn = wave % WN;

  const unsigned xcd = bid & 7u, q8 = nwg >> 3, r8 = nwg & 7u;
  const unsigned wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_n = (int)(wgid % (unsigned)p.tiles_n);
  const unsigned tsp = wgid / (unsigned)p.tiles_n;           // spatial tile
  const int n0 = tile_n * BN;
  const int Wp = p.in_Wp, Hp = p.in_Hp;

  // tile row -> patch row of tap (0,0) and output offset; first pixel (index in the input tensor) of the patch
  int patch_pix0;
  if (pa.flat) {
    const int Pt = pa.P0 + (int)tsp * BM;
    patch_pix0 = Pt - Wp - 1;
    for (int r = tid; r < BM; r += kThreads) {
      const int P = Pt + r;
      const int row = P / Wp, x = P - row * Wp - pa.pad;
      const int rr = row - pa.pad;
      const int img = rr / Hp, y = rr - img * Hp;
      const bool valid = x >= 0 && y < pa.H && img < pa.n_img;
      s_pp[r] = r;
      s_out_off[r] = valid ? ((img * p.out_Hp + y + p.out_pad) * p.out_Wp + x + p.out_pad) * p.out_cstride + p.out_coff : -1;
    }
  } else {
    unsigned t = tsp;
    const int tx = (int)(t % (unsigned)pa.tiles_x); t /= (unsigned)pa.tiles_x;
    const int ty = (int)(t % (unsigned)pa.tiles_y);
    const int img = (int)(t / (unsigned)pa.tiles_y);
    const int y0 = ty * pa.TH, x0 = tx * pa.TW;
    patch_pix0 = (img * Hp + pa.pad + y0 - 1) * Wp + pa.pad + x0 - 1;
    for (int r = tid; r < BM; r += kThreads) {
      int ly, lx;
      if (p.pool) {                                           // window-major: rows 4w .. 4w+3 = the 2x2 window w
        const int w = r >> 2, hw = pa.TW >> 1;
        ly = 2 * (w / hw) + ((r >> 1) & 1);
        lx = 2 * (w % hw) + (r & 1);
      } else {
        ly = r / pa.TW;
        lx = r - ly * pa.TW;
      }
      const int y = y0 + ly, x = x0 + lx;
      const bool valid = ly < pa.TH;                          // TH * TW may be < 256
      s_pp[r] = valid ? ly * pa.PW + lx : 0;
      int off;
      if (p.pool) off = ((img * p.out_Hp + (y >> 1) + p.out_pad) * p.out_Wp + (x >> 1) + p.out_pad) * p.out_cstride + p.out_coff;
      else off = ((img * p.out_Hp + y + p.out_pad) * p.out_Wp + x + p.out_pad) * p.out_cstride + p.out_coff;
      s_out_off[r] = valid ? off : -1;
    }
  }

  // patch pieces of this thread: LDS patch row i = q >> 3 (q = k*512 + tid), slot = q & 7
  int p_voff[kPatchPieces];
#pragma unroll
  for (int k = 0; k < kPatchPieces; ++k) {
    const int q = k * kThreads + tid;
    const int i = q >> 3, slot = q & 7;
    int pix;
    if (pa.flat) {
      pix = patch_pix0 + i;
    } else {
      const int py = i / (pa.TW + 2), px = i - py * (pa.TW + 2);
      pix = patch_pix0 + py * Wp + px;
    }
    // rows outside the allocation belong to positions that store nothing; keep the address inside it
    const unsigned upix = min((unsigned)max(pix, 0), pa.max_pixel);
    p_voff[k] = (int)((upix * (unsigned)p.in_cstride + (unsigned)p.in_coff) * Tr::kEsz) + ((slot ^ patch_key(i)) << 4);
  }
  // B pieces: LDS row (j*MT + r) of a wave's TN-wide group <- weight row (r*NR + j)   (coalesced epilogue, see conv_mfma.hip);
  // weight row n, K step kt at ((n / 64) * KT + kt) * 8 KB + (n % 64) * 128 B
  int b_voff[4];
  static_assert(B_IT <= 4, "BN <= 256");
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int lrow = it * 64 + (tid >> 3);
    const int grp = lrow / TN, loc = lrow % TN;
    const int nrow = n0 + grp * TN + (loc % MT) * NR + (loc / MT);
    b_voff[it] = (int)((unsigned)(nrow >> 6) * (unsigned)p.KT * (unsigned)kWeightBlockBytes + (unsigned)(nrow & 63) * kRowBytes) +
                 (((tid & 7) ^ ((tid >> 4) & 7)) << 4);
  }
  __syncthreads();

  const int fr = lane & (MT - 1), fh = lane / MT;
  int pp[MR];
#pragma unroll
  for (int i = 0; i < MR; ++i) pp[i] = s_pp[wm * TM + i * MT + fr];
  int rd_off_b[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) rd_off_b[s] = fr * kRowBytes + (((kGroups * s + fh) ^ ((fr >> 1) & 7)) << 4);
  const int b_base = wn * TN * kRowBytes;

  // descriptors are rebuilt at the use site with 0 records for pieces that have nothing to fetch
#define RS_A(live_) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (live_) ? p.in_bytes : 0u, 0x00020000)
#define RS_B(live_) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, (live_) ? p.wgt_bytes : 0u, 0x00020000)
  // piece k_ (compile time) of the patch of chunk cc_ into buffer buf_; a wave whose 8 rows lie past the buffer sinks it
#define PATCH_PIECE(k_, buf_, cc_, live_)                                                                            \
  do {                                                                                                               \
    const bool in_ = (k_) * 64 + wave * 8 + 8 <= kPatchRows;                                                         \
    char* d_ = in_ ? smem + (buf_) * kPatchBytes + ((k_) * kThreads + wave * 64) * 16 : s_sink;                      \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(RS_A((live_) && in_), (lds_void*)d_, 16, p_voff[k_], (cc_) * kRowBytes, 0, 0); \
  } while (0)
#define B_PIECE(it_, stage_, soff_, live_)                                                                           \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(RS_B(live_), (lds_void*)(s_b + (stage_) * kBBytes + ((it_) * 64 + wave * 8) * kRowBytes), \
                                           16, b_voff[it_], soff_, 0, 0)

  typename Tr::acc_t acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < EPA; ++e) acc[i][j][e] = 0.f;

  // LDS-DMA order.  A wave's vmcnt retires in issue order, so a transfer holds up every younger one at a counted wait.
  // The patch of the NEXT chunk (six pieces per thread, one per step during taps 0..5) is not needed before that chunk
  // starts and comes from beyond L2 (first touch), while the weights of the next step are needed one step on and mostly
  // hit L2.  Every step issues one group of G = B_IT + 1 instructions: the weights of step + SB - 1 FIRST, then one
  // piece of the next chunk's patch (a zero-record placeholder into the sink when there is none to fetch, so that the
  // groups stay uniform).  The wait at the top of a step leaves the newest SB-2 groups AND the piece behind them in
  // flight: a patch piece gets two steps to land, not one (measured: with the piece issued first, the patch stream alone
  // cost more than the eight times larger weight stream alone).  The prologue (patch of chunk 0, weights of the first SB - 1
  // steps) is waited for as a whole.
  if constexpr (TPS == 3) {
    // ---- one filter row per step -------------------------------------------------------------------------------------------
    // LDS-DMA group of a step (issue order = retire order): the 3 * B_IT weight pieces of step + SB - 1 (taps kx = 0, 1, 2 of its
    // filter row: blocks (3 ky + kx) * chunks + c of the packed weights), then 3 pieces of the next chunk's patch in the chunk's
    // first two steps, placeholders into the sink in its third: the wait at the top of a step leaves the youngest SB - 2 groups and
    // the 3 pieces behind them in flight, so the patch of chunk c + 1 has landed one whole step before it is read.
    constexpr int GR = 3 * B_IT + 3;
    static_assert(kPatchPieces == 6, "three patch pieces in each of a chunk's first two steps");
    const int n_rows = pa.chunks * 3;
    // weights of filter row `row_step` (= chunk * 3 + ky), tap kx: block (3 ky + kx) * chunks + chunk of the packed weights
#define W_SOFF(row_step_, kx_) ((((row_step_) % 3) * 3 + (kx_)) * pa.chunks + (row_step_) / 3) * kWeightBlockBytes
#pragma unroll
    for (int k = 0; k < kPatchPieces; ++k) PATCH_PIECE(k, 0, 0, true);
#pragma unroll
    for (int t = 0; t < SB - 1; ++t) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int it = 0; it < B_IT; ++it)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(RS_B(t < n_rows), (lds_void*)(s_b + t * kBBytes + kx * kTapBytes + (it * 64 + wave * 8) * kRowBytes),
                                                   16, b_voff[it], W_SOFF(min(t, n_rows - 1), kx), 0, 0);
    }
    // everything the prologue issued lands before the first step (no placeholder pieces here: identical back-to-back LDS-DMA
    // instructions into the sink are dead stores to the compiler, which kept one of three and left the first step's counted wait
    // two short -- its weights of taps 1 and 2 could still be in flight)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int cc = 0, ky = 0, st_rd = 0, st_wr = SB - 1;
    for (int step = 0; step < n_rows; ++step) {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((SB - 2) * GR + 3) : "memory");
      __builtin_amdgcn_s_barrier();
      const char* sa = smem + (cc & 1) * kPatchBytes;
      const char* sb = s_b + st_rd * kBBytes + b_base;
      if (++st_rd == SB) st_rd = 0;
      // fragment addresses of the row's three taps: A = patch row pp + ky * PW + kx (k-step 1 = the same ^ 64 bytes)
      int a_off[3][MR];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int i = 0; i < MR; ++i) {
          const int row = pp[i] + ky * pa.PW + kx;
          a_off[kx][i] = row * kRowBytes + ((fh ^ patch_key(row)) << 4);
        }
      // Split precision (TraitsF16X3S): k-step 0 of a tap reads the hi planes (16-byte slots 0..3 of the row chunk), k-step 1 the lo
      // planes (slots 4..7: the same address ^ 64), and the products are hi*hi in k-step 0, lo*hi + hi*lo in k-step 1 - so the lo
      // step needs the hi fragments too while the next tap's are already being read: three register sets instead of two.
      constexpr bool kSplit = IsSplit<Tr>::value;
      constexpr int NS = kSplit ? 3 : 2;
      u32x4 fa[NS][MR], fb[NS][NR];
      // fragments of k-step q_ (= 2 kx + k-step of the tap) into register set set_
#define READ_FRAGS(set_, q_)                                                                                                       \
      do {                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < MR; ++i)                                                                             \
          fa[set_][i] = *reinterpret_cast<const u32x4*>(sa + (a_off[(q_) >> 1][i] ^ (((q_) & 1) << 6)));                            \
        _Pragma("unroll") for (int j = 0; j < NR; ++j)                                                                             \
          fb[set_][j] = *reinterpret_cast<const u32x4*>(sb + ((q_) >> 1) * kTapBytes + j * MT * kRowBytes + rd_off_b[(q_) & 1]);   \
      } while (0)
      READ_FRAGS(0, 0);
      // this step's LDS-DMA group
      {
        const int nxt = step + SB - 1;
        const bool live = nxt < n_rows;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int it = 0; it < B_IT; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(RS_B(live), (lds_void*)(s_b + st_wr * kBBytes + kx * kTapBytes + (it * 64 + wave * 8) * kRowBytes),
                                                     16, b_voff[it], W_SOFF(min(nxt, n_rows - 1), kx), 0, 0);
        if (++st_wr == SB) st_wr = 0;
        const bool more = cc + 1 < pa.chunks && ky < 2;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int piece = ky * 3 + k;               // 0..5 in the chunk's first two steps
          int voff = p_voff[0];
#pragma unroll
          for (int q = 1; q < kPatchPieces; ++q) voff = piece == q ? p_voff[q] : voff;
          const bool in_ = piece * 64 + wave * 8 + 8 <= kPatchRows;
          char* d_ = (more && in_) ? smem + ((cc + 1) & 1) * kPatchBytes + (piece * kThreads + wave * 64) * 16 : s_sink;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(RS_A(more && in_), (lds_void*)d_, 16, voff, (cc + 1) * kRowBytes, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        if (q < 5) READ_FRAGS((q + 1) % NS, q + 1);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < NR; ++j) {
            if constexpr (kSplit) {
              if (q & 1) {
                Tr::mma(fa[q % NS][i], fb[(q + NS - 1) % NS][j], acc[i][j]);       // lo * hi
                Tr::mma(fa[(q + NS - 1) % NS][i], fb[q % NS][j], acc[i][j]);       // hi * lo
              } else {
                Tr::mma(fa[q % NS][i], fb[q % NS][j], acc[i][j]);                  // hi * hi
              }
            } else {
              Tr::mma(fa[q % NS][i], fb[q % NS][j], acc[i][j]);
            }
          }
      }
      // issue order: first fragments | k-step 0: MFMAs with the next reads and the GR DMA instructions between them | k-steps 1..4:
      // MFMAs with the next reads | last k-step
      __builtin_amdgcn_sched_group_barrier(0x100, MR + NR, 0);
      pin_patch_ksteps<Tr, MR, NR, 6, GR>();
      if (++ky == 3) { ky = 0; ++cc; }
    }
#undef READ_FRAGS
#undef W_SOFF
  } else {
  const int n_steps = pa.chunks * 9;
#pragma unroll
  for (int k = 0; k < kPatchPieces; ++k) PATCH_PIECE(k, 0, 0, true);
#pragma unroll
  for (int t = 0; t < SB - 1; ++t) {
    const int soff = ((t % 9) * pa.chunks + t / 9) * kWeightBlockBytes;
#pragma unroll
    for (int it = 0; it < B_IT; ++it) B_PIECE(it, t, soff, t < n_steps);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the prologue lands as a whole (see the row-step form above)

  int tap = 0, cc = 0, ky = 0, kx = 0;
  int ntap = (SB - 1) % 9, ncc = (SB - 1) / 9;      // tap / chunk of step + SB - 1
  int st_rd = 0, st_wr = SB - 1;
  for (int step = 0; step < n_steps; ++step) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((SB - 2) * G + 1) : "memory");
    __builtin_amdgcn_s_barrier();
    const char* sa = smem + (cc & 1) * kPatchBytes;
    const char* sb = s_b + st_rd * kBBytes + b_base;
    if (++st_rd == SB) st_rd = 0;
    const int tapoff = ky * pa.PW + kx;
    int a_off[MR];                                       // byte offset of the k-step-0 fragment; k-step 1 is the same ^ 64
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int row = pp[i] + tapoff;
      a_off[i] = row * kRowBytes + ((fh ^ patch_key(row)) << 4);
    }
    u32x4 fa[2][MR], fb[2][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i) fa[0][i] = *reinterpret_cast<const u32x4*>(sa + a_off[i]);
#pragma unroll
    for (int j = 0; j < NR; ++j) fb[0][j] = *reinterpret_cast<const u32x4*>(sb + j * MT * kRowBytes + rd_off_b[0]);
    // this step's LDS-DMA group (branch free: the patch piece index is the tap, selected with v_cndmask)
    {
      const int soff = (ntap * pa.chunks + ncc) * kWeightBlockBytes;
      const bool live = ncc < pa.chunks;
#pragma unroll
      for (int it = 0; it < B_IT; ++it) B_PIECE(it, st_wr, soff, live);
      if (++ntap == 9) { ntap = 0; ++ncc; }
      if (++st_wr == SB) st_wr = 0;
      const bool more = cc + 1 < pa.chunks && tap < kPatchPieces;
      int voff = p_voff[0];
#pragma unroll
      for (int k = 1; k < kPatchPieces; ++k) voff = tap == k ? p_voff[k] : voff;
      const bool in_ = tap * 64 + wave * 8 + 8 <= kPatchRows;
      char* d_ = (more && in_) ? smem + ((cc + 1) & 1) * kPatchBytes + (tap * kThreads + wave * 64) * 16 : s_sink;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(RS_A(more && in_), (lds_void*)d_, 16, voff, (cc + 1) * kRowBytes, 0, 0);
    }
    static_assert(KS == 2, "k-step 1 fragment = k-step 0 fragment ^ 64 bytes");
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
      if (s2 < KS - 1) {
#pragma unroll
        for (int i = 0; i < MR; ++i) fa[(s2 + 1) & 1][i] = *reinterpret_cast<const u32x4*>(sa + (a_off[i] ^ 64));
#pragma unroll
        for (int j = 0; j < NR; ++j)
          fb[(s2 + 1) & 1][j] = *reinterpret_cast<const u32x4*>(sb + j * MT * kRowBytes + rd_off_b[s2 + 1]);
      }
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) mma_step<Tr, MR, NR>(s2, fa, fb, i, j, acc[i][j]);      // split precision: hi*hi | lo*hi + hi*lo
    }
    // issue order: first fragments | k-step 0: MFMAs with the next reads and the G DMA instructions spaced between them |
    // MFMAs of the last k-step
    __builtin_amdgcn_sched_group_barrier(0x100, MR + NR, 0);
    pin_patch_ksteps<Tr, MR, NR, KS, G>();
    if (++tap == 9) { tap = 0; ++cc; ky = 0; kx = 0; }
    else if (++kx == 3) { kx = 0; ++ky; }
  }
  }
#undef PATCH_PIECE
#undef B_PIECE
#undef RS_A
#undef RS_B

  const int nloc = wn * TN + fr * NR;
  conv_epilogue<Tr, MR, NR, MT, EPA>(p, acc, s_out_off, wm * TM, fh, n0 + nloc, n0 + nloc, 0);
}

template <class Tr, int BN, int WN, int SB, int TPS = 1>
__global__ __launch_bounds__(512, 2) void conv3x3_patch_kernel(PatchArgs pa) {
  conv3x3_patch_tile<Tr, BN, WN, SB, TPS>(pa, pa.c, blockIdx.x, gridDim.x);
}

// Two convolutions over the SAME input tensor geometry (channel slices of one map: the Cout = 20 / 40 heads of a scale read
// slices of the per-scale concatenated tensor, nets/ron_vgg_320.py:406-415,427-428) in ONE launch: workgroups [0, first) run
// `pa.c`, the rest `second`, each exactly as its own launch would.  Alone each fills 200 of the 512 workgroup slots.
template <class Tr, int BN, int WN, int SB, int TPS>
__global__ __launch_bounds__(512, 2) void conv3x3_patch_pair_kernel(PatchArgs pa, ConvArgs second, int first) {
  const int b = (int)blockIdx.x;
  if (b < first) conv3x3_patch_tile<Tr, BN, WN, SB, TPS>(pa, pa.c, (unsigned)b, (unsigned)first);
  else conv3x3_patch_tile<Tr, BN, WN, SB, TPS>(pa, second, (unsigned)(b - first), gridDim.x - (unsigned)first);
}

template <class Tr, int BN, int WN, int SB, int TPS>
int launch_patch_pair_t(const PatchArgs& a, const ConvArgs& second, int first, int grid, hipStream_t s) {
  const size_t lds = 2 * (size_t)kPatchRows * kRowBytes + SB * TPS * (size_t)BN * kRowBytes + 2 * 256 * sizeof(int) + 1024;
  static PerDeviceOnce once;
  RON_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void*>(&conv3x3_patch_pair_kernel<Tr, BN, WN, SB, TPS>), (int)lds));
  RON_LAUNCH((conv3x3_patch_pair_kernel<Tr, BN, WN, SB, TPS>), dim3(grid), dim3(512), lds, s, a, second, first);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

template <class Tr, int BN, int WN, int SB, int TPS = 1>
int launch_patch_t(const PatchArgs& a, int grid, hipStream_t s) {
  const size_t lds = 2 * (size_t)kPatchRows * kRowBytes + SB * TPS * (size_t)BN * kRowBytes + 2 * 256 * sizeof(int) + 1024;
  static PerDeviceOnce once;
  RON_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void*>(&conv3x3_patch_kernel<Tr, BN, WN, SB, TPS>), (int)lds));
  RON_LAUNCH((conv3x3_patch_kernel<Tr, BN, WN, SB, TPS>), dim3(grid), dim3(512), lds, s, a);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

// How an H x W map (input halo `pad`) is cut into 256-row tiles; false when the kernel does not apply.
struct PatchGeom { int flat, TH, TW, tiles_x, tiles_y, tiles_sp; };
static bool patch_geom(int n, int H, int W, int pad, bool pool, PatchGeom* g) {
  if (H % 16 == 0 && W % 16 == 0) { g->flat = 0; g->TH = 16; g->TW = 16; }
  else if (H % 8 == 0 && W % 32 == 0) { g->flat = 0; g->TH = 8; g->TW = 32; }
  else if (!pool && 256 + 2 * (W + pad + 1) <= kPatchRows) { g->flat = 1; g->TH = 0; g->TW = 0; }
  else return false;
  if (g->flat) {
    // positions from the first row that holds pixels to the end of the last image's last row
    const int64_t span = ((int64_t)n * (H + pad) - pad) * (W + pad);
    g->tiles_x = g->tiles_y = 0;
    g->tiles_sp = (int)((span + 255) / 256);
  } else {
    g->tiles_x = W / g->TW; g->tiles_y = H / g->TH;
    g->tiles_sp = n * g->tiles_x * g->tiles_y;
  }
  return true;
}

}  // namespace detail
using namespace detail;

bool conv_patch_applicable(const ConvLaunch& c) {
  PatchGeom g;
  return c.center_from == 0 &&     // centre-tap-only columns: the row-gather kernel only
         c.kh == 3 && c.kw == 3 && c.stride == 1 && c.dil == 1 && c.cpad == 1 && c.up == 0 && c.in.H == c.Ho && c.in.W == c.Wo &&
         c.in.pad >= 1 && c.Npad % 64 == 0 && c.in.C % conv_k_chunk(c.dtype) == 0 &&
         patch_geom(c.in.N, c.in.H, c.in.W, c.in.pad, c.pool != 0, &g);
}

static int patch_bn(int cfg) { return cfg == kCfgPatch64 ? 64 : (cfg == kCfgPatch128 ? 128 : 256); }

// Where the patch kernel is the better choice (tools/sweep_conv.py on MI355X, profiles/r02/sweep_conv_exp_v*.txt): the skinny
// heads (Cout <= 64: objectness_score, loc_pred) once the grid fills the chip.  There the row-gather kernel is bound by
// re-staging the input nine times for almost no arithmetic (b4_loc 68 -> 58 us, b4_obj 67 -> 57 us).  On the wide layers
// the two kernels are level per executed MFMA (the weight stream, not the activation stream, is what the 2-stage pipeline
// cannot hide) and the patch kernel pays the halo positions of the flat tiling (4.8 % on 40 x 40), so the
// row-gather kernel stays the default there.  Below a full grid the row-gather kernel with split-K keeps more CUs busy.
int conv_patch_pick(const ConvLaunch& c) {
  PatchGeom g;
  if (!patch_geom(c.in.N, c.in.H, c.in.W, c.in.pad, c.pool != 0, &g)) return -1;
  if (c.Npad != 64) return -1;
  // split precision: the skinny heads (Cin = 512) only - on conv1_2 (Cin = 64: two chunks of 32 elements per tap) the row-gather
  // kernel measured 7-9 % ahead (tools/sweep_conv.py --dtype f16x3, profiles/r04/sweep_f16x3_patch.txt)
  if (c.dtype == RON_DTYPE_F16X3 && c.in.C < 256) return -1;
  return g.tiles_sp >= 192 ? kCfgPatch64 : -1;
}

// Kernel arguments of launch `c` with N tile BN; *grid = its workgroups.
static int patch_args(const ConvLaunch& c, int cfg, PatchArgs* out, int* grid) {
  RON_REQUIRE(conv_patch_applicable(c), "patch kernel: not a 3x3 / stride 1 / pad 1 conv on a map it can tile");
  RON_REQUIRE(conv_cfg_is_patch(cfg) && cfg < kNumCfgs, "patch kernel: bad tile config %d", cfg);
  RON_REQUIRE(c.out2.base == nullptr, "patch kernel: no second (un-pooled) output");
  const int BN = patch_bn(cfg);
  RON_REQUIRE(c.Npad % BN == 0, "patch kernel: Npad %d not a multiple of the N tile %d", c.Npad, BN);
  const int esz = (int)dtype_size(c.dtype);
  RON_REQUIRE(c.in.bytes > 0 && c.in.bytes < (int64_t)1 << 32 && c.wgt_bytes < (int64_t)1 << 32, "conv: allocations must be < 4 GiB");
  RON_REQUIRE(c.out.pixels() * c.out.cstride < (int64_t)1 << 31, "conv: output too large for 32-bit offsets");
  PatchArgs a = PatchArgs();
  fill_conv_args(c, &a.c);
  RON_REQUIRE((int64_t)c.Npad * a.c.K * esz == c.wgt_bytes, "conv: packed weight size mismatch");
  if (c.pool) RON_REQUIRE(c.res == nullptr && !c.out_f32 && c.out.H == c.Ho / 2 && c.out.W == c.Wo / 2, "conv + fused pool: bad output view");
  PatchGeom g;
  patch_geom(c.in.N, c.in.H, c.in.W, c.in.pad, c.pool != 0, &g);
  a.flat = g.flat; a.TH = g.TH; a.TW = g.TW;
  a.PW = g.flat ? c.in.Wp() : g.TW + 2;
  a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y;
  a.H = c.in.H; a.W = c.in.W; a.pad = c.in.pad; a.n_img = c.in.N;
  a.chunks = c.in.C / conv_k_chunk(c.dtype);
  a.P0 = c.in.pad * c.in.Wp();
  a.max_pixel = (unsigned)(c.in.bytes / ((int64_t)c.in.cstride * esz) - 1);
  a.c.tiles_n = c.Npad / BN;
  *grid = g.tiles_sp * a.c.tiles_n;
  *out = a;
  return RON_OK;
}

// Can `a` and `b` share a launch of the patch kernel (conv3x3_patch_pair_kernel)?  Same input tensor and geometry (channel
// slices may differ), same dtype and N tile, no fused pool, and each big enough for the patch kernel to be the choice at all.
bool conv_patch_pair_applicable(const ConvLaunch& a, const ConvLaunch& b) {
  if (!conv_patch_applicable(a) || !conv_patch_applicable(b) || a.pool || b.pool) return false;
  if (a.dtype != b.dtype || a.in.base != b.in.base || a.in.bytes != b.in.bytes || a.in.N != b.in.N || a.in.H != b.in.H || a.in.W != b.in.W ||
      a.in.pad != b.in.pad || a.in.cstride != b.in.cstride || a.in.C != b.in.C)
    return false;
  const int ca = conv_patch_pick(a), cb = conv_patch_pick(b);
  return ca >= 0 && ca == cb;
}

int launch_conv_patch_pair(const ConvLaunch& ca, const ConvLaunch& cb, hipStream_t stream) {
  RON_REQUIRE(conv_patch_pair_applicable(ca, cb), "patch kernel pair: the two convolutions cannot share a launch");
  const int cfg = conv_patch_pick(ca);
  PatchArgs a, b;
  int grid_a = 0, grid_b = 0, rc;
  if ((rc = patch_args(ca, cfg, &a, &grid_a)) || (rc = patch_args(cb, cfg, &b, &grid_b))) return rc;
  const int grid = grid_a + grid_b;
  const int BN = patch_bn(cfg);
#define RON_PATCH_PAIR(Tr)                                                                          \
  do {                                                                                              \
    if (BN == 256) return launch_patch_pair_t<Tr, 256, 2, 2, 1>(a, b.c, grid_a, grid, stream);      \
    if (BN == 128) return launch_patch_pair_t<Tr, 128, 2, 3, 1>(a, b.c, grid_a, grid, stream);      \
    return launch_patch_pair_t<Tr, 64, 2, 2, 3>(a, b.c, grid_a, grid, stream);                      \
  } while (0)
  if (ca.dtype == RON_DTYPE_BF16) RON_PATCH_PAIR(TraitsBF16S);
  if (ca.dtype == RON_DTYPE_F16) RON_PATCH_PAIR(TraitsF16S);
  if (ca.dtype == RON_DTYPE_F16X3) RON_PATCH_PAIR(TraitsF16X3S);
  RON_PATCH_PAIR(TraitsF32S);
#undef RON_PATCH_PAIR
}

int launch_conv_patch(const ConvLaunch& c, int cfg, hipStream_t stream) {
  PatchArgs a;
  int grid = 0;
  const int rc0 = patch_args(c, cfg, &a, &grid);
  if (rc0) return rc0;
  const int BN = patch_bn(cfg);
#define RON_PATCH_DISPATCH(Tr)                                                      \
  do {                                                                              \
    if (BN == 256) return launch_patch_t<Tr, 256, 2, 2>(a, grid, stream);           \
    if (BN == 128) return launch_patch_t<Tr, 128, 2, 3>(a, grid, stream);           \
    return launch_patch_t<Tr, 64, 2, 2, 3>(a, grid, stream);                        \
  } while (0)
  if (c.dtype == RON_DTYPE_BF16) RON_PATCH_DISPATCH(TraitsBF16S);
  if (c.dtype == RON_DTYPE_F16) RON_PATCH_DISPATCH(TraitsF16S);
  if (c.dtype == RON_DTYPE_F16X3) RON_PATCH_DISPATCH(TraitsF16X3S);
  RON_PATCH_DISPATCH(TraitsF32S);
#undef RON_PATCH_DISPATCH
}

}  // namespace ron
