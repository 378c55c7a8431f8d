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
#include "conv_igemm_tile.h"

namespace ron {
namespace detail {

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
        if (asm_loop_ok(a.kt_split)) return launch_igemm4w(DtypeOf<Tr>::value, 256, a, s);               // four waves, assembly K loop (conv_mfma4w.hip)
      }
      return launch_t<Tr, 256, 256, 4, 2, 2, 1>(a, s);
    case kCfgIgemm128: return launch_t<Tr, 128, 128, 2, 2, 2, 1>(a, s);
    case kCfgIgemm128Early: case kCfgIgemm128EarlyTapsInner: return launch_t<Tr, 128, 128, 2, 2, 2, 2>(a, s);
    case kCfgIgemm128x64: return launch_t<Tr, 128, 64, 2, 2, 2, 2>(a, s);
    case kCfgIgemm256x128:
      if constexpr (AsmLoop<Tr>::value) {
        if (asm_loop_ok(a.kt_split)) return launch_igemm4w(DtypeOf<Tr>::value, 128, a, s);
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
int conv_pick_splitk(int tiles, int KT, int slots, int tile_elems) {
  if (tiles < 1 || tiles * 2 > slots || KT < 16) return 1;
  int sk = slots / tiles;
  int min_steps = 8;
  if (sk > KT / min_steps) sk = KT / min_steps;
  if (sk <= 1) return 1;
  // A split launch pays a second launch and writes, then reads, sk fp32 copies of its output: with many rows and a short K that
  // costs more than the chain it shortens (SSD-512 block8_conv1x1, 16 384 rows x 256 over 16 steps, kernel trace of round 5:
  // 20 us + a 16 us finalize pass for 33 MB of slabs).  Per K step of a workgroup ~1.1 us, ~5 us for the dependent launch,
  // ~4 TB/s for the slab traffic; the split stays where the model says it wins.  Same box, with / without (RON_SPLITK_NO_VETO=1):
  // SSD-512 batch 1 0.697 -> 0.685 ms, batch 4 1.157 -> 1.147, RON and the batch-16 / 32 / 64 benchmark configurations unchanged.
  static const bool veto = getenv("RON_SPLITK_NO_VETO") == nullptr;
  if (veto) {
    const double step_us = 1.1, launch_us = 5.0, bytes_per_us = 4e6;
    const double tile_bytes = 4.0 * tile_elems;       // fp32 slab of one tile (BM x BN of the configuration: 256 x 128 is half of 256 x 256)
    const double t_split = ((KT + sk - 1) / sk) * step_us + launch_us + 2.0 * sk * tiles * tile_bytes / bytes_per_us;
    if (t_split >= KT * step_us) return 1;
  }
  return sk;
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
    if (t256 >= 144) return (taps > 1 && Npad <= kTapsInnerMaxN) ? kCfgIgemm256TapsInner : kCfgIgemm256;      // (round 5: 160 -> 144, sweep_all_cfgs)
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
constexpr int kPanelCols = 8;      // the panel width of wide and tall launches (plain GEMM shapes)
static int pick_m_fastest(const ConvLaunch& c, int BM, int BN, int tiles_m, int tiles_n, int splitk, int panel_steps) {
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
  // (not the transposed conv: its pixel-shuffle epilogue wants the four taps of a pixel block written close in time - measured at
  // batch 4: the groups that carry block4 / block5_deconv_right 8-10 us slower in panel order)
  // Only the four-wave tiles decode it (conv_igemm_tile): `panel_steps` = the K steps of a workgroup when the launch may go to
  // them, 0 when it may not (members of a grouped launch).
  const bool panels_ok = BM == 256 && c.dtype != RON_DTYPE_F32 && panel_steps > 0 && asm_loop_ok(panel_steps) && c.up == 0;
  // experiments (tools/experiments/r06_calls/r06_panels.sh): RON_PANEL_COLS = P forces panels of P column tiles on every launch that
  // can decode them (P = 1: row tiles fastest, 0: column tiles fastest)
  static const char* force = getenv("RON_PANEL_COLS");
  if (force != nullptr && panels_ok) {
    const int P = atoi(force), cols = c.center_from > 0 ? c.center_from / BN : tiles_n;     // (centre-tap-only columns: panels among the long ones)
    return P >= cols ? 0 : P;
  }
  int pick = cost_m < 0.95 * cost_n ? 1 : 0;
  if (panels_ok && tiles_n >= 2 * kPanelCols && tiles_m >= 8 && c.center_from == 0) {
    const double cost_p = std::min(tiles_m, per_xcd / kPanelCols + 1) * a_tile +
                          std::min(tiles_n, kPanelCols * (per_xcd / (tiles_m * kPanelCols) + 1)) * b_tile;
    if (cost_p < 0.9 * std::min(cost_n, cost_m)) pick = kPanelCols;
  }
  // Round 6: the width of the panel from what an XCD's CUs fetch PER ROUND.  Its 32 workgroups run in step through K; with panels of P
  // column tiles they are 32 / P row tiles x P column tiles, and a round streams those row tiles' input lines and those column tiles'
  // weights through a 4 MB L2 that keeps neither for the next round (the weights of ONE 256 x 4608 column tile are 2.4 MB).  Measured
  // (profiles/r06/panel_width_sweep.txt, FETCH_SIZE x 2 + WRITE_SIZE per launch at batch 32): block4 trio3 group 1482 MB with its six long
  // columns fastest, 1002 with P = 2, 1133 / 1166 with 3 / 4; block5 trio3 group 667 -> 547; fc7 group 606 (P = 8) -> 514 (P = 4);
  // 7.40 -> 6.71 GB per step with P = 2 everywhere.  The launches' TIMES did not move (560.4 vs 560.7 us, the step 4.291 vs 4.279 ms):
  // these launches are not short of fabric bandwidth, and the bytes saved did not come back as clock either - the choice is made for the
  // traffic alone, and only where the model sees a clear difference.  A row tile's unique input lines: ~1.3 x its own rows for a 3 x 3
  // filter on these maps (the halo rows), not the 2 x of the column / row estimate above.
  const int cols = c.center_from > 0 ? c.center_from / BN : tiles_n;
  if (panels_ok && cols >= 3) {
    const double a_round = (double)BM * c.in.C * esz * (c.stride > 1 ? taps : (taps > 1 ? 1.3 : 1.0));
    const int resident = std::min(32, std::max(1, tiles_m * cols / 8));
    auto per_round = [&](int P) {
      P = std::max(1, std::min(P, cols));
      return ((resident + P - 1) / P) * a_round + P * b_tile;
    };
    const double now = per_round(pick == 0 ? cols : pick);
    int best = pick;
    double best_cost = now;
    for (int P : {2, 3, 4, 8}) {
      if (P >= cols) break;
      const double t = per_round(P);
      if (t < 0.95 * best_cost) { best = P; best_cost = t; }
    }
    if (best_cost < 0.95 * now) pick = best;
  }
  return pick;
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
  if (c.split_n == 0 && conv_c64_applicable(c)) return kCfgC64Resident;
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
      ((M + 255) / 256) * (c.Npad / 128) >= 256 && KT <= kAsmLoopMaxStepsHost && asm_loop_ok(KT))
    return kCfgIgemm256x128;
  const int cfg = conv_pick_igemm_cfg(M, c.center_from > 0 ? c.center_from : c.Npad, c.kh * c.kw, c.kh * c.kw * c.in.C, may_split);
  // taps innermost: only for the stride-1 convolutions it has been measured and tested on (the stride-2 3x3 convolutions of SSD-512's
  // extra blocks keep the tap-major order).  Centre-tap-only column tiles of such a launch keep the tap-major walk of their one tap.
  if (cfg == kCfgIgemm256TapsInner && (c.stride != 1 || pick_pos_major(c, kCfgIgemm256, 256))) return kCfgIgemm256;
  if (cfg == kCfgIgemm128Early && c.dtype != RON_DTYPE_F32 && c.center_from == 0) {
    // Where 256 x 256 tiles would leave most CUs idle (and K is not long enough to split them, above) the fallback used to be the
    // 128-row tiles: two workgroups per CU, twice the staged bytes per MAC.  Every tile configuration against this function's choice,
    // every layer shape, batches 4 .. 64 (profiles/r05/sweep_256x128_all_layers_batches.txt, sweep_all_cfgs_all_layers_batches.txt):
    // the 256 x 128 four-wave tile where it makes the launch about ONE round of the chip (48 .. 287 tiles; below 129 with K split in
    //    two): conv5_x at batch 32 / conv4_x at 8 / conv3_x at 4 75 -> 55 us (200 tiles), fc7 at batch 16 65 -> 51 (224); from 288
    //    (= 144 tiles of 256 x 256) on that tile wins by 13-17 %, below ~48 the launch is latency either way;
    const int t = ((M + 255) / 256) * (c.Npad / 128);
    if (c.up == 0 && KT <= kAsmLoopMaxStepsHost && asm_loop_ok(KT) && t >= 48 && t < 288) return kCfgIgemm256x128;
    // (The 128 x 64 tile for the smallest launches - SSD-512's tail convolutions 13.5 -> 11.5 us stand-alone - was tried as a third
    // rule: SSD-512 batch 1 -0.8 %, RON batch 2 +0.6 %; not kept.)
  }
  if (cfg == kCfgIgemm128Early && c.kh * c.kw > 1 && c.up == 0 && c.stride == 1) {
    // taps innermost for the 128 x 128 tile too (consecutive steps re-read almost the same input lines): 3-6 % on launches that do
    // not split K (with split-K it loses: conv5_1 at batch 4 +10 %), and where no filter rows can be skipped instead
    const int tiles = ((M + 127) / 128) * (c.Npad / 128);
    const bool splits = may_split && conv_pick_splitk(tiles, KT, igemm_slots(kCfgIgemm128Early), 128 * 128) > 1;
    if (!splits && !pick_pos_major(c, kCfgIgemm128Early, 128)) return kCfgIgemm128EarlyTapsInner;
  }
  return cfg;
}

// Everything launch_conv decides before it launches: the tile configuration and the kernel arguments (split-K, tile order, K order).
static int plan_conv(const ConvLaunch& c, int* cfg_out, ConvArgs* a_out);

int conv_describe(const ConvLaunch& c, int out[4]) {
  int cfg = -1;
  ConvArgs a = ConvArgs();
  const int rc = plan_conv(c, &cfg, &a);
  if (rc != RON_OK) return rc;
  out[0] = cfg;
  const bool gather = !conv_cfg_is_patch(cfg) && cfg != kCfgC64Resident;
  out[1] = gather ? a.splitk : 1; out[2] = gather ? a.m_fastest : 0; out[3] = gather ? a.taps_inner : 0;
  return RON_OK;
}

int launch_conv(const ConvLaunch& c, hipStream_t stream) {
  int cfg = -1;
  ConvArgs a = ConvArgs();
  int rc = plan_conv(c, &cfg, &a);
  if (rc != RON_OK) return rc;
  if (conv_cfg_is_patch(cfg)) return launch_conv_patch(c, cfg, stream);
  if (cfg == kCfgC64Resident) return launch_conv_c64(c, stream);
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

static int plan_conv(const ConvLaunch& c, int* cfg_out, ConvArgs* a_out) {
  ConvArgs& a = *a_out;
  int cfg = c.cfg >= 0 ? c.cfg : conv_pick_cfg(c);
  *cfg_out = cfg;
  RON_REQUIRE(cfg >= 0 && cfg < kNumCfgs, "conv: tile config %d out of range [0, %d)", cfg, kNumCfgs);
  if (conv_cfg_is_patch(cfg) || cfg == kCfgC64Resident) return RON_OK;      // kernels of their own: nothing more to plan here
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
  fill_conv_args(c, &a);
  a.tiles_n = c.Npad / BN;
  a.tiles_total = ((M + BM - 1) / BM) * a.tiles_n;
  a.splitk = 1; a.kt_split = a.KT; a.partial = nullptr;
  if (c.pool) {
    RON_REQUIRE(c.up == 0 && c.res == nullptr && !c.out_f32 && c.Ho % 2 == 0 && c.Wo % 2 == 0 && c.stride == 1,
                "conv + fused pool: plain stride-1 conv on an even map only");
    RON_REQUIRE(c.out.H == c.Ho / 2 && c.out.W == c.Wo / 2, "conv + fused pool: output view must be the pooled map");
  }
  if (c.split_n > 0) {
    RON_REQUIRE(c.out_f32 && !c.pool && c.up == 0 && c.res == nullptr && c.center_from == 0, "conv: two head outputs from a plain fp32-output convolution only");
    RON_REQUIRE(c.split_n % 8 == 0 && c.split_first > 0 && c.split_first <= c.split_n && c.split_n < c.Cout,
                "conv: bad output split (first %d, second from %d, %d columns)", c.split_first, c.split_n, c.Cout);
    RON_REQUIRE(c.out.pad == 0 && c.out.coff == 0 && c.out.cstride == c.out.C && c.out.C == c.split_first && c.out2.base != nullptr &&
                c.out2.pad == 0 && c.out2.coff == 0 && c.out2.cstride == c.out2.C && c.out2.C == c.Cout - c.split_n &&
                c.out2.H == c.Ho && c.out2.W == c.Wo && c.out2.N == c.in.N && c.out.H == c.Ho && c.out.W == c.Wo,
                "conv: the two head outputs must be dense [n][Ho][Wo][C] tensors of %d and %d channels", c.split_first, c.Cout - c.split_n);
    RON_REQUIRE(c.out2.pixels() * c.out2.cstride < (int64_t)1 << 31, "conv: second output too large for 32-bit offsets");
  } else if (c.out2.base != nullptr) {
    RON_REQUIRE(c.pool && c.out2.H == c.Ho && c.out2.W == c.Wo && c.out2.C >= c.Cout && c.out2.N == c.in.N,
                "conv: a second output is the un-pooled map of a fused-pool launch");
    RON_REQUIRE(c.out2.pixels() * c.out2.cstride < (int64_t)1 << 31, "conv: second output too large for 32-bit offsets");
  }
  const int sk = c.splitk >= 0 ? c.splitk : conv_pick_splitk(a.tiles_total, a.KT, igemm_slots(cfg), BM * BN);
  if (sk > 1 && c.up == 0 && !c.pool && c.scratch != nullptr && (int64_t)sk * M * c.Npad * 4 <= c.scratch_bytes) {
    a.kt_split = (a.KT + sk - 1) / sk;
    a.splitk = (a.KT + a.kt_split - 1) / a.kt_split;      // no empty split
    a.partial = (float*)c.scratch;
    if (a.splitk == 1) { a.kt_split = a.KT; a.partial = nullptr; }
  }
  a.m_fastest = pick_m_fastest(c, BM, BN, a.tiles_total / a.tiles_n, a.tiles_n, a.splitk, a.KT);
  a.pos_major = pick_pos_major(c, cfg, BM);
  a.taps_inner = conv_cfg_taps_inner(cfg) ? 1 : 0;
  // the 256 x 128 tile: taps innermost by the rule of the other tiles (stride-1 filters that neither split K nor skip filter rows)
  if (cfg == kCfgIgemm256x128 && c.kh * c.kw > 1 && c.stride == 1 && a.splitk == 1 && !a.pos_major) a.taps_inner = 1;
  if (c.center_from > 0)
    RON_REQUIRE(c.center_from % BN == 0 && (c.kh & 1) && (c.kw & 1) && c.up == 0,
                "conv: centre-tap-only columns need an odd filter and a boundary on the N tile (%d)", BN);
  RON_REQUIRE((int64_t)c.Npad * K * esz == c.wgt_bytes, "conv: packed weight size mismatch");
  return RON_OK;
}

// ---- grouped launches ---------------------------------------------------------------------------------------------
namespace detail {

template <class Tr>
int launch_group_cfg(int cfg, const ConvGroupArgs& g, bool any_split, hipStream_t s) {
  if (cfg == kGroupMixed) return launch_group_mixed<Tr>(g, any_split, s);
  if (cfg == kCfgIgemm128x64) return launch_group_t<Tr, 128, 64, 2, 2, 2, 2>(g, any_split, s);
  if (cfg == kCfgIgemm128) return launch_group_t<Tr, 128, 128, 2, 2, 2, 2>(g, any_split, s);      // pieces issued early, as kCfgIgemm128Early
  if (cfg == kCfgIgemm256) {                                                                       // two launches of < 1 round each as one
    if constexpr (AsmLoop<Tr>::value) {
      int longest = 0;
      for (int k = 0; k < g.n; ++k) longest = std::max(longest, g.op[k].kt_split);
      if (asm_loop_ok(longest)) return launch_igemm4w_group(DtypeOf<Tr>::value, g, any_split, s);
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
  for (int k = 0; k < n; ++k) RON_REQUIRE(ls_in[k].split_n == 0, "conv group: a two-output convolution is a launch of its own");
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
  // the 256 x 256 group runs on the four-wave tiles (which decode panel tile orders) when every member's K chain fits their step table
  bool group_4w = cfg == kCfgIgemm256 && ls[0].dtype != RON_DTYPE_F32;
  for (int k = 0; k < n && group_4w; ++k) {
    const int KT = ls[k].kh * ls[k].kw * ls[k].in.C / conv_k_chunk(ls[k].dtype);
    group_4w = asm_loop_ok((KT + sks[k] - 1) / std::max(sks[k], 1));
  }
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
    a.m_fastest = pick_m_fastest(c, BM, BN, a.tiles_total / a.tiles_n, a.tiles_n, a.splitk, group_4w ? a.kt_split : 0);
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
  const int sk = c.splitk >= 0 ? c.splitk : conv_pick_splitk(tiles, KT, igemm_slots(cfg), igemm_bm(cfg) * igemm_bn(cfg));
  return sk > 1 ? (int64_t)sk * M * c.Npad * 4 : 0;
}

}  // namespace ron
