// DIAGNOSTIC BUILD ONLY (libron_hip_diag.so): the round-1 halo-patch kernel (6 x 40 / 8 x 32 pixel tiles, row-major weights),
// kept for A/B runs against csrc/conv_patch.hip.  tile_cfg 100 / 101 in the diagnostic library.
//
// 3x3 / stride-1 / pad-1 convolution with an LDS-staged input halo patch (gfx950).
//
// The generic kernel (conv_mfma.hip) re-stages the A operand for every filter tap, so the same input pixels travel
// L2 -> LDS nine times.  Under MFMA load the chip holds ~1.4-1.6 GHz and the staging stream, which runs on that clock,
// becomes the bound (ablation, b4_trio 256x256 tile: nothing staged 636 us; only the weights staged 699 us; only the
// activations 692 us; both 854 us).  Here a workgroup owns a TH x TW spatial tile of one image (<= 256 GEMM rows) and BN
// output channels.  Per 128-byte channel chunk it stages the (TH+2) x (TW+2) input patch ONCE (LDS-DMA, double buffered
// across chunks, one piece per tap step) and runs the 9 taps against it: the A fragment of tile row r for tap (ky,kx) is
// patch row pp(r) + ky*(TW+2) + kx, an LDS address shift.  Only the weights (BN x 128 B per step) are staged per tap:
// L2 -> LDS bytes per MFLOP drop from 7.6 (256x256 tile of the generic kernel) to 4.4 at BN = 256.
//
// Patch rows are 128 B; the 16-B chunk c of patch row i sits in slot c ^ ((i>>1)&7) (applied on the DMA source address
// and on the read side, as in the generic kernel; a tap shift changes i, so the key is recomputed per tap).
// MFMA shape (Tr::kMT), fragment double buffering, pinned issue order and everything after the K loop (vector epilogue with
// bias / ReLU / residual / fused 2x2 max-pool / fp32 heads, the B-row permutation behind it) are the generic kernel's.
#include "conv_device.h"

namespace ron {
namespace detail {

constexpr int kCfgPatchDiag = 100;

constexpr int kPatchPieces = 6;                  // LDS-DMA pieces per thread and chunk (512 threads x 16 B = 64 rows each)
constexpr int kPatchRows = 344;                  // rows a patch buffer holds (43 KB): 8 x 42, 10 x 34, 14 x 22 patches fit

// Swizzle key of patch row i (slot = chunk ^ key).  A tap shift moves a fragment's 16 or 32 rows to an arbitrary start row, so the
// key has to be conflict-free for every start (brute-forced over the ds_read_b128 lane groups): (i >> 1) & 7 is for the 32-row
// fragments of the 32x32 MFMA but 2-way conflicted for most starts with the 16-row fragments of the 16x16 MFMA, whose K-group
// bit is the chunk's bit 0; a key that leaves bit 0 alone, ((i >> 1) & 3) << 1, is conflict-free there for every start.
template <int MT> __device__ __forceinline__ int patch_key(int i) { return MT == 16 ? ((i >> 1) & 3) << 1 : (i >> 1) & 7; }

struct PatchArgs {
  ConvArgs c;
  int TH, TW, PW, R;          // tile, patch width (TW + 2), patch rows in use ((TH + 2) * PW <= kPatchRows)
  int tiles_x, tiles_y;       // per image
  int H, W;                   // conv output == input size
  int chunks;                 // Cin / chunk elements
};

// SB weight stages: SB-1 steps of lead for the per-tap weight tiles.
template <class Tr, int BN, int WN, int SB>
__global__ __launch_bounds__(512, 2) void conv3x3_patch_kernel(PatchArgs pa) {
  const ConvArgs& p = pa.c;
  constexpr int BM = 256, WM = 8 / WN, kThreads = 512;
  constexpr int MT = Tr::kMT, kGroups = 64 / MT, KS = 8 / kGroups, EPA = MT * MT / 64;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MR = TM / MT, NR = TN / MT;
  constexpr int B_IT = BN / 64;                   // B pieces per thread and step
  constexpr int G = B_IT + 1;                     // LDS-DMA instructions per thread and step (one patch piece + the weights)
  constexpr int kPatchBytes = kPatchRows * kRowBytes;
  constexpr int kBBytes = BN * kRowBytes;
  static_assert(TM % MT == 0 && TN % MT == 0 && NR <= 8 && SB >= 2, "bad wave tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: [patch 0][patch 1][B stage 0 .. SB-1][pp: BM ints][out_off: BM ints][sink 1 KB]
  char* s_b = smem + 2 * kPatchBytes;
  int* s_pp = reinterpret_cast<int*>(s_b + SB * kBBytes);
  int* s_out_off = s_pp + BM;
  // a zero-record LDS-DMA still writes (zeros): the placeholder pieces that keep the vmcnt groups uniform land here
  char* s_sink = reinterpret_cast<char*>(s_out_off + BM);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const unsigned nwg = gridDim.x, bid = blockIdx.x;
  const unsigned xcd = bid & 7u, q8 = nwg >> 3, r8 = nwg & 7u;
  const unsigned wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_n = (int)(wgid % (unsigned)p.tiles_n);
  unsigned tsp = wgid / (unsigned)p.tiles_n;                 // spatial tile: (img, ty, tx)
  const int tx = (int)(tsp % (unsigned)pa.tiles_x); tsp /= (unsigned)pa.tiles_x;
  const int ty = (int)(tsp % (unsigned)pa.tiles_y);
  const int img = (int)(tsp / (unsigned)pa.tiles_y);
  const int y0 = ty * pa.TH, x0 = tx * pa.TW, n0 = tile_n * BN;

  // tile row -> pixel of the TH x TW tile (window-major when the 2x2 pool is fused), patch row of tap (0,0), output offset
  for (int r = tid; r < BM; r += kThreads) {
    int ly, lx;
    if (p.pool) {
      const int w = r >> 2, hw = pa.TW >> 1;
      ly = 2 * (w / hw) + ((r >> 1) & 1);
      lx = 2 * (w % hw) + (r & 1);
    } else {
      ly = r / pa.TW;
      lx = r - ly * pa.TW;
    }
    const int y = y0 + ly, x = x0 + lx;
    const bool valid = ly < pa.TH && y < pa.H && x < pa.W;
    s_pp[r] = valid ? ly * pa.PW + lx : 0;
    int off;
    if (p.pool) off = ((img * p.out_Hp + (y >> 1) + p.out_pad) * p.out_Wp + (x >> 1) + p.out_pad) * p.out_cstride + p.out_coff;
    else off = ((img * p.out_Hp + y + p.out_pad) * p.out_Wp + x + p.out_pad) * p.out_cstride + p.out_coff;
    s_out_off[r] = valid ? off : -1;
  }

  // patch pieces of this thread: LDS patch row i = q >> 3 (q = k*512 + tid), slot = q & 7
  int p_voff[kPatchPieces];
#pragma unroll
  for (int k = 0; k < kPatchPieces; ++k) {
    const int q = k * kThreads + tid;
    const int i = q >> 3, slot = q & 7;
    const int ic = min(i, pa.R - 1);
    const int py = ic / pa.PW, px = ic - py * pa.PW;
    // patch origin = output (y0-1, x0-1) = padded (y0 - 1 + in_pad, x0 - 1 + in_pad); clamp inside the padded image
    const int gy = min(y0 + py + p.in_org, p.in_Hp - 1), gx = min(x0 + px + p.in_org, p.in_Wp - 1);
    p_voff[k] = (int)((((unsigned)(img * p.in_Hp + gy) * p.in_Wp + gx) * p.in_cstride + p.in_coff) * Tr::kEsz) +
                ((slot ^ patch_key<MT>(i)) << 4);
  }
  // B pieces: LDS row (j*MT + r) of a wave's TN-wide group <- weight row (r*NR + j)   (coalesced epilogue, see conv_mfma.hip)
  int b_voff[4];
  static_assert(B_IT <= 4, "BN <= 256");
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int lrow = it * 64 + (tid >> 3);
    const int grp = lrow / TN, loc = lrow % TN;
    const int nrow = grp * TN + (loc % MT) * NR + (loc / MT);
    b_voff[it] = (n0 + nrow) * p.K * Tr::kEsz + (((tid & 7) ^ ((tid >> 4) & 7)) << 4);
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

  // Every step issues one group of G = 1 + B_IT LDS-DMA instructions: one piece of the NEXT chunk's patch (a placeholder
  // into the sink when there is none to fetch) and the weights of step + SB - 1.  The prologue is shaped the same way, so
  // one counted vmcnt serves every step: all but the newest (SB-2) groups have landed = the weights of this step, and - at
  // a chunk boundary - the patch whose last piece went out three steps earlier.
  const int n_steps = pa.chunks * 9;
#pragma unroll
  for (int k = 0; k < kPatchPieces; ++k) PATCH_PIECE(k, 0, 0, true);
#pragma unroll
  for (int t = 0; t < SB - 1; ++t) {
    if (t > 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(RS_A(false), (lds_void*)s_sink, 16, p_voff[0], 0, 0, 0);
    const int soff = ((t % 9) * pa.chunks + t / 9) * kRowBytes;
#pragma unroll
    for (int it = 0; it < B_IT; ++it) B_PIECE(it, t, soff, t < n_steps);
  }

  int tap = 0, cc = 0, ky = 0, kx = 0;
  int ntap = (SB - 1) % 9, ncc = (SB - 1) / 9;      // tap / chunk of step + SB - 1
  int st_rd = 0, st_wr = SB - 1;
  for (int step = 0; step < n_steps; ++step) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((SB - 2) * G) : "memory");
    __builtin_amdgcn_s_barrier();
    const char* sa = smem + (cc & 1) * kPatchBytes;
    const char* sb = s_b + st_rd * kBBytes + b_base;
    if (++st_rd == SB) st_rd = 0;
    const int tapoff = ky * pa.PW + kx;
    int a_row[MR], a_key[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int row = pp[i] + tapoff;
      a_row[i] = row * kRowBytes;
      a_key[i] = patch_key<MT>(row);
    }
    u32x4 fa[2][MR], fb[2][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i) fa[0][i] = *reinterpret_cast<const u32x4*>(sa + a_row[i] + ((fh ^ a_key[i]) << 4));
#pragma unroll
    for (int j = 0; j < NR; ++j) fb[0][j] = *reinterpret_cast<const u32x4*>(sb + j * MT * kRowBytes + rd_off_b[0]);
    // this step's LDS-DMA group (branch free: the patch piece index is the tap, selected with v_cndmask)
    {
      const bool more = cc + 1 < pa.chunks && tap < kPatchPieces;
      int voff = p_voff[0];
#pragma unroll
      for (int k = 1; k < kPatchPieces; ++k) voff = tap == k ? p_voff[k] : voff;
      const bool in_ = tap * 64 + wave * 8 + 8 <= kPatchRows;
      char* d_ = (more && in_) ? smem + ((cc + 1) & 1) * kPatchBytes + (tap * kThreads + wave * 64) * 16 : s_sink;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(RS_A(more && in_), (lds_void*)d_, 16, voff, (cc + 1) * kRowBytes, 0, 0);
      const int soff = (ntap * pa.chunks + ncc) * kRowBytes;
      const bool live = step + SB - 1 < n_steps;
#pragma unroll
      for (int it = 0; it < B_IT; ++it) B_PIECE(it, st_wr, soff, live);
      if (++ntap == 9) { ntap = 0; ++ncc; }
      if (++st_wr == SB) st_wr = 0;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s < KS - 1) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
          fa[(s + 1) & 1][i] = *reinterpret_cast<const u32x4*>(sa + a_row[i] + (((kGroups * (s + 1) + fh) ^ a_key[i]) << 4));
#pragma unroll
        for (int j = 0; j < NR; ++j)
          fb[(s + 1) & 1][j] = *reinterpret_cast<const u32x4*>(sb + j * MT * kRowBytes + rd_off_b[s + 1]);
      }
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) Tr::mma(fa[s & 1][i], fb[s & 1][j], acc[i][j]);
    }
    // issue order: first fragments | k-step 0: MFMAs with the next reads and the G DMA instructions spaced between them |
    // ... | MFMAs of the last k-step
    {
      constexpr int RD = MR + NR, MM = MR * NR * Tr::kMfmaPerMma;
      __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int ps = s == 0 ? G : 0;
#pragma unroll
        for (int q = 0; q < MM; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (s < KS - 1 && q < RD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (((q + 1) * ps) / MM > (q * ps) / MM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        if (s < KS - 1 && RD > MM) __builtin_amdgcn_sched_group_barrier(0x100, RD - MM, 0);
#pragma unroll
        for (int x = 0; x < 8; ++x)
          if (x < ps - MM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    if (++tap == 9) { tap = 0; ++cc; ky = 0; kx = 0; }
    else if (++kx == 3) { kx = 0; ++ky; }
  }
#undef PATCH_PIECE
#undef B_PIECE
#undef RS_A
#undef RS_B

  // ---- epilogue (as conv_mfma.hip) ----
  const int nloc = wn * TN + fr * NR;
  float bias_v[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) bias_v[j] = p.bias[n0 + nloc + j];
  const int n_valid = p.Cout - (n0 + nloc);
  const int ncol0 = n0 + nloc;
  if (n_valid <= 0) return;
  if (p.pool) {
#pragma unroll
    for (int i = 0; i < MR; ++i) {
#pragma unroll
      for (int t = 0; t < EPA / 4; ++t) {
        const int ooff = s_out_off[wm * TM + i * MT + 8 * t + 4 * fh];
        if (ooff < 0) continue;
        float v[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const float mx = fmaxf(fmaxf(acc[i][j][4 * t], acc[i][j][4 * t + 1]), fmaxf(acc[i][j][4 * t + 2], acc[i][j][4 * t + 3]));
          v[j] = mx + bias_v[j];
          if (p.relu) v[j] = fmaxf(v[j], 0.f);
        }
        const int o = ooff + ncol0;
        if (n_valid >= NR) Tr::template store_vec<NR>(p.out, o, v);
        else {
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
      if (ooff < 0) continue;
      const int o = ooff + ncol0;
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

template <class Tr, int BN, int WN, int SB>
int launch_patch_t(const PatchArgs& a, int grid, hipStream_t s) {
  const size_t lds = 2 * (size_t)kPatchRows * kRowBytes + SB * (size_t)BN * kRowBytes + 2 * 256 * sizeof(int) + 1024;
  static PerDeviceOnce once;
  RON_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void*>(&conv3x3_patch_kernel<Tr, BN, WN, SB>), (int)lds));
  hipLaunchKernelGGL((conv3x3_patch_kernel<Tr, BN, WN, SB>), dim3(grid), dim3(512), lds, s, a);
  RON_HIP_CHECK(hipGetLastError());
  return RON_OK;
}

}  // namespace detail
using namespace detail;

// Spatial tile for an H x W map: <= 256 pixels, patch (TH+2) x (TW+2) <= kPatchRows rows.
static bool pick_tile(int H, int W, bool pool, int* TH, int* TW) {
  (void)H;
  int tw;
  if (W % 32 == 0) tw = 32;                      // 160 / 320-wide maps: 32 x 8
  else if (W % 40 == 0) tw = 40;                 // 40 / 80-wide maps: six rows of 40 (240 of 256 tile rows in use)
  else if (W == 20) tw = 20;                     // 20 x 12
  else return false;
  int th = 256 / tw;
  if (pool) { if (tw % 2) return false; th &= ~1; }
  if ((th + 2) * (tw + 2) > kPatchRows || th < 1) return false;
  *TH = th; *TW = tw;
  return true;
}

bool conv_patch_applicable(const ConvLaunch& c) {
  int th, tw;
  return c.kh == 3 && c.kw == 3 && c.stride == 1 && c.dil == 1 && c.cpad == 1 && c.up == 0 && c.in.H == c.Ho && c.in.W == c.Wo &&
         c.in.pad >= 1 && c.Npad % 64 == 0 && pick_tile(c.in.H, c.in.W, c.pool != 0, &th, &tw);
}

int launch_conv_patch(const ConvLaunch& c, int cfg_id, hipStream_t stream) {
  RON_REQUIRE(conv_patch_applicable(c), "patch kernel: not a 3x3 / stride 1 / pad 1 conv on a supported map");
  const int esz = (int)dtype_size(c.dtype), chunk = conv_k_chunk(c.dtype);
  RON_REQUIRE(c.in.C % chunk == 0, "conv: Cin %d is not a multiple of the K chunk %d", c.in.C, chunk);
  RON_REQUIRE(c.in.bytes > 0 && c.in.bytes < (int64_t)1 << 32 && c.wgt_bytes < (int64_t)1 << 32, "conv: allocations must be < 4 GiB");
  PatchArgs a = PatchArgs();
  ConvArgs& g = a.c;
  g.in = c.in.base; g.in_bytes = (unsigned)c.in.bytes; g.wgt = c.wgt; g.wgt_bytes = (unsigned)c.wgt_bytes;
  g.bias = c.bias; g.out = c.out.base; g.res = c.res;
  g.Ho = c.Ho; g.Wo = c.Wo; g.M = c.in.N * c.Ho * c.Wo;
  g.in_Hp = c.in.Hp(); g.in_Wp = c.in.Wp(); g.in_cstride = c.in.cstride; g.in_org = c.in.pad - 1; g.in_coff = c.in.coff;
  g.Cin = c.in.C; g.kw = 3; g.K = 9 * c.in.C; g.KT = g.K / chunk; g.stride = 1; g.dil = 1;
  g.Cout = c.Cout;
  g.out_Hp = c.out.Hp(); g.out_Wp = c.out.Wp(); g.out_cstride = c.out.cstride; g.out_pad = c.out.pad; g.out_coff = c.out.coff;
  g.relu = c.relu; g.out_f32 = c.out_f32; g.pool = c.pool; g.splitk = 1; g.Npad = c.Npad;
  RON_REQUIRE((int64_t)c.Npad * g.K * esz == c.wgt_bytes, "conv: packed weight size mismatch");
  if (c.pool) RON_REQUIRE(c.res == nullptr && !c.out_f32 && c.out.H == c.Ho / 2 && c.out.W == c.Wo / 2, "conv + fused pool: bad output view");
  pick_tile(c.in.H, c.in.W, c.pool != 0, &a.TH, &a.TW);
  a.PW = a.TW + 2; a.R = (a.TH + 2) * a.PW;
  a.tiles_x = (c.in.W + a.TW - 1) / a.TW; a.tiles_y = (c.in.H + a.TH - 1) / a.TH;
  a.H = c.in.H; a.W = c.in.W; a.chunks = c.in.C / chunk;
  const int BN = c.Npad % 256 == 0 ? 256 : (c.Npad % 128 == 0 ? 128 : 64);
  g.tiles_n = c.Npad / BN;
  const int grid = c.in.N * a.tiles_y * a.tiles_x * g.tiles_n;
#define RON_PATCH_DISPATCH(Tr)                                                                   \
  do {                                                                                           \
    typedef typename SmallShape<Tr>::type TS;                                                    \
    if (BN == 256 && cfg_id == kCfgPatchDiag + 1) return launch_patch_t<Tr, 256, 2, 2>(a, grid, stream); \
    if (BN == 256) return launch_patch_t<TS, 256, 2, 2>(a, grid, stream);                        \
    if (BN == 128) return launch_patch_t<TS, 128, 2, 3>(a, grid, stream);                        \
    return launch_patch_t<TS, 64, 2, 3>(a, grid, stream);                                        \
  } while (0)
  if (c.dtype == RON_DTYPE_BF16) RON_PATCH_DISPATCH(TraitsBF16);
  if (c.dtype == RON_DTYPE_F16) RON_PATCH_DISPATCH(TraitsF16);
  RON_PATCH_DISPATCH(TraitsF32);
#undef RON_PATCH_DISPATCH
}

}  // namespace ron
