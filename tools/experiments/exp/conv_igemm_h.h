// Experimental builds only (make EXP=1, included by conv_mfma.hip under RON_EXP): the row-gather tile with half-chunk stages.
// Measured 10-17 % SLOWER than the shipped two-stage form on every wide layer (profiles/r02/sweep_conv_exp_v7_half_chunk_stages.txt);
// kept as the record of that experiment, parity-checked by tools/check_exp.py.
#pragma once

// ---- half-chunk stages ---------------------------------------------------------------------------------------------
// The same tile with the K step cut in two: a stage holds 64 bytes (32 bf16 / 16 fp32 = ONE MFMA k-step) of every tile row, four
// stages of 32 KB instead of two of 64 KB.  The tile above issues the 64 KB of step kt+1 during step kt and needs all of them at
// the next barrier: what a CU has in flight is at most one step, and the step takes as long as those 64 KB take to arrive
// (profiles/r02/sweep_conv_exp_v6_rows_per_tile.txt: a tile with 1/8 less MFMA work and the same staging takes the same time).
// Here the pieces of half-step q+3 go out during half-step q: three stages (96 KB) in flight, each with two half-steps to land.
// LDS rows are 64 B; 16-byte chunk c of row r sits in slot c ^ key4(r), key4 = {0,2,3,1}[(r >> 2) & 3]: conflict-free for the
// four 16-lane groups of ds_read_b128 (lanes {0-3,12-15,20-27}, ... of a 16-row x 4-chunk fragment).  The packed weights are
// unchanged (128-byte rows in 8-KB blocks): half-step h of K step kt reads bytes [64 h, 64 h + 64) of each row.
__device__ __forceinline__ int key4(int row) {
  const int k = (row >> 2) & 3;
  return (((k ^ (k >> 1)) & 1) << 1) | (k >> 1);
}

constexpr int igemm_h_lds_bytes(int BM, int BN, int S) { return S * (BM + BN) * 64 + 2 * BM * (int)sizeof(int); }

template <class Tr, int BM, int BN, int WM, int WN, int S, bool TI>
__device__ __forceinline__ void conv_igemm_tile_h(const ConvArgs& p, const unsigned bid, const unsigned nwg, char* smem) {
  constexpr int RB = 64;                             // LDS row bytes
  constexpr int MT = Tr::kMT;
  static_assert(MT == 16, "half-chunk stages: 16 x 16 MFMA traits (a fragment = 16 rows x 64 B)");
  constexpr int EPA = MT * MT / 64;
  constexpr int kThreads = WM * WN * 64;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MR = TM / MT, NR = TN / MT;
  constexpr int kRowsPerIt = kThreads / 4;           // tile rows one LDS-DMA pass of the block covers (4 lanes per row)
  constexpr int A_IT = BM / kRowsPerIt, B_IT = BN / kRowsPerIt;
  constexpr int LPT = A_IT + B_IT;
  constexpr int kABytes = BM * RB, kBBytes = BN * RB;
  constexpr int kChunkElems = kRowBytes / Tr::kEsz;  // elements of a whole 128-byte K chunk
  static_assert(BM % kRowsPerIt == 0 && BN % kRowsPerIt == 0 && S >= 3 && S <= 5 && NR <= 8, "bad tile");
  char* s_a = smem;
  char* s_b = smem + S * kABytes;
  int* s_in_off = reinterpret_cast<int*>(smem + S * (kABytes + kBBytes));
  int* s_out_off = s_in_off + BM;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const unsigned xcd = bid & 7u, q8 = nwg >> 3, r8 = nwg & 7u;
  const unsigned wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int zsplit = (int)(wgid / (unsigned)p.tiles_total);
  const unsigned tile = wgid - (unsigned)zsplit * (unsigned)p.tiles_total;
  const int tile_n = (int)(tile % (unsigned)p.tiles_n), tile_m = (int)(tile / (unsigned)p.tiles_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kt0 = zsplit * p.kt_split, kt1 = min(p.KT, kt0 + p.kt_split);

  for (int r = tid; r < BM; r += kThreads) {           // per-row addressing (plain convolutions / transposed: no fused pool here)
    int m = m0 + r;
    const bool valid = m < p.M;
    m = valid ? m : p.M - 1;
    const int hw = p.Ho * p.Wo;
    const int img = m / hw, rem = m - img * hw;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const int os = p.up > 0 ? p.up : 1;
    const int off = ((img * p.out_Hp + oy * os + p.out_pad) * p.out_Wp + ox * os + p.out_pad) * p.out_cstride + p.out_coff;
    const int iy = oy * p.stride + p.in_org, ix = ox * p.stride + p.in_org;
    s_in_off[r] = (int)((((unsigned)(img * p.in_Hp + iy) * p.in_Wp + ix) * p.in_cstride + p.in_coff) * Tr::kEsz);
    s_out_off[r] = valid ? off : -1;
  }
  __syncthreads();

  // LDS-DMA: thread -> (row = it * kRowsPerIt + tid / 4, slot = tid % 4), source chunk (of the 64-byte half) = slot ^ key4(row)
  const int ld_row = tid >> 2;
  int a_voff[4], b_voff[4];
  static_assert(A_IT <= 4 && B_IT <= 4, "tile too large");
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int row = it * kRowsPerIt + ld_row;
    a_voff[it] = s_in_off[row] + (((tid & 3) ^ key4(row)) << 4);
  }
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int lrow = it * kRowsPerIt + ld_row;
    const int grp = lrow / TN, loc = lrow % TN;
    const int nrow = n0 + grp * TN + (loc % MT) * NR + (loc / MT);      // the B-row permutation of the tile above
    b_voff[it] = (int)((unsigned)(nrow >> 6) * (unsigned)p.KT * (unsigned)kWeightBlockBytes + (unsigned)(nrow & 63) * kRowBytes +
                       (((tid & 3) ^ key4(lrow)) << 4));
  }

  // issue pointer: K step (tap ky, kx, channel chunk cc; TI: weight block tb, cb) and half ih of the NEXT half-step to stage
  const int chunks_per_tap = p.Cin / kChunkElems;
  const int n_taps = p.KT / chunks_per_tap;
  const int tap0 = TI ? kt0 % n_taps : kt0 / chunks_per_tap;
  int ky = tap0 / p.kw, kx = tap0 - (tap0 / p.kw) * p.kw;
  int cc = (TI ? kt0 / n_taps : kt0 - tap0 * chunks_per_tap) * kChunkElems;
  int tb = TI ? kt0 % n_taps : 0, cb = TI ? kt0 / n_taps : 0;
  int ikt = kt0, ih = 0, iq = 0;                                         // K step, half, half-step index of the issue pointer
#define RON_H_STAGE()                                                                                                  \
  do {                                                                                                                 \
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, ikt < kt1 ? p.in_bytes : 0u, 0x00020000);   \
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, ikt < kt1 ? p.wgt_bytes : 0u, 0x00020000); \
    const int a_soff = ((ky * p.dil * p.in_Wp + kx * p.dil) * p.in_cstride + cc) * Tr::kEsz + ih * RB;                  \
    const int b_soff = (TI ? tb * chunks_per_tap + cb : ikt) * kWeightBlockBytes + ih * RB;                            \
    char* dst_a = s_a + (iq % S) * kABytes + wave * 1024;                                                              \
    char* dst_b = s_b + (iq % S) * kBBytes + wave * 1024;                                                              \
    _Pragma("unroll") for (int i = 0; i < B_IT; ++i)                                                                   \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(dst_b + i * kRowsPerIt * RB), 16, b_voff[i], b_soff, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < A_IT; ++i)                                                                   \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(dst_a + i * kRowsPerIt * RB), 16, a_voff[i], a_soff, 0, 0); \
    ++iq;                                                                                                              \
    if (++ih == 2) {                                                                                                   \
      ih = 0; ++ikt;                                                                                                   \
      if (TI) {                                                                                                        \
        if (++kx == p.kw) { kx = 0; if (++ky * p.kw >= n_taps) { ky = 0; cc += kChunkElems; } }                          \
        if (++tb == n_taps) { tb = 0; ++cb; }                                                                          \
      } else {                                                                                                         \
        cc += kChunkElems;                                                                                             \
        if (cc >= p.Cin) { cc = 0; if (++kx == p.kw) { kx = 0; ++ky; } }                                                \
      }                                                                                                                \
    }                                                                                                                  \
  } while (0)

  typename Tr::acc_t acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < EPA; ++e) acc[i][j][e] = 0.f;

  const int fr = lane & (MT - 1), fh = lane / MT;
  const int rd_off = fr * RB + ((fh ^ key4(fr)) << 4);       // tile rows of a fragment start at a multiple of 16: key4(row) = key4(fr)
  const int a_base = wm * TM * RB, b_base = wn * TN * RB;

#pragma unroll
  for (int t = 0; t < S - 1; ++t) RON_H_STAGE();
  const int n_half = 2 * (kt1 - kt0);
  for (int q = 0; q < n_half; ++q) {
    wait_vmcnt<(S - 2) * LPT>();            // this wave's pieces of half-step q have landed; the S-2 younger groups may be in flight
    __builtin_amdgcn_s_barrier();           // ... everyone's have, and everyone is done reading half-step q-1
    const char* sbuf_a = s_a + (q % S) * kABytes + a_base + rd_off;
    const char* sbuf_b = s_b + (q % S) * kBBytes + b_base + rd_off;
    u32x4 fa[MR], fb[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) fb[j] = *reinterpret_cast<const u32x4*>(sbuf_b + j * MT * RB);
#pragma unroll
    for (int i = 0; i < MR; ++i) fa[i] = *reinterpret_cast<const u32x4*>(sbuf_a + i * MT * RB);
    RON_H_STAGE();                          // half-step q+S-1 into the stage half-step q-1 occupied
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j) Tr::mma(fa[i], fb[j], acc[i][j]);
  }
#undef RON_H_STAGE

  int tap_off = 0, n_base = n0;
  if (p.up > 0) {
    const int tap = n0 / p.up_cout;
    tap_off = ((tap / p.up) * p.out_Wp + (tap % p.up)) * p.out_cstride;
    n_base = n0 - tap * p.up_cout;
  }
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
  conv_epilogue<Tr, MR, NR, MT, EPA>(p, acc, s_out_off, wm * TM, fh, n0 + nloc, n_base + nloc, tap_off);
}

template <class Tr, int BM, int BN, int WM, int WN, int S, bool TI>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN) / 4) void conv_igemm_h_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  conv_igemm_tile_h<Tr, BM, BN, WM, WN, S, TI>(p, blockIdx.x, gridDim.x, smem);
}

