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
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "conv_mfma.h"

namespace ron {
namespace detail {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct ConvArgs {
  const void* in;
  unsigned in_bytes;
  const void* wgt;
  unsigned wgt_bytes;
  const float* bias;
  void* out;
  const void* res;
  int M, Ho, Wo;
  int in_Hp, in_Wp, in_cstride, in_org;     // in_org = in.pad - cpad  (first tap of output (0,0))
  int in_coff;
  int Cin, kw, KT;                          // KT = kh*kw*Cin / chunk
  int stride, dil;
  int K;                                    // elements per weight row
  int Cout;
  int out_Hp, out_Wp, out_cstride, out_pad, out_coff;
  int up, up_cout;
  int relu, out_f32;
  int tiles_n;
};

struct TraitsBF16 {
  typedef __hip_bfloat16 elem;
  static constexpr int kEsz = 2;
  static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float load(const void* p, int i) {
    return __bfloat162float(reinterpret_cast<const __hip_bfloat16*>(p)[i]);
  }
  static __device__ __forceinline__ void store(void* p, int i, float v) {
    reinterpret_cast<__hip_bfloat16*>(p)[i] = __float2bfloat16(v);
  }
};
struct TraitsF16 {
  typedef _Float16 elem;
  static constexpr int kEsz = 2;
  static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float load(const void* p, int i) {
    return (float)reinterpret_cast<const _Float16*>(p)[i];
  }
  static __device__ __forceinline__ void store(void* p, int i, float v) {
    reinterpret_cast<_Float16*>(p)[i] = (_Float16)v;
  }
};
struct TraitsF32 {
  typedef float elem;
  static constexpr int kEsz = 4;
  static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x16& c) {
    // lane half h holds k = 4*(2s+h) + q, q = 0..3, for both operands: four exact-f32 MFMAs
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], fb[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1], fb[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2], fb[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3], fb[3], c, 0, 0, 0);
  }
  static __device__ __forceinline__ float load(const void* p, int i) { return reinterpret_cast<const float*>(p)[i]; }
  static __device__ __forceinline__ void store(void* p, int i, float v) { reinterpret_cast<float*>(p)[i] = v; }
};

constexpr int kThreads = 256;
constexpr int kRowBytes = 128;   // one LDS row = one K chunk of one tile row

typedef __attribute__((address_space(3))) void lds_void;

template <class Tr, int BM, int BN>
__global__ __launch_bounds__(kThreads, 2) void conv_igemm_kernel(ConvArgs p) {
  constexpr int MR = BM / 64, NR = BN / 64;          // 32x32 accumulators per wave: MR x NR
  constexpr int A_IT = BM / 32, B_IT = BN / 32;      // LDS-DMA wave-instructions per thread and K step
  constexpr int kABytes = BM * kRowBytes, kBBytes = BN * kRowBytes;
  constexpr int kStage = kABytes + kBBytes;
  constexpr int kChunkElems = kRowBytes / Tr::kEsz;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: [stage0: A | B][stage1: A | B][in_off: BM ints][out_off: BM ints]
  int* s_in_off = reinterpret_cast<int*>(smem + 2 * kStage);
  int* s_out_off = s_in_off + BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: workgroups that share an XCD (blockIdx % 8) take consecutive tiles,
  // so the N-tiles that re-read one A tile hit the same L2.
  const unsigned nwg = gridDim.x;
  const unsigned bid = blockIdx.x;
  const unsigned xcd = bid & 7u, q = nwg >> 3, r8 = nwg & 7u;
  const unsigned wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
  const int tile_n = (int)(wgid % (unsigned)p.tiles_n);
  const int tile_m = (int)(wgid / (unsigned)p.tiles_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // per-row addressing, once per tile
  if (tid < BM) {
    int m = m0 + tid;
    const bool valid = m < p.M;
    m = valid ? m : p.M - 1;
    const int hw = p.Ho * p.Wo;
    const int img = m / hw;
    const int rem = m - img * hw;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    const int iy = oy * p.stride + p.in_org, ix = ox * p.stride + p.in_org;
    s_in_off[tid] = (((img * p.in_Hp + iy) * p.in_Wp + ix) * p.in_cstride + p.in_coff) * Tr::kEsz;
    const int os = p.up > 0 ? p.up : 1;
    const int off = ((img * p.out_Hp + oy * os + p.out_pad) * p.out_Wp + ox * os + p.out_pad) * p.out_cstride +
                    p.out_coff;
    s_out_off[tid] = valid ? off : -1;
  }
  __syncthreads();

  // LDS-DMA source offsets (bytes): thread -> (row = it*32 + tid/8, slot = tid%8), source chunk = slot ^ key(row)
  const int ld_row = tid >> 3;
  const int ld_chunk = (tid & 7) ^ ((tid >> 4) & 7);
  // fixed-size arrays on purpose: with a template-dependent bound the LDS-DMA builtin's voffset becomes a
  // type-dependent expression and hipcc (ROCm 7.2) silently drops the kernel's host stub.
  int a_voff[8], b_voff[8];
  static_assert(A_IT <= 8 && B_IT <= 8, "tile too large");
#pragma unroll
  for (int it = 0; it < A_IT; ++it) a_voff[it] = s_in_off[it * 32 + ld_row] + ld_chunk * 16;
#pragma unroll
  for (int it = 0; it < B_IT; ++it) b_voff[it] = (n0 + it * 32 + ld_row) * p.K * Tr::kEsz + ld_chunk * 16;

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, p.wgt_bytes, 0x00020000);

  // K-step bookkeeping (wave-uniform): tap (ky, kx) and channel chunk cc
  int ky = 0, kx = 0, cc = 0;
#define RON_STAGE_LOAD(stage_, kt_)                                                                                  \
  do {                                                                                                               \
    const int a_soff = ((ky * p.dil * p.in_Wp + kx * p.dil) * p.in_cstride + cc) * Tr::kEsz;                         \
    const int b_soff = (kt_) * kRowBytes;                                                                            \
    char* dst = smem + (stage_) * kStage + wave * (8 * kRowBytes);                                                   \
    _Pragma("unroll") for (int it = 0; it < A_IT; ++it)                                                              \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(dst + it * 32 * kRowBytes), 16, a_voff[it],      \
                                                 a_soff, 0, 0);                                                      \
    _Pragma("unroll") for (int it = 0; it < B_IT; ++it)                                                              \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(dst + kABytes + it * 32 * kRowBytes), 16,         \
                                                 b_voff[it], b_soff, 0, 0);                                          \
    cc += kChunkElems;                                                                                               \
    if (cc >= p.Cin) {                                                                                               \
      cc = 0;                                                                                                        \
      if (++kx == p.kw) { kx = 0; ++ky; }                                                                            \
    }                                                                                                                \
  } while (0)

  f32x16 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment read offsets: lane -> row r = lane & 31, K half h = lane >> 5; step s reads chunk 2s+h
  const int fr = lane & 31, fh = lane >> 5;
  int rd_off[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) rd_off[s] = fr * kRowBytes + (((2 * s + fh) ^ ((fr >> 1) & 7)) << 4);
  const int a_base = wm * (BM / 2) * kRowBytes;
  const int b_base = kABytes + wn * (BN / 2) * kRowBytes;

  RON_STAGE_LOAD(0, 0);
  __syncthreads();
  for (int kt = 0; kt < p.KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < p.KT) RON_STAGE_LOAD(cur ^ 1, kt + 1);
    const char* sbuf = smem + cur * kStage;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 fa[MR], fb[NR];
#pragma unroll
      for (int i = 0; i < MR; ++i) fa[i] = *reinterpret_cast<const u32x4*>(sbuf + a_base + i * 32 * kRowBytes + rd_off[s]);
#pragma unroll
      for (int j = 0; j < NR; ++j) fb[j] = *reinterpret_cast<const u32x4*>(sbuf + b_base + j * 32 * kRowBytes + rd_off[s]);
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) Tr::mma(fa[i], fb[j], acc[i][j]);
    }
    __syncthreads();
  }

#undef RON_STAGE_LOAD
  // epilogue.  C/D layout of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
  int tap_off = 0, n_base = n0;
  if (p.up > 0) {
    const int tap = n0 / p.up_cout;                       // BN divides up_cout: uniform per tile
    tap_off = ((tap / p.up) * p.out_Wp + (tap % p.up)) * p.out_cstride;
    n_base = n0 - tap * p.up_cout;
  }
  float bias_v[NR];
  int ncol[NR];
  bool nok[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int nn = wn * (BN / 2) + j * 32 + fr;
    bias_v[j] = p.bias[n0 + nn];
    ncol[j] = n_base + nn;
    nok[j] = (n0 + nn) < p.Cout;
  }
#pragma unroll
  for (int i = 0; i < MR; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rt = wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
      const int ooff = s_out_off[rt];
      if (ooff < 0) continue;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        if (!nok[j]) continue;
        float v = acc[i][j][e] + bias_v[j];
        if (p.relu) v = fmaxf(v, 0.f);
        const int o = ooff + tap_off + ncol[j];
        if (p.res != nullptr) v = fmaxf(v + Tr::load(p.res, o), 0.f);
        if (p.out_f32) reinterpret_cast<float*>(p.out)[o] = v;
        else Tr::store(p.out, o, v);
      }
    }
  }
}

template <class Tr, int BM, int BN>
int launch_t(const ConvArgs& a, int tiles_m, hipStream_t s) {
  const size_t lds = 2 * (BM + BN) * kRowBytes + 2 * BM * sizeof(int);
  static bool attr_set = false;
  if (!attr_set) {
    RON_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<Tr, BM, BN>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<Tr, BM, BN>), dim3(tiles_m * a.tiles_n), dim3(kThreads), lds, s, a);
  RON_HIP_CHECK(hipGetLastError());
  return RON_OK;
}

}  // namespace detail
using namespace detail;

size_t dtype_size(int dtype) { return dtype == RON_DTYPE_F32 ? 4 : 2; }
int conv_k_chunk(int dtype) { return kRowBytes / (int)dtype_size(dtype); }
int conv_n_tile(int cout) { return cout <= 64 ? 64 : 128; }

int launch_conv(const ConvLaunch& c, hipStream_t stream) {
  const int esz = (int)dtype_size(c.dtype);
  const int chunk = conv_k_chunk(c.dtype);
  RON_REQUIRE(c.in.C % chunk == 0, "conv: Cin %d is not a multiple of the K chunk %d", c.in.C, chunk);
  RON_REQUIRE(c.in.pad >= c.cpad, "conv: input halo %d < conv padding %d", c.in.pad, c.cpad);
  RON_REQUIRE(c.in.bytes > 0 && c.in.bytes < (int64_t)1 << 32, "conv: input allocation must be < 4 GiB for buffer addressing");
  RON_REQUIRE(c.wgt_bytes > 0 && c.wgt_bytes < (int64_t)1 << 32, "conv: weight allocation must be < 4 GiB");
  RON_REQUIRE((int64_t)c.out.N * c.out.Hp() * c.out.Wp() * c.out.cstride < (int64_t)1 << 31, "conv: output too large for 32-bit offsets");
  const int BN = conv_n_tile(c.Cout);
  RON_REQUIRE(c.Npad % BN == 0, "conv: Npad %d not a multiple of the N tile %d", c.Npad, BN);
  if (c.up > 0) RON_REQUIRE(c.up_cout % BN == 0, "transposed conv: channels per tap %d not a multiple of %d", c.up_cout, BN);
  ConvArgs a;
  a.in = c.in.base; a.in_bytes = (unsigned)c.in.bytes;
  a.wgt = c.wgt; a.wgt_bytes = (unsigned)c.wgt_bytes;
  a.bias = c.bias; a.out = c.out.base; a.res = c.res;
  a.Ho = c.Ho; a.Wo = c.Wo; a.M = c.in.N * c.Ho * c.Wo;
  a.in_Hp = c.in.Hp(); a.in_Wp = c.in.Wp(); a.in_cstride = c.in.cstride; a.in_org = c.in.pad - c.cpad;
  a.in_coff = c.in.coff;
  a.Cin = c.in.C; a.kw = c.kw; a.K = c.kh * c.kw * c.in.C; a.KT = a.K / chunk;
  a.stride = c.stride; a.dil = c.dil;
  a.Cout = c.Cout;
  a.out_Hp = c.out.Hp(); a.out_Wp = c.out.Wp(); a.out_cstride = c.out.cstride; a.out_pad = c.out.pad;
  a.out_coff = c.out.coff;
  a.up = c.up; a.up_cout = c.up_cout;
  a.relu = c.relu; a.out_f32 = c.out_f32;
  a.tiles_n = c.Npad / BN;
  RON_REQUIRE((int64_t)c.Npad * a.K * esz == c.wgt_bytes, "conv: packed weight size mismatch");
  const int tiles_m = (a.M + 127) / 128;
  if (c.dtype == RON_DTYPE_BF16) return BN == 64 ? launch_t<TraitsBF16, 128, 64>(a, tiles_m, stream) : launch_t<TraitsBF16, 128, 128>(a, tiles_m, stream);
  if (c.dtype == RON_DTYPE_F16) return BN == 64 ? launch_t<TraitsF16, 128, 64>(a, tiles_m, stream) : launch_t<TraitsF16, 128, 128>(a, tiles_m, stream);
  if (c.dtype == RON_DTYPE_F32) return BN == 64 ? launch_t<TraitsF32, 128, 64>(a, tiles_m, stream) : launch_t<TraitsF32, 128, 128>(a, tiles_m, stream);
  ron::set_error("conv: unknown dtype %d", c.dtype);
  return RON_ERR_INVALID;
}

}  // namespace ron
