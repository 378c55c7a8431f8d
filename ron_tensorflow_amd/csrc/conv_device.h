// Device-side pieces shared by the implicit-GEMM conv kernels (conv_mfma.hip, conv_patch.hip): kernel arguments,
// dtype traits (MFMA + conversions), vector stores.
#pragma once
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
typedef __attribute__((ext_vector_type(8))) unsigned u32x8;
typedef __attribute__((ext_vector_type(32))) float f32x32;

template <int N> struct alignas(4 * N) F32Vec { float v[N]; };
template <int N> __device__ __forceinline__ void store_f32_vec(float* p, const float* v) {
  F32Vec<N> t;
#pragma unroll
  for (int j = 0; j < N; ++j) t.v[j] = v[j];
  *reinterpret_cast<F32Vec<N>*>(p) = t;
}
template <class E, int N> struct alignas(sizeof(E) * N) EVec { E v[N]; };

struct ConvArgs {
  const void* in;
  unsigned in_bytes;
  const void* wgt;
  unsigned wgt_bytes;
  const float* bias;
  void* out;
  const void* res;
  int M, Ho, Wo;
  int in_Hp, in_Wp, in_cstride;
  int in_coff;
  int Cin, KT;                              // KT = kh*kw*Cin / chunk
  int K;                                    // elements per weight row
  int Cout;
  int out_Hp, out_Wp, out_cstride, out_coff;
  int up_cout;
  float oscale;                             // accumulator * oscale + bias (2^-k of the weight scale in split-precision mode, else 1)
  int tiles_n;
  // split-K: workgroup z of `splitk` covers K steps [z*kt_split, (z+1)*kt_split) and stores raw fp32 sums to
  // partial[z][m][n] (n < Npad); splitk_finalize_kernel adds the slabs and applies the epilogue.
  int splitk, kt_split, tiles_total, Npad;
  float* partial;
  // The small fields share two dwords: every kernel argument sits in a scalar register from the kernel's first instruction, and
  // ConvArgs was at the size the scalar register file carries.  Round 6, same box: 66 dwords -> 53 took hipcc's sgpr_spill_count from
  // 2-8 (plain tiles), 20-38 (four-wave and grouped tiles), 56-71 (mixed-width groups) to 0 everywhere: batch 1 0.702 -> 0.697 ms,
  // batch 32 two in flight 8 517 / 8 488 -> 8 608 / 8 593 images/s; 28 dwords MORE (multiplier divisions for the set-up) had cost
  // every launch 1-6 % (tools/experiments/README.md, fastdiv_setup.patch).
  unsigned in_org : 5;                      // in.pad - cpad  (first tap of output (0,0))
  unsigned kw : 4, kh : 4;
  unsigned stride : 3, dil : 5;
  unsigned out_pad : 4, out2_pad : 4;
  unsigned up : 3;                          // conv2d_transpose with kernel == stride == up (pixel-shuffle epilogue); 0 = plain conv
  unsigned relu : 1, out_f32 : 1;
  // fused 2x2 / stride-2 max-pool: tile rows are ordered window-major (rows 4q..4q+3 = the four conv outputs of
  // pooled pixel q), which puts a window into four consecutive accumulator registers of one lane.
  unsigned pool : 1;
  // pos_major: rows ordered by output row first (m = (oy * n_img + img) * Wo + ox) instead of image-major: a tile then covers few
  // output rows oy, and the filter rows ky whose taps fall into the zero halo for ALL of them are skipped.  (Round 4: the images used to
  // be the fastest index; consecutive rows were then one image plane apart, a multiple of 16 KB on fc6's input, i.e. on one L2
  // channel.)  fc6 (7x7 on a 10 x 10 map): 31 % of the MACs multiply halo zeros; skipping whole filter rows per tile recovers half.
  // taps_inner, the K order: 0 = tap-major (step = tap * chunks + chunk), 1 = chunk-major with the taps innermost (consecutive steps
  // re-read almost the same input lines one pixel over: the re-reads hit L2).  Centre-tap-only column tiles always walk their one tap
  // tap-major.
  unsigned pos_major : 1, taps_inner : 1;
  unsigned cpad : 6;
  // tile order inside an XCD's run of workgroups: 0 = N fastest (the column tiles that re-read one activation tile share an L2),
  // 1 = M fastest (the row tiles that re-read one weight slice do: fc6's 205 MB of weights are then fetched once, not once per XCD),
  // P >= 2 = panels of P column tiles walked row by row (four-wave tiles only; conv_mfma.hip, pick_m_fastest)
  unsigned m_fastest : 6;
  // fused pool + the full-resolution map too (conv4_3 / conv5_3 feed both their pool and a reverse-connection conv): `out2` is the
  // un-pooled output view, written from the same accumulators; nullptr = pooled map only
  void* out2;
  int out2_Hp, out2_Wp, out2_cstride, out2_coff;
  // Two fp32 head outputs from ONE convolution over a shared input (the loc and cls convolutions of an SSD feature layer,
  // nets/ssd_vgg_300.py:403-431, packed side by side): columns [0, split_first) are channels of `out`, columns [split_n, Cout) are
  // channels [0, Cout - split_n) of `out2` (split_n = split_first rounded up to 8, the columns between are padding).  Both views
  // are un-haloed [n][Ho][Wo][C] tensors, so a row's offset in `out2` is its offset in `out` / out_cstride * out2_cstride.
  // 0: one output.
  int split_n, split_first;
  int n_img, in_H;                          // (position-major rows)
  // Column tiles at or beyond output channel center_from_n (0: none) hold a 1x1 branch whose weights sit in the centre tap of the
  // kh x kw filter, zeros elsewhere: they run the K steps of that tap only.
  int center_from_n;
};

// MFMA shape of a traits class: kMT x kMT output tile per instruction (32: v_mfma_f32_32x32x16, 16 accumulator registers;
// 16: v_mfma_f32_16x16x32, 4 registers).  One u32x4 fragment per lane feeds one instruction either way: lane l holds
// row l % kMT, 16-byte K group l / kMT.  Same FLOP per cycle; the 16x16 form holds a higher clock under load
// (MI355X_MICROARCH.md, DVFS item 7).
struct TraitsBF16 {
  static constexpr bool kIsBf16 = true;
  typedef __hip_bfloat16 elem;
  typedef f32x16 acc_t;
  static constexpr int kMT = 32;
  static constexpr int kEsz = 2;
  static constexpr int kMfmaPerMma = 1;
  static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float load(const void* p, int i) {
    return __bfloat162float(reinterpret_cast<const __hip_bfloat16*>(p)[i]);
  }
  static __device__ __forceinline__ void store(void* p, int i, float v) {
    reinterpret_cast<__hip_bfloat16*>(p)[i] = __float2bfloat16(v);
  }
  template <int N> static __device__ __forceinline__ void store_vec(void* p, int i, const float* v) {
    EVec<__hip_bfloat16, N> t;
#pragma unroll
    for (int j = 0; j < N; ++j) t.v[j] = __float2bfloat16(v[j]);
    *reinterpret_cast<EVec<__hip_bfloat16, N>*>(reinterpret_cast<__hip_bfloat16*>(p) + i) = t;
  }
  template <int N> static __device__ __forceinline__ void load_vec(const void* p, int i, float* v) {
    const EVec<__hip_bfloat16, N> t = *reinterpret_cast<const EVec<__hip_bfloat16, N>*>(reinterpret_cast<const __hip_bfloat16*>(p) + i);
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = __bfloat162float(t.v[j]);
  }
};
struct TraitsF16 {
  static constexpr bool kIsBf16 = false;
  typedef _Float16 elem;
  typedef f32x16 acc_t;
  static constexpr int kMT = 32;
  static constexpr int kEsz = 2;
  static constexpr int kMfmaPerMma = 1;
  static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float load(const void* p, int i) {
    return (float)reinterpret_cast<const _Float16*>(p)[i];
  }
  static __device__ __forceinline__ void store(void* p, int i, float v) {
    reinterpret_cast<_Float16*>(p)[i] = (_Float16)v;
  }
  template <int N> static __device__ __forceinline__ void store_vec(void* p, int i, const float* v) {
    EVec<_Float16, N> t;
#pragma unroll
    for (int j = 0; j < N; ++j) t.v[j] = (_Float16)v[j];
    *reinterpret_cast<EVec<_Float16, N>*>(reinterpret_cast<_Float16*>(p) + i) = t;
  }
  template <int N> static __device__ __forceinline__ void load_vec(const void* p, int i, float* v) {
    const EVec<_Float16, N> t = *reinterpret_cast<const EVec<_Float16, N>*>(reinterpret_cast<const _Float16*>(p) + i);
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = (float)t.v[j];
  }
};
struct TraitsF32 {
  typedef float elem;
  typedef f32x16 acc_t;
  static constexpr int kMT = 32;
  static constexpr int kEsz = 4;
  static constexpr int kMfmaPerMma = 4;
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
  template <int N> static __device__ __forceinline__ void store_vec(void* p, int i, const float* v) {
    store_f32_vec<N>(reinterpret_cast<float*>(p) + i, v);
  }
  template <int N> static __device__ __forceinline__ void load_vec(const void* p, int i, float* v) {
    const F32Vec<N> t = *reinterpret_cast<const F32Vec<N>*>(reinterpret_cast<const float*>(p) + i);
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = t.v[j];
  }
};

// 16x16 forms: everything but the instruction is inherited
struct TraitsBF16S : TraitsBF16 {
  typedef f32x4 acc_t;
  static constexpr int kMT = 16;
  static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c, 0, 0, 0);
  }
};
struct TraitsF16S : TraitsF16 {
  typedef f32x4 acc_t;
  static constexpr int kMT = 16;
  static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  }
};
struct TraitsF32S : TraitsF32 {
  typedef f32x4 acc_t;
  static constexpr int kMT = 16;
  static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& c) {
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], c, 0, 0, 0);
  }
};
// Split precision (RON_DTYPE_F16X3): an element is two f16 planes, hi = rnd(v) and lo = rnd(v - hi); a 128-byte row chunk holds
// 32 elements as [32 x hi][32 x lo].  The K loop sees 4-byte elements like the fp32 mode (same addressing, 32 elements per
// staging step); k-step 0 of a stage reads the hi fragments (16-byte slots 0..3), k-step 1 the lo fragments (slots 4..7), and
// the products are hi*hi in k-step 0, lo*hi + hi*lo in k-step 1 (mma_step below) -- three v_mfma_f32_16x16x32_f16 per
// 16 x 16 x 32 block into one fp32 accumulator.  hi*hi is exact in the matrix core (11 x 11 bits), the dropped lo*lo term is
// 2^-22 of the product.  Element index i of a tensor -> f16 index (i / 32) * 64 + i % 32 (+ 32 for the lo plane): pixel
// strides and channel slices are multiples of 32 elements everywhere.
// One value -> its two f16 planes.  Kernels that store split values run with MODE.FP16_OVFL set (split_mode_on() at their start):
// an fp32 -> f16 conversion that overflows then SATURATES at the largest finite f16 (65504) instead of producing inf, at no cost per
// value (clamping in VALU instructions made the store-bound conv1_x epilogues 11-19 % slower).  Without it hi = +inf, lo = -inf
// would read back as NaN and poison every later layer; saturated, |v| up to 131008 is still representable (hi = 65504, lo = the
// rest, with the f16 spacing of 32 up there) and larger values clip.  True infinities and NaN stay what they are.  Accuracy contract
// (include/ron_hip.h, RON_DTYPE_F16X3): 22 mantissa bits for 2^-3 <= |v| < 65504; below, lo is an f16 subnormal (not flushed by
// the matrix core), i.e. an ABSOLUTE error floor of 2^-25 per stored value.
static __device__ __forceinline__ void split_mode_on() {
  __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);      // hwreg(HW_REG_MODE, offset 23, size 1) = FP16_OVFL
}
static __device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)(v - (float)hi);
}
struct TraitsF16X3S {
  static constexpr bool kIsBf16 = false;
  typedef f32x4 acc_t;
  static constexpr int kMT = 16;
  static constexpr int kEsz = 4;
  static constexpr int kMfmaPerMma = 1;
  static constexpr bool kSplit = true;
  static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ int hidx(int i) { return ((i >> 5) << 6) + (i & 31); }
  static __device__ __forceinline__ float load(const void* p, int i) {
    const _Float16* h = reinterpret_cast<const _Float16*>(p) + hidx(i);
    return (float)h[0] + (float)h[32];
  }
  static __device__ __forceinline__ void store(void* p, int i, float v) {
    _Float16* h = reinterpret_cast<_Float16*>(p) + hidx(i);
    _Float16 hi, lo;
    split_f16(v, hi, lo);
    h[0] = hi;
    h[32] = lo;
  }
  template <int N> static __device__ __forceinline__ void store_vec(void* p, int i, const float* v) {
    static_assert(32 % N == 0, "a vector must not straddle a 32-element chunk");
    EVec<_Float16, N> hi, lo;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      split_f16(v[j], hi.v[j], lo.v[j]);
    }
    _Float16* h = reinterpret_cast<_Float16*>(p) + hidx(i);
    *reinterpret_cast<EVec<_Float16, N>*>(h) = hi;
    *reinterpret_cast<EVec<_Float16, N>*>(h + 32) = lo;
  }
  template <int N> static __device__ __forceinline__ void load_vec(const void* p, int i, float* v) {
    const _Float16* h = reinterpret_cast<const _Float16*>(p) + hidx(i);
    const EVec<_Float16, N> hi = *reinterpret_cast<const EVec<_Float16, N>*>(h);
    const EVec<_Float16, N> lo = *reinterpret_cast<const EVec<_Float16, N>*>(h + 32);
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = (float)hi.v[j] + (float)lo.v[j];
  }
};

// traits whose 256 x 256 tile has the four-wave assembly K loop (kloop4w.inc): bf16, f16 and split precision on v_mfma_f32_16x16x32
template <class Tr> struct AsmLoop { static constexpr bool value = false; };
template <> struct AsmLoop<TraitsBF16S> { static constexpr bool value = true; };
template <> struct AsmLoop<TraitsF16S> { static constexpr bool value = true; };
template <> struct AsmLoop<TraitsF16X3S> { static constexpr bool value = true; };
template <class Tr, class = void> struct IsSplit { static constexpr bool value = false; };
template <class Tr> struct IsSplit<Tr, decltype((void)Tr::kSplit)> { static constexpr bool value = Tr::kSplit; };
// MFMA instructions one (row tile, column tile) pair issues in k-step s of a stage
template <class Tr> constexpr int mfma_in_step(int s) { return IsSplit<Tr>::value ? (s == 0 ? 1 : 2) : Tr::kMfmaPerMma; }
// k-step s of a stage for the pair (i, j); fa / fb: the two fragment register sets (set s & 1 was read for k-step s)
template <class Tr, int MR, int NR>
__device__ __forceinline__ void mma_step(int s, const u32x4 (&fa)[2][MR], const u32x4 (&fb)[2][NR], int i, int j, typename Tr::acc_t& c) {
  if constexpr (IsSplit<Tr>::value) {
    if (s == 0) {
      Tr::mma(fa[0][i], fb[0][j], c);      // hi * hi
    } else {
      Tr::mma(fa[1][i], fb[0][j], c);      // lo * hi
      Tr::mma(fa[0][i], fb[1][j], c);      // hi * lo
    }
  } else {
    Tr::mma(fa[s & 1][i], fb[s & 1][j], c);
  }
}

template <class Tr> struct SmallShape;
template <> struct SmallShape<TraitsBF16> { typedef TraitsBF16S type; };
template <> struct SmallShape<TraitsF16> { typedef TraitsF16S type; };
template <> struct SmallShape<TraitsF32> { typedef TraitsF32S type; };

constexpr int kRowBytes = 128;   // one LDS row = one K chunk of one tile row
constexpr int kWeightBlockBytes = kWeightBlockRows * kRowBytes;   // one (64 rows, K step) block of the packed weights

typedef __attribute__((address_space(3))) void lds_void;

// Everything after the K loop, shared by the row-gather and the halo-patch kernel.  A lane holds, for each of its MR row
// tiles, EPA accumulator registers (rows row0 + i*MT + (e & 3) + 8 * (e >> 2) + 4 * fh of the workgroup tile) times NR
// ADJACENT output channels starting at n_glob (bias / validity) = channel n_store of the output view: + bias, ReLU,
// optional relu(x + residual) (reverse-connection sum, nets/ron_vgg_320.py:425), optional fused 2x2 max-pool (tile rows
// are ordered window-major, so a pool window is four consecutive accumulator registers of one lane), one vector store
// per row (dtype or fp32).  s_out_off[row] = element offset of the row's output pixel, -1 for rows that store nothing.
// `rd.row(i, e, v)` delivers v[j] = accumulator register e of the lane's (row tile i, column tile j), j < NR: from a register array
// (AccArray) or, for the assembly K loop, straight from the accumulation registers (conv_mfma.hip).
template <class Tr, int MR, int NR>
struct AccArray {
  // The row loops specialised per epilogue switch (conv_epilogue_rows) are for the four-wave assembly tiles, where one wave per
  // SIMD hides nothing; the kernels that keep their accumulators in a register array run two waves per SIMD and take the run-time
  // form (specialising all 45 tile instantiations x 6 forms took the build of conv_mfma.hip from 70 s to 4.5 min)
  static constexpr bool kSpecialise = false;
  typename Tr::acc_t (&acc)[MR][NR];
  __device__ __forceinline__ void row(int i, int e, float (&v)[NR]) const {
#pragma unroll
    for (int j = 0; j < NR; ++j) v[j] = acc[i][j][e];
  }
};

// The rows of a lane whose NR channels all exist, for one combination of the launch's epilogue switches: straight-line code per row
// (the switches are wave-uniform and decided once, outside), the rows' output offsets read from LDS up front in one batch.  With one
// wave per SIMD (the four-wave 256 x 256 tile) nothing hides a per-row LDS round trip or a per-value select: the generic row loop took
// 21.6 k cycles per tile there, an eighth of a 72-step tile.
template <class Tr, int MR, int NR, int MT, int EPA, bool RELU, bool RES, bool F32, class Reader>
__device__ __forceinline__ void conv_epilogue_rows(const ConvArgs& p, const Reader& rd, const int* s_out_off, int row0, int fh,
                                                   const float (&bias_v)[NR], int n_store, int tap_off) {
  int ooff[MR][EPA];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int e = 0; e < EPA; ++e) ooff[i][e] = s_out_off[row0 + i * MT + (e & 3) + 8 * (e >> 2) + 4 * fh];
  const float oscale = p.oscale;
#pragma unroll
  for (int i = 0; i < MR; ++i) {
    // The residuals of the EPA rows of this block are loaded together and made to ARRIVE together (the empty asm), in front of the
    // rows' stores: a load waited for inside a row's branch is waited for with `vmcnt(0)`, which is also the previous row's store
    // (conv_epilogue_r, the bias values).  Rows that do not exist read element 0 (any valid address).
    float rv[RES ? EPA : 1][NR];
    if (RES) {
#pragma unroll
      for (int e = 0; e < EPA; ++e) Tr::template load_vec<NR>(p.res, ooff[i][e] < 0 ? 0 : ooff[i][e] + tap_off + n_store, rv[e]);
#pragma unroll
      for (int e = 0; e < EPA; ++e)
#pragma unroll
        for (int j = 0; j < NR; ++j) asm volatile("" : "+v"(rv[e][j]));
    }
#pragma unroll
    for (int e = 0; e < EPA; ++e) {
      float v[NR];
      rd.row(i, e, v);
      const int o = ooff[i][e] + tap_off + n_store;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        v[j] = fmaf(v[j], oscale, bias_v[j]);
        if (RELU) v[j] = fmaxf(v[j], 0.f);
      }
      if (ooff[i][e] < 0) continue;
      if (RES) {
#pragma unroll
        for (int j = 0; j < NR; ++j) v[j] = fmaxf(v[j] + rv[e][j], 0.f);
      }
      if (F32) store_f32_vec<NR>(reinterpret_cast<float*>(p.out) + o, v);
      else Tr::template store_vec<NR>(p.out, o, v);
    }
  }
}

// The fused 2x2 max-pool form (tile rows are window-major: registers 4t..4t+3 of a lane are one window), optionally with the
// un-pooled map as a second output; same structure as above.
template <class Tr, int MR, int NR, int MT, int EPA, bool RELU, bool OUT2, class Reader>
__device__ __forceinline__ void conv_epilogue_pool_rows(const ConvArgs& p, const Reader& rd, const int* s_out_off, const int* s_out2_off,
                                                        int row0, int fh, const float (&bias_v)[NR], int n_store) {
  int ooff[MR][EPA / 4], ooff2[MR][EPA];
#pragma unroll
  for (int i = 0; i < MR; ++i) {
#pragma unroll
    for (int t = 0; t < EPA / 4; ++t) ooff[i][t] = s_out_off[row0 + i * MT + 8 * t + 4 * fh];
    if (OUT2) {
#pragma unroll
      for (int e = 0; e < EPA; ++e) ooff2[i][e] = s_out2_off[row0 + i * MT + (e & 3) + 8 * (e >> 2) + 4 * fh];
    }
  }
  const float oscale = p.oscale;
#pragma unroll
  for (int i = 0; i < MR; ++i) {
#pragma unroll
    for (int t = 0; t < EPA / 4; ++t) {
      float w[4][NR];
#pragma unroll
      for (int u = 0; u < 4; ++u) rd.row(i, 4 * t + u, w[u]);
      if (OUT2) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float v[NR];
#pragma unroll
          for (int j = 0; j < NR; ++j) {
            v[j] = fmaf(w[u][j], oscale, bias_v[j]);
            if (RELU) v[j] = fmaxf(v[j], 0.f);
          }
          if (ooff2[i][4 * t + u] >= 0) Tr::template store_vec<NR>(p.out2, ooff2[i][4 * t + u] + n_store, v);
        }
      }
      // max over the 2x2 window = max over registers 4t..4t+3; relu(max(x) + b) == max(relu(x + b))
      float v[NR];
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const float mx = fmaxf(fmaxf(w[0][j], w[1][j]), fmaxf(w[2][j], w[3][j]));
        v[j] = fmaf(mx, oscale, bias_v[j]);
        if (RELU) v[j] = fmaxf(v[j], 0.f);
      }
      if (ooff[i][t] >= 0) Tr::template store_vec<NR>(p.out, ooff[i][t] + n_store, v);
    }
  }
}

template <class Tr, int MR, int NR, int MT, int EPA, class Reader>
__device__ __forceinline__ void conv_epilogue_r(const ConvArgs& p, const Reader& rd, const int* s_out_off, int row0,
                                                int fh, int n_glob, int n_store, int tap_off, const int* s_out2_off = nullptr) {
  float bias_v[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) bias_v[j] = p.bias[n_glob + j];
  // The bias values must have ARRIVED here, in front of the row loops.  Their first use is inside a row's `offset >= 0` branch; left to
  // itself the compiler waits for them there, in every row, with `s_waitcnt vmcnt(0)` - and vmcnt counts the rows' STORES too, in order:
  // each row then waited for the previous row's store to complete (32 rows x ~390 cycles = the 12.5 k cycles a 256 x 256 tile's
  // epilogue took; found in the ISA, round 5).
#pragma unroll
  for (int j = 0; j < NR; ++j) asm volatile("" : "+v"(bias_v[j]));
  // (ConvArgs::split_n: the lane's channels belong to the first output, or - from split_n on - to the second)
  const bool second = p.split_n > 0 && n_glob >= p.split_n;
  const int n_valid = (p.split_n > 0 && !second ? p.split_first : p.Cout) - n_glob;   // channels of this lane's group that exist (may be <= 0)
  if (n_valid <= 0) return;
  if constexpr (Reader::kSpecialise) {
    if (n_valid >= NR && p.split_n == 0) {
      // every channel of the lane's vector exists (all lanes but the last column tile's tail): the combinations the graphs launch -
      // ReLU layers (1), head logits in fp32 (4), the reverse connection's relu(x + residual) (3), plain (0), pooled ReLU layers -
      // as straight-line row loops; anything else through the run-time form below
      if (p.pool) {
        if (p.relu) {
          if (p.out2 != nullptr && s_out2_off != nullptr)
            conv_epilogue_pool_rows<Tr, MR, NR, MT, EPA, true, true>(p, rd, s_out_off, s_out2_off, row0, fh, bias_v, n_store);
          else
            conv_epilogue_pool_rows<Tr, MR, NR, MT, EPA, true, false>(p, rd, s_out_off, s_out2_off, row0, fh, bias_v, n_store);
          return;
        }
      } else {
        const int sw = (p.relu ? 1 : 0) | (p.res != nullptr ? 2 : 0) | (p.out_f32 ? 4 : 0);
#define RON_EPI_CASE(k_) \
        case k_: conv_epilogue_rows<Tr, MR, NR, MT, EPA, ((k_) & 1) != 0, ((k_) & 2) != 0, ((k_) & 4) != 0>(p, rd, s_out_off, row0, fh, bias_v, n_store, tap_off); return;
        switch (sw) { RON_EPI_CASE(0) RON_EPI_CASE(1) RON_EPI_CASE(3) RON_EPI_CASE(4) default: break; }
#undef RON_EPI_CASE
      }
    }
  }
  // The run-time form (every combination of the switches, full and partial channel vectors): what the register-array tiles run -
  // two waves per SIMD hide its per-row LDS round trips, and one copy of the row loop keeps those kernels small (45 instantiations)
  if (p.pool) {
    const bool out2 = p.out2 != nullptr && s_out2_off != nullptr;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
#pragma unroll
      for (int t = 0; t < EPA / 4; ++t) {
        float w[4][NR];
#pragma unroll
        for (int u = 0; u < 4; ++u) rd.row(i, 4 * t + u, w[u]);
        if (out2) {
          // the un-pooled map as well (s_out2_off[row] = element offset of the row's pixel in out2, channel slice included)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int e = 4 * t + u;
            const int ooff = s_out2_off[row0 + i * MT + (e & 3) + 8 * (e >> 2) + 4 * fh];
            if (ooff < 0) continue;
            float v[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
              v[j] = fmaf(w[u][j], p.oscale, bias_v[j]);
              if (p.relu) v[j] = fmaxf(v[j], 0.f);
            }
            const int o = ooff + n_store;
            if (n_valid >= NR) {
              Tr::template store_vec<NR>(p.out2, o, v);
            } else {
#pragma unroll
              for (int j = 0; j < NR; ++j) if (j < n_valid) Tr::store(p.out2, o + j, v[j]);
            }
          }
        }
        // max over the 2x2 window = max over registers 4t..4t+3; relu(max(x) + b) == max(relu(x + b))
        const int ooff = s_out_off[row0 + i * MT + 8 * t + 4 * fh];
        if (ooff < 0) continue;
        float v[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const float mx = fmaxf(fmaxf(w[0][j], w[1][j]), fmaxf(w[2][j], w[3][j]));
          v[j] = fmaf(mx, p.oscale, bias_v[j]);
          if (p.relu) v[j] = fmaxf(v[j], 0.f);
        }
        const int o = ooff + n_store;
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
      const int ooff = s_out_off[row0 + i * MT + (e & 3) + 8 * (e >> 2) + 4 * fh];
      if (ooff < 0) continue;
      int o = ooff + tap_off + n_store;
      if (second) o = (int)((unsigned)ooff / (unsigned)p.out_cstride) * p.out2_cstride + n_store - p.split_n;
      float v[NR];
      rd.row(i, e, v);
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        v[j] = fmaf(v[j], p.oscale, bias_v[j]);
        if (p.relu) v[j] = fmaxf(v[j], 0.f);
      }
      if (n_valid >= NR) {
        if (p.res != nullptr) {
          float rv[NR];
          Tr::template load_vec<NR>(p.res, o, rv);
#pragma unroll
          for (int j = 0; j < NR; ++j) v[j] = fmaxf(v[j] + rv[j], 0.f);
        }
        if (p.out_f32) store_f32_vec<NR>(reinterpret_cast<float*>(second ? p.out2 : p.out) + o, v);
        else Tr::template store_vec<NR>(p.out, o, v);
      } else {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          if (j < n_valid) {                 // (no run-time `break` in a loop that is to be unrolled: hipcc then refuses)
            float x = v[j];
            if (p.res != nullptr) x = fmaxf(x + Tr::load(p.res, o + j), 0.f);
            if (p.out_f32) reinterpret_cast<float*>(second ? p.out2 : p.out)[o + j] = x;
            else Tr::store(p.out, o + j, x);
          }
        }
      }
    }
  }
}

template <class Tr, int MR, int NR, int MT, int EPA>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, typename Tr::acc_t (&acc)[MR][NR], const int* s_out_off, int row0,
                                              int fh, int n_glob, int n_store, int tap_off, const int* s_out2_off = nullptr) {
  conv_epilogue_r<Tr, MR, NR, MT, EPA>(p, AccArray<Tr, MR, NR>{acc}, s_out_off, row0, fh, n_glob, n_store, tap_off, s_out2_off);
}

// Geometry / pointers of a launch -> kernel arguments (tiling fields are the caller's).
inline void fill_conv_args(const ConvLaunch& c, ConvArgs* out) {
  ConvArgs a = ConvArgs();
  const int chunk = kRowBytes / (int)dtype_size(c.dtype);
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
  a.oscale = c.oscale;
  a.Npad = c.Npad;
  a.splitk = 1; a.kt_split = a.KT; a.partial = nullptr;
  a.pool = c.pool;
  a.split_n = c.split_n; a.split_first = c.split_first;
  a.out2 = c.out2.base;
  a.out2_Hp = c.out2.Hp(); a.out2_Wp = c.out2.Wp(); a.out2_cstride = c.out2.cstride; a.out2_pad = c.out2.pad; a.out2_coff = c.out2.coff;
  a.m_fastest = 0;
  a.pos_major = 0; a.n_img = c.in.N; a.in_H = c.in.H; a.cpad = c.cpad; a.kh = c.kh;
  a.center_from_n = c.center_from;
  a.taps_inner = 0;
  *out = a;
}

}  // namespace detail
}  // namespace ron
