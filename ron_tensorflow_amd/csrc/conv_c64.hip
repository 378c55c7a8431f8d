// 3x3 / stride 1 / pad 1 convolution of a 64-channel map with the weights RESIDENT in LDS (gfx950, bf16 / f16).
//
// Cin = 64 means K = 576: nine K steps.  The row-gather kernel (conv_mfma.hip) re-stages 16 KB of weights per step and tile for
// that little work and spends a third of a tile's time filling and draining its pipeline (conv2_1 at batch 32: 175 us,
// 0.69 PFLOP/s, the L2 -> LDS path of a CU at the rate the wide layers reach with twice the arithmetic per byte).  Here the
// loop of stem2_kernel's conv1_2 phase (stem.hip) is its own kernel:
//   * a persistent workgroup owns one 64-channel slice of the outputs ("half": Cout / 64 of them) and keeps that slice's nine
//     taps (72 KB, the exact LDS image packed by pack_conv_c64_weights) in LDS for its whole life;
//   * per 8 x 32 pixel tile it stages the 10 x 34 x 64-channel input patch once (LDS-DMA, 43 KB, double buffered: the next
//     tile's patch is in flight during this tile's taps) and runs the nine taps against it - a tap is an LDS address shift;
//   * wave w owns tile row w: a fragment's rows are consecutive patch rows; 16x16x32 MFMAs (on random data the chip holds a
//     higher clock under them than under 32x32x16: conv2_1 115 -> 110 us, a 64-output layer 270 -> 250 us), 16-row fragments,
//     chunk swizzle ((row >> 1) & 3) << 1: conflict-free ds_read_b128 at every tap shift (the LDS-DMA applies it on the source
//     address).  The first form (32x32x16, 32-row fragments, (row >> 1) & 7) and the ablation
//     switches and cycle stamps the numbers below were taken with are in tools/experiments/ (README.md there);
//   * one barrier per tile;
//   * epilogue: + bias, ReLU, a lane holds four adjacent channels of eight pixels: 8-byte stores, 16 lanes = one 128-B line.  The
//     stores of tile t go out between the MFMAs of tile t+1, one per K step: the layer writes 210 MB at batch 32, and with every
//     workgroup storing its tile at the same moment between two tap loops nothing computed while HBM took the burst (stores
//     alone 36 us, patch staging alone 31 us, both 75 us of a 113 us launch);
//   * a wave's vmcnt counts loads and stores, which do not retire in order with respect to each other: the one wait per tile
//     (after the taps) is for everything - the next patch and the previous tile's stores, both issued a tap loop earlier.
// The workgroups of the halves of one tile sequence share an XCD (they stage the same patches: the second read hits L2).
// Activations: zero-halo NHWC (conv_mfma.h TensorView), so a patch never needs a bounds check.
#include "conv_device.h"
#include <stdlib.h>

#include "pack.h"

namespace ron {
namespace detail {

constexpr int kC6TH = 8, kC6TW = 32;
constexpr int kC6PW = kC6TW + 2, kC6PH = kC6TH + 2, kC6Rows = kC6PW * kC6PH;     // 34 x 10 = 340 patch rows of 128 B
constexpr int kC6Pieces = (kC6Rows * 128 + 1023) / 1024;                         // 43 wave-instructions of 1 KB
constexpr int kC6PatchBytes = kC6Pieces * 1024;                                  // 44032: 340 rows + 4 rows of overshoot
constexpr int kC6WBytes = 9 * 64 * 128;                                          // 72 KB
constexpr int kC6Lds = kC6WBytes + 2 * kC6PatchBytes;                            // 161 792 of 163 840

struct C64Args {
  const void* in;
  unsigned in_bytes;
  int in_Hp, in_Wp, in_pad;
  const u32x4* wimg;          // [halves][9 taps][64 rows][128 B]
  const float* bias;
  unsigned short* out;
  unsigned out_bytes;
  int out_Hp, out_Wp, out_pad, out_cstride, out_coff;
  int n_img, H, W, halves, relu;
  int n_slots;                // tile sequences (a multiple of 8): grid = n_slots * halves
};

typedef __attribute__((ext_vector_type(2))) float f32x2_c64;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_c64;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_c64;
// two fp32 -> one dword of two storage-type values (round to nearest even), low half = first argument
struct C64BF16 : TraitsBF16 {
  static __device__ __forceinline__ unsigned cvt2(float a, float b) {      // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_c64{a, b}, bf16x2_c64));
  }
};
struct C64F16 : TraitsF16 {
  static __device__ __forceinline__ unsigned cvt2(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_c64{a, b}, f16x2_c64));
  }
};


// ---- the same kernel on 16x16x32 MFMAs --------------------------------------------------------------------------------
// On random data the chip holds a higher clock under v_mfma_f32_16x16x32 than under 32x32x16 (CDNA4 guide, DVFS: 1.12-1.14 x the
// FLOP/s with operands from LDS).  A fragment is 16 rows x 64 B: the patch and weight images use the key ((row >> 1) & 3) << 1
// (conflict-free ds_read_b128 for 16 consecutive rows at every tap shift; the (row >> 1) & 7 key of the 32-row form is 2-way for
// odd shifts); weight-image row (j * 16 + c) holds output channel 4c + j, so a lane's four column tiles are four adjacent channels:
// one 8-byte store per pixel.  18 K steps (9 taps x 2) of 2 x 4 MFMAs per tile row, fragments read two steps ahead.
__device__ __forceinline__ int c64_key16(int row) { return ((row >> 1) & 3) << 1; }

struct C64BF16S : TraitsBF16S {
  static __device__ __forceinline__ unsigned cvt2(float a, float b) { return C64BF16::cvt2(a, b); }
};
struct C64F16S : TraitsF16S {
  static __device__ __forceinline__ unsigned cvt2(float a, float b) { return C64F16::cvt2(a, b); }
};
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_c64;

template <class Tr>
__global__ __launch_bounds__(512) void conv3x3_c64_kernel16(C64Args p) {
  constexpr int NW = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_w = smem;
  char* s_p = smem + kC6WBytes;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, kg = lane >> 4;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int half = idx % p.halves;
  const int slot = (idx / p.halves) * 8 + xcd;
  const int tiles_x = p.W / kC6TW, tiles_y = p.H / kC6TH;
  const int n_tiles = p.n_img * tiles_y * tiles_x;

  int voff[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int q = min((wave + NW * k) * 8 + (lane >> 3), kC6Rows - 1);
    const int chunk = (lane & 7) ^ c64_key16(q);
    voff[k] = ((q / kC6PW) * p.in_Wp + q % kC6PW) * 128 + chunk * 16;
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
#define RON_C64_STAGE(tile_, buf_)                                                                                    \
  do {                                                                                                                \
    const int t_ = (tile_);                                                                                           \
    const int fx = t_ % tiles_x, fy = (t_ / tiles_x) % tiles_y, fimg = t_ / (tiles_x * tiles_y);                      \
    const int soff = ((fimg * p.in_Hp + p.in_pad + fy * kC6TH - 1) * p.in_Wp + p.in_pad + fx * kC6TW - 1) * 128;      \
    char* dst_ = s_p + (buf_) * kC6PatchBytes + wave * 1024;                                                          \
    _Pragma("unroll") for (int k = 0; k < 6; ++k)                                                                     \
      if (wave + NW * k < kC6Pieces)                                                                                  \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst_ + k * NW * 1024), 16, voff[k], soff, 0, 0);     \
  } while (0)

  if (slot < n_tiles) RON_C64_STAGE(slot, 0);
  {
    const u32x4* wsrc = p.wimg + (size_t)half * (kC6WBytes / 16);
    for (int i = tid; i < kC6WBytes / 16; i += 512) reinterpret_cast<u32x4*>(s_w)[i] = wsrc[i];
  }
  const int n0 = half * 64;
  float bias_v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bias_v[j] = p.bias[n0 + 4 * c16 + j];
  // stores: per-lane offset = pixel 4 kg of a 16-pixel block, channels n0 + 4 c16 .. + 3; tile / row / block / register part scalar
  const int st_voff = (4 * kg * p.out_cstride + p.out_coff + n0 + 4 * c16) * 2;
  const int st_row = __builtin_amdgcn_readfirstlane(wave) * p.out_Wp * p.out_cstride * 2;
  const float lo = p.relu ? 0.f : -__builtin_huge_valf();
  f32x4 prev[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) prev[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int prev_off = 0;
  unsigned prev_records = 0;
  // store q (0 .. 7): pixel block a = q >> 2, accumulator register e = q & 3 -> pixel 16 a + 4 kg + e of the row
#define RON_C64_STORE16(q_)                                                                                           \
  do {                                                                                                                \
    const int a_ = (q_) >> 2, e_ = (q_) & 3;                                                                          \
    const float v0_ = fmaxf(prev[a_][0][e_] + bias_v[0], lo), v1_ = fmaxf(prev[a_][1][e_] + bias_v[1], lo);           \
    const float v2_ = fmaxf(prev[a_][2][e_] + bias_v[2], lo), v3_ = fmaxf(prev[a_][3][e_] + bias_v[3], lo);           \
    __builtin_amdgcn_raw_buffer_store_b64(u32x2_c64{Tr::cvt2(v0_, v1_), Tr::cvt2(v2_, v3_)}, rs_st, st_voff,          \
                                          prev_off + (16 * a_ + e_) * p.out_cstride * 2, 0);                          \
  } while (0)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // fragment read offsets: A row = patch row + c16, chunk (4 ks + kg) ^ key(row); B row = tap * 64 + j * 16 + c16 (key of j * 16 + c16)
  const int b_off0 = c16 * 128 + ((kg ^ c64_key16(c16)) << 4);              // ks = 0; rows j * 16 + c16: key(j * 16 + c16) = key(c16)
  const int b_off1 = c16 * 128 + (((4 + kg) ^ c64_key16(c16)) << 4);        // ks = 1
  int buf = 0;
  for (int tile = slot; tile < n_tiles; tile += p.n_slots, buf ^= 1) {
    __builtin_amdgcn_s_barrier();
    if (tile + p.n_slots < n_tiles) RON_C64_STAGE(tile + p.n_slots, buf ^ 1);
    const __amdgpu_buffer_rsrc_t rs_st = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, prev_records, 0x00020000);
    const char* sp = s_p + buf * kC6PatchBytes;
    f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 fa[3][2], fb[3][4];
    auto frag_read = [&](int i) {                 // K step i = tap * 2 + ks
      const int tap = i >> 1, ks = i & 1, r3 = i % 3;
      const char* pb = s_w + tap * 8192 + (ks ? b_off1 : b_off0);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int prow = (wave + tap / 3) * kC6PW + 16 * a + c16 + tap % 3;
        fa[r3][a] = *reinterpret_cast<const u32x4*>(sp + prow * 128 + (((4 * ks + kg) ^ c64_key16(prow)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[r3][j] = *reinterpret_cast<const u32x4*>(pb + j * 16 * 128);
    };
    frag_read(0);
    frag_read(1);
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      if (i + 2 < 18) frag_read(i + 2);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) Tr::mma(fa[i % 3][a], fb[i % 3][j], acc[a][j]);
      if (i >= 1 && i < 9) RON_C64_STORE16(i >= 1 && i < 9 ? i - 1 : 0);
    }
    // pin the order: per K step eight MFMAs, the six reads of step i + 2 between them, one store behind steps 1 .. 8
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
    for (int i = 0; i < 18; ++i) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i + 2 < 18 && q < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      if (i >= 1 && i < 9) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, img = tile / (tiles_x * tiles_y);
    prev_off = ((img * p.out_Hp + p.out_pad + ty * kC6TH) * p.out_Wp + p.out_pad + tx * kC6TW) * p.out_cstride * 2 + st_row;
    prev_records = p.out_bytes;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) prev[a][j] = acc[a][j];
  }
  {
    const __amdgpu_buffer_rsrc_t rs_st = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, prev_records, 0x00020000);
#pragma unroll
    for (int q = 0; q < 8; ++q) RON_C64_STORE16(q);
  }
#undef RON_C64_STORE16
#undef RON_C64_STAGE
}

template <class Tr>
int launch_c64_16_t(const C64Args& a, hipStream_t s) {
  static PerDeviceOnce once;
  RON_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void*>(&conv3x3_c64_kernel16<Tr>), kC6Lds));
  RON_LAUNCH((conv3x3_c64_kernel16<Tr>), dim3(a.n_slots * a.halves), dim3(512), kC6Lds, s, a);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}


}  // namespace detail
using namespace detail;

// LDS images for conv3x3_c64_kernel from fp32 rows [npad][K = 9 * 64] (row n, k = tap * 64 + cin): per 64-channel half, tap-major,
// row (j * 16 + c) of a tap holds output channel 64 * half + 4c + j, its 64 input channels as 8 chunks of 16 B, chunk q in slot
// q ^ (((row >> 1) & 3) << 1) (the 16x16x32 MFMA's fragments, conflict-free at every tap shift).

std::vector<uint8_t> pack_conv_c64_weights(const std::vector<float>& rows, int npad, int dtype) {
  const int halves = npad / 64;
  std::vector<uint16_t> img((size_t)halves * 9 * 64 * 64, 0);
  for (int hf = 0; hf < halves; ++hf)
    for (int tap = 0; tap < 9; ++tap)
      for (int row = 0; row < 64; ++row) {
        const int ch = 64 * hf + 4 * (row % 16) + row / 16;
        for (int cin = 0; cin < 64; ++cin) {
          const float v = rows[(size_t)ch * 576 + tap * 64 + cin];
          const int chunk = cin / 8, slot = chunk ^ (((row >> 1) & 3) << 1);
          img[(((size_t)hf * 9 + tap) * 64 + row) * 64 + slot * 8 + cin % 8] = dtype == RON_DTYPE_BF16 ? f32_to_bf16_rne(v) : f32_to_f16_rne(v);
        }
      }
  std::vector<uint8_t> out(img.size() * 2);
  memcpy(out.data(), img.data(), out.size());
  return out;
}

bool conv_c64_applicable(const ConvLaunch& c) {
  return c.wgt_c64 != nullptr && c.out2.base == nullptr && c.center_from == 0 && (c.dtype == RON_DTYPE_BF16 || c.dtype == RON_DTYPE_F16) && c.kh == 3 && c.kw == 3 && c.stride == 1 &&
         c.dil == 1 && c.cpad == 1 && c.up == 0 && !c.pool && c.res == nullptr && !c.out_f32 && c.in.C == 64 && c.in.cstride == 64 &&
         c.in.coff == 0 && c.in.pad >= 1 && c.Npad == c.Cout && c.Cout % 64 == 0 && (c.Cout / 64 == 1 || c.Cout / 64 == 2 || c.Cout / 64 == 4) &&
         c.Ho % kC6TH == 0 && c.Wo % kC6TW == 0 && c.Ho == c.in.H && c.Wo == c.in.W && c.out.cstride % 4 == 0 && c.out.coff % 4 == 0 &&
         c.in.bytes > 0 && c.in.bytes < (int64_t)1 << 31 && c.out.bytes > 0 && c.out.bytes < (int64_t)1 << 31;   // 32-bit buffer offsets
}

int launch_conv_c64(const ConvLaunch& c, hipStream_t stream) {
  RON_REQUIRE(conv_c64_applicable(c), "resident-weight kernel: 3x3 / stride 1 / pad 1 on a 64-channel bf16 / f16 map whose size the 8 x 32 tile "
              "divides, 64 / 128 / 256 outputs, tensors < 2 GiB, packed weight image present");
  C64Args a;
  a.in = c.in.base; a.in_bytes = (unsigned)c.in.bytes;
  a.in_Hp = c.in.Hp(); a.in_Wp = c.in.Wp(); a.in_pad = c.in.pad;
  a.wimg = (const u32x4*)c.wgt_c64; a.bias = c.bias;
  a.out = (unsigned short*)c.out.base; a.out_bytes = (unsigned)c.out.bytes;
  a.out_Hp = c.out.Hp(); a.out_Wp = c.out.Wp(); a.out_pad = c.out.pad; a.out_cstride = c.out.cstride; a.out_coff = c.out.coff;
  a.n_img = c.in.N; a.H = c.Ho; a.W = c.Wo; a.halves = c.Cout / 64; a.relu = c.relu;
  const int n_tiles = c.in.N * (c.Ho / kC6TH) * (c.Wo / kC6TW);
  a.n_slots = std::min(256 / a.halves, (n_tiles + 7) / 8 * 8);
  return c.dtype == RON_DTYPE_BF16 ? launch_c64_16_t<C64BF16S>(a, stream) : launch_c64_16_t<C64F16S>(a, stream);
}

}  // namespace ron
