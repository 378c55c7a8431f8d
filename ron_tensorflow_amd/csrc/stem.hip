// conv1_1: 3x3 SAME conv, 3 -> 64 channels, + bias + ReLU, straight from the caller's fp32 NHWC image into the
// halo bf16/f16 activation tensor (nets/ron_vgg_320.py:454 / :530, first slim.conv2d of conv1).
//
// K = 27 is too thin for the LDS-DMA implicit-GEMM kernel (its rows are 128-byte chunks), so this layer is its own
// kernel: HBM-bound on the 128 B/pixel output (13 MB/image), the 4 MFMAs per 32 pixels are noise.
//   * one wave = 32 consecutive pixels of one image row; it stages the 3 x 34 x 3 fp32 input patch in its private LDS
//     slice (zero outside the image), then every lane gathers its 16 A values (pixel r = lane & 31, k = 8h+j and
//     16+8h+j, k = ty*9 + tx*3 + c, zero for k >= 27) with stride-3 LDS reads (conflict free) and packs them to bf16;
//   * B (weights, [64][32] after padding K) lives in 4 registers per lane for the whole kernel;
//   * MFMA column r of accumulator t is output channel 2r + t, so a lane's two accumulators are adjacent channels:
//     one dword store per pixel row, 32 lanes = the pixel's full 128-byte channel vector.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <stdlib.h>

#include <vector>

#include "pack.h"

namespace ron {
namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

// cvt2: two fp32 -> one dword of two storage-type values (round to nearest even; bf16: ONE v_cvt_pk_bf16_f32), low half first
struct StemBF16 {
  static __device__ __forceinline__ unsigned short cvt(float v) { return __builtin_bit_cast(unsigned short, __float2bfloat16(v)); }
  static __device__ __forceinline__ unsigned cvt2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2)); }
  static __device__ __forceinline__ float tof(unsigned v) { return __uint_as_float(v << 16); }
  static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ void mma16(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c, 0, 0, 0);
  }
};
struct StemF16 {
  static __device__ __forceinline__ unsigned short cvt(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }
  static __device__ __forceinline__ unsigned cvt2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, f16x2)); }
  static __device__ __forceinline__ float tof(unsigned v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)v); }
  static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ void mma16(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  }
};

constexpr int kPatchW = 34 * 3;          // floats per staged row: pixels x0-1 .. x0+32, 3 channels
constexpr int kPatch = 3 * kPatchW;      // 306 floats per wave

template <class Tr>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, int n_img, int H, int W,
                                                        const u32x4* __restrict__ wfrag, const float* __restrict__ bias2,
                                                        unsigned* __restrict__ out, int out_Hp, int out_Wp, int out_pad) {
  __shared__ float s_in[4][kPatch + 14];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  // weights: [t][s][lane] 16-byte fragments; bias pairs (channel 2r, 2r+1)
  u32x4 wb[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s) wb[t][s] = wfrag[(t * 2 + s) * 64 + lane];
  const float b0 = bias2[2 * r], b1 = bias2[2 * r + 1];
  // LDS read offsets of this lane's 16 A values (floats, relative to the wave's patch)
  int a_off[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * s + 8 * h + j;
      const int ty = k / 9, rem = k - ty * 9;
      a_off[s][j] = k < 27 ? ty * kPatchW + r * 3 + rem : -1;
    }
  float* patch = s_in[wave];
  const int tiles_per_row = W / 32;
  const long long n_tiles = (long long)n_img * H * tiles_per_row;
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < n_tiles; tile += (long long)gridDim.x * 4) {
    const int tx = (int)(tile % tiles_per_row);
    const long long row = tile / tiles_per_row;
    const int y = (int)(row % H);
    const long long img = row / H;
    const int x0 = tx * 32;
    // stage rows y-1..y+1, pixels x0-1..x0+32 (zero outside the image)
    for (int i = lane; i < kPatch; i += 64) {
      const int ty = i / kPatchW, rem = i - ty * kPatchW;
      const int px = rem / 3, c = rem - px * 3;
      const int yy = y + ty - 1, xx = x0 + px - 1;
      float v = 0.f;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = x[((img * H + yy) * (long long)W + xx) * 3 + c];
      patch[i] = v;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): this wave's LDS writes are done (wave-private slice)
    __builtin_amdgcn_wave_barrier();
    u32x4 fa[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      unsigned short e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = Tr::cvt(a_off[s][j] >= 0 ? patch[a_off[s][j]] : 0.f);
      fa[s] = u32x4{(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                    (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
    }
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
      Tr::mma(fa[0], wb[t][0], acc[t]);
      Tr::mma(fa[1], wb[t][1], acc[t]);
    }
    // pixel row p = (e & 3) + 8 * (e >> 2) + 4 * h ; lanes r = 0..31 cover channels 0..63 as dwords
    const long long obase = ((img * out_Hp + y + out_pad) * (long long)out_Wp + x0 + out_pad) * 32;   // in dwords (64 ch * 2 B)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int p = (e & 3) + 8 * (e >> 2) + 4 * h;
      const unsigned lo = Tr::cvt(fmaxf(acc[0][e] + b0, 0.f)), hi = Tr::cvt(fmaxf(acc[1][e] + b1, 0.f));
      out[obase + (long long)p * 32 + r] = lo | (hi << 16);
    }
    __builtin_amdgcn_wave_barrier();          // patch is rewritten by the next iteration
  }
}


// conv1_1 in split precision (RON_DTYPE_F16X3, conv_device.h): the same tiling; A and B are two f16 planes each (hi = rnd(v),
// lo = rnd(v - hi); the weights times 2^k so that their lo plane is a normal f16, undone by `oscale`), a product is three MFMAs
// hi*hi + lo*hi + hi*lo into the fp32 accumulator, and a pixel's 64 outputs are stored as two 128-byte chunks [32 x hi][32 x lo].
// Replaces im2col (173 us at batch 32) + a K = 32 GEMM through the row-gather kernel (276 us): HBM-bound on its 838 MB of output.
__global__ __launch_bounds__(256) void stem_conv_split_kernel(const float* __restrict__ x, int n_img, int H, int W,
                                                              const u32x4* __restrict__ wfrag, const float* __restrict__ bias2, float oscale,
                                                              unsigned* __restrict__ out, int out_Hp, int out_Wp, int out_pad) {
  __shared__ float s_in[4][kPatch + 14];
  __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);      // MODE.FP16_OVFL: overflowing f16 conversions saturate (conv_device.h)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  u32x4 wb[2][2][2];                    // [plane: hi, lo][t][s]
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) wb[pl][t][s] = wfrag[((pl * 2 + t) * 2 + s) * 64 + lane];
  const float b0 = bias2[2 * r], b1 = bias2[2 * r + 1];
  int a_off[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * s + 8 * h + j;
      const int ty = k / 9, rem = k - ty * 9;
      a_off[s][j] = k < 27 ? ty * kPatchW + r * 3 + rem : -1;
    }
  float* patch = s_in[wave];
  const int tiles_per_row = W / 32;
  const long long n_tiles = (long long)n_img * H * tiles_per_row;
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < n_tiles; tile += (long long)gridDim.x * 4) {
    const int tx = (int)(tile % tiles_per_row);
    const long long row = tile / tiles_per_row;
    const int y = (int)(row % H);
    const long long img = row / H;
    const int x0 = tx * 32;
    for (int i = lane; i < kPatch; i += 64) {
      const int ty = i / kPatchW, rem = i - ty * kPatchW;
      const int px = rem / 3, c = rem - px * 3;
      const int yy = y + ty - 1, xx = x0 + px - 1;
      float v = 0.f;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = x[((img * H + yy) * (long long)W + xx) * 3 + c];
      patch[i] = v;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    u32x4 fa[2][2];                     // [plane][s]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      unsigned short eh[8], el[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = a_off[s][j] >= 0 ? patch[a_off[s][j]] : 0.f;
        const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
        eh[j] = __builtin_bit_cast(unsigned short, hi);
        el[j] = __builtin_bit_cast(unsigned short, lo);
      }
      fa[0][s] = u32x4{(unsigned)eh[0] | ((unsigned)eh[1] << 16), (unsigned)eh[2] | ((unsigned)eh[3] << 16),
                       (unsigned)eh[4] | ((unsigned)eh[5] << 16), (unsigned)eh[6] | ((unsigned)eh[7] << 16)};
      fa[1][s] = u32x4{(unsigned)el[0] | ((unsigned)el[1] << 16), (unsigned)el[2] | ((unsigned)el[3] << 16),
                       (unsigned)el[4] | ((unsigned)el[5] << 16), (unsigned)el[6] | ((unsigned)el[7] << 16)};
    }
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        StemF16::mma(fa[0][s], wb[0][t][s], acc[t]);      // hi * hi
        StemF16::mma(fa[1][s], wb[0][t][s], acc[t]);      // lo * hi
        StemF16::mma(fa[0][s], wb[1][t][s], acc[t]);      // hi * lo
      }
    }
    // a pixel = 64 dwords: channel pair r (channels 2r, 2r + 1) -> chunk r / 16: hi dword at chunk * 32 + r % 16, lo dword 16 further
    const long long obase = ((img * out_Hp + y + out_pad) * (long long)out_Wp + x0 + out_pad) * 64;
    const int od = (r >> 4) * 32 + (r & 15);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int p = (e & 3) + 8 * (e >> 2) + 4 * h;
      const float v0 = fmaxf(fmaf(acc[0][e], oscale, b0), 0.f), v1 = fmaxf(fmaf(acc[1][e], oscale, b1), 0.f);
      const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
      const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
      out[obase + (long long)p * 64 + od] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
      out[obase + (long long)p * 64 + od + 16] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
    }
    __builtin_amdgcn_wave_barrier();
  }
}


// ---------------------------------------------------------------------------------------------------------------
// conv1_1 + conv1_2 + pool1 in one kernel (RON_CFG_FUSE_POOLS, bf16 / f16).
//
// As separate launches the stem writes the 64-channel conv1_1 map (13 MB per image) and conv1_2 stages it back nine
// times: 9 % of the step at batch 32.  Here a workgroup owns an 8 x 32 pixel tile of conv1_2's output:
//   A. the 12 x 36 x 3 fp32 image patch goes to LDS (zero outside the image);
//   B. conv1_1 (+bias, ReLU) of the 10 x 34 halo patch is computed with MFMAs (K = 27 -> 32) and stored as bf16 rows of
//      128 B (64 channels) in LDS, swizzled like the generic kernel's stages, zero outside the image (= conv1_2's padding);
//   C. conv1_2 runs its 9 taps straight from that patch: wave w owns tile row w (32 consecutive pixels = two 16-row fragments
//      of consecutive patch rows; chunk swizzle ((row >> 1) & 3) << 1: conflict-free ds_read_b128 for any tap shift), on
//      16x16x32 MFMAs (the chip holds a higher clock under them than under the 32x32x16 form this loop started with: same
//      cycles, conv_c64.hip), the weights of all 9 taps (72 KB) stay resident in LDS for the life of the (persistent)
//      workgroup -- no staging in the loop;
//   D. + bias, ReLU, 2x2 max-pool (horizontal pairs are adjacent accumulator registers, vertical pairs meet in LDS),
//      16-byte stores of the pooled 4 x 16 tile.
// HBM traffic: image in (1.2 MB / image), pool1 out (3.3 MB / image).
// Round 2 tried the overlap DESIGN.md 3.2 had queued: waves 4-7 as producers (image fetch + conv1_1 of tile t+1 into a second
// patch buffer) beside waves 0-3 as consumers (conv1_2 of tile t, two rows each, pooling in registers), three taps' weights in
// registers to make room for the second buffer, one barrier per tile.  Correct (same tests), but 335 us instead of 310 us at
// batch 32: conv1_1 on the fly costs ~1000 cycles per group of 16 patch pixels (gather + convert of 27 taps, bias / ReLU /
// convert / inside-test per output), and four producer waves - one per SIMD, nothing to hide their latencies behind - take longer
// over a tile's 22 groups than the consumers' 4600 MFMA cycles.  The sequential form below stays.
constexpr int kS2TH = 8, kS2TW = 32;
constexpr int kS2PW = kS2TW + 2, kS2PH = kS2TH + 2, kS2Rows = kS2PW * kS2PH;       // 34 x 10 = 340 patch rows
constexpr int kS2IW = kS2TW + 4, kS2IH = kS2TH + 4;                                 // 36 x 12 image patch
constexpr int kS2W2Bytes = 9 * 64 * 128;                                            // 72 KB
constexpr int kS2PatchBytes = kS2Rows * 128;
constexpr int kS2ImgFloats = kS2IH * kS2IW * 3;
constexpr int kS2PoolBytes = 8 * 16 * 128;
constexpr int kS2DumpBytes = 512;                                                   // where the stores of lanes past the patch go
constexpr int kS2Lds = kS2W2Bytes + kS2PatchBytes + kS2ImgFloats * 4 + kS2PoolBytes + kS2DumpBytes;

template <class Tr>
__global__ __launch_bounds__(512) void stem2_kernel(const float* __restrict__ x, int n_img, int H, int W,
                                                    const u32x4* __restrict__ w1frag, const float* __restrict__ bias1,
                                                    const u32x4* __restrict__ w2img, const float* __restrict__ bias2,
                                                    unsigned short* __restrict__ out, int out_Hp, int out_Wp, int out_pad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_w2 = smem;
  char* s_p = s_w2 + kS2W2Bytes;
  float* s_img = reinterpret_cast<float*>(s_p + kS2PatchBytes + kS2DumpBytes);   // the dump area sits behind the patch
  char* s_pool = reinterpret_cast<char*>(s_img + kS2ImgFloats);
  constexpr unsigned kDumpOff = (unsigned)kS2PatchBytes;             // relative to s_p
  static_assert(kDumpOff + kS2DumpBytes <= 0xFFFFu, "store offsets are 16 bits");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (lane -> c16 = lane & 15, kg = lane >> 4 below: the 16x16x32 fragment and accumulator layout)

  // conv1_2 weights: the host packed the exact LDS image (tap, permuted output row, swizzled 16-byte chunks)
  for (int i = tid; i < kS2W2Bytes / 16; i += 512) reinterpret_cast<u32x4*>(s_w2)[i] = w2img[i];
  // conv1_1 runs on 16x16x32 MFMAs (K = 27 -> 32 is ONE instruction): groups of 16 patch pixels, so the 340-pixel patch splits
  // 3 / 3 / ... over the 8 waves instead of 2 / 1 groups of 32.  Weights in registers: n-tile t, lane (c = l & 15, kg = l >> 4)
  // holds W[k = 8 kg + j][channel 4c + t]: a lane's four accumulators are adjacent channels (one 8-byte LDS store per pixel).
  const int c16 = lane & 15, kg = lane >> 4;
  u32x4 wb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) wb[t] = w1frag[t * 64 + lane];
  float b1[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) b1[t] = bias1[4 * c16 + t];
  float b2[4];                                   // conv1_2 bias of this lane's four adjacent output channels 4 c16 .. 4 c16 + 3
#pragma unroll
  for (int j = 0; j < 4; ++j) b2[j] = bias2[4 * (lane & 15) + j];
  int a_off[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * kg + j;
    const int ty = k / 9, rem = k - ty * 9;
    a_off[j] = k < 27 ? ty * (kS2IW * 3) + rem : 0;                // rem = tx*3 + c; k >= 27: any address, the value is dropped
  }
  const bool k_pad = kg == 3;                                       // this lane's elements j >= 3 are K padding (k = 27 .. 31)
  // phase B bookkeeping is the same for every tile: this wave's (up to three) groups of 16 patch pixels, the gather base of the
  // lane's pixel, and per accumulator register the LDS store offset + patch coordinates of the pixel it holds
  // (bits 0-15 offset, 16-19 patch row, 20-25 patch column).  No branches in phase B: the groups past the patch (22 groups
  // of 16 over 8 waves x 3) and the pixels past its end compute like the others and store into a dump area, so the three
  // groups of a wave are straight-line code the compiler interleaves.
  static_assert((kS2Rows + 15) / 16 <= 3 * 8, "8 waves x 3 groups of 16 cover the patch");
  int gb[3];
  unsigned st[3][4];
#pragma unroll
  for (int gi = 0; gi < 3; ++gi) {
    const int g = wave + 8 * gi;
    const int q = min(g * 16 + c16, kS2Rows - 1);
    gb[gi] = ((q / kS2PW) * kS2IW + q % kS2PW) * 3;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int qq = g * 16 + 4 * kg + e;
      const int qy = qq / kS2PW, qx = qq - qy * kS2PW;
      const unsigned off = (unsigned)(qq * 128 + ((((c16 >> 1) ^ (((qq >> 1) & 3) << 1))) << 4) + (c16 & 1) * 8);
      st[gi][e] = qq < kS2Rows ? (off | ((unsigned)qy << 16) | ((unsigned)qx << 20)) : (kDumpOff + (unsigned)lane * 8u);
    }
  }
  const int tiles_x = W / kS2TW, tiles_y = H / kS2TH;
  const int n_tiles = n_img * tiles_y * tiles_x;
  // ---- A: image patch (rows y0-2 .. y0+9, columns x0-2 .. x0+33) of a tile: 1296 floats, <= 3 per thread.  The patch of
  // tile t+1 is fetched into registers while tile t computes and dropped into LDS once phase B of tile t has read its own.
  constexpr int kImgPer = (kS2ImgFloats + 511) / 512;
  int i_iy[kImgPer], i_ix[kImgPer];
#pragma unroll
  for (int k = 0; k < kImgPer; ++k) {
    const int i = min(tid + k * 512, kS2ImgFloats - 1);
    i_iy[k] = i / (kS2IW * 3);
    i_ix[k] = i - i_iy[k] * (kS2IW * 3);                             // ix * 3 + c
  }
  float pre[kImgPer];
#define RON_S2_FETCH(tile_)                                                                                   \
  do {                                                                                                        \
    const int t_ = min((tile_), n_tiles - 1);      /* past the last tile: fetched again, never used */        \
    const int fx = t_ % tiles_x, fy = (t_ / tiles_x) % tiles_y, fimg = t_ / (tiles_x * tiles_y);              \
    _Pragma("unroll") for (int k = 0; k < kImgPer; ++k) {                                                     \
      const int yy = fy * kS2TH - 2 + i_iy[k], xc = (fx * kS2TW - 2) * 3 + i_ix[k];                           \
      const int yc = min(max(yy, 0), H - 1), xcc = min(max(xc, 0), W * 3 - 1);   /* unconditional load */      \
      const float v_ = x[((long long)fimg * H + yc) * W * 3 + xcc];                                           \
      /* a bit mask, not a select: hipcc turns the select back into a branch around the load */               \
      pre[k] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v_) & (unsigned)-(int)(yy == yc && xc == xcc)); \
    }                                                                                                         \
  } while (0)
#define RON_S2_STORE()                                                                                        \
  do {                                                                                                        \
    _Pragma("unroll") for (int k = 0; k < kImgPer; ++k)                                                       \
      if (tid + k * 512 < kS2ImgFloats) s_img[tid + k * 512] = pre[k];                                        \
  } while (0)
  RON_S2_FETCH(blockIdx.x);
  RON_S2_STORE();
  __syncthreads();
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int tx = tile % tiles_x;
    const int ty_ = (tile / tiles_x) % tiles_y;
    const int img = tile / (tiles_x * tiles_y);
    const int y0 = ty_ * kS2TH, x0 = tx * kS2TW;
    RON_S2_FETCH(tile + gridDim.x);                   // in flight during phase B
    // ---- B: conv1_1 of the 340 patch pixels, 32 at a time
#pragma unroll
    for (int gi = 0; gi < 3; ++gi) {
      const float* base = s_img + gb[gi];
      float gv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = base[a_off[j]];
        gv[j] = j >= 3 && k_pad ? 0.f : g;
      }
      const u32x4 fa = u32x4{Tr::cvt2(gv[0], gv[1]), Tr::cvt2(gv[2], gv[3]), Tr::cvt2(gv[4], gv[5]), Tr::cvt2(gv[6], gv[7])};
      f32x4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        Tr::mma16(fa, wb[t], acc[t]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned m = st[gi][e];
        const int iy = y0 - 1 + (int)((m >> 16) & 15u), ix = x0 - 1 + (int)((m >> 20) & 63u);
        const unsigned inside = (unsigned)-(int)((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W);   // mask (no branch)
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = fmaxf(acc[t][e] + b1[t], 0.f);
        *reinterpret_cast<u32x2*>(s_p + (m & 0xFFFFu)) = u32x2{Tr::cvt2(v[0], v[1]) & inside, Tr::cvt2(v[2], v[3]) & inside};
      }
    }
    __syncthreads();
    RON_S2_STORE();                                   // s_img is free: the next tile's patch (read after two more barriers)
    // ---- C: conv1_2, wave = tile row, 9 taps x 4 k-steps x 2 column tiles
    f32x4 acc2[2][4];                                  // [16-pixel block of the row][16-channel block]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc2[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int key_b = ((c16 >> 1) & 3) << 1;           // weight-image rows j * 16 + c16: the key of c16
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const char* pb = s_w2 + tap * 8192 + c16 * 128;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 fa[2], fb[4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int prow = (wave + tap / 3) * kS2PW + 16 * a + c16 + tap % 3;
          fa[a] = *reinterpret_cast<const u32x4*>(s_p + prow * 128 + (((4 * ks + kg) ^ (((prow >> 1) & 3) << 1)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const u32x4*>(pb + j * 16 * 128 + (((4 * ks + kg) ^ key_b) << 4));
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int j = 0; j < 4; ++j) Tr::mma16(fa[a], fb[j], acc2[a][j]);
      }
    }
    // ---- D: bias, ReLU, horizontal pool in registers -> LDS; vertical pool + store.  Accumulator register e of block a holds
    // pixel 16 a + 4 kg + e of the row, channels 4 c16 + j: pooled column 8 a + 2 kg + hp from registers 2 hp, 2 hp + 1
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {
        float m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = fmaxf(fmaxf(acc2[a][j][2 * hp], acc2[a][j][2 * hp + 1]) + b2[j], 0.f);
        const int mc = 8 * a + 2 * kg + hp;                             // pooled column 0..15
        *reinterpret_cast<u32x2*>(s_pool + (wave * 16 + mc) * 128 + c16 * 8) = u32x2{Tr::cvt2(m[0], m[1]), Tr::cvt2(m[2], m[3])};
      }
    __syncthreads();
    {
      const int yp = tid >> 7, m = (tid >> 3) & 15, c8 = tid & 7;     // 4 pooled rows x 16 columns x 8 chunks of 8 channels
      const u32x4 a = *reinterpret_cast<const u32x4*>(s_pool + ((2 * yp) * 16 + m) * 128 + c8 * 16);
      const u32x4 b = *reinterpret_cast<const u32x4*>(s_pool + ((2 * yp + 1) * 16 + m) * 128 + c8 * 16);
      u32x4 o;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        o[d] = Tr::cvt2(fmaxf(Tr::tof(a[d] & 0xFFFFu), Tr::tof(b[d] & 0xFFFFu)), fmaxf(Tr::tof(a[d] >> 16), Tr::tof(b[d] >> 16)));
      }
      const long long opix = ((long long)img * out_Hp + (y0 >> 1) + yp + out_pad) * out_Wp + (x0 >> 1) + m + out_pad;
      *reinterpret_cast<u32x4*>(out + opix * 64 + c8 * 8) = o;
    }
    // the next tile's phase B writes s_p (all reads of this tile's patch are behind the barrier above); its phase D
    // writes s_pool two barriers from here
  }
#undef RON_S2_FETCH
#undef RON_S2_STORE
}
}  // namespace

// Weight fragments for stem_conv_kernel from the HWIO [3,3,3,64] filter: fragment (t, s), lane (r, h), element j holds
// W[k = 16s + 8h + j][channel 2r + t] (zero for k >= 27), in the ctx dtype (bf16 / f16).
void stem_pack_weights(const float* hwio, int dtype, std::vector<uint16_t>* frags) {
  frags->assign(4 * 64 * 8, 0);
  for (int t = 0; t < 2; ++t)
    for (int s = 0; s < 2; ++s)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int r = lane & 31, h = lane >> 5;
          const int k = 16 * s + 8 * h + j, ch = 2 * r + t;
          const float v = k < 27 ? hwio[(size_t)k * 64 + ch] : 0.f;
          (*frags)[((size_t)(t * 2 + s) * 64 + lane) * 8 + j] = dtype == RON_DTYPE_BF16 ? f32_to_bf16_rne(v) : f32_to_f16_rne(v);
        }
}

// split precision: [plane hi / lo][t][s][lane] fragments of w * 2^k; returns 2^-k for the epilogue
float stem_pack_weights_split(const float* hwio, std::vector<uint16_t>* frags) {
  const int k = split_weight_exponent(std::vector<float>(hwio, hwio + 27 * 64));
  frags->assign(2 * 4 * 64 * 8, 0);
  for (int t = 0; t < 2; ++t)
    for (int s = 0; s < 2; ++s)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int r = lane & 31, h = lane >> 5;
          const int kk = 16 * s + 8 * h + j, ch = 2 * r + t;
          const float v = kk < 27 ? ldexpf(hwio[(size_t)kk * 64 + ch], k) : 0.f;
          const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
          (*frags)[((size_t)((0 * 2 + t) * 2 + s) * 64 + lane) * 8 + j] = f32_to_f16_rne((float)hi);
          (*frags)[((size_t)((1 * 2 + t) * 2 + s) * 64 + lane) * 8 + j] = f32_to_f16_rne((float)lo);
        }
  return ldexpf(1.f, -k);
}

int launch_stem_conv(const float* x, int n, int h, int w, int dtype, const void* d_wfrag, const float* d_bias,
                     const TensorView& out, hipStream_t s, float oscale) {
  RON_REQUIRE(dtype == RON_DTYPE_BF16 || dtype == RON_DTYPE_F16 || dtype == RON_DTYPE_F16X3, "stem kernel: bf16 / f16 / f16x3 only");
  RON_REQUIRE(w % 32 == 0 && out.C == 64 && out.cstride == 64 && out.coff == 0 && out.H == h && out.W == w, "stem kernel: bad shape");
  const long long tiles = (long long)n * h * (w / 32);
  const int grid = (int)std::min<long long>((tiles + 3) / 4, 256 * 8);
  if (dtype == RON_DTYPE_F16X3)
    RON_LAUNCH(stem_conv_split_kernel, dim3(grid), dim3(256), 0, s, x, n, h, w, (const u32x4*)d_wfrag, d_bias, oscale,
                       (unsigned*)out.base, out.Hp(), out.Wp(), out.pad);
  else if (dtype == RON_DTYPE_BF16)
    RON_LAUNCH(stem_conv_kernel<StemBF16>, dim3(grid), dim3(256), 0, s, x, n, h, w, (const u32x4*)d_wfrag, d_bias,
                       (unsigned*)out.base, out.Hp(), out.Wp(), out.pad);
  else
    RON_LAUNCH(stem_conv_kernel<StemF16>, dim3(grid), dim3(256), 0, s, x, n, h, w, (const u32x4*)d_wfrag, d_bias,
                       (unsigned*)out.base, out.Hp(), out.Wp(), out.pad);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

}  // namespace ron

namespace ron {

// conv1_1 weight fragments for stem2_kernel (16x16x32 MFMA) from the HWIO [3,3,3,64] filter: n-tile t, lane (c, kg), element j
// holds W[k = 8 kg + j][channel 4c + t] (zero for k >= 27).
void stem2_pack_w1(const float* hwio, int dtype, std::vector<uint16_t>* frags) {
  frags->assign(4 * 64 * 8, 0);
  for (int t = 0; t < 4; ++t)
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < 8; ++j) {
        const int c = lane & 15, kg = lane >> 4, k = 8 * kg + j, ch = 4 * c + t;
        const float v = k < 27 ? hwio[(size_t)k * 64 + ch] : 0.f;
        (*frags)[((size_t)t * 64 + lane) * 8 + j] = dtype == RON_DTYPE_BF16 ? f32_to_bf16_rne(v) : f32_to_f16_rne(v);
      }
}

// LDS image of the conv1_2 weights for stem2_kernel from the HWIO [3,3,64,64] filter: tap-major, row (j*16 + c) of a tap
// holds output channel 4c + j, 64 input channels = 8 chunks of 16 B, chunk k in slot k ^ (((row >> 1) & 3) << 1).
void stem2_pack_weights(const float* hwio, int dtype, std::vector<uint16_t>* img) {
  img->assign(9 * 64 * 64, 0);
  for (int tap = 0; tap < 9; ++tap)
    for (int row = 0; row < 64; ++row) {
      const int j = row / 16, c = row % 16, ch = 4 * c + j;
      for (int cin = 0; cin < 64; ++cin) {
        const float v = hwio[((size_t)tap * 64 + cin) * 64 + ch];
        const int chunk = cin / 8, slot = chunk ^ (((row >> 1) & 3) << 1);
        (*img)[((size_t)tap * 64 + row) * 64 + slot * 8 + cin % 8] = dtype == RON_DTYPE_BF16 ? f32_to_bf16_rne(v) : f32_to_f16_rne(v);
      }
    }
}

int launch_stem2(const float* x, int n, int h, int w, int dtype, const void* d_w1frag, const float* d_bias1,
                 const void* d_w2img, const float* d_bias2, const TensorView& out, hipStream_t s) {
  RON_REQUIRE(dtype == RON_DTYPE_BF16 || dtype == RON_DTYPE_F16, "stem2 kernel: bf16 / f16 only");
  RON_REQUIRE(w % kS2TW == 0 && h % kS2TH == 0 && out.C == 64 && out.cstride == 64 && out.coff == 0 && out.H == h / 2 && out.W == w / 2,
              "stem2 kernel: bad shape");
  const int tiles = n * (h / kS2TH) * (w / kS2TW);
  const int grid = std::min(tiles, 256);
  static PerDeviceOnce attr_set[2];                // the attribute is per device (common.h)
  const int which = dtype == RON_DTYPE_BF16 ? 0 : 1;
  RON_HIP_CHECK(attr_set[which].max_dynamic_lds(which == 0 ? reinterpret_cast<const void*>(&stem2_kernel<StemBF16>)
                                                           : reinterpret_cast<const void*>(&stem2_kernel<StemF16>), kS2Lds));
  if (which == 0)
    RON_LAUNCH(stem2_kernel<StemBF16>, dim3(grid), dim3(512), kS2Lds, s, x, n, h, w, (const u32x4*)d_w1frag, d_bias1,
                       (const u32x4*)d_w2img, d_bias2, (unsigned short*)out.base, out.Hp(), out.Wp(), out.pad);
  else
    RON_LAUNCH(stem2_kernel<StemF16>, dim3(grid), dim3(512), kS2Lds, s, x, n, h, w, (const u32x4*)d_w1frag, d_bias1,
                       (const u32x4*)d_w2img, d_bias2, (unsigned short*)out.base, out.Hp(), out.Wp(), out.pad);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

}  // namespace ron
