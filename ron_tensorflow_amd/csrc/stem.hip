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

#include <vector>

#include "pack.h"

namespace ron {
namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct StemBF16 {
  static __device__ __forceinline__ unsigned short cvt(float v) { return __builtin_bit_cast(unsigned short, __float2bfloat16(v)); }
  static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c, 0, 0, 0);
  }
};
struct StemF16 {
  static __device__ __forceinline__ unsigned short cvt(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }
  static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
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

int launch_stem_conv(const float* x, int n, int h, int w, int dtype, const void* d_wfrag, const float* d_bias,
                     const TensorView& out, hipStream_t s) {
  RON_REQUIRE(dtype == RON_DTYPE_BF16 || dtype == RON_DTYPE_F16, "stem kernel: bf16 / f16 only");
  RON_REQUIRE(w % 32 == 0 && out.C == 64 && out.cstride == 64 && out.coff == 0 && out.H == h && out.W == w, "stem kernel: bad shape");
  const long long tiles = (long long)n * h * (w / 32);
  const int grid = (int)std::min<long long>((tiles + 3) / 4, 256 * 8);
  if (dtype == RON_DTYPE_BF16)
    hipLaunchKernelGGL(stem_conv_kernel<StemBF16>, dim3(grid), dim3(256), 0, s, x, n, h, w, (const u32x4*)d_wfrag, d_bias,
                       (unsigned*)out.base, out.Hp(), out.Wp(), out.pad);
  else
    hipLaunchKernelGGL(stem_conv_kernel<StemF16>, dim3(grid), dim3(256), 0, s, x, n, h, w, (const u32x4*)d_wfrag, d_bias,
                       (unsigned*)out.base, out.Hp(), out.Wp(), out.pad);
  RON_HIP_CHECK(hipGetLastError());
  return RON_OK;
}

}  // namespace ron
