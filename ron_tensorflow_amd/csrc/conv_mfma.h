// Host-side description of one implicit-GEMM convolution launch (see conv_mfma.hip).
#pragma once
#include <vector>

#include "common.h"

namespace ron {

// NHWC activation tensor in HBM.  Every activation carries a zero halo of `pad` pixels so the
// 3x3 / 7x7 / dilated taps never need a bounds check: memory is [N][H+2pad][W+2pad][cstride],
// element type = the ctx dtype (bf16 / f16 / f32).  A view may address a channel slice
// [coff, coff + C) of a wider tensor.
struct TensorView {
  void* base = nullptr;      // start of the allocation (pixel (0,-pad,-pad), channel 0)
  int64_t bytes = 0;         // size of the allocation
  int N = 0, H = 0, W = 0, C = 0;
  int pad = 0;
  int cstride = 0;           // elements per pixel in memory
  int coff = 0;              // first channel of the view
  int Hp() const { return H + 2 * pad; }
  int Wp() const { return W + 2 * pad; }
};

struct ConvLaunch {
  int dtype = RON_DTYPE_BF16;
  TensorView in, out;
  const void* res = nullptr;     // residual with the geometry of `out` (same dtype), or null
  const void* wgt = nullptr;     // packed [Npad][K] (K = kh*kw*Cin contiguous), dtype elements
  int64_t wgt_bytes = 0;
  const float* bias = nullptr;   // [Npad] fp32 (never null; zeros when the layer has none)
  int Cout = 0;                  // GEMM N that is stored (<= Npad)
  int Npad = 0;                  // rows of wgt, multiple of the N tile
  int kh = 1, kw = 1, stride = 1, dil = 1, cpad = 0;
  int relu = 0;
  int out_f32 = 0;               // store fp32 (head logits) instead of dtype
  // conv2d_transpose with kernel == stride == `up` (pixel shuffle epilogue); 0 = plain conv.
  int up = 0;
  int up_cout = 0;               // channels per tap of the transposed conv (Cout = up*up*up_cout)
  int Ho = 0, Wo = 0;            // GEMM rows = N*Ho*Wo (input grid for a transposed conv)
  int pool = 0;                  // fuse slim.max_pool2d [2,2] into the epilogue: `out` is the pooled map
  int cfg = -1;                  // tile configuration index (conv_mfma.hip kCfgs); -1 = pick by shape
  int splitk = -1;               // split-K factor; -1 = pick by grid size, 1 = off
  void* scratch = nullptr;       // fp32 slabs for split-K (conv_scratch_bytes); null disables split-K
  int64_t scratch_bytes = 0;
  unsigned long long* dbg = nullptr;   // stamp builds (tile cfgs 27-29): 4 x u64 per wave of the grid
};

int launch_conv(const ConvLaunch& c, hipStream_t stream);
// 3x3 / stride 1 / pad 1 with the input halo patch staged once per channel chunk (conv_patch.hip); cfg = kCfgPatch forces it
constexpr int kCfgPatch = 100;
bool conv_patch_applicable(const ConvLaunch& c);
int launch_conv_patch(const ConvLaunch& c, hipStream_t stream);
// Elements along K one staging step covers for this dtype (Cin must be a multiple of it).
int conv_k_chunk(int dtype);
int conv_n_tile(int cout);       // granularity Cout is padded to (64 or 128)
int conv_num_cfgs();
int conv_pick_cfg(int M, int Npad, int K);
int conv_pick_splitk(int tiles, int KT, int slots);
int64_t conv_scratch_bytes(int M, int Npad, int K, int dtype, int cfg, int splitk);
size_t dtype_size(int dtype);

// conv1_1 (stem.hip): 3 -> 64 channels straight from the fp32 image, bf16 / f16 only
void stem_pack_weights(const float* hwio, int dtype, std::vector<uint16_t>* frags);
int launch_stem_conv(const float* x, int n, int h, int w, int dtype, const void* d_wfrag, const float* d_bias,
                     const TensorView& out, hipStream_t s);

// conv1_1 + conv1_2 + pool1 fused (stem.hip): image -> pool1, bf16 / f16 only
void stem2_pack_weights(const float* hwio, int dtype, std::vector<uint16_t>* lds_image);
void stem2_pack_w1(const float* hwio, int dtype, std::vector<uint16_t>* frags);
int launch_stem2(const float* x, int n, int h, int w, int dtype, const void* d_w1frag, const float* d_bias1,
                 const void* d_w2img, const float* d_bias2, const TensorView& out, hipStream_t s);

// helpers (elementwise.hip)
int launch_im2col_c3(const float* x, int n, int h, int w, int dtype, void* out, int kchunk, hipStream_t s);
int launch_maxpool2x2(const TensorView& in, const TensorView& out, int dtype, hipStream_t s);
int launch_maxpool3x3s1(const TensorView& in, const TensorView& out, int dtype, hipStream_t s);
int launch_l2norm(const TensorView& in, const TensorView& out, const float* d_gamma, int dtype, hipStream_t s);
int launch_pack_input(const float* x, const TensorView& out, int dtype, hipStream_t s);      // dense fp32 -> view
int launch_fill_random(const TensorView& out, int dtype, unsigned seed, hipStream_t s);       // interior <- U[-1,1)
int launch_unpack(const TensorView& in, int dtype, int in_is_f32, float* y, hipStream_t s);  // view -> dense fp32

}  // namespace ron
