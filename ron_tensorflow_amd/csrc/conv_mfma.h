// Host-side description of one implicit-GEMM convolution launch (see conv_mfma.hip).
#pragma once
#include <vector>

#include "common.h"

namespace ron {

// NHWC activation tensor in HBM with SHARED zero halos: rows are W + pad pixels long (the right halo of a row is the
// left halo of the next one), images are H + pad rows apart (the bottom halo of an image is the top halo of the next),
// one more band of `pad` rows + `pad` pixels closes the allocation.  Pixel (n, y, x) sits at row n*(H+pad) + pad + y,
// column pad + x; a tap that leaves the image lands in a halo pixel whichever side it leaves on, so the 3x3 / 7x7 /
// dilated taps never need a bounds check.  Kernels write interiors only; halos are zeroed once at allocation.
// The flat view matters to conv_patch.hip: positions that are consecutive in memory are consecutive pixels of the map
// with only `pad` halo pixels per row between them (4.8 % of a 40 x 40 map, not 9.3 %).
// Element type = the ctx dtype (bf16 / f16 / f32).  A view may address a channel slice [coff, coff + C) of a wider tensor.
struct TensorView {
  void* base = nullptr;      // start of the allocation (row 0, column 0, channel 0)
  int64_t bytes = 0;         // size of the allocation
  int N = 0, H = 0, W = 0, C = 0;
  int pad = 0;
  int cstride = 0;           // elements per pixel in memory
  int coff = 0;              // first channel of the view
  int Hp() const { return H + pad; }     // rows from one image to the next
  int Wp() const { return W + pad; }     // pixels per row
  int64_t pixels() const { return halo_pixels(N, H, W, pad); }
  static int64_t halo_pixels(int n, int h, int w, int pad) { return ((int64_t)n * (h + pad) + pad) * (w + pad) + pad; }
};

struct ConvLaunch {
  int dtype = RON_DTYPE_BF16;
  TensorView in, out;
  const void* res = nullptr;     // residual with the geometry of `out` (same dtype), or null
  const void* wgt = nullptr;     // packed [Npad][K] (K = kh*kw*Cin contiguous), dtype elements
  int64_t wgt_bytes = 0;
  const void* wgt_c64 = nullptr; // the same weights as LDS images for the resident-weight kernel (conv_c64.hip), or null
  const float* bias = nullptr;   // [Npad] fp32 (never null; zeros when the layer has none)
  float oscale = 1.f;            // out = acc * oscale + bias: 2^-k of the packed weights' scale (RON_DTYPE_F16X3), else 1
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
  TensorView out2;               // with pool: the un-pooled map too (base == nullptr: none), same dtype, geometry Ho x Wo x Cout
  // Two fp32 outputs from one convolution (ConvArgs::split_n): packed columns [0, split_first) -> `out`, [split_n, Cout) -> `out2`,
  // both un-haloed fp32 views (out_f32 = 1).  Row-gather kernel only, a launch of its own; split-K allowed (the finalize pass routes the
  // columns the same way: block4 of SSD-512 at small batches splits).  0: one output.
  int split_n = 0, split_first = 0;
  int center_from = 0;           // > 0: output channels >= center_from have weights in the CENTRE tap only (a 1x1 branch packed beside
                                 // 3x3 ones: nets/ron_vgg_320.py:378-397); their column tiles run that tap's K steps alone
  int halo_skip = 1;             // position-major rows + per-tile skipping of filter rows that only see the halo, where it pays (0: never)
  int cfg = -1;                  // tile configuration (kCfg* below); -1 = pick by shape
  int splitk = -1;               // split-K factor; -1 = pick by grid size, 1 = off
  void* scratch = nullptr;       // fp32 slabs for split-K (conv_scratch_bytes); null disables split-K
  int64_t scratch_bytes = 0;
};

constexpr int kWeightBlockRows = 64;     // packed weights: [Npad / 64][K steps][64 rows][128 B] (pack.h, block_rows)

// Tile configurations of the shipped library: exactly the ones conv_pick_cfg() can return (ron_conv_num_tile_cfgs()).
enum {
  kCfgIgemm256 = 0,        // row-gather kernel (conv_mfma.hip), 256 x 256 tile, 8 waves, 128 KB LDS
  kCfgIgemm128 = 1,        // 128 x 128, 4 waves, 2 workgroups / CU
  kCfgIgemm128Early = 2,   // ... with a stage's LDS-DMA pieces issued during its first k-step
  kCfgIgemm128x64 = 3,     // 128 x 64 (Cout <= 64 and the skinny heads)
  kCfgPatch256 = 4,        // halo-patch kernel (conv_patch.hip), 256 positions x 256 channels
  kCfgPatch128 = 5,        // ... x 128 channels, 3 weight stages
  kCfgPatch64 = 6,         // ... x 64 channels, 3 weight stages
  kCfgIgemm256TapsInner = 7,   // 256 x 256 with K ordered chunk-major, taps innermost (layers with <= 2 column tiles)
  kCfgC64Resident = 8,     // 3x3 on a 64-channel map with the weights resident in LDS (conv_c64.hip)
  kCfgIgemm128EarlyTapsInner = 9,   // 128 x 128, early issue, K ordered chunk-major with the taps innermost (launches that do not split K)
  kCfgIgemm256x128 = 10,   // 256 x 128 on four waves with the assembly K loop (bf16 / f16 / f16x3): layers with Cout = 128 (conv2_x)
  kNumCfgs = 11
};
inline bool conv_cfg_taps_inner(int cfg) { return cfg == kCfgIgemm256TapsInner || cfg == kCfgIgemm128EarlyTapsInner; }
inline bool conv_cfg_is_patch(int cfg) { return cfg >= kCfgPatch256 && cfg <= kCfgPatch64; }

int launch_conv(const ConvLaunch& c, hipStream_t stream);
// what launch_conv would do with `c`: out = {tile configuration (kCfg*), split-K factor, tile order (ConvArgs::m_fastest), K order (taps_inner)}
int conv_describe(const ConvLaunch& c, int out[4]);
// 3x3 / stride 1 / pad 1 on a 64-channel map, weights resident in LDS (conv_c64.hip); pack_conv_c64_weights builds wgt_c64
bool conv_c64_applicable(const ConvLaunch& c);
int launch_conv_c64(const ConvLaunch& c, hipStream_t stream);
std::vector<uint8_t> pack_conv_c64_weights(const std::vector<float>& rows, int npad, int dtype);
// 3x3 / stride 1 / pad 1 with the input halo patch staged once per channel chunk (conv_patch.hip)
bool conv_patch_applicable(const ConvLaunch& c);
int conv_patch_pick(const ConvLaunch& c);          // kCfgPatch* when the patch kernel is the better choice for this launch, else -1
int launch_conv_patch(const ConvLaunch& c, int cfg, hipStream_t stream);
// two convolutions over channel slices of the same map as ONE launch of the patch kernel (the skinny heads of a scale)
bool conv_patch_pair_applicable(const ConvLaunch& a, const ConvLaunch& b);
int launch_conv_patch_pair(const ConvLaunch& a, const ConvLaunch& b, hipStream_t stream);
// Elements along K one staging step covers for this dtype (Cin must be a multiple of it).
int conv_k_chunk(int dtype);
int conv_n_tile(int cout);       // granularity Cout is padded to (64 or 128)
int conv_num_cfgs();
int conv_pick_cfg(const ConvLaunch& c);
int conv_pick_splitk(int tiles, int KT, int slots, int tile_elems);      // tile_elems = BM x BN of the configuration (slab bytes of the cost model)
int64_t conv_scratch_bytes(const ConvLaunch& c);   // fp32 split-K slabs this launch can ask for (0: none)
// Several mutually independent convolutions as ONE launch of the row-gather kernel (tile kCfgIgemm128x64, kCfgIgemm128 or kGroupMixed)
// sk_plan: the members' split-K factors from conv_group_plan (depends on the members' geometry and the batch only: callers cache it;
// the mixed-width form models the launch's schedule, ~1 ms of host time), or null to compute them here
int launch_conv_group(const ConvLaunch* ls, int n, int cfg, void* scratch, int64_t scratch_bytes, hipStream_t stream, const int* sk_plan = nullptr);
void conv_group_plan(const ConvLaunch* ls, int n, int cfg, int* sk);
int64_t conv_group_scratch_bytes(const ConvLaunch* ls, int n, int cfg, const int* sk_plan = nullptr);   // sk_plan: from conv_group_plan, or null
constexpr int kMaxConvGroup = 8;
// launch_conv_group only: every member on the 128-row tile of its own width (128 x 128 when Npad % 128 == 0, else 128 x 64)
constexpr int kGroupMixed = 64;
size_t dtype_size(int dtype);
inline bool dtype_is_half(int dtype) { return dtype == RON_DTYPE_BF16 || dtype == RON_DTYPE_F16; }   // 2-byte elements

// conv1_1 (stem.hip): 3 -> 64 channels straight from the fp32 image (bf16 / f16; f16x3: the split-precision form)
void stem_pack_weights(const float* hwio, int dtype, std::vector<uint16_t>* frags);
float stem_pack_weights_split(const float* hwio, std::vector<uint16_t>* frags);     // RON_DTYPE_F16X3: returns the epilogue's 2^-k
int launch_stem_conv(const float* x, int n, int h, int w, int dtype, const void* d_wfrag, const float* d_bias,
                     const TensorView& out, hipStream_t s, float oscale = 1.f);

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
