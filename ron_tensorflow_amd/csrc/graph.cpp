// ron_ctx: the RON-320 conv stack as a fixed launch plan over pre-allocated HBM tensors.
//
// Restates the graph of nets/ron_vgg_320.py (ron_net :434-508, ron_net_reducedfc :510-580,
// reverse_connection_module_with_pred :418-432, pred_cls_module :378-404, reg_bbox_module
// :406-415) with the slim layer semantics of ron_arg_scope (:595-629), re-planned for the
// MFMA implicit-GEMM kernel:
//   * inference BatchNorm (eps 1e-5) is folded into the preceding conv at load time;
//   * per scale, everything that reads the reference map (objectness hidden layer, box hidden layer,
//     both inception-1 branches) runs as ONE conv with 2048 output channels, the 1x1 branch in the centre
//     tap of its rows (those column tiles run that tap's K steps only); inception-2 likewise; the
//     BatchNorm after each concat is split per branch and folded; consumers read channel slices;
//   * the 2x2 stride-2 transposed conv is a GEMM with a pixel-shuffle epilogue; the left conv of a
//     reverse connection writes its half of relu(left + up) first and the transposed conv adds its
//     half in place, so only the (small) transposed convs sit on the coarse -> fine chain;
//   * head logits are written as fp32 straight into the caller's buffers.
// Every activation lives in HBM as NHWC with a zero halo (conv_mfma.h); all buffers are
// allocated once for max_batch images (activations of the full variant: ~150 MB / image).
#include <math.h>

#include <array>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "pack.h"

using namespace ron;

namespace {

constexpr float kBnEps = 1e-5f;
constexpr int kCarrierPlanMinBatch = 24;   // contexts from this max_batch on pack the small head convolutions into the partial rounds of 256 x 256 launches
constexpr int kLevelPlanMaxBatch = 12;      // contexts up to this max_batch run the heads one launch per dependency level (plan_groups)
const char* kFeatLayers[4] = {"block7", "block6", "block5", "block4"};

struct Var {
  std::string name;
  std::vector<int64_t> shape;
  std::vector<float> data;
  bool loaded = false;
  int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

struct Tensor {
  std::string name;
  int H, W, C, pad;
  int cstride = 0;           // elements per pixel in memory (>= C): wide tensors get +64 so that the pixel stride
                             // is not a power of two (4 KiB strides serialise on a few L2 channels)
  void* d = nullptr;
  int64_t bytes = 0;
};

struct PackedConv {
  void* d_w = nullptr;
  int64_t w_bytes = 0;
  void* d_w_c64 = nullptr;    // 3x3 on 64 input channels (conv2_1): the weights once more as LDS images for conv_c64.hip
  float* d_bias = nullptr;
  float oscale = 1.f;         // accumulator scale of the epilogue (split-precision weights are stored times a power of two)
  int Npad = 0, Cout = 0;
  int split_n = 0, split_first = 0;   // two head convolutions packed side by side (pack_box_pair): ConvLaunch::split_n
};

enum OpKind { OP_IM2COL, OP_CONV, OP_POOL, OP_STEM, OP_POOL3, OP_L2NORM, OP_STEM2 };

struct Op {
  OpKind kind;
  std::string name;
  int in = -1, out = -1, res = -1;      // tensor indices; out == -2: caller head buffer
  int in_coff = 0, in_C = 0;            // channel slice of the input
  int out_coff = 0, out_C = 0;          // channel slice of the output (out_C = 0: the whole tensor)
  int packed = -1;
  int kh = 1, kw = 1, stride = 1, dil = 1, cpad = 0, relu = 0;
  int up = 0, up_cout = 0, pool = 0;
  int center_from = 0;                  // output channels >= this hold a 1x1 branch in the centre tap of the 3x3 filter (0: none)
  int fuse_next_pool = 0;               // the next op is this conv's 2x2 pool and both maps are needed (conv4_3, conv5_3): one launch
                                        // with two outputs whenever the conv would not split K (decided per batch in ron_forward)
  int head_kind = -1, head_layer = -1;  // 0 cls, 1 obj, 2 loc
  int head_kind2 = -1;                  // a second head output of the same layer from the same launch (pack_box_pair), or -1
  int Ho = 0, Wo = 0;
  int lane = 0;                         // 0 = the caller's stream; 1..3 = side streams (independent head branches)
  int group = -1;                       // >= 0: launched together with the neighbouring ops of the same group (launch_conv_group)
  int group_cfg = 0;                    // tile configuration of that grouped launch
  double flops = 0;                     // algorithmic 2*MAC per image of this launch
  double act_bytes = 0;                 // algorithmic HBM bytes per image: input read once + output written once
  double wgt_bytes = 0;                 // ... plus the packed weights, once per launch
};

struct OpTiming {
  double ms = 0;
  int launches = 0;
};

}  // namespace

struct ron_ctx {
  ron_config cfg;
  int c6 = 0;                 // fc6 / fc7 channels
  int num_anchors = 10;                 // RON: anchors per cell on every scale
  int feat[4] = {0, 0, 0, 0};
  // head layers, generic (RON: 4 scales x 10 anchors + objectness; SSD-512: 7 scales, 4/6 anchors, no objectness)
  int n_feat = 4;
  int feat_h[RON_MAX_LAYERS] = {}, feat_w[RON_MAX_LAYERS] = {}, feat_A[RON_MAX_LAYERS] = {};
  bool has_obj = true;
  bool is_ssd() const { return cfg.variant == RON_VARIANT_SSD512; }
  const char* scope() const { return is_ssd() ? "ssd_512_vgg" : "ron_320_vgg"; }
  float* d_l2_gamma = nullptr;          // SSD block4 L2Normalization scale
  std::vector<Var> vars;
  std::map<std::string, int> var_index;
  std::vector<Tensor> tensors;
  std::map<std::string, int> tensor_index;
  std::vector<PackedConv> packed;
  std::vector<Op> ops;
  bool finalized = false;
  double flops_per_image = 0;
  // anchors (device) + head buffers / workspace for ron_detect
  float* d_anchor[RON_MAX_LAYERS][4] = {};
  float* d_head[3][RON_MAX_LAYERS] = {};
  void* d_post_ws = nullptr;
  int64_t post_ws_bytes = 0;
  bool post_ws_dirty = false;     // a ron_detect call failed after its select pass may have run: the self-cleaning counters are re-zeroed on the next call
  void* d_stem_w = nullptr;             // conv1_1 fragments + bias for the dedicated stem kernel (bf16 / f16)
  float* d_stem_b = nullptr;
  float stem_oscale = 1.f;              // split precision: 2^-k of the stem weights' scale
  void* d_stem2_w = nullptr;            // conv1_2 weights as the LDS image of stem2_kernel (conv1_1 + conv1_2 + pool1 fused)
  void* d_stem2_w1 = nullptr;           // conv1_1 weights as 16x16x32 fragments for stem2_kernel
  float* d_stem2_b = nullptr;
  void* d_splitk[4] = {};               // fp32 slabs of the split-K launches, one set per stream lane
  int64_t splitk_bytes[4] = {};
  // RON_CFG_MULTI_STREAM: the heads of the three coarse scales run on side streams beside the main chain
  hipStream_t side[4] = {};
  hipEvent_t lane_ready[4] = {}, lane_done[4] = {};
  // optional per-launch timing (ron_profile_*): event pairs recorded on the caller's stream
  int profiling = 0;                    // calls still to be recorded (ron_profile_enable)
  std::vector<OpTiming> timing;                       // ops.size() + 1 (last = post-processing)
  std::vector<std::string> labels;                    // ron_profile_get names, one per op, built once (stable until ron_destroy)
  int grouped_launches = 0;                           // grouped launches in the plan (0: one launch per convolution)
  // split-K factors of the grouped launches, planned once per (first op of the group, batch) - conv_group_plan models the launch's
  // schedule, which is far too slow for the enqueue path: [op index] -> [batch] -> factors (first = 0: not planned yet)
  std::map<int, std::vector<std::array<int, kMaxConvGroup>>> group_sk;
  // conv4_3 / conv5_3 with their pool from the same launch (Op::fuse_next_pool): decided once per (op, batch) - the answer is "the
  // two-output launch would not split K", which takes a tile-configuration pick: [op index] -> [batch] -> -1 unknown / 0 / 1
  std::map<int, std::vector<signed char>> fuse_pool_ok;
  std::vector<std::vector<hipEvent_t>> pending;       // per recorded call: one event per stamp ...
  std::vector<std::vector<int>> pending_ops;          // ... and what it marks: op index (its start), -1 = end of a lane,
                                                      //     -2 / -3 = start / end of the post-processing stage
  std::vector<hipEvent_t> event_pool;
  // ron_clone: an execution slot that borrows the packed weights of `weights_owner` (its own activations, scratch, streams)
  ron_ctx* weights_owner = nullptr;
  int clones = 0;                       // live slots that borrow this context's weights

  int esz() const { return (int)dtype_size(cfg.dtype); }
  int add_tensor(const std::string& name, int H, int W, int C, int pad) {
    Tensor t;
    t.name = name; t.H = H; t.W = W; t.C = C; t.pad = pad;
    t.cstride = C >= 1024 ? C + 64 : C;
    tensors.push_back(t);
    tensor_index[name] = (int)tensors.size() - 1;
    return (int)tensors.size() - 1;
  }
  void add_var(const std::string& rel, std::vector<int64_t> shape) {
    Var v;
    v.name = std::string(scope()) + "/" + rel;
    v.shape = shape;
    var_index[v.name] = (int)vars.size();
    vars.push_back(v);
  }
  const Var& var(const std::string& rel) const { return vars[var_index.at(std::string(scope()) + "/" + rel)]; }
  TensorView view(int t, int n, int coff = 0, int C = -1) const {
    const Tensor& T = tensors[t];
    TensorView v;
    v.base = T.d; v.bytes = T.bytes; v.N = n; v.H = T.H; v.W = T.W; v.pad = T.pad; v.cstride = T.cstride;
    v.coff = coff; v.C = C < 0 ? T.C : C;
    return v;
  }
};

namespace {

void add_bn_vars(ron_ctx* c, const std::string& scope, int ch) {
  for (const char* n : {"beta", "gamma", "moving_mean", "moving_variance"}) c->add_var(scope + "/BatchNorm/" + n, {ch});
}

// The variable list (names/shapes as TensorFlow stores them; SURVEY.md 8b weight contract).
void declare_variables(ron_ctx* c) {
  const int nc = c->cfg.num_classes, A = c->num_anchors, c6 = c->c6;
  const int widths[5] = {64, 128, 256, 512, 512};
  const int reps[5] = {2, 2, 3, 3, 3};
  int cin = 3;
  for (int b = 0; b < 5; ++b)
    for (int r = 0; r < reps[b]; ++r) {
      const std::string s = "conv" + std::to_string(b + 1) + "/conv" + std::to_string(b + 1) + "_" + std::to_string(r + 1);
      c->add_var(s + "/weights", {3, 3, cin, widths[b]});
      c->add_var(s + "/biases", {widths[b]});
      cin = widths[b];
    }
  const int k6 = c->cfg.variant == RON_VARIANT_FULL ? 7 : 3;
  c->add_var("fc6/weights", {k6, k6, 512, c6});
  c->add_var("fc6/biases", {c6});
  c->add_var("fc7/weights", {1, 1, c6, c6});
  c->add_var("fc7/biases", {c6});
  for (int i = 0; i < 4; ++i) {
    const std::string L = std::string("reverse_module/") + kFeatLayers[i] + "_reverse";
    const int left_c = i < 2 ? c6 : 512;
    const int k = i == 0 ? 2 : 3;
    c->add_var(L + "_conv_left/weights", {k, k, left_c, 512});
    add_bn_vars(c, L + "_conv_left", 512);
    if (i > 0) {
      c->add_var(L + "_deconv_right/weights", {2, 2, 512, 512});
      c->add_var(L + "_deconv_right/biases", {512});
    }
    c->add_var(L + "_objectness/weights", {3, 3, 512, 512});
    add_bn_vars(c, L + "_objectness", 512);
    c->add_var(L + "_objectness_score/weights", {3, 3, 512, 2 * A});
    c->add_var(L + "_objectness_score/biases", {2 * A});
    for (int blk = 1; blk <= 2; ++blk) {
      const std::string I = L + "_inception" + std::to_string(blk);
      const int ic = blk == 1 ? 512 : 1024;
      c->add_var(I + "/Branch_0/Conv2d_3x3/weights", {3, 3, ic, 512});
      c->add_var(I + "/Branch_0/Conv2d_3x3/biases", {512});
      c->add_var(I + "/Branch_1/Conv2d_1x1/weights", {1, 1, ic, 512});
      c->add_var(I + "/Branch_1/Conv2d_1x1/biases", {512});
      add_bn_vars(c, I, 1024);
    }
    c->add_var(L + "_inception2/Conv2d_pred_3x3/weights", {3, 3, 1024, A * nc});
    c->add_var(L + "_inception2/Conv2d_pred_3x3/biases", {A * nc});
    c->add_var(L + "/Conv2d_0_3x3/weights", {3, 3, 512, 512});
    add_bn_vars(c, L + "/Conv2d_0_3x3", 512);
    c->add_var(L + "/Conv2d_1_3x3/weights", {3, 3, 512, 4 * A});
    c->add_var(L + "/Conv2d_1_3x3/biases", {4 * A});
  }
}

// ---- weight assembly: fp32 rows [npad][K] + bias [npad] -----------------------------------
struct Rows {
  int K = 0, npad = 0, kh = 0, kw = 0, cin = 0;
  std::vector<float> w, b;
  void init(int kh_, int kw_, int cin_, int n_real, int ntile) {
    kh = kh_; kw = kw_; cin = cin_; K = kh * kw * cin; npad = round_up(n_real, ntile);
    w.assign((size_t)npad * K, 0.f);
    b.assign(npad, 0.f);
  }
  // place an HWIO filter (fh x fw, centred) at output rows [n_off, n_off + cout)
  void place(const Var& wv, int n_off) {
    const int fh = (int)wv.shape[0], fw = (int)wv.shape[1], ci = (int)wv.shape[2], co = (int)wv.shape[3];
    const int oy = (kh - fh) / 2, ox = (kw - fw) / 2;
    for (int y = 0; y < fh; ++y)
      for (int x = 0; x < fw; ++x)
        for (int c = 0; c < ci; ++c) {
          const float* src = &wv.data[(((size_t)y * fw + x) * ci + c) * co];
          const size_t k = ((size_t)(y + oy) * kw + (x + ox)) * cin + c;
          for (int n = 0; n < co; ++n) w[(size_t)(n_off + n) * K + k] = src[n];
        }
  }
  void add_bias(const Var& bv, int n_off) { for (size_t n = 0; n < bv.data.size(); ++n) b[n_off + n] += bv.data[n]; }
  // y = gamma * (x - mean) / sqrt(var + eps) + beta  folded into rows [n_off, n_off + ch)
  void fold_bn(const ron_ctx* c, const std::string& scope, int n_off, int ch, int bn_off = 0) {
    const Var& be = c->var(scope + "/BatchNorm/beta"); const Var& ga = c->var(scope + "/BatchNorm/gamma");
    const Var& mu = c->var(scope + "/BatchNorm/moving_mean"); const Var& va = c->var(scope + "/BatchNorm/moving_variance");
    for (int n = 0; n < ch; ++n) {
      const int q = bn_off + n;
      const float s = ga.data[q] / sqrtf(va.data[q] + kBnEps);
      float* row = &w[(size_t)(n_off + n) * K];
      for (int k = 0; k < K; ++k) row[k] *= s;
      b[n_off + n] = (b[n_off + n] - mu.data[q]) * s + be.data[q];
    }
  }
};

int upload(ron_ctx* c, const Rows& r, int cout) {
  PackedConv p;
  std::vector<uint8_t> bytes = pack_conv_weights(r.w, r.npad, c->cfg.dtype, &p.oscale);
  p.w_bytes = (int64_t)bytes.size();
  p.Npad = r.npad; p.Cout = cout;
  RON_HIP_CHECK(ron::dev_malloc(&p.d_w, bytes.size()));
  RON_HIP_CHECK(ron::dev_memcpy(p.d_w, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
  if (dtype_is_half(c->cfg.dtype) && r.kh == 3 && r.kw == 3 && r.cin == 64 && r.npad == cout && cout % 64 == 0) {
    const std::vector<uint8_t> img = pack_conv_c64_weights(r.w, r.npad, c->cfg.dtype);
    RON_HIP_CHECK(ron::dev_malloc(&p.d_w_c64, img.size()));
    RON_HIP_CHECK(ron::dev_memcpy(p.d_w_c64, img.data(), img.size(), hipMemcpyHostToDevice));
  }
  RON_HIP_CHECK(ron::dev_malloc((void**)&p.d_bias, r.b.size() * sizeof(float)));
  RON_HIP_CHECK(ron::dev_memcpy(p.d_bias, r.b.data(), r.b.size() * sizeof(float), hipMemcpyHostToDevice));
  c->packed.push_back(p);
  return (int)c->packed.size() - 1;
}

// returns packed index (>= 0) or negative status
int pack_plain(ron_ctx* c, const std::string& scope, bool bn) {
  const Var& w = c->var(scope + "/weights");
  const int cout = (int)w.shape[3];
  Rows r;
  r.init((int)w.shape[0], (int)w.shape[1], (int)w.shape[2], cout, conv_n_tile(cout));
  r.place(w, 0);
  if (bn) r.fold_bn(c, scope, 0, cout); else r.add_bias(c->var(scope + "/biases"), 0);
  return upload(c, r, cout);
}

// The class and box convolutions of an SSD feature layer (nets/ssd_vgg_300.py:403-431: both 3x3 over the same map) as ONE
// convolution: rows [0, A*classes) conv_cls, rows [split_n, split_n + 4A) conv_loc, split_n = A*classes rounded up to 8 (a lane's
// vector of adjacent channels then never straddles the two outputs).  The input is staged once instead of twice and the box
// columns ride in what would be padding of the class convolution's last column tile (block4: 84 + 16 -> 104 of 128 columns).
int pack_box_pair(ron_ctx* c, const std::string& L) {
  const Var& wc = c->var(L + "/conv_cls/weights");
  const Var& wl = c->var(L + "/conv_loc/weights");
  const int n_cls = (int)wc.shape[3], n_loc = (int)wl.shape[3], split_n = round_up(n_cls, 8);
  Rows r;
  r.init((int)wc.shape[0], (int)wc.shape[1], (int)wc.shape[2], split_n + n_loc, conv_n_tile(split_n + n_loc));
  r.place(wc, 0);
  r.add_bias(c->var(L + "/conv_cls/biases"), 0);
  r.place(wl, split_n);
  r.add_bias(c->var(L + "/conv_loc/biases"), split_n);
  const int idx = upload(c, r, split_n + n_loc);
  if (idx >= 0) { c->packed[idx].split_n = split_n; c->packed[idx].split_first = n_cls; }
  return idx;
}

int pack_stem(ron_ctx* c, const std::string& scope) {
  const Var& w = c->var(scope + "/weights");
  const int cout = (int)w.shape[3], chunk = conv_k_chunk(c->cfg.dtype);
  Rows r;
  r.init(1, 1, chunk, cout, conv_n_tile(cout));
  for (int k = 0; k < 27; ++k) for (int n = 0; n < cout; ++n) r.w[(size_t)n * chunk + k] = w.data[(size_t)k * cout + n];
  r.add_bias(c->var(scope + "/biases"), 0);
  return upload(c, r, cout);
}

int pack_deconv(ron_ctx* c, const std::string& scope) {
  const Var& w = c->var(scope + "/weights");     // [kh, kw, Cout, Cin]
  const Var& bv = c->var(scope + "/biases");
  const int taps = (int)(w.shape[0] * w.shape[1]), co = (int)w.shape[2], ci = (int)w.shape[3];
  Rows r;
  r.init(1, 1, ci, taps * co, conv_n_tile(taps * co));
  memcpy(r.w.data(), w.data.data(), w.data.size() * sizeof(float));
  for (int t = 0; t < taps; ++t) for (int n = 0; n < co; ++n) r.b[t * co + n] = bv.data[n];
  return upload(c, r, taps * co);
}

// Everything that reads the reference map of a scale, as ONE convolution with 2048 outputs = the per-scale "hcat" tensor:
// rows 0..511 objectness hidden (3x3 conv + BN), 512..1023 box hidden (3x3 conv + BN), 1024..1535 inception-1 branch 0 (3x3 + bias,
// BN channels 0..511 of the concat), 1536..2047 inception-1 branch 1 (1x1 + bias, BN channels 512..1023) -- the 1x1 filter sits in
// the CENTRE tap of its rows and those column tiles run that tap's K steps only (ConvLaunch::center_from = 1536), so it costs
// its own MACs, not nine times them.
int pack_trio3(ron_ctx* c, const std::string& L) {
  Rows r;
  r.init(3, 3, 512, 2048, 256);
  r.place(c->var(L + "_objectness/weights"), 0);
  r.fold_bn(c, L + "_objectness", 0, 512);
  r.place(c->var(L + "/Conv2d_0_3x3/weights"), 512);
  r.fold_bn(c, L + "/Conv2d_0_3x3", 512, 512);
  r.place(c->var(L + "_inception1/Branch_0/Conv2d_3x3/weights"), 1024);
  r.add_bias(c->var(L + "_inception1/Branch_0/Conv2d_3x3/biases"), 1024);
  r.fold_bn(c, L + "_inception1", 1024, 512, 0);
  r.place(c->var(L + "_inception1/Branch_1/Conv2d_1x1/weights"), 1536);
  r.add_bias(c->var(L + "_inception1/Branch_1/Conv2d_1x1/biases"), 1536);
  r.fold_bn(c, L + "_inception1", 1536, 512, 512);
  return upload(c, r, 2048);
}

// Both branches of an "inception" block (nets/ron_vgg_320.py:378-397) as one convolution over the block's input: rows 0..511 the 3x3
// branch, 512..1023 the 1x1 branch in the centre tap (center_from = 512); each conv + bias, then its half of the BatchNorm that
// follows the concat, ReLU in the kernel epilogue.
int pack_inception(ron_ctx* c, const std::string& I) {
  const Var& w3 = c->var(I + "/Branch_0/Conv2d_3x3/weights");
  Rows r;
  r.init(3, 3, (int)w3.shape[2], 1024, 256);
  r.place(w3, 0);
  r.add_bias(c->var(I + "/Branch_0/Conv2d_3x3/biases"), 0);
  r.fold_bn(c, I, 0, 512, 0);
  r.place(c->var(I + "/Branch_1/Conv2d_1x1/weights"), 512);
  r.add_bias(c->var(I + "/Branch_1/Conv2d_1x1/biases"), 512);
  r.fold_bn(c, I, 512, 512, 512);
  return upload(c, r, 1024);
}


// ---------------------------------------------------------------------------------------------------------
// SSD-512 (nets/ssd_vgg_512.py:364-460, multibox heads nets/ssd_vgg_300.py:403-431)
// ---------------------------------------------------------------------------------------------------------
const char* kSsdFeat[7] = {"block4", "block7", "block8", "block9", "block10", "block11", "block12"};
const int kSsdAnchors[7] = {4, 6, 6, 6, 6, 4, 4};       // len(sizes) + len(ratios), nets/ssd_vgg_512.py:86-99
const int kSsdFeatC[7] = {512, 1024, 512, 256, 256, 256, 256};
// Feature layers whose loc + cls convolutions run as one two-output launch (block4: 64 x 64, block7: 32 x 32).  The small maps'
// heads stay two members of a grouped launch with the next block's 1x1 (plan_groups).
const int kSsdPairedHeads = 2;

void declare_variables_ssd(ron_ctx* c) {
  const int nc = c->cfg.num_classes;
  const int widths[5] = {64, 128, 256, 512, 512};
  const int reps[5] = {2, 2, 3, 3, 3};
  int cin = 3;
  for (int b = 0; b < 5; ++b)
    for (int r = 0; r < reps[b]; ++r) {
      const std::string s = "conv" + std::to_string(b + 1) + "/conv" + std::to_string(b + 1) + "_" + std::to_string(r + 1);
      c->add_var(s + "/weights", {3, 3, cin, widths[b]});
      c->add_var(s + "/biases", {widths[b]});
      cin = widths[b];
    }
  c->add_var("conv6/weights", {3, 3, 512, 1024});
  c->add_var("conv6/biases", {1024});
  c->add_var("conv7/weights", {1, 1, 1024, 1024});
  c->add_var("conv7/biases", {1024});
  const int mid[5] = {256, 128, 128, 128, 128}, outc[5] = {512, 256, 256, 256, 256}, inc[5] = {1024, 512, 256, 256, 256};
  for (int b = 0; b < 5; ++b) {
    const std::string B = "block" + std::to_string(8 + b);
    c->add_var(B + "/conv1x1/weights", {1, 1, inc[b], mid[b]});
    c->add_var(B + "/conv1x1/biases", {mid[b]});
    const int k = b == 4 ? 4 : 3;
    c->add_var(B + (b == 4 ? "/conv4x4" : "/conv3x3") + "/weights", {k, k, mid[b], outc[b]});
    c->add_var(B + (b == 4 ? "/conv4x4" : "/conv3x3") + "/biases", {outc[b]});
  }
  for (int i = 0; i < 7; ++i) {
    const std::string L = std::string(kSsdFeat[i]) + "_box";
    if (i == 0) c->add_var(L + "/L2Normalization/gamma", {512});
    c->add_var(L + "/conv_loc/weights", {3, 3, kSsdFeatC[i], kSsdAnchors[i] * 4});
    c->add_var(L + "/conv_loc/biases", {kSsdAnchors[i] * 4});
    c->add_var(L + "/conv_cls/weights", {3, 3, kSsdFeatC[i], kSsdAnchors[i] * nc});
    c->add_var(L + "/conv_cls/biases", {kSsdAnchors[i] * nc});
  }
}

void declare_tensors_ssd(ron_ctx* c) {
  const int H = c->cfg.img_h, W = c->cfg.img_w;
  c->add_tensor("im2col", H, W, conv_k_chunk(c->cfg.dtype), 0);
  const int widths[5] = {64, 128, 256, 512, 512};
  const int reps[5] = {2, 2, 3, 3, 3};
  int h = H, w = W;
  for (int b = 0; b < 5; ++b) {
    for (int r = 0; r < reps[b]; ++r)
      c->add_tensor("conv" + std::to_string(b + 1) + "_" + std::to_string(r + 1), h, w, widths[b], 1);
    if (b < 4) { h /= 2; w /= 2; }
    c->add_tensor("pool" + std::to_string(b + 1), h, w, widths[b], b == 4 ? 6 : 1);   // pool5 feeds the rate-6 conv
  }
  c->add_tensor("block4_norm", H / 8, W / 8, 512, 1);
  c->add_tensor("conv6", h, w, 1024, 0);
  c->add_tensor("conv7", h, w, 1024, 1);
  const int mid[5] = {256, 128, 128, 128, 128}, outc[5] = {512, 256, 256, 256, 256};
  for (int b = 0; b < 5; ++b) {
    const std::string B = "block" + std::to_string(8 + b);
    c->add_tensor(B + "_mid", h, w, mid[b], 1);            // pad2d(1) of the reference = the halo
    if (b < 4) { h /= 2; w /= 2; } else { h = 1; w = 1; }
    c->add_tensor(B, h, w, outc[b], 1);
  }
  c->n_feat = 7;
  c->has_obj = false;
  const int fh[7] = {H / 8, H / 16, H / 32, H / 64, H / 128, H / 256, 1}, fw[7] = {W / 8, W / 16, W / 32, W / 64, W / 128, W / 256, 1};
  for (int i = 0; i < 7; ++i) { c->feat_h[i] = fh[i]; c->feat_w[i] = fw[i]; c->feat_A[i] = kSsdAnchors[i]; }
}

// SSDNet.default_params anchors (nets/ssd_vgg_512.py:79-102) with ssd_anchor_one_layer (:286-338): anchors per cell are
// [s0 square, sqrt(s0*s1) square, s0 at each ratio]; centres as in the RON version.
int make_anchors_ssd(ron_ctx* c) {
  const int H = c->cfg.img_h, W = c->cfg.img_w;
  const double sizes[7][2] = {{20.48, 51.2}, {51.2, 133.12}, {133.12, 215.04}, {215.04, 296.96}, {296.96, 378.88}, {378.88, 460.8}, {460.8, 542.72}};
  const double ratios[7][4] = {{2, .5, 0, 0}, {2, .5, 3, 1. / 3}, {2, .5, 3, 1. / 3}, {2, .5, 3, 1. / 3}, {2, .5, 3, 1. / 3}, {2, .5, 0, 0}, {2, .5, 0, 0}};
  const int n_ratios[7] = {2, 4, 4, 4, 4, 2, 2};
  const double steps[7] = {8, 16, 32, 64, 128, 256, 512};
  for (int i = 0; i < 7; ++i) {
    const int fh = c->feat_h[i], fw = c->feat_w[i], A = c->feat_A[i];
    std::vector<float> y(fh * fw), x(fh * fw), hh(A), ww(A);
    int rc = ron_ssd_anchor_one_layer(H, W, fh, fw, sizes[i], 2, ratios[i], n_ratios[i], steps[i], 0.5, y.data(), x.data(), hh.data(), ww.data());
    if (rc) return rc;
    const std::vector<float>* src[4] = {&y, &x, &hh, &ww};
    for (int k = 0; k < 4; ++k) {
      RON_HIP_CHECK(ron::dev_malloc((void**)&c->d_anchor[i][k], src[k]->size() * sizeof(float)));
      RON_HIP_CHECK(ron::dev_memcpy(c->d_anchor[i][k], src[k]->data(), src[k]->size() * sizeof(float), hipMemcpyHostToDevice));
    }
  }
  return RON_OK;
}

// Launch order of the heads with the small convolutions grouped (launch_conv_group).  The reference builds the scales one after the
// other (nets/ron_vgg_320.py:495-506); the true dependencies are listed at the RON tables below.  Every braced set only reads what
// earlier entries wrote, and its members write disjoint tensors / channel slices.  History of the measurements behind the plans:
// HISTORY.md (rounds 2-3: T64 / T128 groups by dependency level) and DESIGN.md 3.2 (round 4: mixed-width groups with carriers).
void plan_groups(ron_ctx* c) {
  if (c->cfg.flags & (RON_CFG_MULTI_STREAM | RON_CFG_NO_GROUPS)) return;
  struct Slot { int cfg; std::vector<const char*> names; };     // cfg < 0: launches of their own
  const int T64 = kCfgIgemm128x64;     // tiny convolutions (Npad = 64)
  // SSD-512 (nets/ssd_vgg_512.py:395-458): blocks 8-12 are a chain of 1x1 -> 3x3 stride-2 convolutions on 16x16 ... 1x1 maps,
  // each a 13-20 us launch at batch 16; the two box convolutions of a block only need that block's output, so they share a
  // launch with the next block's 1x1 (20 small launches -> 11).  The block4 / block7 heads are real work and stay alone (round 4:
  // as carriers of the chain's small launches in mixed-width groups they measured -0.8 % images/s - block4_box_conv_loc then leaves
  // the halo-patch kernel: 114 us for {block8_conv1x1, block4_box_conv_loc} where the two take 40 + 45 us on their own).
  const std::vector<Slot> ssd_order = {
      {-1, {"conv6"}}, {-1, {"conv7"}},
      {-1, {"block8_conv1x1"}}, {-1, {"block8_conv3x3"}},
      {T64, {"block8_box_conv_loc", "block8_box_conv_cls", "block9_conv1x1"}}, {-1, {"block9_conv3x3"}},
      {T64, {"block9_box_conv_loc", "block9_box_conv_cls", "block10_conv1x1"}}, {-1, {"block10_conv3x3"}},
      {T64, {"block10_box_conv_loc", "block10_box_conv_cls", "block11_conv1x1"}}, {-1, {"block11_conv3x3"}},
      {T64, {"block11_box_conv_loc", "block11_box_conv_cls", "block12_conv1x1"}}, {-1, {"block12_conv4x4"}},
      {T64, {"block12_box_conv_loc", "block12_box_conv_cls"}},
      {-1, {"block4_l2norm"}}, {-1, {"block4_box_conv_cls_loc"}}, {-1, {"block7_box_conv_cls_loc"}},
  };
  // RON heads.  Dependencies after the round-4 re-formulation of the reverse connection (the LEFT conv of a scale reads a backbone
  // map only; the transposed conv adds its half in place, ron_finalize_weights):
  //   conv_left(i)                       <- backbone (fc7 / fc6 / conv5_3 / conv4_3)        i = block7, 6, 5, 4
  //   deconv_right(i) -> ref(i)          <- ref(i-1), conv_left(i)
  //   trio3(i) -> hcat(i)                <- ref(i)
  //   {objectness_score, inception2, loc_pred}(i) <- hcat(i);   cls_pred(i) <- inception2(i)
  // so the coarse -> fine chain is ref7 -> deconv6 -> deconv5 -> deconv4 (three small GEMMs), and a dependency LEVEL is
  // {cls_pred(i-2), obj / inc2 / loc(i-1), trio3(i), deconv_right(i+1)}.  MIX = one launch of the row-gather kernel with the tile width
  // chosen per member (kGroupMixed): the latency-bound launches of the 5x5 / 10x10 scales (M = 800 / 3200 rows at batch 32: 18-50 us
  // apiece, mostly pipeline fill and split-K hand-off) ride in the partial rounds of the level's large member.  The members that are
  // several full rounds of 256 x 256 tiles on their own (block4: conv_left, trio3, inception2, cls_pred) keep launches of their own.
  const int MIX = kGroupMixed;
  // Batch plan.  G256 = one launch of 256 x 256 tiles (launch_conv_group, kCfgIgemm256): a member that is several rounds of the chip
  // on its own ends with a partial round (400 tiles = 1.56 rounds, 1200 = 4.69), and the members beside it - whose tiles have the same
  // or twice the K length - are dispatched into it (longest chains first); everything here has Npad % 256 == 0.  The skinny heads
  // (Npad = 64) cannot ride on 256-wide tiles: the ones of the three coarse scales share one mixed-width launch.
  const int G256 = kCfgIgemm256;
  const std::vector<Slot> ron_order = {
      // the left conv of block4 reads conv4_3 only: its 400 tiles (1.56 rounds) carry conv5_1's 100 (0.39 of a round on its own)
      {G256, {"conv5_1", "block4_conv_left"}},
      {-1, {"conv5_2"}}, {-1, {"conv5_3"}}, {-1, {"pool5"}}, {-1, {"fc6"}},
      // fc7 (208 tiles) and the left conv of block6 (26 tiles x 576 K steps: split-K) both read fc6 and are each short of one round.
      // (The left conv of block5 on the 48 CUs fc6 leaves idle - {fc6, block5_conv_left} - measured 531 us for the pair where fc6 alone
      // takes 513: -42 us per step with one batch in flight, but -0.85 % images/s with two, where the other batch uses those CUs.)
      {G256, {"fc7", "block6_conv_left"}},
      {MIX, {"block7_conv_left", "block5_conv_left"}},
      {MIX, {"block7_trio3", "block6_deconv_right"}},
      {MIX, {"block6_trio3", "block7_inception2", "block5_deconv_right"}},
      {G256, {"block5_trio3", "block6_inception2", "block7_cls_pred", "block4_deconv_right"}},
      {MIX, {"block7_objectness_score", "block7_loc_pred", "block6_objectness_score", "block6_loc_pred", "block5_objectness_score",
             "block5_loc_pred"}},
      {G256, {"block4_trio3", "block5_inception2", "block6_cls_pred"}},
      // Cout = 20 / 40 over channel slices of the same tensor: one launch of the halo-patch kernel (400 workgroups; 200 each alone)
      {kCfgPatch64, {"block4_objectness_score", "block4_loc_pred"}},
      {G256, {"block4_inception2", "block5_cls_pred"}},
      {-1, {"block4_cls_pred"}},
  };
  // Medium batches (max_batch kLevelPlanMaxBatch + 1 .. kCarrierPlanMinBatch - 1): the large members are not several rounds of 256 x 256 tiles there, so a
  // dependency level is one mixed-width launch (128-row tiles), the free left convs being the carriers of the first levels
  // (batch 8 / 16, one in flight: 2.02 -> 1.94 ms, 3.18 -> 3.08 ms; the 256 x 256 groups above measured 2.08 / 3.23 there).
  const std::vector<Slot> ron_mid = {
      {G256, {"fc7", "block6_conv_left"}},
      {MIX, {"block7_conv_left", "block5_conv_left"}},
      {MIX, {"block7_trio3", "block6_deconv_right"}},
      {MIX, {"block7_objectness_score", "block7_inception2", "block7_loc_pred", "block6_trio3", "block5_deconv_right"}},
      {-1, {"block4_conv_left"}},
      {MIX, {"block7_cls_pred", "block6_objectness_score", "block6_inception2", "block6_loc_pred", "block5_trio3", "block4_deconv_right"}},
      {MIX, {"block6_cls_pred", "block5_objectness_score", "block5_loc_pred", "block5_inception2"}},
      {-1, {"block4_trio3"}},
      {-1, {"block5_cls_pred"}},
      {kCfgPatch64, {"block4_objectness_score", "block4_loc_pred"}},
      {-1, {"block4_inception2"}},
      {-1, {"block4_cls_pred"}},
  };
  // Small batches (RON_CFG_LEVEL_GROUPS, the default when max_batch <= kLevelPlanMaxBatch): every head convolution is a
  // latency-bound launch of a few hundred workgroups (25-40 us each with its split-K finalize, whatever its size), so
  // the heads go out one launch per dependency level, the large convolutions included: 16 head launches -> 7.  Since the left
  // convs left the chain (round 4) this wins up to batch 12 with one or two batches in flight (batch 8: 1.76 vs 1.91 ms, batch 12:
  // 2.44 vs 2.64 ms); from 16 on two batches in flight prefer the plans above (2.47 vs 2.56 ms), one in flight still this one.
  const std::vector<Slot> ron_levels = {
      {MIX, {"block7_conv_left", "block6_conv_left", "block5_conv_left", "block4_conv_left"}},
      {MIX, {"block7_trio3", "block6_deconv_right"}},
      {MIX, {"block7_objectness_score", "block7_inception2", "block7_loc_pred", "block6_trio3", "block5_deconv_right"}},
      {MIX, {"block7_cls_pred", "block6_objectness_score", "block6_inception2", "block6_loc_pred", "block5_trio3", "block4_deconv_right"}},
      {MIX, {"block6_cls_pred", "block5_objectness_score", "block5_inception2", "block5_loc_pred", "block4_trio3"}},
      {MIX, {"block5_cls_pred", "block4_objectness_score", "block4_inception2", "block4_loc_pred"}},
      {-1, {"block4_cls_pred"}},
  };
  const bool levels = !(c->cfg.flags & RON_CFG_BATCH_GROUPS) &&
                      ((c->cfg.flags & RON_CFG_LEVEL_GROUPS) || c->cfg.max_batch <= kLevelPlanMaxBatch);
  const std::vector<Slot>& order = c->is_ssd() ? ssd_order : (levels ? ron_levels : (c->cfg.max_batch >= kCarrierPlanMinBatch ? ron_order : ron_mid));
  std::map<std::string, int> at;
  for (size_t i = 0; i < c->ops.size(); ++i) at[c->ops[i].name] = (int)i;
  size_t first_head = c->ops.size(), n_named = 0;
  for (const Slot& s : order)
    for (const char* nm : s.names) {
      auto it = at.find(nm);
      if (it == at.end()) {                               // not the graph this plan was written for: keep the plain order, loudly
        fprintf(stderr, "libron_hip: grouped launch plan names op '%s' which the graph does not have: one launch per convolution\n", nm);
        return;
      }
      first_head = std::min(first_head, (size_t)it->second);
      ++n_named;
    }
  if (first_head + n_named != c->ops.size()) {            // the heads must be exactly the tail of the op list
    fprintf(stderr, "libron_hip: grouped launch plan covers %zu ops, the graph has %zu after '%s': one launch per convolution\n",
            n_named, c->ops.size() - first_head, c->ops[first_head].name.c_str());
    return;
  }
  std::vector<Op> planned(c->ops.begin(), c->ops.begin() + first_head);
  int gid = 0;
  for (const Slot& s : order) {
    for (const char* nm : s.names) {
      Op o = c->ops[at[nm]];
      if (s.cfg >= 0) { o.group = gid; o.group_cfg = s.cfg; }
      planned.push_back(o);
    }
    if (s.cfg >= 0) ++gid;
  }
  c->grouped_launches = gid;
  c->ops.swap(planned);
}

Op conv_op(const std::string& name, int in, int out, int packed, int k, int cpad, int relu, int Ho, int Wo) {
  Op o;
  o.kind = OP_CONV; o.name = name; o.in = in; o.out = out; o.packed = packed;
  o.kh = o.kw = k; o.cpad = cpad; o.relu = relu; o.Ho = Ho; o.Wo = Wo;
  return o;
}

double conv_flops(const Var& w, int out_pixels) { return 2.0 * (double)w.numel() * out_pixels; }

}  // namespace

// ------------------------------------------------------------------------------------------
extern "C" int ron_create(ron_ctx** out, const ron_config* cfg) {
  RON_REQUIRE(out && cfg, "NULL argument");
  RON_REQUIRE(cfg->variant >= RON_VARIANT_REDUCEDFC && cfg->variant <= RON_VARIANT_SSD512, "unknown variant %d", cfg->variant);
  RON_REQUIRE(cfg->dtype >= 0 && cfg->dtype <= RON_DTYPE_F16X3, "unknown dtype %d", cfg->dtype);
  RON_REQUIRE(cfg->img_h > 0 && cfg->img_h % 64 == 0 && cfg->img_w > 0 && cfg->img_w % 64 == 0, "image size must be a multiple of 64");
  if (cfg->variant == RON_VARIANT_SSD512) RON_REQUIRE(cfg->img_h == 512 && cfg->img_w == 512, "SSD-512 runs on 512 x 512 inputs");
  RON_REQUIRE(cfg->num_classes >= 2 && cfg->num_classes <= RON_MAX_CLASSES, "num_classes %d out of range [2, %d]", cfg->num_classes, RON_MAX_CLASSES);
  RON_REQUIRE(cfg->max_batch >= 1, "max_batch must be >= 1");
  RON_HIP_CHECK(ron::dev_set_device(cfg->device));
  std::unique_ptr<ron_ctx> c(new ron_ctx());
  c->cfg = *cfg;
  c->c6 = cfg->variant == RON_VARIANT_FULL ? 4096 : 1024;
  const int H = cfg->img_h, W = cfg->img_w;
  if (c->is_ssd()) {
    declare_variables_ssd(c.get());
    declare_tensors_ssd(c.get());
  } else {
  declare_variables(c.get());

  // ---- tensors ----
  const int chunk = conv_k_chunk(cfg->dtype);
  c->add_tensor("im2col", H, W, chunk, 0);
  const int widths[5] = {64, 128, 256, 512, 512};
  const int reps[5] = {2, 2, 3, 3, 3};
  int h = H, w = W;
  for (int b = 0; b < 5; ++b) {
    for (int r = 0; r < reps[b]; ++r) {
      const bool last = r == reps[b] - 1;
      // the block output feeds the pool (no halo needed); block4/block5 also feed a 3x3 left conv
      const int pad = (!last || b >= 3) ? 1 : 0;
      c->add_tensor("conv" + std::to_string(b + 1) + "_" + std::to_string(r + 1), h, w, widths[b], pad);
    }
    h /= 2; w /= 2;
    c->add_tensor("pool" + std::to_string(b + 1), h, w, widths[b], b == 4 ? 3 : 1);
  }
  c->add_tensor("fc6", h, w, c->c6, 1);
  c->add_tensor("fc7", h, w, c->c6, 0);
  for (int i = 0; i < 4; ++i) {
    const int s_h = (H / 64) << i, s_w = (W / 64) << i;
    c->feat[i] = s_h;
    c->feat_h[i] = s_h; c->feat_w[i] = s_w; c->feat_A[i] = c->num_anchors;
    const std::string L = kFeatLayers[i];
    c->add_tensor(L + "_ref", s_h, s_w, 512, 1);
    c->add_tensor(L + "_hcat", s_h, s_w, 2048, 1);
    c->add_tensor(L + "_inc2", s_h, s_w, 1024, 1);
  }
  }
  for (auto& t : c->tensors) {
    if (t.name == "im2col" && cfg->dtype != RON_DTYPE_F32) continue;      // bf16 / f16 / f16x3 use the stem kernel
    if ((cfg->flags & RON_CFG_FUSE_POOLS) && (t.name == "conv1_2" || t.name == "conv2_2" || t.name == "conv3_3")) continue;
    if ((cfg->flags & RON_CFG_FUSE_POOLS) && !(cfg->flags & RON_CFG_NO_STEM2) && dtype_is_half(cfg->dtype) && H % 8 == 0 &&
        W % 32 == 0 && t.name == "conv1_1") continue;         // conv1_1 + conv1_2 + pool1 run fused (stem2_kernel)
    t.bytes = TensorView::halo_pixels(cfg->max_batch, t.H, t.W, t.pad) * t.cstride * c->esz();     // shared halos, conv_mfma.h
    if (t.bytes >= ((int64_t)1 << 32)) {
      ron::set_error("tensor %s needs %lld bytes for max_batch %d: above the 4 GiB buffer-addressing limit; lower max_batch",
                     t.name.c_str(), (long long)t.bytes, cfg->max_batch);
      for (auto& u : c->tensors) if (u.d) (void)ron::dev_free(u.d);
      return RON_ERR_INVALID;
    }
    RON_HIP_CHECK(ron::dev_malloc(&t.d, (size_t)t.bytes));
    RON_HIP_CHECK(ron::dev_memset(t.d, 0, (size_t)t.bytes));     // halos stay zero forever: kernels write interiors only
  }
  // ---- anchors (RONNet.default_params, nets/ron_vgg_320.py:97-124) ----
  if (c->is_ssd()) {
    int rc = make_anchors_ssd(c.get());
    if (rc) return rc;
    *out = c.release();
    return RON_OK;
  }
  const double sizes[4][2] = {{224., 256.}, {160., 192.}, {96., 128.}, {32., 64.}};
  const double ratios[5] = {1., 2., 3., 1. / 2, 1. / 3};
  const double steps[4] = {64, 32, 16, 8};
  for (int i = 0; i < 4; ++i) {
    const int fh = (H / 64) << i, fw = (W / 64) << i;
    std::vector<float> y(fh * fw), x(fh * fw), hh(10), ww(10);
    int rc = ron_anchor_one_layer(H, W, fh, fw, sizes[i], 2, ratios, 5, steps[i], 0.5, y.data(), x.data(), hh.data(), ww.data());
    if (rc) return rc;
    const std::vector<float>* src[4] = {&y, &x, &hh, &ww};
    for (int k = 0; k < 4; ++k) {
      RON_HIP_CHECK(ron::dev_malloc((void**)&c->d_anchor[i][k], src[k]->size() * sizeof(float)));
      RON_HIP_CHECK(ron::dev_memcpy(c->d_anchor[i][k], src[k]->data(), src[k]->size() * sizeof(float), hipMemcpyHostToDevice));
    }
  }
  *out = c.release();
  return RON_OK;
}

extern "C" int ron_destroy(ron_ctx* c) {
  if (!c) return RON_OK;
  if (c->clones > 0) { ron::set_error("ron_destroy: %d execution slot(s) still borrow this context's weights", c->clones); return RON_ERR_STATE; }
  const bool borrowed = c->weights_owner != nullptr;
  if (borrowed) {                        // the weights belong to the owner
    --c->weights_owner->clones;
    c->packed.clear();
    c->d_l2_gamma = nullptr; c->d_stem_w = nullptr; c->d_stem_b = nullptr; c->d_stem2_w = nullptr; c->d_stem2_b = nullptr; c->d_stem2_w1 = nullptr;
  }
  for (auto& t : c->tensors) if (t.d) (void)ron::dev_free(t.d);
  for (auto& p : c->packed) { if (p.d_w) (void)ron::dev_free(p.d_w); if (p.d_w_c64) (void)ron::dev_free(p.d_w_c64); if (p.d_bias) (void)ron::dev_free(p.d_bias); }
  for (int i = 0; i < RON_MAX_LAYERS; ++i) for (int k = 0; k < 4; ++k) if (c->d_anchor[i][k]) (void)ron::dev_free(c->d_anchor[i][k]);
  for (int k = 0; k < 3; ++k) for (int i = 0; i < RON_MAX_LAYERS; ++i) if (c->d_head[k][i]) (void)ron::dev_free(c->d_head[k][i]);
  if (c->d_l2_gamma) (void)ron::dev_free(c->d_l2_gamma);
  for (auto& call : c->pending) for (hipEvent_t e : call) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  if (c->d_post_ws) (void)ron::dev_free(c->d_post_ws);
  for (int l = 0; l < 4; ++l) {
    if (c->d_splitk[l]) (void)ron::dev_free(c->d_splitk[l]);
    if (c->side[l]) (void)hipStreamDestroy(c->side[l]);
    if (c->lane_ready[l]) (void)hipEventDestroy(c->lane_ready[l]);
    if (c->lane_done[l]) (void)hipEventDestroy(c->lane_done[l]);
  }
  if (c->d_stem_w) (void)ron::dev_free(c->d_stem_w);
  if (c->d_stem_b) (void)ron::dev_free(c->d_stem_b);
  if (c->d_stem2_w) (void)ron::dev_free(c->d_stem2_w);
  if (c->d_stem2_b) (void)ron::dev_free(c->d_stem2_b);
  if (c->d_stem2_w1) (void)ron::dev_free(c->d_stem2_w1);
  delete c;
  return RON_OK;
}

extern "C" int ron_num_variables(const ron_ctx* c) { return c ? (int)c->vars.size() : RON_ERR_INVALID; }

extern "C" int ron_variable_info(const ron_ctx* c, int i, const char** name, int64_t shape[4], int* ndim) {
  RON_REQUIRE(c && i >= 0 && i < (int)c->vars.size(), "variable index out of range");
  if (name) *name = c->vars[i].name.c_str();
  if (ndim) *ndim = (int)c->vars[i].shape.size();
  if (shape) for (size_t k = 0; k < c->vars[i].shape.size(); ++k) shape[k] = c->vars[i].shape[k];
  return RON_OK;
}

extern "C" int ron_load_weight(ron_ctx* c, const char* tf_name, const float* host_ptr, const int64_t* shape, int ndim) {
  RON_REQUIRE(c && tf_name && host_ptr && shape, "NULL argument");
  if (c->finalized) { ron::set_error("weights are already finalized"); return RON_ERR_STATE; }
  auto it = c->var_index.find(tf_name);
  if (it == c->var_index.end()) { ron::set_error("unknown variable '%s'", tf_name); return RON_ERR_UNKNOWN_NAME; }
  Var& v = c->vars[it->second];
  bool same = ndim == (int)v.shape.size();
  for (int k = 0; same && k < ndim; ++k) same = shape[k] == v.shape[k];
  if (!same) { ron::set_error("variable '%s': shape mismatch", tf_name); return RON_ERR_INVALID; }
  v.data.assign(host_ptr, host_ptr + v.numel());
  v.loaded = true;
  return RON_OK;
}

static int slot_resources(ron_ctx* c);

extern "C" int ron_finalize_weights(ron_ctx* c) {
  RON_REQUIRE(c, "NULL ctx");
  if (c->finalized) { ron::set_error("weights are already finalized"); return RON_ERR_STATE; }
  for (auto& v : c->vars)
    if (!v.loaded) { ron::set_error("variable '%s' was not loaded", v.name.c_str()); return RON_ERR_STATE; }
  RON_HIP_CHECK(ron::dev_set_device(c->cfg.device));
  const int H = c->cfg.img_h, W = c->cfg.img_w;
  auto T = [&](const std::string& n) { return c->tensor_index.at(n); };
  double flops = 0, mark = 0;
  int rc;
#define ATTR() do { c->ops.back().flops += flops - mark; mark = flops; } while (0)
#define PACK(expr) do { rc = (expr); if (rc < 0) return rc; } while (0)
  // ---- VGG-16 body ----
  const bool use_stem = dtype_is_half(c->cfg.dtype) || c->cfg.dtype == RON_DTYPE_F16X3;      // fp32: conv1_1 as im2col + GEMM
  if (!use_stem) {
    Op o; o.kind = OP_IM2COL; o.name = "im2col"; o.out = T("im2col");
    c->ops.push_back(o);
  }
  const int reps[5] = {2, 2, 3, 3, 3};
  int h = H, w = W, prev = T("im2col");
  for (int b = 0; b < 5; ++b) {
    for (int r = 0; r < reps[b]; ++r) {
      const std::string nm = "conv" + std::to_string(b + 1) + "_" + std::to_string(r + 1);
      const std::string scope = "conv" + std::to_string(b + 1) + "/" + nm;
      const bool stem = b == 0 && r == 0;
      if (stem && use_stem) {
        std::vector<uint16_t> frags;
        if (c->cfg.dtype == RON_DTYPE_F16X3) c->stem_oscale = stem_pack_weights_split(c->var(scope + "/weights").data.data(), &frags);
        else stem_pack_weights(c->var(scope + "/weights").data.data(), c->cfg.dtype, &frags);
        RON_HIP_CHECK(ron::dev_malloc(&c->d_stem_w, frags.size() * 2));
        RON_HIP_CHECK(ron::dev_memcpy(c->d_stem_w, frags.data(), frags.size() * 2, hipMemcpyHostToDevice));
        RON_HIP_CHECK(ron::dev_malloc((void**)&c->d_stem_b, 64 * sizeof(float)));
        RON_HIP_CHECK(ron::dev_memcpy(c->d_stem_b, c->var(scope + "/biases").data.data(), 64 * sizeof(float), hipMemcpyHostToDevice));
        Op o; o.kind = OP_STEM; o.name = nm; o.out = T(nm);
        c->ops.push_back(o);
      } else {
        PACK(stem ? pack_stem(c, scope) : pack_plain(c, scope, false));
        c->ops.push_back(conv_op(nm, prev, T(nm), rc, stem ? 1 : 3, stem ? 0 : 1, 1, h, w));
      }
      flops += conv_flops(c->var(scope + "/weights"), h * w); ATTR();
      prev = T(nm);
    }
    const std::string pname = "pool" + std::to_string(b + 1);
    if (c->is_ssd() && b == 4) {              // SSD: pool5 is 3x3 stride 1 (nets/ssd_vgg_512.py:391)
      Op p; p.kind = OP_POOL3; p.name = pname; p.in = prev; p.out = T(p.name);
      c->ops.push_back(p);
      prev = T(pname);
      continue;
    }
    if ((c->cfg.flags & RON_CFG_FUSE_POOLS) && b < 3 && c->ops.back().kind == OP_CONV) {
      c->ops.back().pool = 1;                 // block1..3 feed nothing but their pool: never written at full size
      c->ops.back().out = T(pname);
      c->ops.back().name += "+" + pname;
      const size_t n_ops = c->ops.size();
      if (b == 0 && dtype_is_half(c->cfg.dtype) && n_ops >= 2 && c->ops[n_ops - 2].kind == OP_STEM && H % 8 == 0 && W % 32 == 0 &&
          !(c->cfg.flags & RON_CFG_NO_STEM2)) {
        // conv1_1 + conv1_2 + pool1 as one kernel (stem.hip): neither full-resolution 64-channel map touches HBM
        std::vector<uint16_t> img;
        stem2_pack_weights(c->var("conv1/conv1_2/weights").data.data(), c->cfg.dtype, &img);
        RON_HIP_CHECK(ron::dev_malloc(&c->d_stem2_w, img.size() * 2));
        RON_HIP_CHECK(ron::dev_memcpy(c->d_stem2_w, img.data(), img.size() * 2, hipMemcpyHostToDevice));
        RON_HIP_CHECK(ron::dev_malloc((void**)&c->d_stem2_b, 64 * sizeof(float)));
        RON_HIP_CHECK(ron::dev_memcpy(c->d_stem2_b, c->var("conv1/conv1_2/biases").data.data(), 64 * sizeof(float), hipMemcpyHostToDevice));
        // conv1_1 as 16x16x32 fragments for the fused kernel (the stand-alone stem kernel keeps its 32x32 fragments in d_stem_w)
        stem2_pack_w1(c->var("conv1/conv1_1/weights").data.data(), c->cfg.dtype, &img);
        RON_HIP_CHECK(ron::dev_malloc(&c->d_stem2_w1, img.size() * 2));
        RON_HIP_CHECK(ron::dev_memcpy(c->d_stem2_w1, img.data(), img.size() * 2, hipMemcpyHostToDevice));
        Op f; f.kind = OP_STEM2; f.name = "conv1_1+conv1_2+pool1"; f.out = T(pname);
        f.flops = c->ops[n_ops - 2].flops + c->ops[n_ops - 1].flops;
        c->ops.pop_back(); c->ops.pop_back();
        c->ops.push_back(f);
      }
    } else {
      if ((c->cfg.flags & RON_CFG_FUSE_POOLS) && c->ops.back().kind == OP_CONV) c->ops.back().fuse_next_pool = 1;
      Op p; p.kind = OP_POOL; p.name = pname; p.in = prev; p.out = T(p.name);
      c->ops.push_back(p);
    }
    prev = T(pname);
    h /= 2; w /= 2;
  }
  if (c->is_ssd()) {
    // ---- conv6 (3x3 rate 6), conv7 (1x1), blocks 8-12 (1x1 then pad2d(1) + 3x3 stride 2 VALID; block12: 4x4 VALID) ----
    PACK(pack_plain(c, "conv6", false));
    { Op o = conv_op("conv6", prev, T("conv6"), rc, 3, 6, 1, h, w); o.dil = 6; c->ops.push_back(o); }
    flops += conv_flops(c->var("conv6/weights"), h * w); ATTR();
    PACK(pack_plain(c, "conv7", false));
    c->ops.push_back(conv_op("conv7", T("conv6"), T("conv7"), rc, 1, 0, 1, h, w));
    flops += conv_flops(c->var("conv7/weights"), h * w); ATTR();
    int src = T("conv7");
    for (int b = 0; b < 5; ++b) {
      const std::string B = "block" + std::to_string(8 + b);
      PACK(pack_plain(c, B + "/conv1x1", false));
      c->ops.push_back(conv_op(B + "_conv1x1", src, T(B + "_mid"), rc, 1, 0, 1, h, w));
      flops += conv_flops(c->var(B + "/conv1x1/weights"), h * w); ATTR();
      const std::string cs = B + (b == 4 ? "/conv4x4" : "/conv3x3");
      PACK(pack_plain(c, cs, false));
      const int ho = b == 4 ? 1 : h / 2, wo = b == 4 ? 1 : w / 2;
      Op o = conv_op(B + (b == 4 ? "_conv4x4" : "_conv3x3"), T(B + "_mid"), T(B), rc, b == 4 ? 4 : 3, 1, 1, ho, wo);
      o.stride = b == 4 ? 1 : 2;
      c->ops.push_back(o);
      flops += conv_flops(c->var(cs + "/weights"), ho * wo); ATTR();
      src = T(B); h = ho; w = wo;
    }
    // ---- multibox heads (nets/ssd_vgg_300.py:403-431) ----
    {
      const Var& g = c->var("block4_box/L2Normalization/gamma");
      RON_HIP_CHECK(ron::dev_malloc((void**)&c->d_l2_gamma, g.data.size() * sizeof(float)));
      RON_HIP_CHECK(ron::dev_memcpy(c->d_l2_gamma, g.data.data(), g.data.size() * sizeof(float), hipMemcpyHostToDevice));
      Op o; o.kind = OP_L2NORM; o.name = "block4_l2norm"; o.in = T("conv4_3"); o.out = T("block4_norm");
      c->ops.push_back(o);
    }
    const char* feat_src[7] = {"block4_norm", "conv7", "block8", "block9", "block10", "block11", "block12"};
    for (int i = 0; i < 7; ++i) {
      const std::string L = std::string(kSsdFeat[i]) + "_box";
      const int fh = c->feat_h[i], fw = c->feat_w[i];
      if (i < kSsdPairedHeads) {
        // the two large maps: loc and cls as one launch with two outputs (pack_box_pair)
        PACK(pack_box_pair(c, L));
        Op o = conv_op(L + "_conv_cls_loc", T(feat_src[i]), -2, rc, 3, 1, 0, fh, fw);
        o.head_kind = 0; o.head_kind2 = 2; o.head_layer = i;
        c->ops.push_back(o);
        flops += conv_flops(c->var(L + "/conv_loc/weights"), fh * fw) + conv_flops(c->var(L + "/conv_cls/weights"), fh * fw); ATTR();
        continue;
      }
      PACK(pack_plain(c, L + "/conv_loc", false));
      { Op o = conv_op(L + "_conv_loc", T(feat_src[i]), -2, rc, 3, 1, 0, fh, fw); o.head_kind = 2; o.head_layer = i; c->ops.push_back(o); }
      flops += conv_flops(c->var(L + "/conv_loc/weights"), fh * fw); ATTR();
      PACK(pack_plain(c, L + "/conv_cls", false));
      { Op o = conv_op(L + "_conv_cls", T(feat_src[i]), -2, rc, 3, 1, 0, fh, fw); o.head_kind = 0; o.head_layer = i; c->ops.push_back(o); }
      flops += conv_flops(c->var(L + "/conv_cls/weights"), fh * fw); ATTR();
    }
  } else {
  // ---- fc6 / fc7 ----
  PACK(pack_plain(c, "fc6", false));
  {
    Op o = c->cfg.variant == RON_VARIANT_FULL ? conv_op("fc6", prev, T("fc6"), rc, 7, 3, 1, h, w)
                                               : conv_op("fc6", prev, T("fc6"), rc, 3, 3, 1, h, w);
    if (c->cfg.variant != RON_VARIANT_FULL) o.dil = 3;
    c->ops.push_back(o);
    flops += conv_flops(c->var("fc6/weights"), h * w); ATTR();
  }
  PACK(pack_plain(c, "fc7", false));
  c->ops.push_back(conv_op("fc7", T("fc6"), T("fc7"), rc, 1, 0, 1, h, w));
  flops += conv_flops(c->var("fc7/weights"), h * w); ATTR();
  // ---- reverse connections + heads, coarse -> fine ----
  const char* left_src[4] = {"fc7", "fc6", "conv5_3", "conv4_3"};
  for (int i = 0; i < 4; ++i) {
    const std::string Ln = kFeatLayers[i];
    const std::string L = "reverse_module/" + Ln + "_reverse";
    const int sh = c->feat[i], sw = (W / 64) << i;
    if (i == 0) {
      PACK(pack_plain(c, L + "_conv_left", true));
      Op o = conv_op(Ln + "_conv_left", T(left_src[i]), T(Ln + "_ref"), rc, 2, 0, 1, sh, sw);
      o.stride = 2;
      c->ops.push_back(o);
      flops += conv_flops(c->var(L + "_conv_left/weights"), sh * sw); ATTR();
    } else {
      // relu(relu(BN(conv_left(backbone map))) + relu(deconv_right(coarser reference map) + b))  (nets/ron_vgg_320.py:420-425).
      // The LEFT conv writes its half into the reference map's tensor first - it reads a backbone map only, so it is OFF the
      // coarse -> fine chain and free to be launched early / beside anything - and the transposed conv, the cheap one that IS on the
      // chain, adds its half in place (pixel-shuffle epilogue with the residual at the same address).
      PACK(pack_plain(c, L + "_conv_left", true));
      c->ops.push_back(conv_op(Ln + "_conv_left", T(left_src[i]), T(Ln + "_ref"), rc, 3, 1, 1, sh, sw));
      flops += conv_flops(c->var(L + "_conv_left/weights"), sh * sw); ATTR();
      PACK(pack_deconv(c, L + "_deconv_right"));
      Op d = conv_op(Ln + "_deconv_right", T(std::string(kFeatLayers[i - 1]) + "_ref"), T(Ln + "_ref"), rc, 1, 0, 1, sh / 2, sw / 2);
      d.up = 2; d.up_cout = 512;
      d.res = T(Ln + "_ref");                 // in place: ref = relu(left + up)
      c->ops.push_back(d);
      flops += conv_flops(c->var(L + "_deconv_right/weights"), (sh / 2) * (sw / 2)); ATTR();
    }
    // hcat channels: [0,512) objectness hidden | [512,1024) box hidden | [1024,1536) inception-1 3x3 | [1536,2048) inception-1 1x1
    PACK(pack_trio3(c, L));
    {
      Op o = conv_op(Ln + "_trio3", T(Ln + "_ref"), T(Ln + "_hcat"), rc, 3, 1, 1, sh, sw);
      o.center_from = 1536;
      c->ops.push_back(o);
      flops += conv_flops(c->var(L + "_objectness/weights"), sh * sw) + conv_flops(c->var(L + "/Conv2d_0_3x3/weights"), sh * sw) +
               conv_flops(c->var(L + "_inception1/Branch_0/Conv2d_3x3/weights"), sh * sw) +
               conv_flops(c->var(L + "_inception1/Branch_1/Conv2d_1x1/weights"), sh * sw); ATTR();
    }
    PACK(pack_plain(c, L + "_objectness_score", false));
    {
      Op o = conv_op(Ln + "_objectness_score", T(Ln + "_hcat"), -2, rc, 3, 1, 0, sh, sw);
      o.in_coff = 0; o.in_C = 512; o.head_kind = 1; o.head_layer = i;
      c->ops.push_back(o);
      flops += conv_flops(c->var(L + "_objectness_score/weights"), sh * sw); ATTR();
    }
    PACK(pack_inception(c, L + "_inception2"));
    {
      Op o = conv_op(Ln + "_inception2", T(Ln + "_hcat"), T(Ln + "_inc2"), rc, 3, 1, 1, sh, sw);
      o.in_coff = 1024; o.in_C = 1024; o.center_from = 512;
      c->ops.push_back(o);
      flops += conv_flops(c->var(L + "_inception2/Branch_0/Conv2d_3x3/weights"), sh * sw) +
               conv_flops(c->var(L + "_inception2/Branch_1/Conv2d_1x1/weights"), sh * sw); ATTR();
    }
    PACK(pack_plain(c, L + "_inception2/Conv2d_pred_3x3", false));
    {
      Op o = conv_op(Ln + "_cls_pred", T(Ln + "_inc2"), -2, rc, 3, 1, 0, sh, sw);
      o.head_kind = 0; o.head_layer = i;
      c->ops.push_back(o);
      flops += conv_flops(c->var(L + "_inception2/Conv2d_pred_3x3/weights"), sh * sw); ATTR();
    }
    PACK(pack_plain(c, L + "/Conv2d_1_3x3", false));
    {
      Op o = conv_op(Ln + "_loc_pred", T(Ln + "_hcat"), -2, rc, 3, 1, 0, sh, sw);
      o.in_coff = 512; o.in_C = 512; o.head_kind = 2; o.head_layer = i;
      c->ops.push_back(o);
      flops += conv_flops(c->var(L + "/Conv2d_1_3x3/weights"), sh * sw); ATTR();
    }
  }
  }   // RON tail
#undef PACK
#undef ATTR
  c->flops_per_image = flops;
  plan_groups(c);
  // stream lanes: heads of block7 / block6 / block5 are independent of the main chain once their reference map exists
  if ((c->cfg.flags & RON_CFG_MULTI_STREAM) && !c->is_ssd()) {
    for (Op& o : c->ops)
      for (int i = 0; i < 3; ++i) {
        const std::string pre = std::string(kFeatLayers[i]) + "_";
        if (o.name.compare(0, pre.size(), pre) == 0 && o.name.find("_conv_left") == std::string::npos &&
            o.name.find("_deconv_right") == std::string::npos)
          o.lane = i + 1;
      }
  }
  for (Op& o : c->ops) {
    if (o.kind != OP_CONV) continue;
    const PackedConv& pk = c->packed[o.packed];
    const Tensor& ti = c->tensors[o.in];
    const int cin = o.in_C > 0 ? o.in_C : ti.C;
    const double out_esz = o.out == -2 ? 4.0 : (double)c->esz();
    const int os = o.up > 0 ? o.up * o.up : 1;
    const double out_px = o.pool ? (double)o.Ho * o.Wo / 4 : (double)o.Ho * o.Wo * os;
    const double out_ch = o.up > 0 ? (double)o.up_cout : (double)(pk.Cout - (pk.split_n - pk.split_first));
    o.act_bytes = (double)ti.H * ti.W * cin * c->esz() + out_px * out_ch * out_esz + (o.res >= 0 ? out_px * out_ch * c->esz() : 0.0);
    o.wgt_bytes = (double)pk.w_bytes;
  }
  for (auto& v : c->vars) { v.data.clear(); v.data.shrink_to_fit(); }
  if ((rc = slot_resources(c))) return rc;
  c->finalized = true;
  return RON_OK;
}

// The launch description of conv op `o` at batch n (heads == nullptr: geometry only, for sizing).
static int describe_conv(const ron_ctx* c, const Op& o, int n, const ron_heads* heads, ConvLaunch* out_l) {
  const PackedConv& p = c->packed[o.packed];
  ConvLaunch L;
  L.dtype = c->cfg.dtype;
  L.in = c->view(o.in, n, o.in_coff, o.in_C > 0 ? o.in_C : -1);
  if (o.out == -2) {
    float* dst = nullptr;
    if (heads != nullptr) {
      const float* const* arr = o.head_kind == 0 ? heads->cls : (o.head_kind == 1 ? heads->obj : heads->loc);
      dst = const_cast<float*>(arr[o.head_layer]);
      RON_REQUIRE(dst != nullptr, "ron_forward: head buffer (kind %d, layer %d) is NULL", o.head_kind, o.head_layer);
    }
    TensorView v;
    const int A = c->feat_A[o.head_layer];
    v.base = dst; v.N = n; v.H = o.Ho; v.W = o.Wo; v.pad = 0; v.coff = 0;
    v.C = o.head_kind == 0 ? A * c->cfg.num_classes : (o.head_kind == 1 ? 2 * A : 4 * A);
    v.cstride = v.C;
    v.bytes = (int64_t)n * v.H * v.W * v.C * 4;
    L.out = v;
    L.out_f32 = 1;
    if (o.head_kind2 >= 0) {
      // second head output of the launch (pack_box_pair): dense like the first
      float* dst2 = nullptr;
      if (heads != nullptr) {
        const float* const* arr2 = o.head_kind2 == 0 ? heads->cls : (o.head_kind2 == 1 ? heads->obj : heads->loc);
        dst2 = const_cast<float*>(arr2[o.head_layer]);
        RON_REQUIRE(dst2 != nullptr, "ron_forward: head buffer (kind %d, layer %d) is NULL", o.head_kind2, o.head_layer);
      }
      TensorView v2 = v;
      v2.base = dst2 != nullptr ? dst2 : reinterpret_cast<float*>(16);        // (geometry only: any non-null address)
      v2.C = o.head_kind2 == 0 ? A * c->cfg.num_classes : (o.head_kind2 == 1 ? 2 * A : 4 * A);
      v2.cstride = v2.C;
      v2.bytes = (int64_t)n * v2.H * v2.W * v2.C * 4;
      L.out2 = v2;
      L.split_n = p.split_n; L.split_first = p.split_first;
    }
  } else {
    L.out = c->view(o.out, n, o.out_coff, o.out_C > 0 ? o.out_C : -1);
  }
  L.res = o.res >= 0 ? c->tensors[o.res].d : nullptr;
  L.wgt = p.d_w; L.wgt_bytes = p.w_bytes; L.wgt_c64 = p.d_w_c64; L.bias = p.d_bias; L.oscale = p.oscale; L.Cout = p.Cout; L.Npad = p.Npad;
  L.kh = o.kh; L.kw = o.kw; L.stride = o.stride; L.dil = o.dil; L.cpad = o.cpad; L.relu = o.relu;
  L.up = o.up; L.up_cout = o.up_cout; L.Ho = o.Ho; L.Wo = o.Wo; L.pool = o.pool;
  L.center_from = o.center_from;
  L.scratch = c->d_splitk[o.lane]; L.scratch_bytes = c->splitk_bytes[o.lane];
  L.halo_skip = (c->cfg.flags & RON_CFG_NO_HALO_SKIP) ? 0 : 1;
  *out_l = L;
  return RON_OK;
}

// Streams, events, split-K scratch and timing slots of one execution slot (after c->ops / c->packed are in place).
static int slot_resources(ron_ctx* c) {
  if ((c->cfg.flags & RON_CFG_MULTI_STREAM) && !c->is_ssd()) {
    for (int l = 1; l < 4; ++l) {
      RON_HIP_CHECK(hipStreamCreateWithFlags(&c->side[l], hipStreamNonBlocking));
      RON_HIP_CHECK(hipEventCreateWithFlags(&c->lane_ready[l], hipEventDisableTiming));
      RON_HIP_CHECK(hipEventCreateWithFlags(&c->lane_done[l], hipEventDisableTiming));
    }
  }
  // split-K scratch: the largest slab set any launch (or grouped launch) of a lane can ask for at max_batch
  for (size_t i = 0; i < c->ops.size();) {
    const Op& o = c->ops[i];
    size_t j = i + 1;
    if (o.kind == OP_CONV && o.group >= 0) while (j < c->ops.size() && c->ops[j].group == o.group) ++j;
    if (o.kind == OP_CONV && (o.group >= 0 || o.up == 0)) {
      // a clone takes the plans (and with them the scratch sizes) of the slot it was cloned from: the schedule model of a grouped
      // launch costs ~1 ms per (group, batch)
      const ron_ctx* from = c->weights_owner;
      std::vector<std::array<int, kMaxConvGroup>>* plans = nullptr;
      if (o.group >= 0) {
        plans = &c->group_sk[(int)i];
        if (from != nullptr && from->group_sk.count((int)i)) *plans = from->group_sk.at((int)i);
        else plans->assign(c->cfg.max_batch + 1, std::array<int, kMaxConvGroup>{});
      }
      for (int nb = 1; nb <= c->cfg.max_batch && from == nullptr; ++nb) {
        ConvLaunch L[kMaxConvGroup];
        for (size_t k = i; k < j; ++k) describe_conv(c, c->ops[k], nb, nullptr, &L[k - i]);
        int64_t b;
        if (o.group >= 0) {
          // the plan ron_forward will launch with, and its scratch, from ONE run of the model
          conv_group_plan(L, (int)(j - i), o.group_cfg, (*plans)[nb].data());
          b = conv_group_scratch_bytes(L, (int)(j - i), o.group_cfg, (*plans)[nb].data());
        } else {
          b = conv_scratch_bytes(L[0]);
        }
        if (b > c->splitk_bytes[o.lane]) c->splitk_bytes[o.lane] = b;
      }
    }
    i = j;
  }
  if (c->weights_owner != nullptr) for (int l = 0; l < 4; ++l) c->splitk_bytes[l] = c->weights_owner->splitk_bytes[l];
  for (int l = 0; l < 4; ++l)
    if (c->splitk_bytes[l] > 0) RON_HIP_CHECK(ron::dev_malloc(&c->d_splitk[l], (size_t)c->splitk_bytes[l]));
  // ron_detect's head buffers and post-processing workspace, for max_batch: allocated (and the workspace zeroed) here, so that the
  // first ron_detect is as free of host synchronisation as every later one (include/ron_hip.h, Ownership)
  {
    const int mb = c->cfg.max_batch;
    for (int i = 0; i < c->n_feat; ++i) {
      const int A = c->feat_A[i];
      const size_t cells = (size_t)mb * c->feat_h[i] * c->feat_w[i];
      RON_HIP_CHECK(ron::dev_malloc((void**)&c->d_head[0][i], cells * A * c->cfg.num_classes * sizeof(float)));
      if (c->has_obj) RON_HIP_CHECK(ron::dev_malloc((void**)&c->d_head[1][i], cells * A * 2 * sizeof(float)));
      RON_HIP_CHECK(ron::dev_malloc((void**)&c->d_head[2][i], cells * A * 4 * sizeof(float)));
    }
    ron_heads hd;
    memset(&hd, 0, sizeof(hd));
    int rc = ron_heads_describe(c, &hd);
    if (rc) return rc;
    for (int i = 0; i < c->n_feat; ++i) { hd.cls[i] = c->d_head[0][i]; hd.obj[i] = c->d_head[1][i]; hd.loc[i] = c->d_head[2][i]; }
    c->post_ws_bytes = ron_post_np_workspace_bytes(&hd, mb);
    if (c->post_ws_bytes <= 0) return RON_ERR_INVALID;      // (ron_last_error says why)
    RON_HIP_CHECK(ron::dev_malloc(&c->d_post_ws, (size_t)c->post_ws_bytes));
    RON_HIP_CHECK(ron::dev_memset(c->d_post_ws, 0, (size_t)c->post_ws_bytes));      // once: the kernels keep the counters clean
  }
  c->timing.assign(c->ops.size() + 1, OpTiming());
  // names ron_profile_get hands out: a grouped launch is reported on its first member as "group[first+N]", the other members as
  // "(name)"; built once so that the pointers stay valid until ron_destroy
  c->labels.assign(c->ops.size() + 1, std::string());
  for (size_t i = 0; i < c->ops.size(); ++i) {
    const Op& o = c->ops[i];
    const bool member = o.group >= 0 && i > 0 && c->ops[i - 1].group == o.group;
    if (member) { c->labels[i] = "(" + o.name + ")"; continue; }
    int extra = 0;
    for (size_t j = i + 1; o.group >= 0 && j < c->ops.size() && c->ops[j].group == o.group; ++j) ++extra;
    c->labels[i] = extra > 0 ? "group[" + o.name + "+" + std::to_string(extra) + "]" : o.name;
  }
  c->labels[c->ops.size()] = "post_np";
  return RON_OK;
}

// A second execution slot over the same weights: own activations, head buffers, scratch and streams, so that two
// (or more) batches can be in flight on different streams; the packed weights stay with `src`.
extern "C" int ron_clone(ron_ctx* src, ron_ctx** out) {
  RON_REQUIRE(src && out, "NULL argument");
  if (!src->finalized) { ron::set_error("ron_clone before ron_finalize_weights"); return RON_ERR_STATE; }
  ron_ctx* owner = src->weights_owner ? src->weights_owner : src;
  ron_ctx* c = nullptr;
  int rc = ron_create(&c, &src->cfg);
  if (rc) return rc;
  c->packed = owner->packed;
  c->ops = owner->ops;
  c->d_l2_gamma = owner->d_l2_gamma; c->d_stem_w = owner->d_stem_w; c->d_stem_b = owner->d_stem_b;
  c->d_stem2_w = owner->d_stem2_w; c->d_stem2_b = owner->d_stem2_b; c->d_stem2_w1 = owner->d_stem2_w1;
  c->stem_oscale = owner->stem_oscale;
  c->flops_per_image = owner->flops_per_image;
  c->grouped_launches = owner->grouped_launches;
  c->weights_owner = owner;
  ++owner->clones;
  for (auto& v : c->vars) v.loaded = true;
  if ((rc = slot_resources(c))) { (void)ron_destroy(c); return rc; }
  c->finalized = true;
  *out = c;
  return RON_OK;
}

extern "C" double ron_flops_per_image(const ron_ctx* c) { return c ? c->flops_per_image : -1.0; }

extern "C" int ron_heads_describe(const ron_ctx* c, ron_heads* hd) {
  RON_REQUIRE(c && hd, "NULL argument");
  hd->num_layers = c->n_feat;
  hd->num_classes = c->cfg.num_classes;
  for (int i = 0; i < c->n_feat; ++i) {
    hd->feat_h[i] = c->feat_h[i];
    hd->feat_w[i] = c->feat_w[i];
    hd->num_anchors[i] = c->feat_A[i];
    hd->anchor_y[i] = c->d_anchor[i][0]; hd->anchor_x[i] = c->d_anchor[i][1];
    hd->anchor_h[i] = c->d_anchor[i][2]; hd->anchor_w[i] = c->d_anchor[i][3];
    if (!c->has_obj) hd->obj[i] = nullptr;
  }
  return RON_OK;
}

extern "C" int ron_forward(ron_ctx* c, const float* d_images, int n, ron_heads* out, void* stream) {
  RON_REQUIRE(c && d_images && out, "NULL argument");
  if (!c->finalized) { ron::set_error("ron_forward before ron_finalize_weights"); return RON_ERR_STATE; }
  RON_REQUIRE(n >= 1 && n <= c->cfg.max_batch, "batch %d outside [1, max_batch=%d]", n, c->cfg.max_batch);
  // launches go to the context's device whatever the caller's current device is (the stream must belong to it)
  DeviceGuard on_device(c->cfg.device);
  RON_HIP_CHECK(on_device.err);
  int rc = ron_heads_describe(c, out);
  if (rc) return rc;
  std::vector<hipEvent_t>* ev = nullptr;
  if (c->profiling > 0 && c->pending.size() < 256) {
    --c->profiling;
    c->pending.emplace_back();
    c->pending_ops.emplace_back();
    ev = &c->pending.back();
  }
  // one event per op start on the op's stream; an op ends where the next op of its lane starts (or at the lane's end stamp)
  auto stamp = [&](hipStream_t st, int what) -> int {
    if (!ev) return RON_OK;
    hipEvent_t e;
    if (!c->event_pool.empty()) { e = c->event_pool.back(); c->event_pool.pop_back(); }
    else RON_HIP_CHECK(hipEventCreate(&e));
    RON_HIP_CHECK(hipEventRecord(e, st));
    ev->push_back(e);
    c->pending_ops.back().push_back(what);
    return RON_OK;
  };
  const hipStream_t main_stream = (hipStream_t)stream;
  bool lane_started[4] = {false, false, false, false};
  for (size_t oi = 0; oi < c->ops.size(); ++oi) {
    const Op& o = c->ops[oi];
    hipStream_t s = main_stream;
    if (o.lane > 0) {
      s = c->side[o.lane];
      if (!lane_started[o.lane]) {          // everything enqueued on the main stream so far (incl. this scale's reference map)
        RON_HIP_CHECK(hipEventRecord(c->lane_ready[o.lane], main_stream));
        RON_HIP_CHECK(hipStreamWaitEvent(s, c->lane_ready[o.lane], 0));
        lane_started[o.lane] = true;
      }
    }
    if ((rc = stamp(s, (int)oi))) return rc;
    if (o.kind == OP_IM2COL) {
      const Tensor& t = c->tensors[o.out];
      if ((rc = launch_im2col_c3(d_images, n, t.H, t.W, c->cfg.dtype, t.d, t.C, s))) return rc;
    } else if (o.kind == OP_STEM) {
      const Tensor& t = c->tensors[o.out];
      if ((rc = launch_stem_conv(d_images, n, t.H, t.W, c->cfg.dtype, c->d_stem_w, c->d_stem_b, c->view(o.out, n), s, c->stem_oscale))) return rc;
    } else if (o.kind == OP_STEM2) {
      const Tensor& t = c->tensors[o.out];
      if ((rc = launch_stem2(d_images, n, 2 * t.H, 2 * t.W, c->cfg.dtype, c->d_stem2_w1, c->d_stem_b, c->d_stem2_w, c->d_stem2_b,
                             c->view(o.out, n), s))) return rc;
    } else if (o.kind == OP_POOL) {
      if ((rc = launch_maxpool2x2(c->view(o.in, n), c->view(o.out, n), c->cfg.dtype, s))) return rc;
    } else if (o.kind == OP_POOL3) {
      if ((rc = launch_maxpool3x3s1(c->view(o.in, n), c->view(o.out, n), c->cfg.dtype, s))) return rc;
    } else if (o.kind == OP_L2NORM) {
      if ((rc = launch_l2norm(c->view(o.in, n), c->view(o.out, n), c->d_l2_gamma, c->cfg.dtype, s))) return rc;
    } else if (o.group >= 0) {
      // this op and the following ones of the same group: one launch (the stamp above times the whole group)
      ConvLaunch L[kMaxConvGroup];
      size_t j = oi;
      for (; j < c->ops.size() && c->ops[j].group == o.group; ++j) {
        RON_REQUIRE(j - oi < (size_t)kMaxConvGroup, "conv group %d has more than %d members", o.group, kMaxConvGroup);
        if ((rc = describe_conv(c, c->ops[j], n, out, &L[j - oi]))) return rc;
      }
      std::vector<std::array<int, kMaxConvGroup>>& plans = c->group_sk[(int)oi];      // filled by slot_resources
      if (plans.empty()) plans.assign(c->cfg.max_batch + 1, std::array<int, kMaxConvGroup>{});
      if (plans[n][0] == 0) conv_group_plan(L, (int)(j - oi), o.group_cfg, plans[n].data());
      if ((rc = launch_conv_group(L, (int)(j - oi), o.group_cfg, c->d_splitk[o.lane], c->splitk_bytes[o.lane], s, plans[n].data()))) {
        std::string msg = ron_last_error();
        ron::set_error("group of %s: %s", o.name.c_str(), msg.c_str());
        return rc;
      }
      oi = j - 1;
    } else {
      ConvLaunch L;
      if ((rc = describe_conv(c, o, n, out, &L))) return rc;
      if (o.fuse_next_pool && oi + 1 < c->ops.size() && c->ops[oi + 1].kind == OP_POOL) {
        // conv4_3 / conv5_3: both the map and its pool are read later.  When this launch does not split K (it does at small
        // batches: the pool epilogue needs whole sums) the pool comes out of the same accumulators and the pool launch is skipped.
        // (The pool inside the split-K finalize pass instead was measured at batch 1: conv4_3 30.9 + pool4 8.7 us -> 40.9 us, no gain.)
        ConvLaunch F = L;
        F.out2 = L.out;                           // the kernel choice of a two-output launch (row-gather kernel only) ...
        std::vector<signed char>& ok = c->fuse_pool_ok[(int)oi];
        if (ok.empty()) ok.assign(c->cfg.max_batch + 1, (signed char)-1);
        if (ok[n] < 0) ok[n] = conv_scratch_bytes(F) == 0 ? 1 : 0;      // ... would not split K at this batch
        if (ok[n] == 1) {
          F.out = c->view(c->ops[oi + 1].out, n);
          F.pool = 1;
          L = F;
          ++oi;
        }
      }
      if ((rc = launch_conv(L, s))) {
        std::string msg = ron_last_error();
        ron::set_error("%s: %s", o.name.c_str(), msg.c_str());
        return rc;
      }
    }
  }
  if (ev) {
    if ((rc = stamp(main_stream, -1))) return rc;
    for (int l = 1; l < 4; ++l) if (lane_started[l] && (rc = stamp(c->side[l], -1 - 10 * l))) return rc;
  }
  for (int l = 1; l < 4; ++l)
    if (lane_started[l]) {                  // join: the caller's stream continues after every side branch
      RON_HIP_CHECK(hipEventRecord(c->lane_done[l], c->side[l]));
      RON_HIP_CHECK(hipStreamWaitEvent(main_stream, c->lane_done[l], 0));
    }
  return RON_OK;
}

// ---- per-launch timing -----------------------------------------------------------------------
extern "C" int ron_profile_enable(ron_ctx* c, int enable) {
  RON_REQUIRE(c, "NULL ctx");
  c->profiling = enable > 0 ? enable : 0;
  return RON_OK;
}

static int profile_collect(ron_ctx* c) {
  for (size_t k = 0; k < c->pending.size(); ++k) {
    auto& call = c->pending[k];
    const auto& what = c->pending_ops[k];
    if (call.empty()) continue;
    for (hipEvent_t e : call) RON_HIP_CHECK(hipEventSynchronize(e));
    // the end of op i = the next stamp recorded on the same lane
    for (size_t i = 0; i < call.size(); ++i) {
      int slot, lane;
      if (what[i] >= 0) { slot = what[i]; lane = c->ops[slot].lane; }
      else if (what[i] == -2) { slot = (int)c->ops.size(); lane = 0; }
      else continue;
      size_t j = i + 1;
      for (; j < call.size(); ++j) {
        const int w = what[j];
        const int lj = w >= 0 ? c->ops[w].lane : (w == -1 || w == -2 || w == -3 ? 0 : (-1 - w) / 10);
        if (lj == lane) break;
      }
      if (j == call.size()) continue;
      float ms = 0.f;
      RON_HIP_CHECK(hipEventElapsedTime(&ms, call[i], call[j]));
      c->timing[slot].ms += ms;
      c->timing[slot].launches += 1;
    }
    for (hipEvent_t e : call) c->event_pool.push_back(e);
  }
  c->pending.clear();
  c->pending_ops.clear();
  return RON_OK;
}

extern "C" int ron_profile_num_ops(const ron_ctx* c) { return c ? (int)c->ops.size() + 1 : RON_ERR_INVALID; }
extern "C" int ron_num_grouped_launches(const ron_ctx* c) { return c ? c->grouped_launches : RON_ERR_INVALID; }

extern "C" int ron_profile_get(ron_ctx* c, int i, const char** name, int* is_conv, double* flops_per_image,
                               double* total_ms, int* launches, double* act_bytes_per_image, double* weight_bytes) {
  RON_REQUIRE(c && i >= 0 && i <= (int)c->ops.size(), "op index out of range");
  int rc = profile_collect(c);
  if (rc) return rc;
  const bool post = i == (int)c->ops.size();
  // a grouped launch is reported on its first member (FLOPs / bytes of the whole group, name "first+N"); the other members
  // report nothing, so sums over the rows stay right
  double fl = 0, ab = 0, wb = 0;
  if (!post) {
    const Op& o = c->ops[i];
    const bool member = o.group >= 0 && i > 0 && c->ops[i - 1].group == o.group;
    if (!member)
      for (size_t j = i; j < c->ops.size() && (j == (size_t)i || (o.group >= 0 && c->ops[j].group == o.group)); ++j) {
        fl += c->ops[j].flops; ab += c->ops[j].act_bytes; wb += c->ops[j].wgt_bytes;
      }
  }
  if (name) *name = c->labels[i].c_str();
  if (is_conv) *is_conv = !post && c->ops[i].kind == OP_CONV;
  if (flops_per_image) *flops_per_image = fl;
  if (total_ms) *total_ms = c->timing[i].ms;
  if (launches) *launches = c->timing[i].launches;
  if (act_bytes_per_image) *act_bytes_per_image = ab;
  if (weight_bytes) *weight_bytes = wb;
  return RON_OK;
}

extern "C" int ron_profile_reset(ron_ctx* c) {
  RON_REQUIRE(c, "NULL ctx");
  int rc = profile_collect(c);
  if (rc) return rc;
  for (auto& t : c->timing) t = OpTiming();
  return RON_OK;
}

extern "C" int ron_end_point_shape(const ron_ctx* c, const char* name, int n, int64_t nhwc[4]) {
  RON_REQUIRE(c && name && nhwc, "NULL argument");
  std::string key = name;
  // end_points of the reference: block1..5 = last conv of the VGG block, block6 = fc6, block7 = fc7
  static const std::map<std::string, std::string> alias = {{"block1", "conv1_2"}, {"block2", "conv2_2"}, {"block3", "conv3_3"},
                                                           {"block4", "conv4_3"}, {"block5", "conv5_3"}, {"block6", "fc6"},
                                                           {"block7", "fc7"}};
  auto a = alias.find(key);
  if (a != alias.end()) key = a->second;
  if (c->is_ssd()) { if (key == "fc6") key = "conv6"; else if (key == "fc7") key = "conv7"; }
  auto it = c->tensor_index.find(key);
  if (it == c->tensor_index.end()) { ron::set_error("unknown end point '%s'", name); return RON_ERR_UNKNOWN_NAME; }
  const Tensor& t = c->tensors[it->second];
  nhwc[0] = n; nhwc[1] = t.H; nhwc[2] = t.W; nhwc[3] = t.C;
  return it->second + 1;     // > 0: tensor index + 1 (internal use), callers test for < 0
}

extern "C" int ron_end_point_copy(ron_ctx* c, const char* name, int n, float* d_out, void* stream) {
  RON_REQUIRE(c && name && d_out, "NULL argument");
  RON_REQUIRE(n >= 1 && n <= c->cfg.max_batch, "bad batch");
  int64_t shp[4];
  const int idx = ron_end_point_shape(c, name, n, shp);
  if (idx < 0) return idx;
  if (c->tensors[idx - 1].d == nullptr) { ron::set_error("end point '%s' is not materialised in this configuration", name); return RON_ERR_UNKNOWN_NAME; }
  return launch_unpack(c->view(idx - 1, n), c->cfg.dtype, 0, d_out, (hipStream_t)stream);
}

extern "C" int ron_detect(ron_ctx* c, const float* d_images, int n, const ron_post_cfg* cfg, ron_detections* out, void* stream) {
  RON_REQUIRE(c && cfg && out, "NULL argument");
  RON_REQUIRE(n >= 1 && n <= c->cfg.max_batch, "batch %d outside [1, max_batch=%d]", n, c->cfg.max_batch);
  const int mb = c->cfg.max_batch;
  ron_heads hd;
  memset(&hd, 0, sizeof(hd));
  DeviceGuard on_device(c->cfg.device);
  RON_HIP_CHECK(on_device.err);
  if (!c->finalized) { ron::set_error("ron_detect before ron_finalize_weights"); return RON_ERR_STATE; }
  (void)mb;
  // (head buffers and workspace: slot_resources, at ron_finalize_weights / ron_clone)
  for (int i = 0; i < c->n_feat; ++i) { hd.cls[i] = c->d_head[0][i]; hd.obj[i] = c->d_head[1][i]; hd.loc[i] = c->d_head[2][i]; }
  if (c->post_ws_dirty) {
    // an earlier call failed between its select pass and the pass that zeroes the counters again: start from a clean workspace
    RON_HIP_CHECK(ron::dev_memset_async(c->d_post_ws, 0, (size_t)c->post_ws_bytes, (hipStream_t)stream));
    c->post_ws_dirty = false;
  }
  int rc = ron_forward(c, d_images, n, &hd, stream);
  if (rc) return rc;
  ron_post_cfg pc = *cfg;
  pc.input_flags = ron::kPostWsClean;      // logits + raw offsets straight from the conv stack; self-cleaning workspace (common.h)
  const bool prof = !c->pending.empty() && !c->pending_ops.back().empty() && c->pending_ops.back().back() <= -1 &&
                    c->pending_ops.back().back() != -3 && c->pending_ops.back().back() != -2;   // this call was recorded
  auto post_stamp = [&](int what) -> int {
    hipEvent_t e;
    if (!c->event_pool.empty()) { e = c->event_pool.back(); c->event_pool.pop_back(); }
    else RON_HIP_CHECK(hipEventCreate(&e));
    RON_HIP_CHECK(hipEventRecord(e, (hipStream_t)stream));
    c->pending.back().push_back(e);
    c->pending_ops.back().push_back(what);
    return RON_OK;
  };
  if (prof && (rc = post_stamp(-2))) return rc;
  rc = ron_post_np(&hd, n, &pc, c->d_post_ws, c->post_ws_bytes, out, nullptr, nullptr, stream);
  if (rc != RON_OK) c->post_ws_dirty = true;
  if (rc == RON_OK && prof) rc = post_stamp(-3);
  return rc;
}
