// Single-operator entry points (ron_conv2d_nhwc, ron_maxpool2x2_nhwc): the same kernels the
// graph launches, wrapped with dense-fp32 <-> halo-tensor conversion so that the parity tests
// can pin each kernel against the oracle.  They allocate scratch and synchronise: test/tool use.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "pack.h"

namespace {

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(int64_t bytes, bool zero) {
    RON_HIP_CHECK(hipMalloc(&p, (size_t)bytes));
    if (zero) RON_HIP_CHECK(hipMemset(p, 0, (size_t)bytes));
    return RON_OK;
  }
};

ron::TensorView make_view(void* base, int n, int h, int w, int c, int pad, int esz) {
  ron::TensorView v;
  v.base = base; v.N = n; v.H = h; v.W = w; v.C = c; v.pad = pad; v.cstride = c; v.coff = 0;
  v.bytes = ron::TensorView::halo_pixels(n, h, w, pad) * c * esz;
  return v;
}

struct ConvSetup {
  ron::ConvLaunch c;
  DevBuf d_w, d_w_c64, d_b, d_in, d_out, d_res, d_scratch;
  int ho = 0, wo = 0;
  bool is_c3 = false;
};

// Packs weights like ron_finalize_weights does, uploads them and allocates the halo tensors.
int setup_conv(const ron_conv_desc* d, const float* w, const float* bias, bool with_residual, ConvSetup* S) {
  using namespace ron;
  RON_REQUIRE(d->dtype >= 0 && d->dtype <= RON_DTYPE_F16X3, "bad dtype");
  const int esz = (int)dtype_size(d->dtype);
  const int chunk = conv_k_chunk(d->dtype);
  S->is_c3 = (!d->transpose && d->cin == 3 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->dilation == 1);
  RON_REQUIRE(S->is_c3 || d->cin % chunk == 0, "cin %d must be a multiple of %d (or the 3-channel 3x3 stem)", d->cin, chunk);
  int cpad = 0;
  ConvLaunch& c = S->c;
  c.dtype = d->dtype;
  c.cfg = d->tile_cfg;
  std::vector<float> rows;
  std::vector<float> bias_pad;
  int cout_gemm;
  if (d->transpose) {
    RON_REQUIRE(d->kh == d->kw && d->kh == d->stride && d->dilation == 1, "transposed conv needs kernel == stride");
    S->ho = d->h * d->stride; S->wo = d->w * d->stride;
    cout_gemm = d->kh * d->kw * d->cout;
    RON_REQUIRE(d->cout % 128 == 0, "transposed conv: cout %d must be a multiple of 128", d->cout);
    c.Npad = cout_gemm;
    // TF layout [kh,kw,cout,cin] -> rows[(t*cout + co)][ci]
    rows.assign((size_t)cout_gemm * d->cin, 0.f);
    memcpy(rows.data(), w, rows.size() * sizeof(float));
    bias_pad.assign(cout_gemm, 0.f);
    if (bias) for (int t = 0; t < d->kh * d->kw; ++t) for (int co = 0; co < d->cout; ++co) bias_pad[t * d->cout + co] = bias[co];
    c.up = d->stride; c.up_cout = d->cout;
    c.kh = c.kw = 1; c.stride = 1; c.dil = 1; c.cpad = 0;
    c.Ho = d->h; c.Wo = d->w;
  } else {
    RON_REQUIRE(d->stride == 1 || (d->h % d->stride == 0 && d->w % d->stride == 0 && d->kh == d->stride),
                "strided conv: only kernel == stride on divisible maps");
    cpad = d->stride == 1 ? ((d->kh - 1) * d->dilation) / 2 : 0;   // SAME
    S->ho = d->h / d->stride; S->wo = d->w / d->stride;
    cout_gemm = d->cout;
    c.Npad = round_up(cout_gemm, conv_n_tile(cout_gemm));
    if (S->is_c3) {
      rows.assign((size_t)c.Npad * chunk, 0.f);
      for (int k = 0; k < 27; ++k) for (int n = 0; n < d->cout; ++n) rows[(size_t)n * chunk + k] = w[(size_t)k * d->cout + n];
      c.kh = c.kw = 1; c.cpad = 0;
    } else {
      hwio_to_rows(w, d->kh, d->kw, d->cin, d->cout, c.Npad, &rows);
      c.kh = d->kh; c.kw = d->kw; c.cpad = cpad;
    }
    bias_pad.assign(c.Npad, 0.f);
    if (bias) for (int n = 0; n < d->cout; ++n) bias_pad[n] = bias[n];
    c.stride = d->stride; c.dil = d->dilation;
    c.Ho = S->ho; c.Wo = S->wo;
  }
  c.Cout = cout_gemm;
  c.relu = d->relu;
  std::vector<uint8_t> wbytes = pack_conv_weights(rows, c.Npad, d->dtype, &c.oscale);
  int rc;
  if ((rc = S->d_w.alloc((int64_t)wbytes.size(), false))) return rc;
  if ((rc = S->d_b.alloc((int64_t)bias_pad.size() * 4, false))) return rc;
  RON_HIP_CHECK(hipMemcpy(S->d_w.p, wbytes.data(), wbytes.size(), hipMemcpyHostToDevice));
  RON_HIP_CHECK(hipMemcpy(S->d_b.p, bias_pad.data(), bias_pad.size() * 4, hipMemcpyHostToDevice));
  c.wgt = S->d_w.p; c.wgt_bytes = (int64_t)wbytes.size(); c.bias = (const float*)S->d_b.p;
  if (!d->transpose && !S->is_c3 && dtype_is_half(d->dtype) && d->kh == 3 && d->kw == 3 && d->cin == 64 && c.Npad == d->cout && d->cout % 64 == 0) {
    // what ron_finalize_weights adds for such a layer: the weights once more as LDS images for the resident-weight kernel
    const std::vector<uint8_t> img = pack_conv_c64_weights(rows, c.Npad, d->dtype);
    if ((rc = S->d_w_c64.alloc((int64_t)img.size(), false))) return rc;
    RON_HIP_CHECK(hipMemcpy(S->d_w_c64.p, img.data(), img.size(), hipMemcpyHostToDevice));
    c.wgt_c64 = S->d_w_c64.p;
  }
  if (S->is_c3) c.in = make_view(nullptr, d->n, d->h, d->w, chunk, 0, esz);
  else {
    const int cs = d->in_cstride > 0 ? d->in_cstride : d->cin;
    RON_REQUIRE(d->in_coff >= 0 && d->in_coff + d->cin <= cs, "bad channel slice");
    c.in = make_view(nullptr, d->n, d->h, d->w, cs, cpad, esz);     // allocation of the wide tensor
    c.in.C = d->cin;
    c.in.coff = d->in_coff;
  }
  if ((rc = S->d_in.alloc(c.in.bytes, true))) return rc;
  c.in.base = S->d_in.p;
  c.pool = d->pool;
  if (d->pool) { S->ho /= 2; S->wo /= 2; }
  // halo 1: exercises padded stores.  Split precision: pixels are whole 32-element (128-byte) chunks
  c.out = make_view(nullptr, d->n, S->ho, S->wo, d->dtype == RON_DTYPE_F16X3 ? round_up(d->cout, 32) : d->cout, 1, esz);
  c.out.C = d->cout;
  if ((rc = S->d_out.alloc(c.out.bytes, true))) return rc;
  c.out.base = S->d_out.p;
  if (with_residual) {
    if ((rc = S->d_res.alloc(c.out.bytes, true))) return rc;
    c.res = S->d_res.p;
  }
  c.splitk = d->splitk;
  c.center_from = d->transpose ? 0 : d->center_from;
  const int64_t sb = conv_scratch_bytes(c);
  if (sb > 0) {
    if ((rc = S->d_scratch.alloc(sb, false))) return rc;
    c.scratch = S->d_scratch.p;
    c.scratch_bytes = sb;
  }
  return RON_OK;
}

}  // namespace

extern "C" int ron_conv2d_nhwc(const ron_conv_desc* d, const float* x, const float* w, const float* bias,
                               const float* residual, float* y, void* stream) {
  using namespace ron;
  RON_REQUIRE(d && x && w && y, "NULL argument");
  hipStream_t s = (hipStream_t)stream;
  ConvSetup S;
  int rc;
  if ((rc = setup_conv(d, w, bias, residual != nullptr, &S))) return rc;
  if (S.is_c3 && dtype_is_half(d->dtype) && d->cout == 64 && d->w % 32 == 0 && d->relu && !residual) {
    // conv1_1 through the dedicated stem kernel (what the graph runs for bf16 / f16)
    std::vector<uint16_t> frags;
    stem_pack_weights(w, d->dtype, &frags);
    std::vector<float> b64(64, 0.f);
    if (bias) memcpy(b64.data(), bias, 64 * sizeof(float));
    DevBuf d_f, d_bb;
    if ((rc = d_f.alloc((int64_t)frags.size() * 2, false))) return rc;
    if ((rc = d_bb.alloc(64 * 4, false))) return rc;
    RON_HIP_CHECK(hipMemcpy(d_f.p, frags.data(), frags.size() * 2, hipMemcpyHostToDevice));
    RON_HIP_CHECK(hipMemcpy(d_bb.p, b64.data(), 64 * 4, hipMemcpyHostToDevice));
    if ((rc = launch_stem_conv(x, d->n, d->h, d->w, d->dtype, d_f.p, (const float*)d_bb.p, S.c.out, s))) return rc;
    if ((rc = launch_unpack(S.c.out, d->dtype, 0, y, s))) return rc;
    RON_HIP_CHECK(hipStreamSynchronize(s));
    return RON_OK;
  }
  if (S.is_c3) {
    if ((rc = launch_im2col_c3(x, d->n, d->h, d->w, d->dtype, S.d_in.p, S.c.in.C, s))) return rc;
  } else {
    if ((rc = launch_pack_input(x, S.c.in, d->dtype, s))) return rc;
  }
  if (residual) {
    TensorView rv = S.c.out;
    rv.base = S.d_res.p;
    if ((rc = launch_pack_input(residual, rv, d->dtype, s))) return rc;
  }
  if ((rc = launch_conv(S.c, s))) return rc;
  if ((rc = launch_unpack(S.c.out, d->dtype, 0, y, s))) return rc;
  RON_HIP_CHECK(hipStreamSynchronize(s));
  return RON_OK;
}

// How ron_conv2d_nhwc would launch `d` (no launch): the decisions of launch_conv for the same ConvLaunch.
extern "C" int ron_conv_plan(const ron_conv_desc* d, int32_t out[4]) {
  using namespace ron;
  RON_REQUIRE(d != nullptr && out != nullptr, "NULL argument");
  RON_REQUIRE(!(d->cin == 3), "the 3-channel stem has a kernel of its own");
  const size_t wn = (size_t)d->kh * d->kw * d->cin * d->cout;
  std::vector<float> w(wn, 0.f);
  ConvSetup S;
  int rc;
  if ((rc = setup_conv(d, w.data(), nullptr, false, &S))) return rc;
  int o[4];
  if ((rc = conv_describe(S.c, o))) return rc;
  for (int i = 0; i < 4; ++i) out[i] = o[i];
  return RON_OK;
}

// Two fp32 head tensors from ONE convolution over a shared input (ConvLaunch::split_n; the graph's pack_box_pair for the class and box
// convolutions of an SSD feature layer): the same packing - first head's columns, the second's from the next multiple of 8 - and
// the same launch, with the tile configuration and the split-K factor the caller's to force.
extern "C" int ron_conv2d_heads_nhwc(const ron_conv_desc* d, int split_first, const float* x, const float* w, const float* bias,
                                     float* y_first, float* y_second, void* stream) {
  using namespace ron;
  RON_REQUIRE(d && x && w && y_first && y_second, "NULL argument");
  RON_REQUIRE(!d->transpose && !d->pool && d->center_from == 0 && d->stride == 1 && d->cin % conv_k_chunk(d->dtype) == 0,
              "two-output convolution: a plain stride-1 convolution");
  RON_REQUIRE(split_first > 0 && split_first < d->cout, "two-output convolution: %d of %d channels in the first output", split_first, d->cout);
  hipStream_t s = (hipStream_t)stream;
  const int n_second = d->cout - split_first, split_n = round_up(split_first, 8);
  // the packed convolution: `split_n + n_second` columns, the ones between the two heads zero
  ron_conv_desc dp = *d;
  dp.cout = split_n + n_second;
  const int K = d->kh * d->kw * d->cin;
  std::vector<float> wp((size_t)K * dp.cout, 0.f), bp(dp.cout, 0.f);
  for (int k = 0; k < K; ++k) {
    for (int n = 0; n < split_first; ++n) wp[(size_t)k * dp.cout + n] = w[(size_t)k * d->cout + n];
    for (int n = 0; n < n_second; ++n) wp[(size_t)k * dp.cout + split_n + n] = w[(size_t)k * d->cout + split_first + n];
  }
  if (bias) {
    for (int n = 0; n < split_first; ++n) bp[n] = bias[n];
    for (int n = 0; n < n_second; ++n) bp[split_n + n] = bias[split_first + n];
  }
  ConvSetup S;
  int rc;
  if ((rc = setup_conv(&dp, wp.data(), bp.data(), false, &S))) return rc;
  ConvLaunch& c = S.c;
  TensorView v;
  v.base = y_first; v.N = d->n; v.H = S.ho; v.W = S.wo; v.pad = 0; v.coff = 0; v.C = split_first; v.cstride = split_first;
  v.bytes = (int64_t)d->n * S.ho * S.wo * split_first * 4;
  c.out = v;
  v.base = y_second; v.C = n_second; v.cstride = n_second;
  v.bytes = (int64_t)d->n * S.ho * S.wo * n_second * 4;
  c.out2 = v;
  c.out_f32 = 1;
  c.split_n = split_n; c.split_first = split_first;
  // (setup_conv sized the split-K scratch for a one-output launch, which may have gone to a kernel that never splits)
  DevBuf scratch2;
  const int64_t sb = conv_scratch_bytes(c);
  if (sb > c.scratch_bytes) {
    if ((rc = scratch2.alloc(sb, false))) return rc;
    c.scratch = scratch2.p;
    c.scratch_bytes = sb;
  }
  if ((rc = launch_pack_input(x, c.in, d->dtype, s))) return rc;
  if ((rc = launch_conv(c, s))) return rc;
  RON_HIP_CHECK(hipStreamSynchronize(s));
  return RON_OK;
}

// Times the conv kernel alone on random data (tools/sweep_conv.py): ms per launch over `iters` launches.
extern "C" int ron_conv2d_bench(const ron_conv_desc* d, int warmup, int iters, float* ms_per_launch) {
  using namespace ron;
  RON_REQUIRE(d && ms_per_launch && iters > 0 && warmup >= 0, "bad argument");
  RON_REQUIRE(!(d->cin == 3), "the 3-channel stem is not benchmarked through this entry");
  const size_t wn = (size_t)d->kh * d->kw * d->cin * d->cout;
  std::vector<float> w(wn), b(d->cout);
  uint32_t st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 32768.f - 1.f; };
  const float scale = 1.f / sqrtf((float)(d->kh * d->kw * d->cin));
  // RON_BENCH_ZERO=1 (tools/sweep_conv.py --zeros): all-zero operands instead of random ones - the same instruction stream at the
  // clock the chip holds when the data paths do not toggle (MI355X_MICROARCH.md, DVFS): how much of a kernel's time is power
  const bool zeros = getenv("RON_BENCH_ZERO") != nullptr;
  for (auto& v : w) v = zeros ? 0.f : rnd() * scale;
  for (auto& v : b) v = zeros ? 0.f : rnd() * 0.1f;
  ConvSetup S;
  int rc;
  if ((rc = setup_conv(d, w.data(), b.data(), false, &S))) return rc;
  if (!zeros && (rc = launch_fill_random(S.c.in, d->dtype, 777u, nullptr))) return rc;
  hipEvent_t e0, e1;
  RON_HIP_CHECK(hipEventCreate(&e0));
  RON_HIP_CHECK(hipEventCreate(&e1));
  for (int i = 0; i < warmup; ++i) if ((rc = launch_conv(S.c, nullptr))) return rc;
  RON_HIP_CHECK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters; ++i) if ((rc = launch_conv(S.c, nullptr))) return rc;
  RON_HIP_CHECK(hipEventRecord(e1, nullptr));
  RON_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  RON_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  *ms_per_launch = ms / iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return RON_OK;
}

extern "C" int ron_maxpool2x2_nhwc(const float* x, int n, int h, int w, int c, int dtype, float* y, void* stream) {
  using namespace ron;
  RON_REQUIRE(x && y && n > 0 && h > 0 && w > 0 && c > 0 && h % 2 == 0 && w % 2 == 0, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  const int esz = (int)dtype_size(dtype);
  TensorView vin = make_view(nullptr, n, h, w, c, 1, esz);
  TensorView vout = make_view(nullptr, n, h / 2, w / 2, c, 3, esz);
  DevBuf d_in, d_out;
  int rc;
  if ((rc = d_in.alloc(vin.bytes, true))) return rc;
  if ((rc = d_out.alloc(vout.bytes, true))) return rc;
  vin.base = d_in.p; vout.base = d_out.p;
  if ((rc = launch_pack_input(x, vin, dtype, s))) return rc;
  if ((rc = launch_maxpool2x2(vin, vout, dtype, s))) return rc;
  if ((rc = launch_unpack(vout, dtype, 0, y, s))) return rc;
  RON_HIP_CHECK(hipStreamSynchronize(s));
  return RON_OK;
}

extern "C" int ron_conv_num_tile_cfgs(void) { return ron::conv_num_cfgs(); }
