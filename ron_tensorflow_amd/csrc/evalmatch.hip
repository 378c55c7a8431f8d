// Ground-truth matching of the TF-evaluation detections: tfe.bboxes_matching_batch (tf_extended/bboxes.py:316-450)
// with tfe.bboxes_jaccard (:527-555) and tfe_math.safe_divide (math.py:25-38).
//
// The reference walks one (image, class) list in score order inside a tf.while_loop; every step is an argmax over
// the image's ground-truth boxes and a read-modify-write of the "already matched" flags, so the walk is inherently
// serial per list and parallel over the N * (C-1) lists and over the ground-truth boxes.  One wave per list, one lane
// per ground-truth box (chunks of 64), the matched flags live in one ballot mask per chunk.  Byte work, latency bound:
// 200 steps * ~6 shuffles; the whole batch is a single launch of N * (C-1) independent waves.
//
// Compiled with -ffp-contract=off: the jaccard decides tp / fp, it has to round like the float32 TF kernels do.
#include <hip/hip_runtime.h>

#include "common.h"

namespace ron {
namespace {

constexpr int kMaxGtChunks = RON_MAX_GT / 64;

__global__ __launch_bounds__(64) void bboxes_matching_kernel(const float* __restrict__ scores, const float* __restrict__ bboxes,
                                                             int K, const int32_t* __restrict__ glabels,
                                                             const float* __restrict__ gbboxes,
                                                             const uint8_t* __restrict__ gdifficults, int G, float thr,
                                                             int32_t* __restrict__ n_gbboxes, uint8_t* __restrict__ tp,
                                                             uint8_t* __restrict__ fp) {
  const int c1 = blockIdx.x, img = blockIdx.y, C1 = gridDim.x;
  const int label = c1 + 1;
  const int lane = threadIdx.x;
  const long long list = (long long)img * C1 + c1;
  // this lane's ground-truth boxes (one per chunk)
  float g0[kMaxGtChunks], g1[kMaxGtChunks], g2[kMaxGtChunks], g3[kMaxGtChunks], garea[kMaxGtChunks], same[kMaxGtChunks];
  unsigned long long diff_mask[kMaxGtChunks], matched[kMaxGtChunks];
  int n_gb = 0;
#pragma unroll
  for (int ch = 0; ch < kMaxGtChunks; ++ch) {
    const int g = ch * 64 + lane;
    const bool in = g < G;
    const long long gi = (long long)img * G + (in ? g : 0);
    const int lab = in ? glabels[gi] : -1;
    const bool d = in ? gdifficults[gi] != 0 : true;
    g0[ch] = in ? gbboxes[gi * 4 + 0] : 0.f;
    g1[ch] = in ? gbboxes[gi * 4 + 1] : 0.f;
    g2[ch] = in ? gbboxes[gi * 4 + 2] : 0.f;
    g3[ch] = in ? gbboxes[gi * 4 + 3] : 0.f;
    garea[ch] = (g2[ch] - g0[ch]) * (g3[ch] - g1[ch]);
    same[ch] = lab == label ? 1.f : 0.f;
    diff_mask[ch] = __ballot(d);
    matched[ch] = 0ull;
    n_gb += __popcll(__ballot(lab == label && !d));
  }
  if (lane == 0) n_gbboxes[list] = n_gb;
  const float* sb = bboxes + list * K * 4;
  for (int i = 0; i < K; ++i) {
    const float r0 = sb[i * 4 + 0], r1 = sb[i * 4 + 1], r2 = sb[i * 4 + 2], r3 = sb[i * 4 + 3];
    const float rarea = (r2 - r0) * (r3 - r1);
    float best = -1.f;
    int best_idx = 0;
#pragma unroll
    for (int ch = 0; ch < kMaxGtChunks; ++ch) {
      if (ch * 64 >= G) break;
      const float h = fmaxf(fminf(g2[ch], r2) - fmaxf(g0[ch], r0), 0.f);
      const float w = fmaxf(fminf(g3[ch], r3) - fmaxf(g1[ch], r1), 0.f);
      const float inter = h * w;
      const float uni = (-inter + garea[ch]) + rarea;
      float jac = (uni > 0.f ? inter / uni : 0.f) * same[ch];
      int idx = ch * 64 + lane;
      if (idx >= G) jac = -1.f;
      // wave argmax, first maximum wins (tf.argmax)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float oj = __shfl_xor(jac, off);
        const int oi = __shfl_xor(idx, off);
        if (oj > jac || (oj == jac && oi < idx)) { jac = oj; idx = oi; }
      }
      if (jac > best) { best = jac; best_idx = idx; }
    }
    const int bch = best_idx >> 6;
    const unsigned long long bit = 1ull << (best_idx & 63);
    unsigned long long dm = diff_mask[0], mm = matched[0];
#pragma unroll
    for (int ch = 1; ch < kMaxGtChunks; ++ch)
      if (bch == ch) { dm = diff_mask[ch]; mm = matched[ch]; }
    const bool match = best > thr;
    const bool existing = (mm & bit) != 0;
    const bool not_diff = (dm & bit) == 0;
    if (lane == 0) {
      tp[list * K + i] = not_diff && match && !existing;
      fp[list * K + i] = not_diff && (existing || !match);
    }
    if (not_diff && match) {
#pragma unroll
      for (int ch = 0; ch < kMaxGtChunks; ++ch)
        if (bch == ch) matched[ch] |= bit;
    }
  }
}

}  // namespace
}  // namespace ron

extern "C" int ron_bboxes_matching(const float* scores, const float* bboxes, int n, int num_lists, int k, const int32_t* glabels,
                                   const float* gbboxes, const uint8_t* gdifficults, int g, float matching_threshold,
                                   int32_t* n_gbboxes, uint8_t* tp, uint8_t* fp, void* stream) {
  RON_REQUIRE(scores != nullptr && bboxes != nullptr && glabels != nullptr && gbboxes != nullptr && gdifficults != nullptr,
              "bad argument");
  RON_REQUIRE(n_gbboxes != nullptr && tp != nullptr && fp != nullptr, "bad argument");
  RON_REQUIRE(n > 0 && num_lists > 0 && k > 0, "bad argument");
  RON_REQUIRE(g >= 1 && g <= RON_MAX_GT, "ground-truth boxes per image %d not in [1, %d]", g, RON_MAX_GT);
  RON_LAUNCH(ron::bboxes_matching_kernel, dim3(num_lists, n), dim3(64), 0, (hipStream_t)stream, scores, bboxes, k,
                     glabels, gbboxes, gdifficults, g, matching_threshold, n_gbboxes, tp, fp);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}
