// Anchor grids on the host, bit-exact with the reference's numpy float32 arithmetic
// (nets/ron_vgg_320.py:285-333): centres are float32 throughout, sizes are computed in
// double and rounded once to float32.
#include <math.h>

#include "common.h"

extern "C" int ron_anchor_one_layer(int img_h, int img_w, int feat_h, int feat_w, const double* sizes, int n_sizes,
                                    const double* ratios, int n_ratios, double step, double offset, float* y,
                                    float* x, float* h, float* w) {
  RON_REQUIRE(img_h > 0 && img_w > 0 && feat_h > 0 && feat_w > 0, "bad image / feature shape");
  RON_REQUIRE(sizes && ratios && n_sizes > 0 && n_ratios > 0, "bad sizes / ratios");
  RON_REQUIRE(n_sizes * n_ratios <= RON_MAX_ANCHORS_PER_CELL, "too many anchors per cell");
  RON_REQUIRE(y && x && h && w, "NULL output");
  // ((idx.astype(f32) + offset) * step) / img_shape : python scalars are weak, so every
  // intermediate stays float32.
  const float off = (float)offset, st = (float)step;
  for (int r = 0; r < feat_h; ++r)
    for (int c = 0; c < feat_w; ++c) {
      volatile float ty = (float)r + off;
      volatile float tx = (float)c + off;
      ty = ty * st;
      tx = tx * st;
      y[r * feat_w + c] = ty / (float)img_h;
      x[r * feat_w + c] = tx / (float)img_w;
    }
  for (int i = 0; i < n_ratios; ++i) {
    const double root = sqrt(ratios[i]);
    for (int j = 0; j < n_sizes; ++j) {
      h[i * n_sizes + j] = (float)(sizes[j] / (double)img_h / root);
      w[i * n_sizes + j] = (float)(sizes[j] / (double)img_w * root);
    }
  }
  return RON_OK;
}

// SSD anchors (nets/ssd_vgg_512.py:286-338): same centre grid; per cell [s0, sqrt(s0*s1)] squares then s0 at each ratio.
extern "C" int ron_ssd_anchor_one_layer(int img_h, int img_w, int feat_h, int feat_w, const double* sizes, int n_sizes,
                                        const double* ratios, int n_ratios, double step, double offset, float* y,
                                        float* x, float* h, float* w) {
  RON_REQUIRE(img_h > 0 && img_w > 0 && feat_h > 0 && feat_w > 0, "bad image / feature shape");
  RON_REQUIRE(sizes && ratios && n_sizes >= 1 && n_sizes <= 2 && n_ratios >= 0, "bad sizes / ratios");
  RON_REQUIRE(n_sizes + n_ratios <= RON_MAX_ANCHORS_PER_CELL, "too many anchors per cell");
  RON_REQUIRE(y && x && h && w, "NULL output");
  const float off = (float)offset, st = (float)step;
  for (int r = 0; r < feat_h; ++r)
    for (int c = 0; c < feat_w; ++c) {
      volatile float ty = (float)r + off;
      volatile float tx = (float)c + off;
      ty = ty * st;
      tx = tx * st;
      y[r * feat_w + c] = ty / (float)img_h;
      x[r * feat_w + c] = tx / (float)img_w;
    }
  h[0] = (float)(sizes[0] / (double)img_h);
  w[0] = (float)(sizes[0] / (double)img_w);
  int di = 1;
  if (n_sizes > 1) {
    h[1] = (float)(sqrt(sizes[0] * sizes[1]) / (double)img_h);
    w[1] = (float)(sqrt(sizes[0] * sizes[1]) / (double)img_w);
    di = 2;
  }
  for (int i = 0; i < n_ratios; ++i) {
    h[i + di] = (float)(sizes[0] / (double)img_h / sqrt(ratios[i]));
    w[i + di] = (float)(sizes[0] / (double)img_w * sqrt(ratios[i]));
  }
  return RON_OK;
}
