// Evaluation preprocessing in front of the conv stack: preprocess_for_eval with Resize.WARP_RESIZE
// (preprocessing/ssd_vgg_preprocessing.py:358-425): tf.to_float -> tf_image_whitened (:41-55, means 123/117/104)
// -> tf_image.resize_image (preprocessing/tf_image.py:269-282) = TF1 bilinear resize, align_corners=False, i.e. the
// legacy source coordinate  in = out_index * (in_size / out_size)  with no half-pixel offset.
//
// HBM-bound byte work: every output pixel reads 4 source pixels (12 bytes, L2-resident neighbours) and writes
// 12 bytes of fp32; one thread per output pixel, images of different sizes come packed in one byte buffer with an
// (offset, height, width) table.  -ffp-contract=off: the interpolation rounds once per operation like the TF kernel.
#include <hip/hip_runtime.h>

#include "common.h"

namespace ron {
namespace {

struct Means { float m[3]; };

// geom (optional, 8 ints per image): the source window [cy, cy+ch) x [cx, cx+cw) of the whitened image is bilinearly resized
// to rh x rw and placed at (py, px) of the output, zeros elsewhere.  WARP_RESIZE: the whole image onto the whole output (the
// default when geom is null); CENTRAL_CROP: ch == rh (scale 1, exact copy); PAD_AND_RESIZE: whole image, rh x rw = floor(factor * hw).
__global__ __launch_bounds__(256) void preprocess_eval_kernel(const uint8_t* __restrict__ packed, const int64_t* __restrict__ offsets,
                                                              const int32_t* __restrict__ hw, const int32_t* __restrict__ geom,
                                                              int out_h, int out_w, Means mean, float* __restrict__ out) {
  const int img = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= out_h * out_w) return;
  int oy = p / out_w, ox = p - oy * out_w;
  const int H = hw[2 * img], W = hw[2 * img + 1];
  int cy = 0, cx = 0, ch = H, cw = W, rh = out_h, rw = out_w;
  float* o = out + ((long long)img * out_h * out_w + p) * 3;
  if (geom != nullptr) {
    const int32_t* g = geom + 8 * img;
    cy = g[0]; cx = g[1]; ch = g[2]; cw = g[3]; rh = g[6]; rw = g[7];
    oy -= g[4]; ox -= g[5];
    if (oy < 0 || oy >= rh || ox < 0 || ox >= rw) {          // tf.image.pad_to_bounding_box pads the whitened image with 0
      o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
      return;
    }
  }
  const uint8_t* src = packed + offsets[img] + ((long long)cy * W + cx) * 3;
  const float sy = (float)ch / (float)rh, sx = (float)cw / (float)rw;
  const float in_y = (float)oy * sy, in_x = (float)ox * sx;
  const int y0 = (int)floorf(in_y), x0 = (int)floorf(in_x);
  const int y1 = min(y0 + 1, ch - 1), x1 = min(x0 + 1, cw - 1);
  const float ly = in_y - (float)y0, lx = in_x - (float)x0;
  const uint8_t* p00 = src + ((long long)y0 * W + x0) * 3;
  const uint8_t* p01 = src + ((long long)y0 * W + x1) * 3;
  const uint8_t* p10 = src + ((long long)y1 * W + x0) * 3;
  const uint8_t* p11 = src + ((long long)y1 * W + x1) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float tl = (float)p00[c] - mean.m[c], tr = (float)p01[c] - mean.m[c];
    const float bl = (float)p10[c] - mean.m[c], br = (float)p11[c] - mean.m[c];
    const float top = tl + (tr - tl) * lx;
    const float bot = bl + (br - bl) * lx;
    o[c] = top + (bot - top) * ly;
  }
}

}  // namespace
}  // namespace ron

extern "C" int ron_preprocess_eval(const uint8_t* packed, const int64_t* offsets, const int32_t* hw, int n, int out_h, int out_w,
                                   const float* means, float* out, void* stream) {
  return ron_preprocess_eval_geom(packed, offsets, hw, nullptr, n, out_h, out_w, means, out, stream);
}

extern "C" int ron_preprocess_eval_geom(const uint8_t* packed, const int64_t* offsets, const int32_t* hw, const int32_t* geom, int n,
                                        int out_h, int out_w, const float* means, float* out, void* stream) {
  RON_REQUIRE(packed != nullptr && offsets != nullptr && hw != nullptr && means != nullptr && out != nullptr, "bad argument");
  RON_REQUIRE(n > 0 && out_h > 0 && out_w > 0, "bad argument");
  ron::Means m;
  for (int c = 0; c < 3; ++c) m.m[c] = means[c];
  const int px = out_h * out_w;
  RON_LAUNCH(ron::preprocess_eval_kernel, dim3((px + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, packed, offsets, hw,
                     geom, out_h, out_w, m, out);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}
