// Host-side weight packing: TF layouts (HWIO conv, [kh,kw,Cout,Cin] transposed conv) -> the
// [Npad][K] K-contiguous rows the implicit-GEMM kernel streams, rounded to the ctx dtype.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "conv_mfma.h"

namespace ron {

static inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x0040u);   // keep NaN a NaN
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static inline uint16_t f32_to_f16_rne(float f) {
  const _Float16 h = (_Float16)f;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}

// rows: fp32 [npad][k]  ->  dtype bytes
static inline std::vector<uint8_t> cast_rows(const std::vector<float>& rows, int dtype) {
  std::vector<uint8_t> out(rows.size() * dtype_size(dtype));
  if (dtype == RON_DTYPE_F32) {
    memcpy(out.data(), rows.data(), out.size());
  } else {
    uint16_t* o = reinterpret_cast<uint16_t*>(out.data());
    for (size_t i = 0; i < rows.size(); ++i) o[i] = dtype == RON_DTYPE_BF16 ? f32_to_bf16_rne(rows[i]) : f32_to_f16_rne(rows[i]);
  }
  return out;
}

static inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

// HWIO [kh,kw,cin,cout] -> rows[n][(ky*kw+kx)*cin + c], n < npad (extra rows zero)
static inline void hwio_to_rows(const float* w, int kh, int kw, int cin, int cout, int npad, std::vector<float>* rows) {
  const int K = kh * kw * cin;
  rows->assign((size_t)npad * K, 0.f);
  for (int t = 0; t < kh * kw; ++t)
    for (int c = 0; c < cin; ++c)
      for (int n = 0; n < cout; ++n) (*rows)[(size_t)n * K + t * cin + c] = w[((size_t)t * cin + c) * cout + n];
}

}  // namespace ron
