// Host-side weight packing: TF layouts (HWIO conv, [kh,kw,Cout,Cin] transposed conv) -> the
// [Npad][K] K-contiguous rows the implicit-GEMM kernel streams, rounded to the ctx dtype.
#pragma once
#include <math.h>
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

// RON_DTYPE_F16X3: power-of-two scale 2^k that brings the largest |w| of a layer into [2^14, 2^15): the hi plane stays below the
// f16 maximum (65504) and the lo plane of every weight down to 2^-17 of the largest one is a normal f16 (full 22 bits).
static inline int split_weight_exponent(const std::vector<float>& rows) {
  float mx = 0.f;
  for (float v : rows) mx = fmaxf(mx, fabsf(v));
  if (!(mx > 0.f) || !isfinite(mx)) return 0;
  int e;
  frexpf(mx, &e);                     // mx = m * 2^e, m in [0.5, 1)
  const int k = 15 - e;               // mx * 2^k in [2^14, 2^15)
  return k < -40 ? -40 : (k > 60 ? 60 : k);
}

// rows: fp32 [npad][k], k % 32 == 0 -> split-precision bytes: per 32 elements [32 x f16 hi][32 x f16 lo] of w * 2^k
static inline std::vector<uint8_t> split_rows(const std::vector<float>& rows, int k) {
  std::vector<uint8_t> out(rows.size() * 4);
  uint16_t* o = reinterpret_cast<uint16_t*>(out.data());
  for (size_t i = 0; i < rows.size(); ++i) {
    const float v = ldexpf(rows[i], k);
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const size_t at = (i / 32) * 64 + i % 32;
    memcpy(&o[at], &hi, 2);
    memcpy(&o[at + 32], &lo, 2);
  }
  return out;
}

// Row-major [npad][K] bytes -> 64-row x 128-byte blocks [npad / 64][K steps][64][128 B] (npad % 64 == 0, K * esz % 128 == 0):
// the (64 rows, one K step) unit both conv kernels stage is 8 KB of consecutive memory, so a wave's LDS-DMA instruction
// (8 rows x 128 B) reads one contiguous kilobyte instead of eight lines that are K * esz bytes apart.
static inline std::vector<uint8_t> block_rows(const std::vector<uint8_t>& rows, int npad, int64_t row_bytes) {
  const int64_t steps = row_bytes / 128;
  std::vector<uint8_t> out(rows.size());
  for (int n = 0; n < npad; ++n)
    for (int64_t kt = 0; kt < steps; ++kt)
      memcpy(&out[(((int64_t)(n / 64) * steps + kt) * 64 + n % 64) * 128], &rows[(int64_t)n * row_bytes + kt * 128], 128);
  return out;
}

// fp32 rows [npad][K] -> what the conv kernels read: rounded to the dtype, blocked.  *oscale = what the epilogue multiplies the
// accumulator by (1, or 2^-k in split-precision mode where the packed weights are w * 2^k).
static inline std::vector<uint8_t> pack_conv_weights(const std::vector<float>& rows, int npad, int dtype, float* oscale) {
  *oscale = 1.f;
  const int64_t K = (int64_t)(rows.size() / (size_t)npad);
  if (dtype == RON_DTYPE_F16X3) {
    const int k = split_weight_exponent(rows);
    *oscale = ldexpf(1.f, -k);
    return block_rows(split_rows(rows, k), npad, K * 4);
  }
  return block_rows(cast_rows(rows, dtype), npad, K * (int64_t)dtype_size(dtype));
}

// HWIO [kh,kw,cin,cout] -> rows[n][(ky*kw+kx)*cin + c], n < npad (extra rows zero)
static inline void hwio_to_rows(const float* w, int kh, int kw, int cin, int cout, int npad, std::vector<float>* rows) {
  const int K = kh * kw * cin;
  rows->assign((size_t)npad * K, 0.f);
  for (int t = 0; t < kh * kw; ++t)
    for (int c = 0; c < cin; ++c)
      for (int n = 0; n < cout; ++n) (*rows)[(size_t)n * K + t * cin + c] = w[((size_t)t * cin + c) * cout + n];
}

}  // namespace ron
