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

// fp32 rows [npad][K] -> what the conv kernels read: rounded to the dtype, blocked
static inline std::vector<uint8_t> pack_conv_weights(const std::vector<float>& rows, int npad, int dtype) {
#ifdef RON_DIAG      // libron_hip_diag.so: the round-1 kernels of csrc/diag read row-major weights
  (void)npad;
  return cast_rows(rows, dtype);
#else
  const int64_t K = (int64_t)(rows.size() / (size_t)npad);
  return block_rows(cast_rows(rows, dtype), npad, K * (int64_t)dtype_size(dtype));
#endif
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
