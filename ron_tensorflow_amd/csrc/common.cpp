#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace ron {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }

bool plan_only() {
  static const bool on = getenv("RON_PLAN_ONLY") != nullptr;
  return on;
}
// dry run: addresses from a counter (4 KB apart at least, never handed to HIP, never dereferenced on the host)
static std::atomic<uint64_t> g_fake_next{0x100000000ull};
hipError_t dev_malloc(void** p, size_t bytes) {
  if (!plan_only()) return hipMalloc(p, bytes);
  *p = reinterpret_cast<void*>(g_fake_next.fetch_add((uint64_t)align_up((int64_t)bytes + 1, 4096)));
  return hipSuccess;
}
hipError_t dev_free(void* p) { return plan_only() ? hipSuccess : hipFree(p); }
hipError_t dev_memset(void* p, int v, size_t bytes) { return plan_only() ? hipSuccess : hipMemset(p, v, bytes); }
hipError_t dev_memset_async(void* p, int v, size_t bytes, hipStream_t s) { return plan_only() ? hipSuccess : hipMemsetAsync(p, v, bytes, s); }
hipError_t dev_memcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  return plan_only() ? hipSuccess : hipMemcpy(dst, src, bytes, kind);
}
hipError_t dev_set_device(int device) { return plan_only() ? hipSuccess : hipSetDevice(device); }
hipError_t launch_error() { return plan_only() ? hipSuccess : hipGetLastError(); }
}  // namespace ron

extern "C" const char* ron_last_error(void) { return ron::get_error(); }
extern "C" int ron_abi_version(void) { return 2; }   // 2: ron_conv_desc.center_from, RON_DTYPE_F16X3, ron_num_grouped_launches

// CRC32C (Castagnoli, reflected 0x82F63B78), slicing-by-8 on the host: the checksum of TensorFlow's tensor-bundle files
// (ron_tensorflow_amd/checkpoint.py); `crc` chains calls (0 for the first).
extern "C" uint32_t ron_crc32c(const void* data, uint64_t nbytes, uint32_t crc) {
  static uint32_t tab[8][256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) tab[t][i] = (tab[t - 1][i] >> 8) ^ tab[0][tab[t - 1][i] & 0xFF];
    ready = true;
  }
  const unsigned char* p = static_cast<const unsigned char*>(data);
  crc = ~crc;
  while (nbytes >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= crc;
    crc = tab[7][lo & 0xFF] ^ tab[6][(lo >> 8) & 0xFF] ^ tab[5][(lo >> 16) & 0xFF] ^ tab[4][lo >> 24] ^
          tab[3][hi & 0xFF] ^ tab[2][(hi >> 8) & 0xFF] ^ tab[1][(hi >> 16) & 0xFF] ^ tab[0][hi >> 24];
    p += 8;
    nbytes -= 8;
  }
  while (nbytes--) crc = tab[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
  return ~crc;
}
