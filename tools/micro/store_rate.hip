// Micro-benchmark: how fast can a CU store a 256 x 256 bf16 output tile (128 KB)?  The conv epilogue of the four-wave tile takes
// 12.5 k cycles for it (tools/experiments/tile_phase_stamps.patch).  Variants: plain / nontemporal stores, 4 or 8 waves,
// 256-byte row segments (the epilogue's shape) or 1 KB contiguous per instruction, 256 or 64 workgroups (chip-wide or per-CU limit?).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/store_rate.hip -o /tmp/store_rate && /tmp/store_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int NT, int SHAPE>
__global__ void k(u32x4* out, unsigned long long* cyc, int row_stride16) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
  // tile = 256 rows x 512 bytes (32 x 16-byte units); row r of workgroup b at (b * 256 + r) * row_stride16 units
  const size_t base = (size_t)blockIdx.x * 256 * row_stride16;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  const int per_wave_rows = 256 / nw * 2;      // each wave covers (256 / nw * 2) half-rows of 256 bytes... see below
  if (SHAPE == 0) {
    // epilogue shape: a wave owns a 128-column half (256 bytes) of rows; instruction = 4 rows x 256 B (lanes 0-15 one row)
    const int half = wave & 1, rgrp = wave >> 1, ngrp = nw >> 1;
    for (int i = 0; i < 256 / ngrp / 4; ++i) {
      const int r = (rgrp * (256 / ngrp / 4) + i) * 4 + (lane >> 4);
      u32x4* p = out + base + (size_t)r * row_stride16 + half * 16 + (lane & 15);
      if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
  } else {
    // 1 KB contiguous per instruction: 2 full rows of 512 B
    for (int i = 0; i < 128 / nw; ++i) {
      const int r = (wave * (128 / nw) + i) * 2 + (lane >> 5);
      u32x4* p = out + base + (size_t)r * row_stride16 + (lane & 31);
      if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_readcyclecounter();
  (void)per_wave_rows;
  if (tid == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = t2 - t0; }
}

template <int NT, int SHAPE>
void run(const char* name, int nwg, int threads, u32x4* out, unsigned long long* dcyc, int row_stride16) {
  std::vector<unsigned long long> h(nwg * 2);
  double best_issue = 1e18, best_all = 1e18;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL((k<NT, SHAPE>), dim3(nwg), dim3(threads), 0, 0, out, dcyc, row_stride16);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), dcyc, nwg * 16, hipMemcpyDeviceToHost);
    std::vector<double> a, b;
    for (int i = 0; i < nwg; ++i) { a.push_back((double)h[2 * i]); b.push_back((double)h[2 * i + 1]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    best_issue = std::min(best_issue, a[nwg / 2]); best_all = std::min(best_all, b[nwg / 2]);
  }
  printf("%-44s wgs %3d threads %3d : issue %7.0f cycles, with drain %7.0f  (%.1f B/cycle/CU)\n", name, nwg, threads, best_issue, best_all,
         131072.0 / best_all);
}

int main() {
  const int row_stride16 = 64;          // a 512-channel bf16 map: 1 KB per pixel
  u32x4* out; unsigned long long* dcyc;
  hipMalloc(&out, (size_t)256 * 256 * row_stride16 * 16);
  hipMalloc(&dcyc, 256 * 16);
  for (int nwg : {256, 64, 8}) {
    run<0, 0>("plain, 4 rows x 256 B per instruction", nwg, 256, out, dcyc, row_stride16);
    run<1, 0>("nontemporal, 4 rows x 256 B", nwg, 256, out, dcyc, row_stride16);
    run<0, 1>("plain, 2 rows x 512 B (1 KB contiguous)", nwg, 256, out, dcyc, row_stride16);
    run<1, 1>("nontemporal, 2 rows x 512 B", nwg, 256, out, dcyc, row_stride16);
    run<0, 0>("plain, 4 rows x 256 B, 8 waves", nwg, 512, out, dcyc, row_stride16);
  }
  return 0;
}
