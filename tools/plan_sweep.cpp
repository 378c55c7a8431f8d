// Dry-run sweep of the host planners of libron_hip under AddressSanitizer / UBSan (make -C ron_tensorflow_amd/csrc asan).
//
// With RON_PLAN_ONLY=1 (csrc/common.h) the library makes every host-side decision and no HIP call.  For every
// (variant, dtype, head plan, max_batch) below: ron_create -> ron_load_weight (constant weights) -> ron_finalize_weights, which builds
// the graph (describe_conv geometry), the grouped launch plans (plan_groups) and, for EVERY batch 1..max_batch, the split-K plans and
// scratch sizes (conv_pick_cfg, pick_pos_major, conv_group_plan -> group_splitks_scheduled); then ron_detect at a few batch sizes,
// which walks launch_conv / launch_conv_group up to the launch itself (tile counts, split-K slices, tile order, entry lists, the
// post-processing's workspace layout) and ron_clone (a second slot taking over the plans).  Index tables overrun or integer
// overflow in any of it ends the run with a sanitizer report and a non-zero exit code.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../include/ron_hip.h"

// The objects of this binary are compiled host-only: the registration hooks hipcc emits per translation unit must not reach the HIP
// runtime (there is no device code to register, and no GPU where this runs).  Defined here, they take precedence over libamdhip64's.
extern "C" {
void** __hipRegisterFatBinary(const void*) { static void* handle = nullptr; return &handle; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipRegisterManagedVar(void*, void**, void*, const char*, size_t, unsigned) {}
}

#define CHECK(expr)                                                                        \
  do {                                                                                     \
    const int rc_ = (expr);                                                                \
    if (rc_ != 0) {                                                                        \
      fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, ron_last_error());                     \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)

static int run(int variant, int dtype, uint32_t flags, int max_batch, bool detect) {
  ron_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.variant = variant; cfg.dtype = dtype;
  cfg.img_h = cfg.img_w = variant == RON_VARIANT_SSD512 ? 512 : 320;
  cfg.num_classes = 21; cfg.max_batch = max_batch; cfg.device = 0; cfg.flags = flags;
  ron_ctx* c = nullptr;
  CHECK(ron_create(&c, &cfg));
  const int nv = ron_num_variables(c);
  std::vector<float> buf;
  for (int i = 0; i < nv; ++i) {
    const char* name = nullptr;
    int64_t shape[4] = {0, 0, 0, 0};
    int nd = 0;
    CHECK(ron_variable_info(c, i, &name, shape, &nd));
    size_t n = 1;
    for (int k = 0; k < nd; ++k) n *= (size_t)shape[k];
    if (buf.size() < n) buf.resize(n);
    const std::string s = name;
    const float v = s.find("moving_variance") != std::string::npos || s.find("gamma") != std::string::npos ? 1.f : 0.01f;
    for (size_t k = 0; k < n; ++k) buf[k] = v;
    CHECK(ron_load_weight(c, name, buf.data(), shape, nd));
  }
  CHECK(ron_finalize_weights(c));
  if (detect) {
    ron_post_cfg pc;
    memset(&pc, 0, sizeof(pc));
    pc.objectness_thres = 0.03f; pc.select_threshold = 0.01f; pc.nms_threshold = 0.45f; pc.top_k = 400;
    pc.bbox_img[2] = pc.bbox_img[3] = 1.f;
    pc.prior_scaling[0] = pc.prior_scaling[1] = 0.1f; pc.prior_scaling[2] = pc.prior_scaling[3] = 0.2f;
    // fake device addresses (never dereferenced in a dry run), sized like the real buffers
    ron_detections det;
    memset(&det, 0, sizeof(det));
    det.capacity = 400;
    det.classes = (int32_t*)0x7000000000ull; det.scores = (float*)0x7100000000ull; det.bboxes = (float*)0x7200000000ull;
    det.anchor_index = (int32_t*)0x7300000000ull; det.count = (int32_t*)0x7400000000ull;
    const float* images = (const float*)0x7500000000ull;
    const int batches[4] = {1, (max_batch + 1) / 2, max_batch > 1 ? max_batch - 1 : 1, max_batch};
    for (int b : batches) CHECK(ron_detect(c, images, b, &pc, &det, nullptr));
    ron_ctx* slot = nullptr;
    CHECK(ron_clone(c, &slot));
    CHECK(ron_detect(slot, images, max_batch, &pc, &det, nullptr));
    CHECK(ron_destroy(slot));
  }
  CHECK(ron_destroy(c));
  return 0;
}

int main(int argc, char** argv) {
  if (getenv("RON_PLAN_ONLY") == nullptr) {
    fprintf(stderr, "plan_sweep: run with RON_PLAN_ONLY=1 (a dry run: this binary holds no device code)\n");
    return 2;
  }
  const bool quick = argc > 1 && strcmp(argv[1], "--quick") == 0;
  const int batches_all[] = {1, 2, 3, 4, 6, 8, 12, 13, 16, 23, 24, 32, 48, 64};
  const uint32_t plans[] = {0u, RON_CFG_LEVEL_GROUPS, RON_CFG_BATCH_GROUPS, RON_CFG_NO_GROUPS, RON_CFG_NO_HALO_SKIP};
  int runs = 0;
  // reducedfc and SSD-512: every head plan x every batch size of the ladder (the plan tables change at 12 / 13 and 23 / 24)
  for (int variant : {(int)RON_VARIANT_REDUCEDFC, (int)RON_VARIANT_SSD512}) {
    for (uint32_t plan : plans) {
      if (variant == RON_VARIANT_SSD512 && (plan == RON_CFG_LEVEL_GROUPS || plan == RON_CFG_BATCH_GROUPS)) continue;
      for (int mb : batches_all) {
        if (quick && mb != 1 && mb != 13 && mb != 32) continue;
        if (variant == RON_VARIANT_SSD512 && mb > 32) continue;       // (conv1_x of 64 images at 512 x 512 is beyond 4 GiB)
        for (int dtype : {(int)RON_DTYPE_BF16, (int)RON_DTYPE_F16X3}) {
          if (dtype == RON_DTYPE_F16X3 && (plan != 0u || (mb != 1 && mb != 32))) continue;
          if (run(variant, dtype, RON_CFG_FUSE_POOLS | plan, mb, true)) return 1;
          ++runs;
        }
      }
    }
  }
  // the full VGG-16 variant (229 M parameters to fold and pack per context): the three plan regimes, fp32 once
  for (int mb : {1, 13, 32}) {
    if (quick && mb != 32) continue;
    if (run(RON_VARIANT_FULL, RON_DTYPE_BF16, RON_CFG_FUSE_POOLS, mb, true)) return 1;
    ++runs;
  }
  if (!quick) {
    if (run(RON_VARIANT_REDUCEDFC, RON_DTYPE_F32, 0u, 4, true)) return 1;
    if (run(RON_VARIANT_REDUCEDFC, RON_DTYPE_F16, RON_CFG_FUSE_POOLS | RON_CFG_NO_STEM2, 32, true)) return 1;
    runs += 2;
  }
  printf("plan_sweep: %d contexts planned, no sanitizer report\n", runs);
  return 0;
}
