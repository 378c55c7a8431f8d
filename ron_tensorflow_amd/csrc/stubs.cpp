// Entry points declared in ron_hip.h that are not implemented yet fail loudly.
#include "common.h"
#define RON_STUB(name, ...)                                   \
  extern "C" int name(__VA_ARGS__) {                          \
    ron::set_error(#name " is not implemented yet");          \
    return RON_ERR_UNSUPPORTED;                               \
  }
extern "C" int64_t ron_post_tfe_workspace_bytes(const ron_heads*, int) { return -1; }
RON_STUB(ron_post_tfe, const ron_heads*, int, const ron_tfe_cfg*, void*, int64_t, float*, float*, void*)
