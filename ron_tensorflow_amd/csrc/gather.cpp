// The path's one exchange step from C (`ncclAllGather` of the fixed-size detection records over xGMI): a C host that
// shards a batch over the GPUs of a node packs each rank's detections (ron_pack_records) and gathers them with ONE in-place-capable
// all-gather on the stream of its choice.  The Python host does the same through torch.distributed (parallel.gather_detections).
//
// RCCL is not a link-time dependency of libron_hip.so (a single-GPU host never loads it): ncclAllGather is looked up at the first
// call - in the process's global scope first (a C host that links librccl), then in an RCCL that is already loaded under its
// soname (e.g. the one PyTorch brought), then by loading librccl.so.1.  The communicator must come from that same library.
#include <dlfcn.h>
#include <stddef.h>

#include "common.h"

namespace {
typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, hipStream_t);
constexpr int kNcclFloat32 = 7;      // rccl.h: ncclFloat32 = 7

AllGatherFn find_all_gather() {
  if (void* f = dlsym(RTLD_DEFAULT, "ncclAllGather")) return reinterpret_cast<AllGatherFn>(f);
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    if (void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))
      if (void* f = dlsym(h, "ncclAllGather")) return reinterpret_cast<AllGatherFn>(f);
  }
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))
      if (void* f = dlsym(h, "ncclAllGather")) return reinterpret_cast<AllGatherFn>(f);
  }
  return nullptr;
}
}  // namespace

extern "C" int ron_gather_records(const float* records, int n, int capacity, float* gathered, void* nccl_comm, void* stream) {
  RON_REQUIRE(records != nullptr && gathered != nullptr && nccl_comm != nullptr, "ron_gather_records: NULL argument");
  RON_REQUIRE(n > 0 && capacity > 0, "ron_gather_records: %d images of capacity %d", n, capacity);
  static AllGatherFn all_gather = find_all_gather();
  if (all_gather == nullptr) {
    ron::set_error("ron_gather_records: no RCCL in this process (ncclAllGather not found: link or load librccl.so.1)");
    return RON_ERR_UNSUPPORTED;
  }
  const size_t count = (size_t)n * (size_t)(capacity + 1) * 7;        // floats per rank: [n][capacity + 1][7], ron_pack_records
  const int rc = all_gather(records, gathered, count, kNcclFloat32, nccl_comm, (hipStream_t)stream);
  if (rc != 0) { ron::set_error("ron_gather_records: ncclAllGather returned %d", rc); return RON_ERR_HIP; }
  return RON_OK;
}
