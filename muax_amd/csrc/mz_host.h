// mz_host.h -- host-side helpers shared by the translation units of libmzsearch.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <string>

#include "../../include/mzsearch.h"

namespace mz {
struct StepArgs;
struct JumpArgs;
}  // namespace mz
namespace mzh {
// internals of a handle for the translation units that launch their own kernels on its step-wise tree (defined in
// mz_api.hip): MZS_OK and the kernel argument blocks of the rooted tree with cached decisions, or an error (message set
// on the handle) when the handle has no such tree
int step_view(mzs_handle* h, mz::StepArgs* sa, mz::JumpArgs* ja, int* policy, const char* who, int* device = nullptr);
int fail_handle(mzs_handle* h, int code, const char* msg);
// message of the last failure of an entry point that has no handle (mzs_last_error(NULL)); defined in mz_api.hip
extern thread_local std::string g_create_error;
inline int fail_global(int code, const char* fmt, const char* a = "") {
  char buf[512];
  snprintf(buf, sizeof buf, fmt, a);
  g_create_error = buf;
  return code;
}
// per-device record of what hipFuncSetAttribute(MaxDynamicSharedMemorySize) has already granted ONE kernel (one static
// LdsGrant per kernel instance).  Atomic: host threads that race can at worst both set the attribute (idempotent);
// device ordinals beyond the table are never recorded, so the attribute is then set on every call (no aliasing)
struct LdsGrant {
  std::atomic<size_t> have[64];
  bool covers(int device, size_t lds) const {
    return device >= 0 && device < 64 && have[device].load(std::memory_order_acquire) >= lds;
  }
  void note(int device, size_t lds) {
    if (device < 0 || device >= 64) return;
    size_t cur = have[device].load(std::memory_order_relaxed);
    while (cur < lds && !have[device].compare_exchange_weak(cur, lds, std::memory_order_release)) {
    }
  }
};
}  // namespace mzh

#define MZS_HIPG(call)                                                                             \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) return mzh::fail_global(MZS_E_RUNTIME, #call ": %s", hipGetErrorString(e_)); \
  } while (0)
