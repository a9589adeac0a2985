// mz_host.h -- host-side helpers shared by the translation units of libmzsearch.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

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
int step_view(mzs_handle* h, mz::StepArgs* sa, mz::JumpArgs* ja, int* policy, const char* who);
int fail_handle(mzs_handle* h, int code, const char* msg);
// message of the last failure of an entry point that has no handle (mzs_last_error(NULL)); defined in mz_api.hip
extern thread_local std::string g_create_error;
inline int fail_global(int code, const char* fmt, const char* a = "") {
  char buf[512];
  snprintf(buf, sizeof buf, fmt, a);
  g_create_error = buf;
  return code;
}
}  // namespace mzh

#define MZS_HIPG(call)                                                                             \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) return mzh::fail_global(MZS_E_RUNTIME, #call ": %s", hipGetErrorString(e_)); \
  } while (0)
