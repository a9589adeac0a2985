// mz_host.h -- host-side helpers shared by the translation units of libmzsearch.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>

#include "../../include/mzsearch.h"

namespace mzh {
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
