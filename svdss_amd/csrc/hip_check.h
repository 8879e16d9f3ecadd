// hip_check.h -- error plumbing shared by the translation units of libsvdss_hip.so: the message of the last failed HIP
// call of this thread (svdss_last_hip_error) and the macro that records it and turns it into a SVDSS_E* code.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/svdss_hip.h"

extern thread_local std::string g_svdss_hip_err;   // defined in index_api.hip

#define HIPCHK(expr)                                                              \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      g_svdss_hip_err = std::string(#expr) + ": " + hipGetErrorString(e_);        \
      return (e_ == hipErrorOutOfMemory) ? SVDSS_ENOMEM : SVDSS_EHIP;             \
    }                                                                             \
  } while (0)
