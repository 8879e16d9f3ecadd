// hip_check.h -- error plumbing shared by the translation units of libsvdss_hip.so: the message of the last failed HIP
// call of this thread (svdss_last_hip_error) and the macro that records it and turns it into a SVDSS_E* code.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
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

// A non-blocking stream of the library -- restricted to a range of the device's compute units when the environment
// variable `env` holds "first,count" (bits [first, first + count) of hipExtStreamCreateWithCUMask's mask).  On gfx950
// bit b stands for CU b / 8 of XCD b % 8 (tools/cu_mask_probe.hip), so a contiguous range is the same share of every
// XCD.  What for: the search kernel and the call-side DP kernels each fill the chip with long-lived wavefronts; side by
// side on all CUs they take turns, on disjoint CUs they run at the same time (SVDSS_SEARCH_CUS /
// SVDSS_CALL_CUS).  NOTE: hipExtStreamCreateWithCUMask has no flags argument -- a masked stream is a *blocking* stream
// (it synchronises with work on the null stream: torch's default stream, a plain hipMemcpy), unlike the non-blocking
// stream returned without the variable: the knob is a developer measurement aid, and what it measured (HISTORY.md 5a:
// CU partitioning never wins) was measured with no null-stream work in flight.
inline hipError_t svdss_make_stream(hipStream_t* st, const char* env) {
  int first = 0, count = 0;
  const char* e = env ? getenv(env) : nullptr;
  if (e && sscanf(e, "%d,%d", &first, &count) == 2 && count > 0 && first >= 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
      const int n_cu = prop.multiProcessorCount;
      uint32_t mask[32] = {0};
      const int words = (n_cu + 31) / 32;
      int set = 0;
      for (int b = first; b < first + count && b < n_cu && b < 1024; ++b) { mask[b / 32] |= 1u << (b % 32); ++set; }
      if (set > 0 && words <= 32) return hipExtStreamCreateWithCUMask(st, (uint32_t)words, mask);
    }
  }
  return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}
