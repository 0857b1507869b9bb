// Host-side helpers shared by the C-ABI translation units: per-thread error text and the
// HIP_TRY macro that turns a hipError_t into NDT_ERR_HIP without throwing.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace ndt {

inline std::string& last_error() {
  thread_local std::string e;
  return e;
}
inline void set_error(const char* msg) { last_error() = msg ? msg : ""; }

}  // namespace ndt

#define HIP_TRY(expr)                                                            \
  do {                                                                           \
    const hipError_t _e = (expr);                                                \
    if (_e != hipSuccess) {                                                      \
      ::ndt::last_error() = std::string(#expr) + ": " + hipGetErrorString(_e);   \
      (void)hipGetLastError();                                                   \
      return NDT_ERR_HIP;                                                        \
    }                                                                            \
  } while (0)
