// Shared host-side helpers for libgnan_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "gnan_hip.h"

namespace gnan {

char* last_error_buf();  // thread-local, 512 bytes

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GNAN_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return GNAN_OK;
}

constexpr int kWave = 64;  // gfx950 wavefront

}  // namespace gnan

#define GNAN_REQUIRE(cond, ...)                                   \
  do {                                                            \
    if (!(cond)) return gnan::fail(GNAN_ERR_BAD_ARG, __VA_ARGS__); \
  } while (0)
