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

// True in exactly ONE workgroup of the grid: the last to get here.  `counter` is zero before the first launch of the pass and the
// last workgroup leaves it zero (no fill per launch: a captured hipMemset replays once on this runtime, csrc/graph_fix.hip).  What
// the other workgroups stored before the call is visible to the last one after it.  One thread per workgroup fences: an agent-scope
// fence is an L2 write-back / invalidate on this multi-die part (csrc/small_graph_body.hpp) — callers keep grids of such passes
// to a few thousand workgroups (kMaxArriveBlocks) and take the two-launch route beyond.  Every thread of the workgroup calls it.
constexpr int64_t kMaxArriveBlocks = 128;
__device__ __forceinline__ bool last_block(unsigned* counter) {
  __shared__ unsigned s_prev;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    s_prev = atomicAdd(counter, 1u);
  }
  __syncthreads();
  if (s_prev != gridDim.x * gridDim.y * gridDim.z - 1u) return false;
  if (threadIdx.x == 0) {
    *counter = 0u;
    __threadfence();
  }
  __syncthreads();
  return true;
}

// The 256 threads of the last workgroup add the n per-workgroup partial sums (float64) in the order of the single-workgroup
// "final" kernels they replace — thread t takes t, t + 256, ...; fixed tree — so the result is bit-identical to the two-launch route.
__device__ __forceinline__ double sum_partials_256(const double* partial, int64_t n) {
  __shared__ double red_final[256];
  double s = 0.0;
  for (int64_t b = threadIdx.x; b < n; b += 256) s += partial[b];
  __syncthreads();
  red_final[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (static_cast<int>(threadIdx.x) < st) red_final[threadIdx.x] += red_final[threadIdx.x + st];
    __syncthreads();
  }
  return red_final[0];
}

}  // namespace gnan

#define GNAN_REQUIRE(cond, ...)                                   \
  do {                                                            \
    if (!(cond)) return gnan::fail(GNAN_ERR_BAD_ARG, __VA_ARGS__); \
  } while (0)
