// Dense node_distances / normalization_matrix  ->  uint8 hop codes + per-row shell counts (gfx950).
//
// The reference feeds two dense fp32 N x N matrices whose entries are piecewise constant over hop
// shells (pre_process_datasets.py:112-121).  One workgroup per row re-derives that structure:
// pass 1 turns nd into codes and histograms them in LDS, pass 2 (optional) checks that the supplied
// normalisation matrix really is the shell count.  Integer work: results are bit-exact.
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void dense_to_code_kernel(const float* __restrict__ nd,
                                                            const float* __restrict__ norm, int64_t n_cols,
                                                            int64_t in_stride, uint8_t* __restrict__ code,
                                                            int32_t* __restrict__ cnt, int32_t* __restrict__ status) {
  __shared__ int hist[GNAN_MAX_CODES];
  __shared__ int flags, max_hop;
  const int64_t i = blockIdx.x;
  for (int t = threadIdx.x; t < GNAN_MAX_CODES; t += blockDim.x) hist[t] = 0;
  if (threadIdx.x == 0) { flags = 0; max_hop = 0; }
  __syncthreads();
  const float* row = nd + i * in_stride;
  int my_flags = 0, my_max = 0;
  for (int64_t j = threadIdx.x; j < n_cols; j += blockDim.x) {
    const float u = row[j];
    int c;
    if (u == 0.0f) {
      c = GNAN_MAX_CODES - 1;
    } else {
      const float h = rintf(1.0f / u);                       // hop + 1
      const bool ok = h >= 1.0f && h <= 255.0f && (1.0f / h) == u;  // nd must be float32(1/(1+hop))
      if (!ok) my_flags |= 1;
      c = ok ? static_cast<int>(h) - 1 : GNAN_MAX_CODES - 1;
      my_max = c > my_max && ok ? c : my_max;
    }
    code[i * n_cols + j] = static_cast<uint8_t>(c);
    atomicAdd(&hist[c], 1);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < GNAN_MAX_CODES; t += blockDim.x) cnt[i * GNAN_MAX_CODES + t] = hist[t];
  if (norm) {
    const float* nrow = norm + i * in_stride;
    for (int64_t j = threadIdx.x; j < n_cols; j += blockDim.x) {
      const int c = code[i * n_cols + j];
      if (nrow[j] != static_cast<float>(hist[c])) my_flags |= 2;
    }
  }
  if (my_flags) atomicOr(&flags, my_flags);
  if (my_max) atomicMax(&max_hop, my_max);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (flags) atomicOr(&status[0], flags);
    atomicMax(&status[1], max_hop);
  }
}

}  // namespace

extern "C" int gnan_dense_to_code(const float* nd, const float* norm, int64_t n_rows, int64_t n_cols,
                                  int64_t in_stride, uint8_t* code, int32_t* cnt, int32_t* status,
                                  gnan_stream_t stream) {
  GNAN_REQUIRE(n_rows >= 0 && n_cols >= 0, "dense_to_code: negative size");
  if (n_rows == 0) return GNAN_OK;
  GNAN_REQUIRE(nd && code && cnt && status, "dense_to_code: null pointer");
  GNAN_REQUIRE(in_stride >= n_cols, "dense_to_code: row stride smaller than n_cols");
  GNAN_REQUIRE(n_rows <= 0x7fffffffLL, "dense_to_code: too many rows for one launch");
  hipLaunchKernelGGL(dense_to_code_kernel, dim3(static_cast<unsigned>(n_rows)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), nd, norm, n_cols, in_stride, code, cnt, status);
  return gnan::check_launch("dense_to_code_kernel");
}

// ---- library-wide entry points -------------------------------------------------------------
namespace gnan {
char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace gnan

extern "C" int gnan_abi_version(void) { return GNAN_ABI_VERSION; }
extern "C" const char* gnan_last_error(void) { return gnan::last_error_buf(); }
