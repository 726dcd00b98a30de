// All-pairs hop distances on the GPU -> dense uint8 hop codes + per-row shell counts (gfx950).
//
// Replaces, for graphs small enough to hold N x N bytes, what pre_process_datasets.py:104-142 does on the host
// (scipy Dijkstra over the directed unit-weight adjacency, then N^2 Python lambda calls to count equal entries):
//   code[i, j] = hop(i -> j)   (255 if unreachable or beyond max_hops)        cnt[i, d] = #{ j : code[i, j] == d }
// One workgroup per source walks a level-synchronous BFS: frontier queues and the visited bitmap live in a
// per-workgroup slice of the caller's workspace (L2-resident), the output row doubles as the distance array.
// Integer work: bit-exact against the reference's matrices (node_distances = 1/(1+code), unreachable -> 0).
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void bfs_dense_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                        int32_t n, int32_t max_hops, uint8_t* __restrict__ code,
                                                        int32_t* __restrict__ cnt, int32_t* __restrict__ ws,
                                                        int32_t* __restrict__ status) {
  __shared__ int hist[GNAN_MAX_CODES];
  __shared__ int cur_n, next_n, overflow;
  const int tid = threadIdx.x;
  const int words = (n + 31) / 32;
  int32_t* q0 = ws + static_cast<int64_t>(blockIdx.x) * (2 * static_cast<int64_t>(n) + words);
  int32_t* q1 = q0 + n;
  unsigned* seen = reinterpret_cast<unsigned*>(q1 + n);
  if (tid == 0) overflow = 0;
  for (int src = blockIdx.x; src < n; src += gridDim.x) {
    uint8_t* row = code + static_cast<int64_t>(src) * n;
    for (int j = tid; j < n; j += 256) row[j] = 255;
    for (int j = tid; j < words; j += 256) seen[j] = 0u;
    for (int j = tid; j < GNAN_MAX_CODES; j += 256) hist[j] = 0;
    __syncthreads();
    if (tid == 0) {
      row[src] = 0;
      seen[src >> 5] = 1u << (src & 31);
      q0[0] = src;
      cur_n = 1;
      next_n = 0;
    }
    __syncthreads();
    int32_t* cur = q0;
    int32_t* nxt = q1;
    int level = 0;
    while (cur_n > 0 && level < max_hops) {
      const int cn = cur_n;
      const uint8_t lvl = static_cast<uint8_t>(level + 1 < 255 ? level + 1 : 255);
      if (level + 1 >= 255 && tid == 0) overflow = 1;          // hop 255 collides with the "unreachable" code
      for (int idx = tid; idx < cn; idx += 256) {
        const int u = cur[idx];
        for (int e = rowptr[u]; e < rowptr[u + 1]; ++e) {
          const int v = col[e];
          const unsigned bit = 1u << (v & 31);
          if (!(atomicOr(&seen[v >> 5], bit) & bit)) {          // first visit wins
            row[v] = lvl;
            nxt[atomicAdd(&next_n, 1)] = v;
          }
        }
      }
      __syncthreads();
      if (tid == 0) {
        cur_n = next_n;
        next_n = 0;
      }
      int32_t* t = cur; cur = nxt; nxt = t;
      ++level;
      __syncthreads();
    }
    for (int j = tid; j < n; j += 256) atomicAdd(&hist[row[j]], 1);
    __syncthreads();
    for (int j = tid; j < GNAN_MAX_CODES; j += 256) cnt[static_cast<int64_t>(src) * GNAN_MAX_CODES + j] = hist[j];
    if (tid == 0) {
      int mx = 0;
      for (int d = 0; d < GNAN_MAX_CODES - 1; ++d) mx = hist[d] ? d : mx;
      atomicMax(&status[1], mx);
      if (overflow) atomicOr(&status[0], 1);
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" size_t gnan_bfs_dense_workspace_bytes(int32_t n) {
  const size_t blocks = n < 1024 ? n : 1024;
  return blocks * (2 * static_cast<size_t>(n) + (n + 31) / 32) * sizeof(int32_t);
}

extern "C" int gnan_bfs_dense(const int32_t* rowptr, const int32_t* col, int32_t n, int32_t max_hops, uint8_t* code,
                              int32_t* cnt, int32_t* status, void* workspace, size_t workspace_bytes,
                              gnan_stream_t stream) {
  GNAN_REQUIRE(n >= 0, "bfs_dense: negative size");
  if (n == 0) return GNAN_OK;
  GNAN_REQUIRE(rowptr && col && code && cnt && status && workspace, "bfs_dense: null pointer");
  GNAN_REQUIRE(max_hops >= 0, "bfs_dense: max_hops must be >= 0");
  if (workspace_bytes < gnan_bfs_dense_workspace_bytes(n))
    return gnan::fail(GNAN_ERR_WORKSPACE, "bfs_dense: workspace %zu B < required %zu B", workspace_bytes,
                      gnan_bfs_dense_workspace_bytes(n));
  const int blocks = n < 1024 ? n : 1024;
  hipLaunchKernelGGL(bfs_dense_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), rowptr, col, n,
                     max_hops > 254 ? 254 : max_hops, code, cnt, static_cast<int32_t*>(workspace), status);
  return gnan::check_launch("bfs_dense_kernel");
}
