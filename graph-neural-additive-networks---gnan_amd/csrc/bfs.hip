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

extern "C" int gnan_bfs_dense(const gnan_bfs_dense_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "bfs_dense: null args");
  const int32_t* rowptr = a->rowptr;
  const int32_t* col = a->col;
  const int32_t n = a->n, max_hops = a->max_hops;
  uint8_t* code = a->code;
  int32_t* cnt = a->cnt;
  int32_t* status = a->status;
  void* workspace = a->workspace;
  const size_t workspace_bytes = a->workspace_bytes;
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

// =============================================================================================
// K-hop truncated BFS -> hop-coded CSR, for graphs too large for N x N bytes (SURVEY f-1).
//
// The reference's preprocessing (pre_process_datasets.py:104-142) is all-pairs; truncated at K hops (every pair
// farther apart falls into the rest bucket, exactly what zeroing the dense matrices beyond K hops means) the lists
// stay short and the dense matrices are never formed.  Persistent workgroups; a workgroup owns a visited bitmap of
// N bits and ONE queue that holds the whole ball of the current source in BFS order (hop 0, hop 1, ...): expanding
// hop d walks queue[lvl_end[d-1] .. lvl_end[d]) — 8 lanes per frontier node stride over its adjacency — and appends
// first visits at the tail.  The queue then is the output row, and the list of bits to clear.
// Two passes with the same arguments: count (level sizes -> the caller's prefix sum), then fill.
// =============================================================================================
namespace {

struct KhopParams {
  const void* rowptr;
  int rowptr_is64;
  const int32_t* col;
  int64_t n;
  int max_hops;
  int64_t row_lo, row_hi;
  int32_t* level_cnt;        // [rows, max_hops + 1]
  const int64_t* out_rowptr; // fill pass only
  int32_t* out_col;
  uint8_t* out_code;
  int queue_cap;
  int32_t* status;
  int32_t* queues;           // [workgroups, queue_cap]
  unsigned* seen;            // [workgroups, words]
  int64_t words;
};

__device__ __forceinline__ int64_t adj_ptr(const KhopParams& p, int64_t i) {
  return p.rowptr_is64 ? static_cast<const int64_t*>(p.rowptr)[i] : static_cast<const int32_t*>(p.rowptr)[i];
}

__global__ __launch_bounds__(256) void bfs_khop_kernel(const KhopParams p) {
  __shared__ int tail, over;
  __shared__ int lvl_end[GNAN_MAX_CODES];
  const int tid = threadIdx.x;
  const int grp = tid >> 3, sub = tid & 7;
  int32_t* q = p.queues + static_cast<int64_t>(blockIdx.x) * p.queue_cap;
  unsigned* seen = p.seen + static_cast<int64_t>(blockIdx.x) * p.words;
  for (int64_t j = tid; j < p.words; j += 256) seen[j] = 0u;
  if (tid == 0) over = 0;
  __syncthreads();
  const int K = p.max_hops;
  for (int64_t r = p.row_lo + blockIdx.x; r < p.row_hi; r += gridDim.x) {
    if (tid == 0) {
      q[0] = static_cast<int32_t>(r);
      seen[r >> 5] = 1u << (r & 31);
      tail = 1;
      lvl_end[0] = 1;
    }
    int start = 0;
    for (int level = 1; level <= K; ++level) {
      __syncthreads();                                   // lvl_end[level - 1] is final
      const int end = lvl_end[level - 1];
      if (end == start) {                                // empty frontier: all farther levels are empty too
        if (tid == 0)
          for (int d = level; d <= K; ++d) lvl_end[d] = end;
        break;
      }
      for (int f = start + grp; f < end; f += 32) {
        const int u = q[f];
        const int64_t e1 = adj_ptr(p, u + 1);
        for (int64_t e = adj_ptr(p, u) + sub; e < e1; e += 8) {
          const int v = p.col[e];
          const unsigned bit = 1u << (v & 31);
          if (!(atomicOr(&seen[v >> 5], bit) & bit)) {   // first visit wins
            const int pos = atomicAdd(&tail, 1);
            if (pos < p.queue_cap) q[pos] = v; else over = 1;
          }
        }
      }
      __syncthreads();
      if (tid == 0) {
        if (tail > p.queue_cap) tail = p.queue_cap;
        lvl_end[level] = tail;
      }
      start = end;
    }
    __syncthreads();
    const int total = lvl_end[K];
    const int64_t local = r - p.row_lo;
    if (p.out_rowptr == nullptr) {
      for (int d = tid; d <= K; d += 256)
        p.level_cnt[local * (K + 1) + d] = lvl_end[d] - (d ? lvl_end[d - 1] : 0);
    } else {
      const int64_t base = p.out_rowptr[local];
      for (int t = tid; t < total; t += 256) {
        int d = 0;
        while (t >= lvl_end[d]) ++d;
        p.out_col[base + t] = q[t];
        p.out_code[base + t] = static_cast<uint8_t>(d);
      }
    }
    if (over) {                                          // bits of unrecorded nodes are set: wipe everything
      for (int64_t j = tid; j < p.words; j += 256) seen[j] = 0u;
    } else {
      for (int t = tid; t < total; t += 256) seen[q[t] >> 5] = 0u;   // every set bit of a word belongs to this row
    }
    __syncthreads();
  }
  if (tid == 0 && over) atomicOr(&p.status[0], 1);
}

}  // namespace

extern "C" size_t gnan_bfs_khop_workspace_bytes(int64_t n, int32_t queue_cap, int32_t n_workgroups) {
  if (n <= 0 || queue_cap <= 0 || n_workgroups <= 0) return 0;
  const size_t words = static_cast<size_t>((n + 31) / 32);
  return static_cast<size_t>(n_workgroups) * (static_cast<size_t>(queue_cap) + words) * sizeof(int32_t);
}

extern "C" int gnan_bfs_khop(const gnan_bfs_khop_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "bfs_khop: null args");
  const void* rowptr = a->rowptr;
  const int32_t rowptr_is64 = a->rowptr_is64, max_hops = a->max_hops, queue_cap = a->queue_cap, n_workgroups = a->n_workgroups;
  const int32_t* col = a->col;
  const int64_t n = a->n, row_lo = a->row_lo, row_hi = a->row_hi;
  int32_t* level_cnt = a->level_cnt;
  const int64_t* out_rowptr = a->out_rowptr;
  int32_t* out_col = a->out_col;
  uint8_t* out_code = a->out_code;
  int32_t* status = a->status;
  void* workspace = a->workspace;
  const size_t workspace_bytes = a->workspace_bytes;
  GNAN_REQUIRE(n >= 0 && n <= 0x7fffffffLL, "bfs_khop: n must fit int32 node ids");
  GNAN_REQUIRE(row_lo >= 0 && row_lo <= row_hi && row_hi <= n, "bfs_khop: bad row range");
  if (row_hi == row_lo) return GNAN_OK;
  GNAN_REQUIRE(max_hops >= 1 && max_hops <= GNAN_MAX_CODES - 2, "bfs_khop: max_hops must be in [1, %d]", GNAN_MAX_CODES - 2);
  GNAN_REQUIRE(rowptr && col && status && workspace, "bfs_khop: null pointer");
  GNAN_REQUIRE(queue_cap >= 1 && n_workgroups >= 1, "bfs_khop: queue_cap and n_workgroups must be >= 1");
  GNAN_REQUIRE(out_rowptr ? (out_col && out_code) : level_cnt != nullptr, "bfs_khop: count pass needs level_cnt, fill pass out_col/out_code");
  const size_t need = gnan_bfs_khop_workspace_bytes(n, queue_cap, n_workgroups);
  if (workspace_bytes < need)
    return gnan::fail(GNAN_ERR_WORKSPACE, "bfs_khop: workspace %zu B < required %zu B", workspace_bytes, need);
  KhopParams p;
  p.rowptr = rowptr; p.rowptr_is64 = rowptr_is64; p.col = col; p.n = n; p.max_hops = max_hops;
  p.row_lo = row_lo; p.row_hi = row_hi; p.level_cnt = level_cnt; p.out_rowptr = out_rowptr;
  p.out_col = out_col; p.out_code = out_code; p.queue_cap = queue_cap; p.status = status;
  p.words = (n + 31) / 32;
  p.queues = static_cast<int32_t*>(workspace);
  p.seen = reinterpret_cast<unsigned*>(p.queues + static_cast<int64_t>(n_workgroups) * queue_cap);
  const int64_t rows = row_hi - row_lo;
  const unsigned blocks = static_cast<unsigned>(rows < n_workgroups ? rows : n_workgroups);
  hipLaunchKernelGGL(bfs_khop_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return gnan::check_launch("bfs_khop_kernel");
}
