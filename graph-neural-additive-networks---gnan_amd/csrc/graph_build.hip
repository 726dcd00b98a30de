// The degree-sorted copy of a hop-coded CSR (HopGraph.degree_sorted_copy: what the aggregation kernels walk) as library
// kernels — a stable radix sort of the rows by their number of listed pairs, a scan, one copy pass — instead of a chain of
// framework sorts / bincounts / cumsums / gathers (graph.py, rounds 1-5: 60-130 ms per graph warm, 234 ms in a fresh process
// on the driver's box; the framework's kernels are loaded lazily, these sit in libgnan_hip.so).  Index work only, bit-exact.
#include "common.hpp"

#include <cstdint>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace {

using gnan::kWave;

__device__ __forceinline__ int64_t load_ptr(const void* rowptr, int is64, int64_t i) {
  return is64 ? static_cast<const int64_t*>(rowptr)[i] : static_cast<int64_t>(static_cast<const int32_t*>(rowptr)[i]);
}

__global__ __launch_bounds__(256) void degree_keys_kernel(const void* rowptr, int is64, int64_t n, unsigned* keys, int32_t* ids) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  keys[i] = static_cast<unsigned>(load_ptr(rowptr, is64, i + 1) - load_ptr(rowptr, is64, i));
  ids[i] = static_cast<int32_t>(i);
}

// rowptr_s[q + 1] = inclusive sum of the sorted degrees (written in the caller's index width); rowptr_s[0] = 0
__global__ __launch_bounds__(256) void store_rowptr_kernel(const int64_t* incl, int64_t n, void* rowptr_s, int is64) {
  const int64_t q = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (q > n) return;
  const int64_t v = q == 0 ? 0 : incl[q - 1];
  if (is64) static_cast<int64_t*>(rowptr_s)[q] = v;
  else static_cast<int32_t*>(rowptr_s)[q] = static_cast<int32_t>(v);
}

struct CopyParams {
  int64_t n, nnz;
  const void* rowptr;
  int is64;
  const int32_t* col;
  const uint8_t* code;
  const int32_t* cnt;
  int D, pack_shift;
  const int32_t* order;
  const unsigned* deg_s;
  const int64_t* incl;
  int32_t* col_s;
  uint8_t* code_s;
  int32_t* colp_s;
  int32_t* cnt_s;
  unsigned long_threshold;      // rows with more pairs are copied by a whole wavefront (copy_long_rows_kernel)
};

__device__ __forceinline__ void copy_pair(const CopyParams& p, int64_t src, int64_t dst) {
  const int32_t c = p.col[src];
  const uint8_t d = p.code[src];
  p.col_s[dst] = c;
  p.code_s[dst] = d;
  if (p.colp_s) p.colp_s[dst] = static_cast<int32_t>(static_cast<unsigned>(c) | (static_cast<unsigned>(d) << p.pack_shift));
}

// one THREAD per sorted row (the rows of a wavefront have equal lengths: no divergence; their destinations are adjacent)
__global__ __launch_bounds__(256) void copy_rows_kernel(const CopyParams p) {
  const int64_t q = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (q >= p.n) return;
  const int64_t i = p.order[q];
  if (p.cnt_s) {
    for (int d = 0; d < p.D; ++d) p.cnt_s[q * p.D + d] = p.cnt[i * p.D + d];
  }
  const unsigned deg = p.deg_s[q];
  if (deg > p.long_threshold) return;
  const int64_t src = load_ptr(p.rowptr, p.is64, i), dst = q == 0 ? 0 : p.incl[q - 1];
  for (unsigned k = 0; k < deg; ++k) copy_pair(p, src + k, dst + k);
}

// the long rows sit at the END of the sorted order: waves walk backwards from the last row while the rows are long
__global__ __launch_bounds__(256) void copy_long_rows_kernel(const CopyParams p) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t waves = static_cast<int64_t>(gridDim.x) * (256 / kWave);
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * (256 / kWave) + threadIdx.x / kWave; w < p.n; w += waves) {
    const int64_t q = p.n - 1 - w;
    const unsigned deg = p.deg_s[q];
    if (deg <= p.long_threshold) return;                  // sorted ascending: every row before q is short too
    const int64_t src = load_ptr(p.rowptr, p.is64, p.order[q]), dst = q == 0 ? 0 : p.incl[q - 1];
    for (unsigned k = lane; k < deg; k += kWave) copy_pair(p, src + k, dst + k);
  }
}

struct ToI64 {
  __device__ int64_t operator()(unsigned v) const { return static_cast<int64_t>(v); }
};

struct Layout {
  size_t keys_in, keys_out, ids_in, incl, temp, temp_bytes, total;
};

Layout layout(int64_t n) {
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  Layout l{};
  size_t at = 0;
  l.keys_in = at; at += up(static_cast<size_t>(n) * 4);
  l.keys_out = at; at += up(static_cast<size_t>(n) * 4);
  l.ids_in = at; at += up(static_cast<size_t>(n) * 4);
  l.incl = at; at += up(static_cast<size_t>(n) * 8);
  size_t sort_bytes = 0, scan_bytes = 0;
  unsigned* kn = nullptr;
  int32_t* vn = nullptr;
  int64_t* on = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, kn, kn, vn, vn, static_cast<size_t>(n));
  auto in = rocprim::make_transform_iterator(kn, ToI64{});
  (void)rocprim::inclusive_scan(nullptr, scan_bytes, in, on, static_cast<size_t>(n), rocprim::plus<int64_t>());
  l.temp = at;
  l.temp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  at += up(l.temp_bytes);
  l.total = at;
  return l;
}

}  // namespace

extern "C" size_t gnan_degree_sorted_csr_workspace_bytes(int64_t n_rows) {
  if (n_rows <= 0) return 256;
  return layout(n_rows).total;
}

extern "C" int gnan_degree_sorted_csr(const gnan_sorted_csr_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "gnan_degree_sorted_csr: null arguments");
  GNAN_REQUIRE(a->n_rows >= 0 && a->n_rows < (int64_t{1} << 31) && a->nnz >= 0, "gnan_degree_sorted_csr: bad sizes");
  if (a->n_rows == 0) return GNAN_OK;
  GNAN_REQUIRE(a->rowptr && a->order && a->rowptr_s && (a->nnz == 0 || (a->col && a->code && a->col_s && a->code_s)),
               "gnan_degree_sorted_csr: null pointer");
  GNAN_REQUIRE(a->cnt_s == nullptr || (a->cnt && a->D >= 1), "gnan_degree_sorted_csr: cnt_s without cnt");
  GNAN_REQUIRE(a->colp_s == nullptr || (a->pack_shift > 0 && a->pack_shift < 32), "gnan_degree_sorted_csr: bad pack_shift");
  const int64_t n = a->n_rows;
  const Layout l = layout(n);
  GNAN_REQUIRE(a->workspace && a->workspace_bytes >= l.total && (reinterpret_cast<uintptr_t>(a->workspace) % 256) == 0,
               "gnan_degree_sorted_csr: workspace %zu B < required %zu B (256-byte aligned)", a->workspace_bytes, l.total);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(a->workspace);
  unsigned* keys_in = reinterpret_cast<unsigned*>(ws + l.keys_in);
  unsigned* keys_out = reinterpret_cast<unsigned*>(ws + l.keys_out);
  int32_t* ids_in = reinterpret_cast<int32_t*>(ws + l.ids_in);
  int64_t* incl = reinterpret_cast<int64_t*>(ws + l.incl);
  const unsigned blocks = static_cast<unsigned>((n + 255) / 256);
  hipLaunchKernelGGL(degree_keys_kernel, dim3(blocks), dim3(256), 0, st, a->rowptr, a->rowptr_is64, n, keys_in, ids_in);
  if (int rc = gnan::check_launch("degree_keys_kernel")) return rc;
  size_t temp_bytes = l.temp_bytes;
  // stable: rows of equal length keep their order (torch.argsort(deg, stable=True))
  hipError_t e = rocprim::radix_sort_pairs(ws + l.temp, temp_bytes, keys_in, keys_out, ids_in, a->order, static_cast<size_t>(n), 0, 32, st);
  if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "gnan_degree_sorted_csr: radix sort: %s", hipGetErrorString(e));
  temp_bytes = l.temp_bytes;
  auto in = rocprim::make_transform_iterator(keys_out, ToI64{});
  e = rocprim::inclusive_scan(ws + l.temp, temp_bytes, in, incl, static_cast<size_t>(n), rocprim::plus<int64_t>(), st);
  if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "gnan_degree_sorted_csr: scan: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(store_rowptr_kernel, dim3(static_cast<unsigned>((n + 1 + 255) / 256)), dim3(256), 0, st, incl, n, a->rowptr_s,
                     a->rowptr_is64);
  if (int rc = gnan::check_launch("store_rowptr_kernel")) return rc;
  CopyParams p{};
  p.n = n; p.nnz = a->nnz; p.rowptr = a->rowptr; p.is64 = a->rowptr_is64; p.col = a->col; p.code = a->code; p.cnt = a->cnt;
  p.D = a->D; p.pack_shift = a->pack_shift; p.order = a->order; p.deg_s = keys_out; p.incl = incl; p.col_s = a->col_s;
  p.code_s = a->code_s; p.colp_s = a->colp_s; p.cnt_s = a->cnt_s; p.long_threshold = 128;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(blocks), dim3(256), 0, st, p);
  if (int rc = gnan::check_launch("copy_rows_kernel")) return rc;
  hipLaunchKernelGGL(copy_long_rows_kernel, dim3(blocks < 4096 ? blocks : 4096), dim3(256), 0, st, p);
  return gnan::check_launch("copy_long_rows_kernel");
}
