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

// =============================================================================================
// The bucketed copy of the pairs behind gnan_spmm_pb_fwd (HopGraph.pb_plan), pair-level work as library kernels.
// Row-level arrays (slots, bins: n_rows-sized scans) stay with the caller; what touches every listed pair happens here:
//   gnan_pb_plan_rows   per row: how many pairs carry hop code 0, the column and position of the (last) one
//   gnan_pb_plan_keys   per pair: its tile (row bin, column block), its column inside the block, its accumulator inside the
//                       bin; tile histogram
//   gnan_pb_plan_fill   stable radix sort of the pairs by tile, scatter into the padded bin-major layout, pads, chunk offsets
// The framework route (graph.py: ~25 passes over 8-byte temporaries of nnz elements) took 124 ms warm and 270-420 ms in a
// fresh process on the 10M-node graph — mostly first-touch allocations of ~8 GB of temporaries.
// =============================================================================================
namespace {

constexpr int kPbLongRow = 256;       // rows with more pairs are walked by a whole workgroup (a thread alone would take a hub row's 10^5 pairs
                                      // one dependent load after the other: 100+ ms for the longest row of the 10M-node R-MAT graph)

__global__ __launch_bounds__(256) void pb_rows_kernel(const void* rowptr, int is64, const int32_t* col, const uint8_t* code, int64_t n,
                                                      int32_t* c0, int32_t* self_col, int32_t* self_pos) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t lo = load_ptr(rowptr, is64, i), hi = load_ptr(rowptr, is64, i + 1);
  if (hi - lo > kPbLongRow) return;                     // pb_rows_long_kernel
  int cnt = 0, sc = -1, sp = -1;
  for (int64_t e = lo; e < hi; ++e) {
    if (code[e] == 0) {
      ++cnt;
      sc = col[e];
      sp = static_cast<int>(e - lo);
    }
  }
  c0[i] = cnt;
  self_col[i] = sc;
  self_pos[i] = sp;
}

// one workgroup per long row: count of code-0 pairs and the position of the LAST one (as the serial walk finds it)
__global__ __launch_bounds__(256) void pb_rows_long_kernel(const void* rowptr, int is64, const int32_t* col, const uint8_t* code,
                                                           const int32_t* long_rows, int32_t* c0, int32_t* self_col, int32_t* self_pos) {
  __shared__ int s_cnt, s_pos;
  const int64_t i = long_rows[blockIdx.x];
  const int64_t lo = load_ptr(rowptr, is64, i), hi = load_ptr(rowptr, is64, i + 1);
  if (threadIdx.x == 0) {
    s_cnt = 0;
    s_pos = -1;
  }
  __syncthreads();
  int cnt = 0, sp = -1;
  for (int64_t e = lo + threadIdx.x; e < hi; e += 256) {
    if (code[e] == 0) {
      ++cnt;
      sp = static_cast<int>(e - lo);
    }
  }
  if (cnt) {
    atomicAdd(&s_cnt, cnt);
    atomicMax(&s_pos, sp);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    c0[i] = s_cnt;
    self_pos[i] = s_pos;
    self_col[i] = s_pos >= 0 ? col[lo + s_pos] : -1;
  }
}

struct KeyParams {
  const void* rowptr;
  int is64;
  const int32_t* col;
  const uint8_t* code;
  int64_t n;
  const int32_t* self_pos;     // null: no pair is left out
  int code_base, n_acc;
  const int32_t* slot_ptr;     // [n + 1]
  const int32_t* bin_of_row;   // [n]
  const int32_t* bin_slot0;    // [n_bins]
  int n_cb, cb_width;
  unsigned n_tiles;
  unsigned* key;
  unsigned* val;
  uint16_t* tmp_src;
  uint16_t* tmp_dst;
  unsigned* tile_cnt;          // [n_tiles + 1], zeroed by the caller (entry n_tiles stays 0)
};

__device__ __forceinline__ void pb_key_of_pair(const KeyParams& p, int64_t e, int e_local, int sp, int b, int s0, int slot) {
  unsigned key = p.n_tiles;
  if (e_local != sp) {
    const int c = p.col[e];
    const int cb = c / p.cb_width;
    key = static_cast<unsigned>(b) * static_cast<unsigned>(p.n_cb) + static_cast<unsigned>(cb);
    p.tmp_src[e] = static_cast<uint16_t>(c - cb * p.cb_width);
    p.tmp_dst[e] = static_cast<uint16_t>((s0 + slot) * p.n_acc + (static_cast<int>(p.code[e]) - p.code_base));
  }
  p.key[e] = key;
  p.val[e] = static_cast<unsigned>(e);
  if (key != p.n_tiles) atomicAdd(p.tile_cnt + key, 1u);     // (the pairs left out are not counted: one address for 10^7 atomics)
}

__global__ __launch_bounds__(256) void pb_keys_kernel(const KeyParams p) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= p.n) return;
  const int64_t lo = load_ptr(p.rowptr, p.is64, i), hi = load_ptr(p.rowptr, p.is64, i + 1);
  if (hi - lo > kPbLongRow) return;                     // pb_keys_long_kernel
  const int b = p.bin_of_row[i];
  const int s0 = p.slot_ptr[i] - p.bin_slot0[b];
  const int ns = p.slot_ptr[i + 1] - p.slot_ptr[i];
  const int sp = p.self_pos ? p.self_pos[i] : -1;
  int slot = 0;                                          // (rank among the kept pairs) % ns, without a division per pair
  for (int64_t e = lo; e < hi; ++e) {
    const int el = static_cast<int>(e - lo);
    pb_key_of_pair(p, e, el, sp, b, s0, slot);
    if (el != sp) slot = slot + 1 == ns ? 0 : slot + 1;
  }
}

__global__ __launch_bounds__(256) void pb_keys_long_kernel(const KeyParams p, const int32_t* long_rows) {
  const int64_t i = long_rows[blockIdx.x];
  const int64_t lo = load_ptr(p.rowptr, p.is64, i), hi = load_ptr(p.rowptr, p.is64, i + 1);
  const int b = p.bin_of_row[i];
  const int s0 = p.slot_ptr[i] - p.bin_slot0[b];
  const int ns = p.slot_ptr[i + 1] - p.slot_ptr[i];
  const int sp = p.self_pos ? p.self_pos[i] : -1;
  for (int64_t e = lo + threadIdx.x; e < hi; e += 256) {
    const int el = static_cast<int>(e - lo);
    const int k = el - ((sp >= 0 && el > sp) ? 1 : 0);   // rank among the kept pairs
    pb_key_of_pair(p, e, el, sp, b, s0, k % ns);
  }
}

// per tile: the pads behind its entries, and the offsets of its chunks in the column-block-major chunk list
__global__ __launch_bounds__(256) void pb_tiles_kernel(const int32_t* tile_ptr, const unsigned* tile_cnt, const int32_t* chunk_first, int n_bins, int n_cb,
                                                       uint16_t* src16, uint16_t* dst16, int dummy, int32_t* chunk_q) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;          // bin-major tile id
  if (t >= static_cast<int64_t>(n_bins) * n_cb) return;
  const int lo = tile_ptr[t], hi = tile_ptr[t + 1];
  for (int q = lo + static_cast<int>(tile_cnt[t]); q < hi; ++q) {
    src16[q] = 0;
    dst16[q] = static_cast<uint16_t>(dummy);
  }
  const int b = static_cast<int>(t / n_cb), cb = static_cast<int>(t % n_cb);
  int32_t* out = chunk_q + chunk_first[static_cast<int64_t>(cb) * n_bins + b];
  for (int q = lo, k = 0; q < hi; q += 16, ++k) out[k] = q;
}

__global__ __launch_bounds__(256) void pb_scatter_kernel(const unsigned* key_s, const unsigned* val_s, int64_t m, const int32_t* tile_ptr,
                                                         const int32_t* tile_start, const uint16_t* tmp_src, const uint16_t* tmp_dst,
                                                         uint16_t* src16, uint16_t* dst16) {
  const int64_t s = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (s >= m) return;
  const unsigned t = key_s[s];
  const unsigned e = val_s[s];
  const int64_t pos = static_cast<int64_t>(tile_ptr[t]) + (s - tile_start[t]);
  src16[pos] = tmp_src[e];
  dst16[pos] = tmp_dst[e];
}

struct FillLayout {
  size_t key_out, val_out, temp, temp_bytes, total;
};

FillLayout fill_layout(int64_t nnz, unsigned end_bit) {
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  FillLayout l{};
  size_t at = 0;
  l.key_out = at; at += up(static_cast<size_t>(nnz) * 4);
  l.val_out = at; at += up(static_cast<size_t>(nnz) * 4);
  unsigned* kn = nullptr;
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, kn, kn, kn, kn, static_cast<size_t>(nnz), 0, end_bit);
  l.temp = at;
  l.temp_bytes = bytes;
  at += up(bytes);
  l.total = at;
  return l;
}

unsigned bits_for(unsigned max_value) {
  unsigned b = 1;
  while (b < 32 && (max_value >> b) != 0) ++b;
  return b;
}

}  // namespace

extern "C" int32_t gnan_pb_plan_long_row_threshold(void) { return kPbLongRow; }

extern "C" int gnan_pb_plan_rows(const void* rowptr, int32_t rowptr_is64, const int32_t* col, const uint8_t* code, int64_t n_rows,
                                 const int32_t* long_rows, int32_t n_long, int32_t* c0, int32_t* self_col, int32_t* self_pos,
                                 gnan_stream_t stream) {
  GNAN_REQUIRE(n_rows >= 0 && n_rows < (int64_t{1} << 31), "gnan_pb_plan_rows: bad sizes");
  if (n_rows == 0) return GNAN_OK;
  GNAN_REQUIRE(rowptr && col && code && c0 && self_col && self_pos, "gnan_pb_plan_rows: null pointer");
  hipLaunchKernelGGL(pb_rows_kernel, dim3(static_cast<unsigned>((n_rows + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     rowptr, rowptr_is64, col, code, n_rows, c0, self_col, self_pos);
  if (int rc = gnan::check_launch("pb_rows_kernel")) return rc;
  GNAN_REQUIRE(n_long == 0 || long_rows, "gnan_pb_plan_rows: n_long without long_rows");
  if (n_long > 0) {
    hipLaunchKernelGGL(pb_rows_long_kernel, dim3(static_cast<unsigned>(n_long)), dim3(256), 0, static_cast<hipStream_t>(stream), rowptr,
                       rowptr_is64, col, code, long_rows, c0, self_col, self_pos);
    return gnan::check_launch("pb_rows_long_kernel");
  }
  return GNAN_OK;
}

extern "C" int gnan_pb_plan_keys(const gnan_pb_keys_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr && a->n_rows > 0 && a->n_rows < (int64_t{1} << 31), "gnan_pb_plan_keys: bad sizes");
  GNAN_REQUIRE(a->rowptr && a->col && a->code && a->slot_ptr && a->bin_of_row && a->bin_slot0 && a->key && a->val && a->tmp_src &&
                   a->tmp_dst && a->tile_cnt,
               "gnan_pb_plan_keys: null pointer");
  GNAN_REQUIRE(a->n_cb > 0 && a->cb_width > 0 && a->cb_width <= 65536 && a->n_acc >= 1 && a->n_tiles > 0 && a->n_tiles < 0x7fffffffu,
               "gnan_pb_plan_keys: bad plan parameters");
  KeyParams p{};
  p.rowptr = a->rowptr; p.is64 = a->rowptr_is64; p.col = a->col; p.code = a->code; p.n = a->n_rows; p.self_pos = a->self_pos;
  p.code_base = a->code_base; p.n_acc = a->n_acc; p.slot_ptr = a->slot_ptr; p.bin_of_row = a->bin_of_row; p.bin_slot0 = a->bin_slot0;
  p.n_cb = a->n_cb; p.cb_width = a->cb_width; p.n_tiles = a->n_tiles; p.key = a->key; p.val = a->val; p.tmp_src = a->tmp_src;
  p.tmp_dst = a->tmp_dst; p.tile_cnt = a->tile_cnt;
  hipLaunchKernelGGL(pb_keys_kernel, dim3(static_cast<unsigned>((a->n_rows + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  if (int rc = gnan::check_launch("pb_keys_kernel")) return rc;
  GNAN_REQUIRE(a->n_long == 0 || a->long_rows, "gnan_pb_plan_keys: n_long without long_rows");
  if (a->n_long > 0) {
    hipLaunchKernelGGL(pb_keys_long_kernel, dim3(static_cast<unsigned>(a->n_long)), dim3(256), 0, static_cast<hipStream_t>(stream), p, a->long_rows);
    return gnan::check_launch("pb_keys_long_kernel");
  }
  return GNAN_OK;
}

extern "C" size_t gnan_pb_plan_fill_workspace_bytes(int64_t nnz, uint32_t n_tiles) {
  if (nnz <= 0) return 256;
  return fill_layout(nnz, bits_for(n_tiles)).total;
}

extern "C" int gnan_pb_plan_fill(const gnan_pb_fill_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr && a->nnz > 0 && a->nnz < (int64_t{1} << 32) && a->n_kept >= 0 && a->n_kept <= a->nnz, "gnan_pb_plan_fill: bad sizes");
  GNAN_REQUIRE(a->key && a->val && a->tmp_src && a->tmp_dst && a->tile_ptr && a->tile_start && a->tile_cnt && a->chunk_first && a->src16 &&
                   a->dst16 && a->chunk_q && a->workspace,
               "gnan_pb_plan_fill: null pointer");
  const unsigned end_bit = bits_for(a->n_tiles);
  const FillLayout l = fill_layout(a->nnz, end_bit);
  GNAN_REQUIRE(a->workspace_bytes >= l.total && (reinterpret_cast<uintptr_t>(a->workspace) % 256) == 0,
               "gnan_pb_plan_fill: workspace %zu B < required %zu B (256-byte aligned)", a->workspace_bytes, l.total);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(a->workspace);
  unsigned* key_s = reinterpret_cast<unsigned*>(ws + l.key_out);
  unsigned* val_s = reinterpret_cast<unsigned*>(ws + l.val_out);
  size_t temp_bytes = l.temp_bytes;
  // stable: the pairs of a tile keep their CSR order (torch.sort(tile, stable=True)); the pairs left out carry key n_tiles: last
  hipError_t e = rocprim::radix_sort_pairs(ws + l.temp, temp_bytes, a->key, key_s, a->val, val_s, static_cast<size_t>(a->nnz), 0, end_bit, st);
  if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "gnan_pb_plan_fill: radix sort: %s", hipGetErrorString(e));
  const int64_t n_tiles = static_cast<int64_t>(a->n_bins) * a->n_cb;
  hipLaunchKernelGGL(pb_tiles_kernel, dim3(static_cast<unsigned>((n_tiles + 255) / 256)), dim3(256), 0, st, a->tile_ptr, a->tile_cnt,
                     a->chunk_first, a->n_bins, a->n_cb, a->src16, a->dst16, a->dummy, a->chunk_q);
  if (int rc = gnan::check_launch("pb_tiles_kernel")) return rc;
  if (a->n_kept > 0) {
    hipLaunchKernelGGL(pb_scatter_kernel, dim3(static_cast<unsigned>((a->n_kept + 255) / 256)), dim3(256), 0, st, key_s, val_s, a->n_kept,
                       a->tile_ptr, a->tile_start, a->tmp_src, a->tmp_dst, a->src16, a->dst16);
    return gnan::check_launch("pb_scatter_kernel");
  }
  return GNAN_OK;
}

// =============================================================================================
// The transposed adjacency (HopGraph.transposed: the backward's graph) by the library: rows <-> neighbours, the pairs of a
// neighbour in the order of the rows that list it (a STABLE sort by column id, as torch.argsort(col, stable=True) gives).
// =============================================================================================
namespace {

__global__ __launch_bounds__(256) void tr_rows_kernel(const void* rowptr, int is64, const int32_t* col, int64_t n, unsigned* key, unsigned* val,
                                                      unsigned* row_of, unsigned* col_cnt) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t lo = load_ptr(rowptr, is64, i), hi = load_ptr(rowptr, is64, i + 1);
  if (hi - lo > kPbLongRow) return;
  for (int64_t e = lo; e < hi; ++e) {
    const unsigned c = static_cast<unsigned>(col[e]);
    key[e] = c;
    val[e] = static_cast<unsigned>(e);
    row_of[e] = static_cast<unsigned>(i);
    atomicAdd(col_cnt + c, 1u);
  }
}

__global__ __launch_bounds__(256) void tr_rows_long_kernel(const void* rowptr, int is64, const int32_t* col, const int32_t* long_rows,
                                                           unsigned* key, unsigned* val, unsigned* row_of, unsigned* col_cnt) {
  const int64_t i = long_rows[blockIdx.x];
  const int64_t lo = load_ptr(rowptr, is64, i), hi = load_ptr(rowptr, is64, i + 1);
  for (int64_t e = lo + threadIdx.x; e < hi; e += 256) {
    const unsigned c = static_cast<unsigned>(col[e]);
    key[e] = c;
    val[e] = static_cast<unsigned>(e);
    row_of[e] = static_cast<unsigned>(i);
    atomicAdd(col_cnt + c, 1u);
  }
}

__global__ __launch_bounds__(256) void tr_gather_kernel(const unsigned* val_s, const unsigned* row_of, const uint8_t* code, int64_t nnz,
                                                        int32_t* col_t, uint8_t* code_t) {
  const int64_t s = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (s >= nnz) return;
  const unsigned e = val_s[s];
  col_t[s] = static_cast<int32_t>(row_of[e]);
  code_t[s] = code[e];
}

struct TrLayout {
  size_t key, val, row_of, key_out, val_out, cnt, incl, temp, temp_bytes, total;
};

TrLayout tr_layout(int64_t nnz, int64_t n_cols, unsigned end_bit) {
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  TrLayout l{};
  size_t at = 0;
  l.key = at; at += up(static_cast<size_t>(nnz) * 4);
  l.val = at; at += up(static_cast<size_t>(nnz) * 4);
  l.row_of = at; at += up(static_cast<size_t>(nnz) * 4);
  l.key_out = at; at += up(static_cast<size_t>(nnz) * 4);
  l.val_out = at; at += up(static_cast<size_t>(nnz) * 4);
  l.cnt = at; at += up(static_cast<size_t>(n_cols) * 4);
  l.incl = at; at += up(static_cast<size_t>(n_cols) * 8);
  unsigned* kn = nullptr;
  int64_t* on = nullptr;
  size_t sort_bytes = 0, scan_bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, kn, kn, kn, kn, static_cast<size_t>(nnz), 0, end_bit);
  auto in = rocprim::make_transform_iterator(kn, ToI64{});
  (void)rocprim::inclusive_scan(nullptr, scan_bytes, in, on, static_cast<size_t>(n_cols), rocprim::plus<int64_t>());
  l.temp = at;
  l.temp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  at += up(l.temp_bytes);
  l.total = at;
  return l;
}

__global__ __launch_bounds__(256) void zero_u32_kernel(unsigned* p, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) p[i] = 0u;
}

}  // namespace

extern "C" size_t gnan_csr_transpose_workspace_bytes(int64_t nnz, int64_t n_cols) {
  if (nnz <= 0 || n_cols <= 0) return 256;
  return tr_layout(nnz, n_cols, bits_for(static_cast<unsigned>(n_cols))).total;
}

extern "C" int gnan_csr_transpose(const gnan_csr_transpose_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr && a->n_rows > 0 && a->n_cols > 0 && a->n_rows < (int64_t{1} << 31) && a->n_cols < (int64_t{1} << 31) &&
                   a->nnz > 0 && a->nnz < (int64_t{1} << 32),
               "gnan_csr_transpose: bad sizes");
  GNAN_REQUIRE(a->rowptr && a->col && a->code && a->rowptr_t && a->col_t && a->code_t && a->workspace, "gnan_csr_transpose: null pointer");
  GNAN_REQUIRE(a->n_long == 0 || a->long_rows, "gnan_csr_transpose: n_long without long_rows");
  const unsigned end_bit = bits_for(static_cast<unsigned>(a->n_cols));
  const TrLayout l = tr_layout(a->nnz, a->n_cols, end_bit);
  GNAN_REQUIRE(a->workspace_bytes >= l.total && (reinterpret_cast<uintptr_t>(a->workspace) % 256) == 0,
               "gnan_csr_transpose: workspace %zu B < required %zu B (256-byte aligned)", a->workspace_bytes, l.total);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(a->workspace);
  unsigned* key = reinterpret_cast<unsigned*>(ws + l.key);
  unsigned* val = reinterpret_cast<unsigned*>(ws + l.val);
  unsigned* row_of = reinterpret_cast<unsigned*>(ws + l.row_of);
  unsigned* key_s = reinterpret_cast<unsigned*>(ws + l.key_out);
  unsigned* val_s = reinterpret_cast<unsigned*>(ws + l.val_out);
  unsigned* cnt = reinterpret_cast<unsigned*>(ws + l.cnt);
  int64_t* incl = reinterpret_cast<int64_t*>(ws + l.incl);
  hipLaunchKernelGGL(zero_u32_kernel, dim3(static_cast<unsigned>((a->n_cols + 255) / 256)), dim3(256), 0, st, cnt, a->n_cols);
  hipLaunchKernelGGL(tr_rows_kernel, dim3(static_cast<unsigned>((a->n_rows + 255) / 256)), dim3(256), 0, st, a->rowptr, a->rowptr_is64, a->col,
                     a->n_rows, key, val, row_of, cnt);
  if (int rc = gnan::check_launch("tr_rows_kernel")) return rc;
  if (a->n_long > 0) {
    hipLaunchKernelGGL(tr_rows_long_kernel, dim3(static_cast<unsigned>(a->n_long)), dim3(256), 0, st, a->rowptr, a->rowptr_is64, a->col,
                       a->long_rows, key, val, row_of, cnt);
    if (int rc = gnan::check_launch("tr_rows_long_kernel")) return rc;
  }
  size_t temp_bytes = l.temp_bytes;
  hipError_t e = rocprim::radix_sort_pairs(ws + l.temp, temp_bytes, key, key_s, val, val_s, static_cast<size_t>(a->nnz), 0, end_bit, st);
  if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "gnan_csr_transpose: radix sort: %s", hipGetErrorString(e));
  temp_bytes = l.temp_bytes;
  auto in = rocprim::make_transform_iterator(cnt, ToI64{});
  e = rocprim::inclusive_scan(ws + l.temp, temp_bytes, in, incl, static_cast<size_t>(a->n_cols), rocprim::plus<int64_t>(), st);
  if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "gnan_csr_transpose: scan: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(store_rowptr_kernel, dim3(static_cast<unsigned>((a->n_cols + 1 + 255) / 256)), dim3(256), 0, st, incl, a->n_cols,
                     a->rowptr_t, a->rowptr_is64);
  if (int rc = gnan::check_launch("store_rowptr_kernel")) return rc;
  hipLaunchKernelGGL(tr_gather_kernel, dim3(static_cast<unsigned>((a->nnz + 255) / 256)), dim3(256), 0, st, val_s, row_of, a->code, a->nnz,
                     a->col_t, a->code_t);
  return gnan::check_launch("tr_gather_kernel");
}

// =============================================================================================
// The hub-row plan (HopGraph.long_row_plan): the rows with more than `threshold` pairs in ascending order and the prefix of their
// slice counts ceil(pairs / slice_edges) — two passes over the row lengths (count per block of 256 rows, scan, ordered fill).  It
// replaces ~10 framework launches whose first use in a process cost 20 ms of lazily loaded kernels (cold setup of the bench line).
// =============================================================================================
namespace {

__global__ __launch_bounds__(256) void long_count_kernel(const void* rowptr, int is64, int64_t n, int64_t threshold, int* block_cnt) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const bool is_long = i < n && load_ptr(rowptr, is64, i + 1) - load_ptr(rowptr, is64, i) > threshold;
  const int c = __syncthreads_count(is_long ? 1 : 0);
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = c;
}

__global__ __launch_bounds__(256) void long_fill_kernel(const void* rowptr, int is64, int64_t n, int64_t threshold, int64_t slice_edges,
                                                        const int* block_off, int32_t* long_rows, int32_t* n_sl) {
  __shared__ int wave_cnt[4];
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  int64_t deg = 0;
  if (i < n) deg = load_ptr(rowptr, is64, i + 1) - load_ptr(rowptr, is64, i);
  const bool is_long = i < n && deg > threshold;
  const unsigned long long b = __ballot(is_long);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wave] = __popcll(b);
  __syncthreads();
  int before = block_off[blockIdx.x];
  for (int w = 0; w < wave; ++w) before += wave_cnt[w];
  if (is_long) {
    const int at = before + __popcll(b & ((1ull << lane) - 1ull));
    long_rows[at] = static_cast<int32_t>(i);
    n_sl[at] = static_cast<int32_t>((deg + slice_edges - 1) / slice_edges);
  }
}

__global__ void long_ptr0_kernel(int32_t* slice_ptr) { slice_ptr[0] = 0; }

}  // namespace

extern "C" size_t gnan_long_row_plan_workspace_bytes(int64_t n_rows) {
  const size_t nb = static_cast<size_t>((n_rows + 255) / 256);
  size_t scan_bytes = 0;
  int* in = nullptr;
  (void)rocprim::exclusive_scan(nullptr, scan_bytes, in, in, 0, nb + 1, rocprim::plus<int>());
  size_t scan2 = 0;                                   // (phase 2 scans up to n_rows slice counts)
  (void)rocprim::inclusive_scan(nullptr, scan2, in, in, static_cast<size_t>(n_rows > 0 ? n_rows : 1), rocprim::plus<int>());
  scan_bytes = scan_bytes > scan2 ? scan_bytes : scan2;
  return ((nb + 1) * 2 * sizeof(int) + 255) / 256 * 256 + (scan_bytes + 255) / 256 * 256 + 256;
}

// Phase 1: block counts and their exclusive scan; total[0] = number of hub rows (device; the caller reads it back to size phase 2).
extern "C" int gnan_long_row_plan_count(const void* rowptr, int32_t rowptr_is64, int64_t n_rows, int64_t threshold, void* workspace,
                                        size_t workspace_bytes, int32_t* total, gnan_stream_t stream) {
  GNAN_REQUIRE(rowptr && workspace && total && n_rows > 0 && n_rows < (int64_t{1} << 31) && threshold >= 0, "long_row_plan_count: bad arguments");
  GNAN_REQUIRE(workspace_bytes >= gnan_long_row_plan_workspace_bytes(n_rows) && reinterpret_cast<uintptr_t>(workspace) % 256 == 0,
               "long_row_plan_count: workspace of %zu bytes, 256-byte aligned", gnan_long_row_plan_workspace_bytes(n_rows));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t nb = static_cast<size_t>((n_rows + 255) / 256);
  int* cnt = static_cast<int*>(workspace);
  int* off = cnt + (nb + 1);
  char* tmp = static_cast<char*>(workspace) + ((nb + 1) * 2 * sizeof(int) + 255) / 256 * 256;
  hipLaunchKernelGGL(long_count_kernel, dim3(static_cast<unsigned>(nb)), dim3(256), 0, st, rowptr, rowptr_is64, n_rows, threshold, cnt);
  if (int rc = gnan::check_launch("long_count_kernel")) return rc;
  hipLaunchKernelGGL(zero_u32_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<unsigned*>(cnt + nb), int64_t{1});     // the scan's extra slot
  size_t scan_bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, scan_bytes, cnt, off, 0, nb + 1, rocprim::plus<int>());
  hipError_t e = rocprim::exclusive_scan(tmp, scan_bytes, cnt, off, 0, nb + 1, rocprim::plus<int>(), st);
  if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "long_row_plan_count: scan: %s", hipGetErrorString(e));
  e = hipMemcpyAsync(total, off + nb, sizeof(int), hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "long_row_plan_count: copy: %s", hipGetErrorString(e));
  return GNAN_OK;
}

// Phase 2 (same workspace, untouched since phase 1): long_rows [n_long] ascending, slice_ptr [n_long + 1]; slice_ptr[n_long] = slices.
extern "C" int gnan_long_row_plan_fill(const void* rowptr, int32_t rowptr_is64, int64_t n_rows, int64_t threshold, int64_t slice_edges,
                                       int64_t n_long, void* workspace, size_t workspace_bytes, int32_t* long_rows, int32_t* slice_ptr,
                                       gnan_stream_t stream) {
  GNAN_REQUIRE(rowptr && workspace && long_rows && slice_ptr && n_rows > 0 && n_long > 0 && slice_edges > 0, "long_row_plan_fill: bad arguments");
  GNAN_REQUIRE(workspace_bytes >= gnan_long_row_plan_workspace_bytes(n_rows), "long_row_plan_fill: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t nb = static_cast<size_t>((n_rows + 255) / 256);
  int* cnt = static_cast<int*>(workspace);
  int* off = cnt + (nb + 1);
  char* tmp = static_cast<char*>(workspace) + ((nb + 1) * 2 * sizeof(int) + 255) / 256 * 256;
  // the slice counts land in slice_ptr[1 ..] and are scanned in place
  hipLaunchKernelGGL(long_fill_kernel, dim3(static_cast<unsigned>(nb)), dim3(256), 0, st, rowptr, rowptr_is64, n_rows, threshold, slice_edges,
                     off, long_rows, slice_ptr + 1);
  if (int rc = gnan::check_launch("long_fill_kernel")) return rc;
  hipLaunchKernelGGL(long_ptr0_kernel, dim3(1), dim3(1), 0, st, slice_ptr);
  size_t scan_bytes = 0;
  (void)rocprim::inclusive_scan(nullptr, scan_bytes, slice_ptr + 1, slice_ptr + 1, static_cast<size_t>(n_long), rocprim::plus<int>());
  if (scan_bytes > gnan_long_row_plan_workspace_bytes(n_rows) - ((nb + 1) * 2 * sizeof(int) + 255) / 256 * 256)
    return gnan::fail(GNAN_ERR_WORKSPACE, "long_row_plan_fill: scan needs %zu bytes", scan_bytes);
  hipError_t e = rocprim::inclusive_scan(tmp, scan_bytes, slice_ptr + 1, slice_ptr + 1, static_cast<size_t>(n_long), rocprim::plus<int>(), st);
  if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "long_row_plan_fill: scan: %s", hipGetErrorString(e));
  return GNAN_OK;
}
