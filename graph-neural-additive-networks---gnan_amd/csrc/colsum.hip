// Column sums of the operand, total[w] = sum_j S[j, w] — the "everything" term of the rest bucket
// (pairs not listed in the CSR get weight rho(0)/cnt_rest).  One streaming pass over S,
// fp32 per-thread partials, float64 across threads and workgroups, fixed reduction order (deterministic).
#include "common.hpp"

namespace {

constexpr int kBlocks = 1024;

// BF16: S holds bf16 rows (VEC must be 4: 8-B loads), widened to fp32 before summing.
template <int VEC, bool BF16 = false>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ S, int64_t n, int W, int64_t stride,
                                                             double* __restrict__ partial, const int32_t* __restrict__ wcnt = nullptr,
                                                             int64_t wstride = 0) {
  // wcnt (gnan_colsum_weighted): row r counts with weight 1 / max(wcnt[r * wstride], 1)
  // thread (cx, ry): column chunk cx of VEC floats, rows ry, ry + RY, ... inside this workgroup's row range
  const int chunks = (W + VEC - 1) / VEC;
  const int cpb = chunks < 256 ? chunks : 256;     // column chunks handled per pass
  const int RY = 256 / cpb;
  const int cx = threadIdx.x % cpb, ry = threadIdx.x / cpb;
  __shared__ double red[256 * VEC];
  const int64_t rows_per_block = (n + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  for (int c0 = 0; c0 < chunks; c0 += cpb) {
    const int c = c0 + cx;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    if (c < chunks && ry < RY) {
      int64_t r = r0 + ry;
      if constexpr (!BF16 && VEC == 1) if (wcnt == nullptr) {
        // narrow operands (one float per thread and row): eight rows in flight per thread — with one, a 40-MB column of the
        // 10M-node graph took 27 us (1.5 TB/s: request latency, not bandwidth)
        for (; r + 7 * static_cast<int64_t>(RY) < r1; r += 8 * static_cast<int64_t>(RY)) {
          float t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) t[u] = S[(r + u * static_cast<int64_t>(RY)) * stride + c];
#pragma unroll
          for (int u = 0; u < 8; ++u) acc[0] += t[u];
        }
      }
      for (; r < r1; r += RY) {
        const float* ptr = S + r * stride + static_cast<int64_t>(c) * VEC;
        if constexpr (BF16) {
          const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(S) + r * stride +
                                                          static_cast<int64_t>(c) * 4);
          acc[0] += __uint_as_float(t.x << 16); acc[1] += __uint_as_float(t.x & 0xffff0000u);
          acc[2] += __uint_as_float(t.y << 16); acc[3] += __uint_as_float(t.y & 0xffff0000u);
        } else if constexpr (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4*>(ptr);
          if (wcnt) {
            const int q = wcnt[r * wstride];
            const float d = static_cast<float>(q > 1 ? q : 1);
            acc[0] += t.x / d; acc[1] += t.y / d; acc[2] += t.z / d; acc[3] += t.w / d;
          } else {
            acc[0] += t.x; acc[1] += t.y; acc[2] += t.z; acc[3] += t.w;
          }
        } else {
          if (wcnt) {
            const int q = wcnt[r * wstride];
            acc[0] += ptr[0] / static_cast<float>(q > 1 ? q : 1);
          } else {
            acc[0] += ptr[0];
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < VEC; ++v) red[threadIdx.x * VEC + v] = acc[v];
    __syncthreads();
    if (ry == 0 && c < chunks) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        double s = 0.0;
        for (int y = 0; y < RY; ++y) s += red[(y * cpb + cx) * VEC + v];
        if (c * VEC + v < W) partial[static_cast<int64_t>(blockIdx.x) * W + c * VEC + v] = s;
      }
    }
  }
}

// One workgroup per column: 256 threads stride over the per-workgroup partials, then a fixed-order tree.
__global__ __launch_bounds__(256) void colsum_final_kernel(const double* __restrict__ partial, int blocks, int W,
                                                           float* __restrict__ total, const float* __restrict__ scale = nullptr) {
  __shared__ double red[256];
  const int w = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < blocks; b += 256) s += partial[static_cast<int64_t>(b) * W + w];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (static_cast<int>(threadIdx.x) < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) total[w] = static_cast<float>(scale ? red[0] * static_cast<double>(scale[0]) : red[0]);
}

// wt[i, d, c] = lut[d, c] / max(cnt[i, d], 1) - (with_rest ? lut[D - 1, c] / max(cnt[i, D - 1], 1) : 0): the per-node weight table of the
// wide backward (gnan_spmm_fwd with weight_by_col), IEEE divisions and one subtraction as the framework expression it replaces
__global__ __launch_bounds__(256) void weight_table_kernel(const float* __restrict__ lut, const int32_t* __restrict__ cnt, int64_t cnt_stride,
                                                           int64_t n, int D, int Cw, int with_rest, float* __restrict__ wt) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= n * D * Cw) return;
  const int c = static_cast<int>(idx % Cw);
  const int d = static_cast<int>((idx / Cw) % D);
  const int64_t i = idx / (static_cast<int64_t>(Cw) * D);
  const int qd = cnt ? cnt[i * cnt_stride + d] : 1, qr = cnt ? cnt[i * cnt_stride + D - 1] : 1;
  float w = lut[d * Cw + c] / static_cast<float>(qd > 1 ? qd : 1);
  if (with_rest) w -= lut[(D - 1) * Cw + c] / static_cast<float>(qr > 1 ? qr : 1);
  wt[idx] = w;
}

}  // namespace

extern "C" int gnan_weight_table(const float* lut, const int32_t* cnt, int64_t cnt_stride, int64_t n, int32_t D, int32_t Cw,
                                 int32_t with_rest, float* wt, gnan_stream_t stream) {
  GNAN_REQUIRE(n >= 0 && D >= 1 && Cw >= 1, "weight_table: bad sizes");
  GNAN_REQUIRE((lut && wt) || n == 0, "weight_table: null pointer");
  GNAN_REQUIRE(cnt == nullptr || cnt_stride >= D, "weight_table: count rows shorter than D");
  if (n == 0) return GNAN_OK;
  const int64_t blocks = (n * D * Cw + 255) / 256;
  GNAN_REQUIRE(blocks < (int64_t{1} << 31), "weight_table: too many entries for one launch");
  hipLaunchKernelGGL(weight_table_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), lut, cnt,
                     cnt_stride, n, D, Cw, with_rest, wt);
  return gnan::check_launch("weight_table_kernel");
}

extern "C" int gnan_colsum_weighted(const float* S, int64_t n, int32_t W, int64_t stride, const int32_t* cnt, int64_t cnt_stride,
                                    const float* scale, float* total, void* workspace, size_t workspace_bytes, gnan_stream_t stream) {
  GNAN_REQUIRE(n >= 0 && W >= 1, "colsum_weighted: bad sizes");
  GNAN_REQUIRE(S && total && workspace && cnt, "colsum_weighted: null pointer");
  GNAN_REQUIRE(stride >= W && cnt_stride >= 1, "colsum_weighted: row stride smaller than the width");
  if (workspace_bytes < static_cast<size_t>(kBlocks) * W * sizeof(double))
    return gnan::fail(GNAN_ERR_WORKSPACE, "colsum_weighted: workspace too small");
  GNAN_REQUIRE(reinterpret_cast<uintptr_t>(workspace) % 8 == 0, "colsum_weighted: workspace must be 8-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int blocks = static_cast<int>(n / 256 + 1);
  blocks = blocks > kBlocks ? kBlocks : blocks;
  double* partial = static_cast<double*>(workspace);
  const bool vec = W % 4 == 0 && stride % 4 == 0 && reinterpret_cast<uintptr_t>(S) % 16 == 0;
  if (vec) hipLaunchKernelGGL(colsum_partial_kernel<4>, dim3(blocks), dim3(256), 0, st, S, n, W, stride, partial, cnt, cnt_stride);
  else hipLaunchKernelGGL(colsum_partial_kernel<1>, dim3(blocks), dim3(256), 0, st, S, n, W, stride, partial, cnt, cnt_stride);
  if (int rc = gnan::check_launch("colsum_partial_kernel")) return rc;
  hipLaunchKernelGGL(colsum_final_kernel, dim3(W), dim3(256), 0, st, partial, blocks, W, total, scale);
  return gnan::check_launch("colsum_final_kernel");
}

extern "C" size_t gnan_colsum_workspace_bytes(int32_t W) { return static_cast<size_t>(kBlocks) * W * sizeof(double); }

extern "C" int gnan_colsum_bf16(const void* S, int64_t n, int32_t W, int64_t stride, float* total, void* workspace,
                                size_t workspace_bytes, gnan_stream_t stream) {
  GNAN_REQUIRE(n >= 0 && W >= 1, "colsum: bad sizes");
  GNAN_REQUIRE(S && total && workspace, "colsum: null pointer");
  GNAN_REQUIRE(stride >= W && W % 4 == 0 && stride % 4 == 0 && reinterpret_cast<uintptr_t>(S) % 8 == 0,
               "colsum_bf16: needs W %% 4 == 0 and 8-B aligned rows");
  if (workspace_bytes < gnan_colsum_workspace_bytes(W))
    return gnan::fail(GNAN_ERR_WORKSPACE, "colsum: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int blocks = static_cast<int>(n / 256 + 1);
  blocks = blocks > kBlocks ? kBlocks : blocks;
  double* partial = static_cast<double*>(workspace);
  hipLaunchKernelGGL((colsum_partial_kernel<4, true>), dim3(blocks), dim3(256), 0, st, static_cast<const float*>(S), n, W,
                     stride, partial);
  if (int rc = gnan::check_launch("colsum_partial_kernel")) return rc;
  hipLaunchKernelGGL(colsum_final_kernel, dim3(W), dim3(256), 0, st, partial, blocks, W, total);
  return gnan::check_launch("colsum_final_kernel");
}

extern "C" int gnan_colsum(const float* S, int64_t n, int32_t W, int64_t stride, float* total, void* workspace,
                           size_t workspace_bytes, gnan_stream_t stream) {
  GNAN_REQUIRE(n >= 0 && W >= 1, "colsum: bad sizes");
  GNAN_REQUIRE(S && total && workspace, "colsum: null pointer");
  GNAN_REQUIRE(stride >= W, "colsum: row stride smaller than W");
  if (workspace_bytes < gnan_colsum_workspace_bytes(W))
    return gnan::fail(GNAN_ERR_WORKSPACE, "colsum: workspace %zu B < required %zu B", workspace_bytes,
                      gnan_colsum_workspace_bytes(W));
  GNAN_REQUIRE(reinterpret_cast<uintptr_t>(workspace) % 8 == 0, "colsum: workspace must be 8-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int blocks = static_cast<int>(n / 256 + 1);
  blocks = blocks > kBlocks ? kBlocks : blocks;
  double* partial = static_cast<double*>(workspace);
  const bool vec = W % 4 == 0 && stride % 4 == 0 && reinterpret_cast<uintptr_t>(S) % 16 == 0;
  if (vec) {
    hipLaunchKernelGGL(colsum_partial_kernel<4>, dim3(blocks), dim3(256), 0, st, S, n, W, stride, partial);
  } else {
    hipLaunchKernelGGL(colsum_partial_kernel<1>, dim3(blocks), dim3(256), 0, st, S, n, W, stride, partial);
  }
  if (int rc = gnan::check_launch("colsum_partial_kernel")) return rc;
  hipLaunchKernelGGL(colsum_final_kernel, dim3(W), dim3(256), 0, st, partial, blocks, W, total);
  return gnan::check_launch("colsum_final_kernel");
}


// dst[k, :] = src[ids[k], :] for W floats per row: the compact copy of the most listed nodes' operand rows that narrow
// aggregations read instead of the scattered originals (HopGraph.hot_columns).  The destination usually sits right behind
// the operand, in room the table look-up left there (functional.feature_mlps(room_rows=...)).
namespace {
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int64_t src_stride, const int64_t* __restrict__ ids,
                                                          int64_t k, int W, float* __restrict__ dst) {
  const int64_t total = k * W;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < total; e += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t r = e / W;
    dst[e] = src[ids[r] * src_stride + (e - r * W)];
  }
}
}  // namespace

extern "C" int gnan_gather_rows(const float* src, int64_t src_stride, const int64_t* ids, int64_t k, int32_t W, float* dst,
                                gnan_stream_t stream) {
  GNAN_REQUIRE(k >= 0 && W >= 1 && src_stride >= W, "gather_rows: bad sizes");
  if (k == 0) return GNAN_OK;
  GNAN_REQUIRE(src && ids && dst, "gather_rows: null pointer");
  int64_t blocks = (k * W + 255) / 256;
  blocks = blocks > 4096 ? 4096 : blocks;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                     src_stride, ids, k, W, dst);
  return gnan::check_launch("gather_rows_kernel");
}

// ---------------------------------------------------------------------------------------------
// Up to eight small device-to-device copies in ONE launch: the static input slots of a captured graph-task step (features,
// hop codes, shell sizes, label — four cudaMemcpyAsync of a few hundred bytes each cost four launch slots, 10 us of a 45-us
// evaluation pass).  One workgroup per copy; 16-byte words where both ends are aligned, bytes otherwise.
// ---------------------------------------------------------------------------------------------
namespace {
struct MultiCopy {
  const void* src[8];
  void* dst[8];
  int64_t bytes[8];
};

__global__ __launch_bounds__(256) void multi_copy_kernel(const MultiCopy m) {
  const int c = blockIdx.x;
  const char* s = static_cast<const char*>(m.src[c]);
  char* d = static_cast<char*>(m.dst[c]);
  const int64_t n = m.bytes[c];
  if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
    const int64_t words = n / 16;
    for (int64_t i = threadIdx.x; i < words; i += 256) reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
    for (int64_t i = words * 16 + threadIdx.x; i < n; i += 256) d[i] = s[i];
  } else {
    for (int64_t i = threadIdx.x; i < n; i += 256) d[i] = s[i];
  }
}
}  // namespace

extern "C" int gnan_multi_copy(int32_t count, const void* const* src, void* const* dst, const int64_t* bytes, gnan_stream_t stream) {
  GNAN_REQUIRE(count >= 0 && count <= 8, "multi_copy: up to eight copies per call (got %d)", count);
  if (count == 0) return GNAN_OK;
  GNAN_REQUIRE(src && dst && bytes, "multi_copy: null argument arrays");
  MultiCopy m;
  for (int i = 0; i < 8; ++i) {
    m.src[i] = i < count ? src[i] : nullptr;
    m.dst[i] = i < count ? dst[i] : nullptr;
    m.bytes[i] = i < count ? bytes[i] : 0;
    if (i < count) GNAN_REQUIRE(bytes[i] >= 0 && (bytes[i] == 0 || (src[i] && dst[i])), "multi_copy: bad copy %d", i);
  }
  hipLaunchKernelGGL(multi_copy_kernel, dim3(static_cast<unsigned>(count)), dim3(256), 0, static_cast<hipStream_t>(stream), m);
  return gnan::check_launch("multi_copy_kernel");
}


// ---------------------------------------------------------------------------------------------
// out[i, c] = sum over features k of fx[i, k * C + c]  (C in {1, 2, 4}; rows 16-byte aligned, W % 4 == 0): the feature sum of
// KEPT per-feature rows — what the backward pass of a reference-order forward aggregates (GNAN.py:157 applied to the rows of
// models.py:360-365).  LPR lanes share a row, a 16-byte load each (every quad carries the same channel pattern), a butterfly
// over the row's lanes, one store; HBM-bound: W * 4 bytes in, C * 4 out per row.
// ---------------------------------------------------------------------------------------------
namespace {
template <int LPR>
__global__ __launch_bounds__(256) void feature_sum_kernel(const float* __restrict__ fx, int64_t n, int W, int64_t stride, int C,
                                                          float* __restrict__ out, int64_t out_stride) {
  constexpr int G = 256 / LPR;                       // rows per workgroup pass
  const int sub = threadIdx.x % LPR, slot = threadIdx.x / LPR;
  const bool live = sub * 4 < W;
  for (int64_t r0 = static_cast<int64_t>(blockIdx.x) * G; r0 < n; r0 += static_cast<int64_t>(gridDim.x) * G) {
    const int64_t r = r0 + slot;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < n) {
      const float* row = fx + r * stride;
      for (int c4 = sub; c4 * 4 < W; c4 += LPR) {   // (one trip unless a row is wider than 4 * LPR floats)
        const float4 t = *reinterpret_cast<const float4*>(row + c4 * 4);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
      }
    }
    (void)live;
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) {
      acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
      acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
    }
    if (sub == 0 && r < n) {
      float* o = out + r * out_stride;
      if (C == 1) o[0] = (acc.x + acc.y) + (acc.z + acc.w);
      else if (C == 2) { o[0] = acc.x + acc.z; o[1] = acc.y + acc.w; }
      else { o[0] = acc.x; o[1] = acc.y; o[2] = acc.z; o[3] = acc.w; }
    }
  }
}
}  // namespace

extern "C" int gnan_feature_sum(const float* fx, int64_t n, int32_t W, int64_t stride, int32_t C, float* out,
                                int64_t out_stride, gnan_stream_t stream) {
  GNAN_REQUIRE(n >= 0 && W >= 1 && (C == 1 || C == 2 || C == 4), "feature_sum: bad sizes (C must be 1, 2 or 4)");
  if (n == 0) return GNAN_OK;
  GNAN_REQUIRE(fx && out, "feature_sum: null pointer");
  GNAN_REQUIRE(W % 4 == 0 && W % C == 0 && stride % 4 == 0 && stride >= W && out_stride >= C &&
                   reinterpret_cast<uintptr_t>(fx) % 16 == 0,
               "feature_sum: rows must be whole 16-byte quads (W %% 4 == 0, aligned, stride %% 4 == 0)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int lpr = 1;
  while (lpr * 4 < W && lpr < 64) lpr <<= 1;
  const int64_t rows_per_pass = 256 / lpr;
  int64_t blocks = (n + rows_per_pass - 1) / rows_per_pass;
  if (blocks > 256 * 64) blocks = 256 * 64;
  auto go = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, fx, n, W, stride, C, out, out_stride);
    return gnan::check_launch("feature_sum_kernel");
  };
  switch (lpr) {
    case 1: return go(feature_sum_kernel<1>);
    case 2: return go(feature_sum_kernel<2>);
    case 4: return go(feature_sum_kernel<4>);
    case 8: return go(feature_sum_kernel<8>);
    case 16: return go(feature_sum_kernel<16>);
    case 32: return go(feature_sum_kernel<32>);
    default: return go(feature_sum_kernel<64>);
  }
}


// ---------------------------------------------------------------------------------------------
// Y[q, c] += sum over the operand columns w that read-out channel c collects of  wt(i_q, D - 1, w mod Cw) * total[w]
// — the part of the aggregation that depends on the operand only through its column sums (the rest bucket's weight times
// `total`): a multi-rank forward aggregates against zero sums while the all-reduce of `total` is in flight and adds this
// afterwards (gnan_amd/distributed.py).  wt(i, D-1, cw) = lut[i or 0][D - 1][cw] / max(cnt[i][D - 1], 1).
// reduce_cr == 0: c = w (Y has W columns); else c = w mod reduce_cr (the fused read-out's channels).
// Thread = (row, output column); the per-(c, cw) sums of `total` sit in LDS (reduce_cr * Cw <= 64 entries).
// ---------------------------------------------------------------------------------------------
namespace {
struct RestTermParams {
  float* Y;
  int64_t y_stride, n;
  const float* total;
  int W;
  const float* lut;
  int64_t lut_row_stride;     // 0: one table for every row
  int D, Cw;
  const int32_t* cnt;
  int64_t cnt_stride;
  const int32_t* row_ids;
  int reduce_cr;
};

__global__ __launch_bounds__(256) void rest_term_kernel(const RestTermParams p) {
  __shared__ float T[64];                              // [reduce_cr][Cw]
  const int rc = p.reduce_cr;
  if (rc) {
    if (static_cast<int>(threadIdx.x) < rc * p.Cw) {
      const int c = threadIdx.x / p.Cw, cw = threadIdx.x % p.Cw;
      float s = 0.f;
      for (int w = 0; w < p.W; ++w)
        if (w % rc == c && w % p.Cw == cw) s += p.total[w];
      T[threadIdx.x] = s;
    }
    __syncthreads();
  }
  const int Co = rc ? rc : p.W;
  const int64_t items = p.n * Co;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < items; e += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t q = e / Co;
    const int c = static_cast<int>(e - q * Co);
    const int64_t i = p.row_ids ? p.row_ids[q] : q;
    const float* wr = p.lut + i * p.lut_row_stride + static_cast<int64_t>(p.D - 1) * p.Cw;
    float inv = 1.f;
    if (p.cnt) {
      const int cn = p.cnt[i * p.cnt_stride + p.D - 1];
      inv = 1.f / static_cast<float>(cn > 1 ? cn : 1);
    }
    float acc = 0.f;
    if (rc) {
      for (int cw = 0; cw < p.Cw; ++cw) acc = fmaf(wr[cw] * inv, T[c * p.Cw + cw], acc);
    } else {
      acc = wr[c % p.Cw] * inv * p.total[c];
    }
    p.Y[q * p.y_stride + c] += acc;
  }
}
}  // namespace

extern "C" int gnan_rest_term_add(const gnan_rest_term_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "rest_term: null args");
  GNAN_REQUIRE(a->n >= 0 && a->W >= 1 && a->D >= 1 && a->Cw >= 1, "rest_term: bad sizes");
  if (a->n == 0) return GNAN_OK;
  GNAN_REQUIRE(a->Y && a->total && a->lut, "rest_term: null pointer");
  const int rc = a->reduce_cr;
  GNAN_REQUIRE(rc >= 0 && (rc == 0 || (a->W % rc == 0 && rc * a->Cw <= 64)), "rest_term: reduce_cr must divide W and reduce_cr * Cw <= 64");
  GNAN_REQUIRE(a->y_stride >= (rc ? rc : a->W), "rest_term: y_stride smaller than the output width");
  GNAN_REQUIRE(a->cnt == nullptr || a->cnt_stride >= a->D, "rest_term: cnt row stride smaller than D");
  RestTermParams p;
  p.Y = a->Y; p.y_stride = a->y_stride; p.n = a->n; p.total = a->total; p.W = a->W;
  p.lut = a->lut; p.lut_row_stride = a->lut_row_stride; p.D = a->D; p.Cw = a->Cw;
  p.cnt = a->cnt; p.cnt_stride = a->cnt_stride; p.row_ids = a->row_ids; p.reduce_cr = rc;
  const int64_t items = a->n * (rc ? rc : a->W);
  int64_t blocks = (items + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(rest_term_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return gnan::check_launch("rest_term_kernel");
}


// ---------------------------------------------------------------------------------------------
// out[g, c] = sum of Y[i, c] over the rows node_off[g] <= i < node_off[g + 1]: the per-graph read-out of a batch of graphs
// (the scatter_add_ of batched_pyg_main.py:173-181; a graph's nodes are consecutive rows).  One wave per graph, lanes
// stride over (row, channel) pairs, a fixed butterfly at the end: no atomics, bit-reproducible.  C <= 64.
// ---------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void segment_sum_kernel(const float* __restrict__ Y, int64_t y_stride, int C,
                                                          const int32_t* __restrict__ node_off, int n_graphs,
                                                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= n_graphs) return;
  const int64_t lo = node_off[g], hi = node_off[g + 1];
  const int rows_per_pass = 64 / C;                   // lanes [0, rows_per_pass * C) are busy: lane = (row slot, channel)
  const int slot = lane / C, c = lane - slot * C;
  float acc = 0.f;
  if (slot < rows_per_pass)
    for (int64_t i = lo + slot; i < hi; i += rows_per_pass) acc += Y[i * y_stride + c];
  // add the row slots of a channel: lanes c, c + C, c + 2C, ... (fixed order, by lane 'c')
  float total = 0.f;
  for (int s2 = 0; s2 < rows_per_pass; ++s2) total += __shfl(acc, s2 * C + (lane < C ? lane : 0));
  if (lane < C) out[static_cast<int64_t>(g) * C + lane] = total;
}
}  // namespace

extern "C" int gnan_segment_sum(const float* Y, int64_t y_stride, int32_t C, const int32_t* node_off, int32_t n_graphs,
                                float* out, gnan_stream_t stream) {
  GNAN_REQUIRE(n_graphs >= 0 && C >= 1 && C <= 64 && y_stride >= C, "segment_sum: bad sizes (1 <= C <= 64)");
  if (n_graphs == 0) return GNAN_OK;
  GNAN_REQUIRE(Y && node_off && out, "segment_sum: null pointer");
  hipLaunchKernelGGL(segment_sum_kernel, dim3(static_cast<unsigned>((n_graphs + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), Y, y_stride, C, node_off, n_graphs, out);
  return gnan::check_launch("segment_sum_kernel");
}
