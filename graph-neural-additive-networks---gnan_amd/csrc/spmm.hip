// rho(distance)-weighted neighbourhood sum over a hop-coded adjacency (gfx950).
//
// Replaces GNAN.py:65-73 / models.py:368-376 (rho on N^2 pairs, normalisation, bmm, feature sum)
// and the per-node loop GNAN.py:159-170.  See include/gnan_hip.h for the contract.
//
// Mapping (wave64):  an operand row of W floats is covered by LPR lanes x VEC floats
// (16 lanes x float4 for W = 64: one 256-B row = one fully used 16-B/lane request).  A wave
// therefore holds G = 64/LPR independent lane groups:
//   * row blocks  : one group per output row, edges walked sequentially, index pairs fetched
//                   LPR at a time with one coalesced load and broadcast inside the group by
//                   ds_bpermute; no cross-lane reduction, the group stores its own row;
//   * slice blocks: hub rows (degree > long_threshold) are cut into slices; a 256-thread
//                   workgroup owns a slice, every wave streams 64 index pairs per load, its G
//                   groups stride over them, partial rows meet in LDS, and a fix-up kernel adds
//                   the slices in a fixed order (bit-reproducible, no float atomics).
// HBM-bound: per edge 4 B col + 1 B code + W*4 B gathered row; per row rowptr + W*4 B store.
#include "common.hpp"

#include <cstdlib>

namespace {

using gnan::kWave;

struct Params {
  int64_t n_rows, n_cols;
  int64_t nnz;     // listed pairs (length of col / code), 0 = unknown: index runs are then read entry by entry
  const void* rowptr;
  int rowptr_is64;
  const int32_t* col;
  const uint8_t* code;
  const int32_t* row_ids;
  const void* S;   // fp32 rows, or bf16 rows when the kernels are instantiated with VEC == 8
  int W;
  int64_t s_stride;
  const float* lut;
  int64_t lut_row_stride;
  int D, Cw;
  const int32_t* cnt;
  int64_t cnt_stride;
  const float* s_total;
  int weight_by_col, minus_rest;
  int reduce_cr;  // > 0: store only the per-channel sums over columns w = c (mod reduce_cr)
  int scatter_out;  // output row q is stored at Y[row_ids[q]] (rows are PROCESSED in row_ids order, e.g. by degree)
  int s_by_code;    // the operand row of pair (i, c, d) is S[c * D + d]: one pre-weighted row per (node, hop code)
  int packed;       // col entries carry the hop code in their top kPackBits bits (code is not read)
  float* Y;
  int64_t y_stride;
  int64_t long_threshold;
  const int32_t* long_rows;
  const int32_t* long_slice_ptr;
  int n_long, n_slices, slice_edges;
  float* partial;  // [n_slices, 2, W]
  int64_t hot_lo;  // spmm_hot_kernel: operand rows [hot_lo, hot_lo + hot_n) are served from an LDS copy
  int hot_n;
  float* shell_out;  // [n_rows, D - 1] raw per-code sums of the operand over the row's pairs (W == 1, small-D route, a lane per row)
};

__device__ __forceinline__ int64_t load_rowptr(const Params& p, int64_t i) {
  return p.rowptr_is64 ? static_cast<const int64_t*>(p.rowptr)[i]
                       : static_cast<int64_t>(static_cast<const int32_t*>(p.rowptr)[i]);
}

// Processing slot q -> adjacency row (index into rowptr / cnt / a per-row weight table) and output row.
//   no row_ids        : both q                         row_ids, scatter_out 0 : row_ids[q] -> q   (row subset)
//   scatter_out 1     : row_ids[q] -> row_ids[q]       (rows PROCESSED in row_ids order, e.g. by degree, stored in place)
//   scatter_out 2     : q -> row_ids[q]                (the adjacency itself is stored in processing order — a degree-sorted
//                       copy of the CSR — so that rowptr, cnt and the index pairs of neighbouring lane groups are adjacent)
__device__ __forceinline__ int64_t adj_row(const Params& p, int64_t q) {
  return (p.row_ids && p.scatter_out != 2) ? static_cast<int64_t>(p.row_ids[q]) : q;
}
__device__ __forceinline__ int64_t out_row(const Params& p, int64_t q, int64_t i) {
  return p.scatter_out == 2 ? static_cast<int64_t>(p.row_ids[q]) : (p.scatter_out ? i : q);
}

template <int VEC>
struct Vec {
  float v[VEC];
};

template <int VEC>
__device__ __forceinline__ Vec<VEC> load_vec(const float* ptr) {
  Vec<VEC> r;
  if constexpr (VEC == 8) {
    const float4 a = *reinterpret_cast<const float4*>(ptr), b = *reinterpret_cast<const float4*>(ptr + 4);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  } else if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(ptr);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else if constexpr (VEC == 2) {
    const float2 t = *reinterpret_cast<const float2*>(ptr);
    r.v[0] = t.x; r.v[1] = t.y;
  } else {
    r.v[0] = *ptr;
  }
  return r;
}

// Operand rows: fp32 for VEC in {1, 4}; VEC == 8 is the bf16-storage mode (8 bf16 = one 16-B request per lane,
// widened to fp32 in registers; accumulation and output stay fp32).
template <int VEC>
__device__ __forceinline__ Vec<VEC> load_operand(const void* S, int64_t row, int64_t stride, int col) {
  if constexpr (VEC == 8) {
    const uint4 t = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(S) + row * stride + col);
    Vec<VEC> r;
    const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      r.v[2 * i] = __uint_as_float(w[i] << 16);
      r.v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
    return r;
  } else {
    return load_vec<VEC>(static_cast<const float*>(S) + row * stride + col);
  }
}

// A gathered operand chunk as it sits in registers while the load is in flight: bf16 rows stay packed (4 VGPRs for
// 8 values) until they are consumed, so the in-flight window of the bf16 mode costs no more registers than fp32.
template <int VEC>
struct Raw {
  Vec<VEC> f;
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int v = 0; v < VEC; ++v) f.v[v] = 0.f;
  }
  __device__ __forceinline__ void load(const void* S, int64_t row, int64_t stride, int col) {
    f = load_vec<VEC>(static_cast<const float*>(S) + row * stride + col);
  }
  __device__ __forceinline__ Vec<VEC> widen() const { return f; }
};

template <>
struct Raw<8> {
  uint4 t;
  __device__ __forceinline__ void zero() { t = make_uint4(0u, 0u, 0u, 0u); }
  __device__ __forceinline__ void load(const void* S, int64_t row, int64_t stride, int col) {
    t = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(S) + row * stride + col);
  }
  __device__ __forceinline__ Vec<8> widen() const {
    Vec<8> r;
    const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      r.v[2 * i] = __uint_as_float(w[i] << 16);
      r.v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
    return r;
  }
};

template <int VEC>
__device__ __forceinline__ void store_vec(float* ptr, const Vec<VEC>& r) {
  if constexpr (VEC == 8) {
    *reinterpret_cast<float4*>(ptr) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
    *reinterpret_cast<float4*>(ptr + 4) = make_float4(r.v[4], r.v[5], r.v[6], r.v[7]);
    return;
  }
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(ptr) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<float2*>(ptr) = make_float2(r.v[0], r.v[1]);
  } else {
    *ptr = r.v[0];
  }
}

// Weights of adjacency row i for hop code d, for the VEC columns starting at column w0.
//   wt = lut[i*lrs + d*Cw + (w % Cw)] / max(cnt[i, d], 1)          (IEEE division, as torch.div)
template <int VEC>
__device__ __forceinline__ Vec<VEC> row_weights(const Params& p, int64_t i, int d, int w0) {
  Vec<VEC> w;
  const float* l = p.lut + i * p.lut_row_stride + static_cast<int64_t>(d) * p.Cw;
#pragma unroll
  for (int v = 0; v < VEC; ++v) w.v[v] = l[p.Cw == 1 ? 0 : (w0 + v) % p.Cw];
  if (p.cnt) {
    const int c = p.cnt[i * p.cnt_stride + d];
    const float r = static_cast<float>(c > 1 ? c : 1);
#pragma unroll
    for (int v = 0; v < VEC; ++v) w.v[v] = w.v[v] / r;
  }
  return w;
}

// Weight of one listed pair.  Forward: the table row is the output row i.  Transposed use
// (backward w.r.t. S): the table row is the neighbour c (weight_by_col) and the rest-bucket weight
// is subtracted (minus_rest), because d/dS_j of  wt_rest * (total - sum_listed S)  is  -wt_rest.
template <int VEC>
__device__ __forceinline__ Vec<VEC> edge_weights(const Params& p, int64_t i, int c, int d, int w0) {
  const int64_t r = p.weight_by_col ? static_cast<int64_t>(c) : i;
  Vec<VEC> w = row_weights<VEC>(p, r, d, w0);
  if (p.minus_rest) {
    const Vec<VEC> wr = row_weights<VEC>(p, r, p.D - 1, w0);
#pragma unroll
    for (int v = 0; v < VEC; ++v) w.v[v] -= wr.v[v];
  }
  return w;
}

// Per-row weight cache for the common truncated case (Cw == 1, D <= 4): four registers.  The empty asm statements
// keep the select chain a chain: left alone, the optimiser rewrites it as an indexed load from a 4-float stack array,
// which the backend then places in LDS (8 KB per workgroup and a ds_read + wait per listed pair).
struct SmallW {
  float w[4];
  __device__ __forceinline__ float pick(int d) const {
    float r = w[0];
    r = d == 1 ? w[1] : r;
    asm volatile("" : "+v"(r));
    r = d == 2 ? w[2] : r;
    asm volatile("" : "+v"(r));
    r = d >= 3 ? w[3] : r;
    return r;
  }
};

__device__ __forceinline__ SmallW small_weights(const Params& p, int64_t i) {
  SmallW s;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    float v = 0.f;
    if (d < p.D) {
      v = p.lut[i * p.lut_row_stride + d];
      if (p.cnt) {
        const int c = p.cnt[i * p.cnt_stride + d];
        v = v / static_cast<float>(c > 1 ? c : 1);
      }
    }
    s.w[d] = v;
  }
  return s;
}

// A lane's run of N consecutive (col, code) entries as wide loads: N * 4 bytes of column ids (4-byte aligned) and N bytes of
// hop codes (byte aligned; gfx950 runs HSA code in unaligned-access mode) instead of 2 N scalar loads.  With one lane per
// row (W = 1: 16 entries per lane and round) the scalar loads were 32 of the 48 memory instructions of a round, each
// touching ~20 different lines per wavefront.  The caller guarantees e0 + N <= nnz; entries past the row end are read
// (they belong to the next row) and ignored.
// Packed index entries: column id in the low 29 bits, hop code in the top 3 (graphs below 2^29 neighbours, D <= 8).  One
// 4-byte stream instead of a 4-byte and a 1-byte one: the index loads of a lane group are then ONE L2 request per round
// instead of two — 3 % of the W = 64 kernel's requests, 10 % of the bf16 kernel's (8 pairs per round, one request per row).
constexpr int kPackShift = 29;
constexpr unsigned kPackMask = (1u << kPackShift) - 1u;

// (explicit under-aligned vector loads into scalars: routed through __builtin_memcpy into the index arrays, the arrays were
// promoted to LDS — 16 KB per workgroup and a ds_read per listed pair; W = 1 on the arxiv-shaped graph: 37 us against 12)
typedef unsigned uint4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned uint4_a1 __attribute__((ext_vector_type(4), aligned(1)));
typedef unsigned uint2_a1 __attribute__((ext_vector_type(2), aligned(1)));
typedef unsigned uint1_a1 __attribute__((aligned(1)));

template <int N>
__device__ __forceinline__ void load_col_run(const int32_t* col, int (&colv)[N]) {
  static_assert(N % 4 == 0, "whole quads of column ids");
#pragma unroll
  for (int r = 0; r < N; r += 4) {
    const uint4_a4 v = *reinterpret_cast<const uint4_a4*>(col + r);
    colv[r] = static_cast<int>(v.x); colv[r + 1] = static_cast<int>(v.y);
    colv[r + 2] = static_cast<int>(v.z); colv[r + 3] = static_cast<int>(v.w);
  }
}

template <int N>
__device__ __forceinline__ void load_index_run(const int32_t* col, const uint8_t* code, int (&colv)[N], int (&codev)[N]) {
  static_assert(N == 4 || N == 8 || N == 16, "whole dwords of codes");
  load_col_run<N>(col, colv);
  unsigned cw[4] = {0u, 0u, 0u, 0u};
  if constexpr (N == 16) {
    const uint4_a1 v = *reinterpret_cast<const uint4_a1*>(code);
    cw[0] = v.x; cw[1] = v.y; cw[2] = v.z; cw[3] = v.w;
  } else if constexpr (N == 8) {
    const uint2_a1 v = *reinterpret_cast<const uint2_a1*>(code);
    cw[0] = v.x; cw[1] = v.y;
  } else {
    cw[0] = *reinterpret_cast<const uint1_a1*>(code);
  }
#pragma unroll
  for (int r = 0; r < N; ++r) codev[r] = static_cast<int>((cw[r / 4] >> (8 * (r % 4))) & 0xffu);
}

// ---------------------------------------------------------------------------------------------
// rows kernel: one LPR-lane group per output row
// ---------------------------------------------------------------------------------------------
template <int VEC, int LPR, bool DENSE, bool SMALLD, bool BYCODE, bool PACKED = false>
__device__ __forceinline__ void rows_body(const Params& p, const int64_t block_id) {
  constexpr int G = kWave / LPR;     // groups (rows) per wave
  constexpr int TILE = LPR * VEC;    // operand columns one pass covers
  // gathers in flight per lane.  fp32 rows: 1, 2, 3 and 4 measure the same (4.68-4.76 ms on C4: at 8 waves/SIMD the
  // kernel sits on the L2 request rate, not on latency), 8 costs registers, hence waves.  bf16 rows: 2 fits
  // the 64-VGPR budget of 8 waves/SIMD without scratch: 2.93 -> 2.71 ms (1: 2.77, 3: 2.70, 4: 2.93).
  constexpr int UNROLL = VEC == 8 ? 2 : 4;
  constexpr int IW = LPR >= 8 ? LPR : 16;  // index pairs fetched per round by one group (narrow rows: 16)
  constexpr int IPL = IW / LPR;            // ... per lane
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int sub = lane % LPR;
  const int slot = lane / LPR;
  const int64_t q = (block_id * (blockDim.x / kWave) + wave) * G + slot;
  if (q >= p.n_rows) return;
  const int64_t i = adj_row(p, q);
  int64_t lo, hi, code_base;
  if constexpr (DENSE) {
    lo = 0;
    hi = p.n_cols;
    code_base = i * p.n_cols;
  } else {
    lo = load_rowptr(p, i);
    hi = load_rowptr(p, i + 1);
    code_base = 0;
    if (hi - lo > p.long_threshold) return;  // hub row: long kernel
  }
  const int rest = p.D - 1;
  // SMALLD folds the rest bucket into the listed weights:  sum_d w_d s + w_rest (total - sum s)
  //   = sum_d (w_d - w_rest) s + w_rest total,  so no second accumulator for the listed operand rows is needed.
  SmallW sw;
  float w_rest = 0.f;
  if constexpr (SMALLD) {
    sw = small_weights(p, i);
    if (p.s_total) {
      w_rest = sw.pick(rest);
#pragma unroll
      for (int d = 0; d < 4; ++d) sw.w[d] = d < rest ? sw.w[d] - w_rest : 0.f;
    }
  }
  float red[4] = {0.f, 0.f, 0.f, 0.f};  // fused feature sum (reduce_cr in {1, 2, 4}): channel partials of this lane
  // a training forward of a one-column operand keeps the raw per-code sums of its rows (gnan_spmm_args.shell_out): a lane owns a
  // row here, so three more accumulators and a select per pair
  constexpr bool kShell = SMALLD && VEC == 1 && LPR == 1 && !BYCODE && !DENSE;
  float sh[3] = {0.f, 0.f, 0.f};
  // (uniform) one weight per pair from a per-neighbour table, no counts, no rest subtraction: see the index loads below
  const bool pair_weights = !SMALLD && !DENSE && p.weight_by_col && p.Cw == 1 && p.cnt == nullptr && !p.minus_rest && p.lut_row_stride != 0;

  for (int w0 = 0; w0 < p.W; w0 += TILE) {
    const int cw = w0 + sub * VEC;
    const bool col_ok = cw < p.W;
    Vec<VEC> acc, all;
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc.v[v] = all.v[v] = 0.f;

    // The group fetches IW = max(LPR, 16) index pairs per round — IPL per lane — so that narrow operand
    // rows (few lanes per group) still see 16 gathers between two dependent index loads.
    for (int64_t base = lo; base < hi; base += IW) {
      int colv[IPL], codev[IPL];
      bool wide = false;
      if constexpr (!DENSE && IPL % 4 == 0) {
        const int64_t e0 = base + sub * IPL;
        wide = e0 + IPL <= p.nnz;
        if (wide) {
          if constexpr (PACKED) load_col_run<IPL>(p.col + e0, colv);
          else load_index_run<IPL>(p.col + e0, p.code + e0, colv, codev);
        }
      }
      if (!wide) {
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          const int64_t e = base + sub * IPL + r;
          colv[r] = codev[r] = 0;
          if (e < hi) {
            if constexpr (!DENSE) colv[r] = p.col[e];
            if constexpr (!PACKED) codev[r] = p.code[code_base + e];
          }
        }
      }
      if constexpr (PACKED) {
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          codev[r] = static_cast<int>(static_cast<unsigned>(colv[r]) >> kPackShift);
          colv[r] = static_cast<int>(static_cast<unsigned>(colv[r]) & kPackMask);
        }
      }
      const int m = static_cast<int>(hi - base < IW ? hi - base : IW);
      // Per-neighbour weight table (the wide backward pass: weight = wt[c, d], one channel): the lane that holds a pair's index
      // entry fetches its weight too — ONE load instruction per round and group — and hands it out by shuffle like the column id.
      // Read inside the pair loop it was a second memory instruction per pair and lane: 133 -> 271 us on the arxiv shape.
      float wv[IPL];
      if constexpr (!SMALLD) {
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          wv[r] = 0.f;
          if (pair_weights && base + sub * IPL + r < hi) {
            const int dd = codev[r] < rest ? codev[r] : rest;
            wv[r] = p.lut[static_cast<int64_t>(colv[r]) * p.lut_row_stride + dd];
          }
        }
      }
      // IPL > 1: fully unrolled so that the register index j % IPL is static; IPL == 1: plain runtime loop
#pragma unroll(IPL > 1 ? IW / UNROLL : 1)
      for (int j0 = 0; j0 < (IPL > 1 ? IW : m); j0 += UNROLL) {
        if (IPL > 1 && j0 >= m) break;
        Raw<VEC> s[UNROLL];
        int d[UNROLL], c[UNROLL];
        float wp[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
          const int j = j0 + u;              // compile-time after unrolling: lane j / IPL holds it in register j % IPL
          if constexpr (DENSE) {
            c[u] = static_cast<int>(base) + j;
          } else {
            c[u] = __shfl(colv[j % IPL], j / IPL, LPR);
          }
          d[u] = __shfl(codev[j % IPL], j / IPL, LPR);
          d[u] = d[u] < rest ? d[u] : rest;
          if constexpr (!SMALLD) wp[u] = __shfl(wv[j % IPL], j / IPL, LPR);
          s[u].zero();
          if (j < m && col_ok)
            s[u].load(p.S, BYCODE ? static_cast<int64_t>(c[u]) * p.D + d[u] : static_cast<int64_t>(c[u]), p.s_stride, cw);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
          if (j0 + u < m) {
            const Vec<VEC> sv = s[u].widen();
            if constexpr (SMALLD) {
              const float w = sw.pick(d[u]);
#pragma unroll
              for (int v = 0; v < VEC; ++v) acc.v[v] = fmaf(w, sv.v[v], acc.v[v]);
              if constexpr (kShell) {
                sh[0] += d[u] == 0 ? sv.v[0] : 0.f;
                sh[1] += d[u] == 1 ? sv.v[0] : 0.f;
                sh[2] += d[u] == 2 ? sv.v[0] : 0.f;
              }
            } else {
              Vec<VEC> w;
              if (pair_weights) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) w.v[v] = wp[u];
              } else {
                w = edge_weights<VEC>(p, i, c[u], d[u], cw);
              }
#pragma unroll
              for (int v = 0; v < VEC; ++v) {
                acc.v[v] = fmaf(w.v[v], sv.v[v], acc.v[v]);
                all.v[v] += sv.v[v];
              }
            }
          }
        }
      }
    }
    if (col_ok) {
      if (p.s_total) {
        const Vec<VEC> tot = load_vec<VEC>(p.s_total + cw);
        if constexpr (SMALLD) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc.v[v] = fmaf(w_rest, tot.v[v], acc.v[v]);
        } else {
          const Vec<VEC> wr = row_weights<VEC>(p, i, rest, cw);
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc.v[v] = fmaf(wr.v[v], tot.v[v] - all.v[v], acc.v[v]);
        }
      }
      if (p.reduce_cr == 0) {
        store_vec<VEC>(p.Y + out_row(p, q, i) * p.y_stride + cw, acc);
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const int ch = (cw + v) & (p.reduce_cr - 1);  // reduce_cr is a power of two
#pragma unroll
          for (int c = 0; c < 4; ++c) red[c] += ch == c ? acc.v[v] : 0.f;
        }
      }
    }
  }
  if constexpr (kShell) {
    if (p.shell_out) {
      float* t = p.shell_out + out_row(p, q, i) * (p.D - 1);
      for (int dd = 0; dd < p.D - 1; ++dd) t[dd] = dd == 0 ? sh[0] : (dd == 1 ? sh[1] : sh[2]);
    }
  }
  if (p.reduce_cr) {
    // read-out fused into the epilogue (GNAN.py:72-73): add the channel partials of the group's lanes
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) red[c] += __shfl_xor(red[c], off);
    }
    if (sub == 0)
      for (int c = 0; c < p.reduce_cr; ++c) p.Y[out_row(p, q, i) * p.y_stride + c] = red[c];
  }
}

// ---------------------------------------------------------------------------------------------
// long kernel: one 256-thread workgroup per slice of a hub row
// ---------------------------------------------------------------------------------------------
template <int VEC, int LPR, bool SMALLD, bool DENSE, bool BYCODE, bool PACKED = false>
__device__ __forceinline__ void slice_body(const Params& p, const int s) {
  constexpr int G = kWave / LPR;
  constexpr int TILE = LPR * VEC;
  constexpr int NW = 4;  // waves per workgroup
  __shared__ float red[NW][2][TILE];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int sub = lane % LPR;
  const int slot = lane / LPR;
  // which hub row owns slice s: last r with long_slice_ptr[r] <= s
  int a = 0, b = p.n_long;
  while (b - a > 1) {
    const int mid = (a + b) >> 1;
    if (p.long_slice_ptr[mid] <= s) a = mid; else b = mid;
  }
  const int64_t q = p.long_rows[a];
  const int64_t i = adj_row(p, q);
  // dense layout: the "pairs" of row i are all n_cols neighbours, the column is the position, codes sit at i*n_cols
  const int64_t row_lo = DENSE ? 0 : load_rowptr(p, i), row_hi = DENSE ? p.n_cols : load_rowptr(p, i + 1);
  const int64_t code_base = DENSE ? i * p.n_cols : 0;
  const int64_t lo = row_lo + static_cast<int64_t>(s - p.long_slice_ptr[a]) * p.slice_edges;
  const int64_t hi = lo + p.slice_edges < row_hi ? lo + p.slice_edges : row_hi;
  const int rest = p.D - 1;
  SmallW sw;
  if constexpr (SMALLD) sw = small_weights(p, i);

  for (int w0 = 0; w0 < p.W; w0 += TILE) {
    const int cw = w0 + sub * VEC;
    const bool col_ok = cw < p.W;
    Vec<VEC> acc, all;
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc.v[v] = all.v[v] = 0.f;
    // wave `wave` takes 64-edge chunks wave, wave+NW, ...
    for (int64_t base = lo + static_cast<int64_t>(wave) * kWave; base < hi; base += NW * kWave) {
      const int64_t e = base + lane;
      int colv = 0, codev = 0;
      if (e < hi) {
        colv = DENSE ? static_cast<int>(e) : p.col[e];
        if constexpr (!PACKED) codev = p.code[code_base + e];
      }
      if constexpr (PACKED) {
        codev = static_cast<int>(static_cast<unsigned>(colv) >> kPackShift);
        colv = static_cast<int>(static_cast<unsigned>(colv) & kPackMask);
      }
      const int m = static_cast<int>(hi - base < kWave ? hi - base : kWave);
#pragma unroll 4
      for (int t = 0; t < LPR; ++t) {
        const int j = slot + t * G;
        const int c = __shfl(colv, j);
        int d = __shfl(codev, j);
        d = d < rest ? d : rest;
        if (j < m && col_ok) {
          const Vec<VEC> sv = load_operand<VEC>(p.S, BYCODE ? static_cast<int64_t>(c) * p.D + d : static_cast<int64_t>(c), p.s_stride, cw);
          if constexpr (SMALLD) {
            const float w = sw.pick(d);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc.v[v] = fmaf(w, sv.v[v], acc.v[v]);
          } else {
            const Vec<VEC> w = edge_weights<VEC>(p, i, c, d, cw);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc.v[v] = fmaf(w.v[v], sv.v[v], acc.v[v]);
          }
#pragma unroll
          for (int v = 0; v < VEC; ++v) all.v[v] += sv.v[v];
        }
      }
    }
    // groups of one wave -> group 0 (fixed butterfly order), then waves -> LDS -> wave 0
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        acc.v[v] += __shfl_xor(acc.v[v], off);
        all.v[v] += __shfl_xor(all.v[v], off);
      }
    }
    __syncthreads();
    if (slot == 0) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        red[wave][0][sub * VEC + v] = acc.v[v];
        red[wave][1][sub * VEC + v] = all.v[v];
      }
    }
    __syncthreads();
    if (wave == 0 && slot == 0 && col_ok) {
      float* out = p.partial + static_cast<int64_t>(s) * 2 * p.W;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        float x = 0.f, y = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          x += red[w][0][sub * VEC + v];
          y += red[w][1][sub * VEC + v];
        }
        out[cw + v] = x;
        out[p.W + cw + v] = y;
      }
    }
  }
}

// One launch covers everything: workgroups [0, n_slices) take the hub-row slices (they start first,
// so the long-latency slices overlap the bulk), the rest take 4*G ordinary rows each.
// BYCODE (operand row = (neighbour, hop code), the narrow-operand backward) is a template parameter: as a run-time
// flag its address arithmetic cost the W = 64 kernels 4 VGPRs and the bf16 variant 20 B of scratch (bf16 rows 2.85 -> 3.35 ms).
template <int VEC, int LPR, bool DENSE, bool SMALLD, bool BYCODE = false, bool PACKED = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((SMALLD && LPR >= 8) ? 8 : 1)))
void spmm_kernel(const Params p) {
  if constexpr (!DENSE) {
    if (static_cast<int>(blockIdx.x) < p.n_slices) {
      slice_body<VEC, LPR, SMALLD, false, BYCODE, PACKED>(p, static_cast<int>(blockIdx.x));
      return;
    }
    rows_body<VEC, LPR, false, SMALLD, BYCODE, PACKED>(p, static_cast<int64_t>(blockIdx.x) - p.n_slices);
  } else {
    if (p.n_slices > 0) {      // few rows, many neighbours: every row is cut into slices, there are no row blocks
      slice_body<VEC, LPR, SMALLD, true, false>(p, static_cast<int>(blockIdx.x));
      return;
    }
    rows_body<VEC, LPR, true, SMALLD, false>(p, static_cast<int64_t>(blockIdx.x));
  }
}

// ---------------------------------------------------------------------------------------------
// Narrow operand rows (W in {1, 2, 4} floats, the sum-first order of GNAN.py:157: S = f_sums) with the hottest rows in LDS.
//
// A W = 1 aggregation is one 4-byte gather per listed pair, and every gather is a request to the L2 (TCC): the kernel
// sits on the L2 request rate (10M-node R-MAT: 1.1e8 pairs in 1.0 ms = 109 G requests/s; the operand itself is 40 MB and
// never leaves the Infinity Cache).  A power-law graph sends a large share of those requests to very few rows — the
// 32 768 most listed of the 10M nodes receive 42 % of the pairs (tools/hot_coverage.py) — and the host already keeps a
// compact copy of the most listed rows behind the operand, most listed first, with the column ids of the degree-sorted
// copy pointing there (HopGraph.hot_columns).  This kernel loads the head of that copy (hot_n rows, 64 KB: the 16 384 most
// listed at W = 1, 32 % of the pairs) into LDS once per workgroup and serves their gathers from there: no L2 request at
// all for them.  Two 1024-thread workgroups per CU (32 waves, 64 VGPRs: the kernel is as much bound by the latency of its
// dependent rowptr -> index -> gather chains as by the request rate — with one workgroup and a 128-KB table it LOST 7 %),
// persistent: each loads the table once and then walks its share of the hub-row slices and of the row blocks (rows are
// degree-sorted: a round-robin share is balanced).  Sixteen waves = four "virtual" 4-wave workgroups, numbered like the
// blocks of spmm_kernel for the ordinary rows; hub slices share spmm_kernel's partial-sum layout and its fix-up kernel.
// 10M-node R-MAT, W = 1: 1.04 -> 0.91 ms.  One lane per row, the same arithmetic per row as spmm_kernel<VEC, 1, false, true, false, true>
// (weights folded with the rest bucket): ordinary rows come out bit-identical, hub rows add their pairs in another order.
// ---------------------------------------------------------------------------------------------
// Branch-free: a divergent `if (hot) ds_read else global_load` makes the compiler wait at every join, i.e. one gather in
// flight per lane (measured: 1.46 ms against 1.04 ms for the plain kernel).  Both loads are always issued — the hot lanes
// of the global load all read row hot_lo (one line: a single request per wavefront, an L1 hit), the cold lanes of the LDS
// read all read entry 0 (a broadcast) — and the value is selected afterwards.
template <int VEC>
__device__ __forceinline__ Vec<VEC> hot_gather(const Params& p, const float* hot, int c) {
  const int64_t r = static_cast<int64_t>(c) - p.hot_lo;
  const bool is_hot = r >= 0 && r < p.hot_n;
  const Vec<VEC> g = load_vec<VEC>(static_cast<const float*>(p.S) + (is_hot ? p.hot_lo : static_cast<int64_t>(c)) * VEC);
  const int rl = is_hot ? static_cast<int>(r) : 0;
  Vec<VEC> l;
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(hot + rl * 4);
    l.v[0] = t.x; l.v[1] = t.y; l.v[2] = t.z; l.v[3] = t.w;
  } else if constexpr (VEC == 2) {
    const float2 t = *reinterpret_cast<const float2*>(hot + rl * 2);
    l.v[0] = t.x; l.v[1] = t.y;
  } else {
    l.v[0] = hot[rl];
  }
  Vec<VEC> out;
#pragma unroll
  for (int v = 0; v < VEC; ++v) out.v[v] = is_hot ? l.v[v] : g.v[v];
  return out;
}

template <int VEC>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))     // two workgroups = 32 waves per CU
void spmm_hot_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) float hot[];        // [hot_n * VEC]
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int sub = wave >> 2, w4 = wave & 3;                           // virtual 4-wave workgroup, wave inside it
  {
    const float* src = static_cast<const float*>(p.S) + p.hot_lo * p.s_stride;      // rows are contiguous: s_stride == W
    for (int i = tid; i < p.hot_n * VEC; i += 1024) hot[i] = src[i];
  }
  __syncthreads();
  const int rest = p.D - 1;

  // ---- hub-row slices: one WAVE per slice, a contiguous run of slices per wave -----------------------------------------
  // (a workgroup per slice as in spmm_kernel would leave 4 slices in flight per CU, and every slice starts with a chain of
  // dependent loads — which hub row, its bounds, its weights: 45 rounds of that chain cost 0.6 ms.  A wave walks its run
  // front to back, so the row of the next slice is found by stepping, not by searching; no barriers.  The partial sums of
  // a slice are added in lane order by a fixed butterfly: deterministic, though not in spmm_kernel's order.)
  {
    const int n_waves = static_cast<int>(gridDim.x) * 16;
    const int per = (p.n_slices + n_waves - 1) / n_waves;
    const int gw = static_cast<int>(blockIdx.x) * 16 + wave;
    const int s_lo = gw * per, s_hi = s_lo + per < p.n_slices ? s_lo + per : p.n_slices;
    int a = 0;
    if (s_lo < s_hi) {
      int b = p.n_long;
      while (b - a > 1) {
        const int mid = (a + b) >> 1;
        if (p.long_slice_ptr[mid] <= s_lo) a = mid; else b = mid;
      }
    }
    for (int sidx = s_lo; sidx < s_hi; ++sidx) {
      while (p.long_slice_ptr[a + 1] <= sidx) ++a;
      const int64_t q = p.long_rows[a];
      const int64_t i = adj_row(p, q);
      const int64_t row_lo = load_rowptr(p, i), row_hi = load_rowptr(p, i + 1);
      const int64_t lo = row_lo + static_cast<int64_t>(sidx - p.long_slice_ptr[a]) * p.slice_edges;
      const int64_t hi = lo + p.slice_edges < row_hi ? lo + p.slice_edges : row_hi;
      const SmallW sw = small_weights(p, i);
      Vec<VEC> acc, all;
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc.v[v] = all.v[v] = 0.f;
      constexpr int SF = 8;                              // pairs in flight per lane
      for (int64_t base = lo + lane; base < hi; base += SF * kWave) {
        unsigned ce[SF];
#pragma unroll
        for (int k = 0; k < SF; ++k) {
          const int64_t e = base + static_cast<int64_t>(k) * kWave;
          ce[k] = e < hi ? static_cast<unsigned>(p.col[e]) : 0u;
        }
        // no branch around a gather (a guarded region ends in s_waitcnt 0: one gather in flight): pairs past the end read the
        // first hot row — an LDS hit, no request — and are dropped by a select
        Vec<VEC> sv[SF];
#pragma unroll
        for (int k = 0; k < SF; ++k) {
          const bool ok = base + static_cast<int64_t>(k) * kWave < hi;
          sv[k] = hot_gather<VEC>(p, hot, ok ? static_cast<int>(ce[k] & kPackMask) : static_cast<int>(p.hot_lo));
        }
#pragma unroll
        for (int k = 0; k < SF; ++k) {
          const bool ok = base + static_cast<int64_t>(k) * kWave < hi;
          int d = static_cast<int>(ce[k] >> kPackShift);
          d = d < rest ? d : rest;
          const float w = sw.pick(d);
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            acc.v[v] = ok ? fmaf(w, sv[k].v[v], acc.v[v]) : acc.v[v];
            all.v[v] = ok ? all.v[v] + sv[k].v[v] : all.v[v];
          }
        }
      }
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          acc.v[v] += __shfl_xor(acc.v[v], off);
          all.v[v] += __shfl_xor(all.v[v], off);
        }
      }
      if (lane == 0) {
        float* out = p.partial + static_cast<int64_t>(sidx) * 2 * p.W;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          out[v] = acc.v[v];
          out[p.W + v] = all.v[v];
        }
      }
    }
  }

  // ---- ordinary rows: virtual workgroup = 256 rows, one lane per row (as rows_body<VEC, 1, false, true, false, true>) ----
  const int64_t n_vblocks = (p.n_rows + 255) / 256;
  for (int64_t vb = static_cast<int64_t>(blockIdx.x) * 4 + sub; vb < n_vblocks; vb += static_cast<int64_t>(gridDim.x) * 4) {
    const int64_t q = (vb * 4 + w4) * kWave + lane;
    if (q >= p.n_rows) continue;
    const int64_t i = adj_row(p, q);
    const int64_t lo = load_rowptr(p, i), hi = load_rowptr(p, i + 1);
    if (hi - lo > p.long_threshold) continue;           // hub row: sliced above
    SmallW sw = small_weights(p, i);
    float w_rest = 0.f;
    if (p.s_total) {
      w_rest = sw.pick(rest);
#pragma unroll
      for (int d = 0; d < 4; ++d) sw.w[d] = d < rest ? sw.w[d] - w_rest : 0.f;
    }
    Vec<VEC> acc;
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc.v[v] = 0.f;
    for (int64_t base = lo; base < hi; base += 16) {
      int colv[16];
      if (base + 16 <= p.nnz) {
        load_col_run<16>(p.col + base, colv);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) colv[r] = base + r < hi ? p.col[base + r] : 0;
      }
      const int m = static_cast<int>(hi - base < 16 ? hi - base : 16);
      constexpr int FLY = VEC == 1 ? 8 : (VEC == 2 ? 4 : 2);      // gathers in flight per lane (64 VGPRs: 8 waves per SIMD)
#pragma unroll
      for (int j0 = 0; j0 < 16; j0 += FLY) {
        if (j0 >= m) break;
        Vec<VEC> sv[FLY];
        int d[FLY];
#pragma unroll
        for (int u = 0; u < FLY; ++u) {
          const unsigned ce = static_cast<unsigned>(colv[j0 + u]);
          d[u] = static_cast<int>(ce >> kPackShift);
          d[u] = d[u] < rest ? d[u] : rest;
          // (no branch around a gather, see the slice loop: entries past the row end read the first hot row from LDS)
          sv[u] = hot_gather<VEC>(p, hot, j0 + u < m ? static_cast<int>(ce & kPackMask) : static_cast<int>(p.hot_lo));
        }
#pragma unroll
        for (int u = 0; u < FLY; ++u) {
          const float w = sw.pick(d[u]);
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc.v[v] = j0 + u < m ? fmaf(w, sv[u].v[v], acc.v[v]) : acc.v[v];
        }
      }
    }
    if (p.s_total) {
      const Vec<VEC> tot = load_vec<VEC>(p.s_total);
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc.v[v] = fmaf(w_rest, tot.v[v], acc.v[v]);
    }
    store_vec<VEC>(p.Y + out_row(p, q, i) * p.y_stride, acc);
  }
}

// fix-up: add a hub row's slices in a fixed order, apply the rest-bucket term, store the row.
// One WAVE per hub row, four rows per workgroup, no barriers: a power-law graph has tens of thousands of hub rows with
// one or two slices each (R-MAT 10M/100M: 33 068 rows, 45 284 slices, at most 114 per row), so a workgroup per row was
// bound by the workgroup launch rate (0.10 ms).  Lane = column (64 columns per pass); narrower operands put
// K = 64 / W' lanes on a column (W' = W rounded up to a power of two), lane k takes slices k, k + K, ...; loads are
// issued eight at a time; the K partial sums meet in a fixed butterfly.  One fixed order per row: bit-reproducible.
__global__ __launch_bounds__(256) void spmm_long_fixup_kernel(const Params p) {
  const int lane = threadIdx.x & (kWave - 1);
  const int r = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  if (r >= p.n_long) return;
  const int64_t q = p.long_rows[r];
  const int64_t i = adj_row(p, q);
  const int s0 = p.long_slice_ptr[r], s1 = p.long_slice_ptr[r + 1];
  int wp = 1;
  while (wp < p.W && wp < kWave) wp <<= 1;
  const int K = kWave / wp;              // slice lanes per column
  const int k = lane / wp;
  float chan = 0.f;                      // lane c < reduce_cr accumulates channel c over the column passes
  for (int w0 = 0; w0 < p.W; w0 += kWave) {
    const int w = w0 + lane % wp;
    float acc = 0.f, all = 0.f;
    if (w < p.W) {
      const float* src = p.partial + w;
      const int64_t row = 2 * static_cast<int64_t>(p.W);
      int s = s0 + k;
      for (; s + 7 * K < s1; s += 8 * K) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a[u] = src[(s + u * K) * row];
          b[u] = src[(s + u * K) * row + p.W];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          acc += a[u];
          all += b[u];
        }
      }
      for (; s < s1; s += K) {
        acc += src[s * row];
        all += src[s * row + p.W];
      }
    }
    for (int off = wp; off < kWave; off <<= 1) {   // slice lanes of a column: fixed butterfly, every lane ends with the sum
      acc += __shfl_xor(acc, off);
      all += __shfl_xor(all, off);
    }
    const bool owner = k == 0 && w < p.W;
    if (owner) {
      if (p.s_total) {
        const Vec<1> wr = row_weights<1>(p, i, p.D - 1, w);
        acc = fmaf(wr.v[0], p.s_total[w] - all, acc);
      }
      if (p.reduce_cr == 0) p.Y[out_row(p, q, i) * p.y_stride + w] = acc;
    }
    if (p.reduce_cr) {  // fixed butterfly: offsets stay multiples of reduce_cr, so channels never mix
      float v = owner ? acc : 0.f;
      for (int off = kWave / 2; off >= p.reduce_cr; off >>= 1) v += __shfl_xor(v, off);
      chan += v;
    }
  }
  if (p.reduce_cr && lane < p.reduce_cr) p.Y[out_row(p, q, i) * p.y_stride + lane] = chan;
}

constexpr int kHotLdsFloats = 16384;   // 64 KB of hot operand rows per workgroup (the default dynamic-LDS limit): two workgroups per CU

// can the persistent hot-row kernel take this call?  (what the host wrapper sets up: functional.spmm_launch, narrow walk)
bool hot_kernel_applies(const gnan_spmm_args* a) {
  auto aligned = [](const void* ptr, size_t n) { return (reinterpret_cast<uintptr_t>(ptr) % n) == 0; };
  const int W = a->W;
  return a->hot_rows > 0 && a->rowptr != nullptr && a->packed_index && a->s_dtype == GNAN_F32 && (W == 1 || W == 2 || W == 4) &&
         a->s_stride == W && a->Cw == 1 && a->D <= 4 && !a->weight_by_col && !a->minus_rest && !a->s_by_code && a->reduce_cr == 0 &&
         aligned(a->S, 16) && aligned(a->Y, 4 * static_cast<size_t>(W)) && a->y_stride % W == 0 &&
         (!a->s_total || aligned(a->s_total, 4 * static_cast<size_t>(W))) && a->hot_lo >= 0 &&
         a->hot_lo + a->hot_rows <= a->n_cols && static_cast<int64_t>(a->hot_rows) * W <= kHotLdsFloats && a->nnz > 0;
}

template <int VEC>
int launch_hot(const Params& p, hipStream_t st) {
  static int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  const size_t lds = static_cast<size_t>(p.hot_n) * VEC * sizeof(float);
  hipLaunchKernelGGL((spmm_hot_kernel<VEC>), dim3(static_cast<unsigned>(cus) * 2), dim3(1024), lds, st, p);
  if (int rc = gnan::check_launch("spmm_hot_kernel")) return rc;
  if (p.n_slices > 0) {
    hipLaunchKernelGGL(spmm_long_fixup_kernel, dim3(static_cast<unsigned>((p.n_long + 3) / 4)), dim3(256), 0, st, p);
    if (int rc = gnan::check_launch("spmm_long_fixup_kernel")) return rc;
  }
  return GNAN_OK;
}

// ---------------------------------------------------------------------------------------------
// shell sums:  T[q, d, :] = sum_{e in row, code_e == d} S[col_e, :]   (and the rest bucket)
// Needed only by the backward pass (gradient w.r.t. the weight table):  dwt[q, d, c] =
// sum_{w = c mod Cw} dY[q, w] * T[q, d, w].  Same traversal as the rows kernel; a lane owns its
// columns of T[q, :, :], so plain read-modify-write on global memory is race-free.
// ---------------------------------------------------------------------------------------------
template <int VEC, int LPR, bool DENSE>
__global__ __launch_bounds__(256) void spmm_shell_sums_kernel(const Params p) {
  constexpr int G = kWave / LPR;
  constexpr int TILE = LPR * VEC;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int sub = lane % LPR;
  const int slot = lane / LPR;
  const int64_t q = (static_cast<int64_t>(blockIdx.x) * (blockDim.x / kWave) + wave) * G + slot;
  if (q >= p.n_rows) return;
  const int64_t i = adj_row(p, q);
  int64_t lo, hi, code_base;
  if constexpr (DENSE) {
    lo = 0; hi = p.n_cols; code_base = i * p.n_cols;
  } else {
    lo = load_rowptr(p, i); hi = load_rowptr(p, i + 1); code_base = 0;
  }
  const int rest = p.D - 1;
  float* T = p.Y + q * static_cast<int64_t>(p.D) * p.W;
  for (int w0 = 0; w0 < p.W; w0 += TILE) {
    const int cw = w0 + sub * VEC;
    const bool col_ok = cw < p.W;
    Vec<VEC> all;
#pragma unroll
    for (int v = 0; v < VEC; ++v) all.v[v] = 0.f;
    for (int64_t base = lo; base < hi; base += LPR) {
      const int64_t e = base + sub;
      int colv = 0, codev = 0;
      if (e < hi) {
        if constexpr (!DENSE) colv = p.col[e];
        codev = p.code[code_base + e];
      }
      const int m = static_cast<int>(hi - base < LPR ? hi - base : LPR);
      for (int j = 0; j < m; ++j) {
        int c;
        if constexpr (DENSE) c = static_cast<int>(base) + j; else c = __shfl(colv, j, LPR);
        int d = __shfl(codev, j, LPR);
        d = d < rest ? d : rest;
        if (col_ok) {
          const Vec<VEC> sv = load_operand<VEC>(p.S, c, p.s_stride, cw);
          float* t = T + static_cast<int64_t>(d) * p.W + cw;
          Vec<VEC> cur = load_vec<VEC>(t);
#pragma unroll
          for (int v = 0; v < VEC; ++v) { cur.v[v] += sv.v[v]; all.v[v] += sv.v[v]; }
          store_vec<VEC>(t, cur);
        }
      }
    }
    if (col_ok && p.s_total) {
      const Vec<VEC> tot = load_vec<VEC>(p.s_total + cw);
      Vec<VEC> r;
#pragma unroll
      for (int v = 0; v < VEC; ++v) r.v[v] = tot.v[v] - all.v[v];
      store_vec<VEC>(T + static_cast<int64_t>(rest) * p.W + cw, r);
    }
  }
}

template <int VEC, int LPR>
int launch_shell(const Params& p, bool dense, hipStream_t st) {
  constexpr int G = kWave / LPR;
  const int rows_per_block = 4 * G;
  const int64_t blocks = (p.n_rows + rows_per_block - 1) / rows_per_block;
  if (blocks > 0x7fffffffLL) return gnan::fail(GNAN_ERR_UNSUPPORTED, "shell_sums: too many rows for one launch");
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
  if (dense) {
    hipLaunchKernelGGL((spmm_shell_sums_kernel<VEC, LPR, true>), grid, block, 0, st, p);
  } else {
    hipLaunchKernelGGL((spmm_shell_sums_kernel<VEC, LPR, false>), grid, block, 0, st, p);
  }
  return gnan::check_launch("spmm_shell_sums_kernel");
}

template <int VEC>
int launch_shell_lpr(const Params& p, int lpr, bool dense, hipStream_t st) {
  switch (lpr) {
    case 1: return launch_shell<VEC, 1>(p, dense, st);
    case 2: return launch_shell<VEC, 2>(p, dense, st);
    case 4: return launch_shell<VEC, 4>(p, dense, st);
    case 8: return launch_shell<VEC, 8>(p, dense, st);
    case 16: return launch_shell<VEC, 16>(p, dense, st);
    case 32: return launch_shell<VEC, 32>(p, dense, st);
    default: return launch_shell<VEC, 64>(p, dense, st);
  }
}

// ---------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------
template <int VEC, int LPR>
int launch(const Params& p, bool dense, bool smalld, hipStream_t st) {
  constexpr int G = kWave / LPR;
  const int rows_per_block = 4 * G;
  const int n_slices = p.n_slices;
  const int64_t blocks = dense && n_slices > 0 ? n_slices : (p.n_rows + rows_per_block - 1) / rows_per_block + n_slices;
  if (blocks > 0x7fffffffLL) return gnan::fail(GNAN_ERR_UNSUPPORTED, "spmm: too many rows for one launch");
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
  if (p.s_by_code) {
    if constexpr (VEC <= 4 && VEC * LPR <= 32) {  // validate(): fp32 rows of at most 32 columns, CSR layout
      if (smalld) {
        hipLaunchKernelGGL((spmm_kernel<VEC, LPR, false, true, true>), grid, block, 0, st, p);
      } else {
        hipLaunchKernelGGL((spmm_kernel<VEC, LPR, false, false, true>), grid, block, 0, st, p);
      }
    } else {
      return gnan::fail(GNAN_ERR_UNSUPPORTED, "spmm: s_by_code covers operand rows of at most 32 columns");
    }
  } else if (dense) {
    hipLaunchKernelGGL((spmm_kernel<VEC, LPR, true, false>), grid, block, 0, st, p);
  } else if (smalld) {
    if (p.packed) {
      hipLaunchKernelGGL((spmm_kernel<VEC, LPR, false, true, false, true>), grid, block, 0, st, p);
    } else {
      hipLaunchKernelGGL((spmm_kernel<VEC, LPR, false, true>), grid, block, 0, st, p);
    }
  } else {
    hipLaunchKernelGGL((spmm_kernel<VEC, LPR, false, false>), grid, block, 0, st, p);
  }
  if (int rc = gnan::check_launch("spmm_kernel")) return rc;
  if (n_slices > 0) {
    hipLaunchKernelGGL(spmm_long_fixup_kernel, dim3(static_cast<unsigned>((p.n_long + 3) / 4)), dim3(256), 0, st, p);
    if (int rc = gnan::check_launch("spmm_long_fixup_kernel")) return rc;
  }
  return GNAN_OK;
}

template <int VEC>
int launch_lpr(const Params& p, int lpr, bool dense, bool smalld, hipStream_t st) {
  switch (lpr) {
    case 1: return launch<VEC, 1>(p, dense, smalld, st);
    case 2: return launch<VEC, 2>(p, dense, smalld, st);
    case 4: return launch<VEC, 4>(p, dense, smalld, st);
    case 8: return launch<VEC, 8>(p, dense, smalld, st);
    case 16: return launch<VEC, 16>(p, dense, smalld, st);
    case 32: return launch<VEC, 32>(p, dense, smalld, st);
    default: return launch<VEC, 64>(p, dense, smalld, st);
  }
}

int validate(const gnan_spmm_args* a) {
  GNAN_REQUIRE(a != nullptr, "spmm: null args");
  GNAN_REQUIRE(a->n_rows >= 0 && a->n_cols >= 0, "spmm: negative size");
  GNAN_REQUIRE(a->W >= 1, "spmm: W must be >= 1 (got %d)", a->W);
  GNAN_REQUIRE(a->D >= 1 && a->D <= GNAN_MAX_CODES, "spmm: D must be in [1, %d] (got %d)", GNAN_MAX_CODES, a->D);
  GNAN_REQUIRE(a->Cw >= 1, "spmm: Cw must be >= 1");
  GNAN_REQUIRE(a->n_cols <= 0x7fffffffLL, "spmm: n_cols exceeds int32 column ids");
  if (a->n_rows == 0) return GNAN_OK;
  GNAN_REQUIRE(a->S && a->lut && a->Y && (a->code || a->packed_index), "spmm: null S / lut / Y / code");
  GNAN_REQUIRE((a->rowptr == nullptr) == (a->col == nullptr), "spmm: rowptr and col must both be set (CSR) or both NULL (dense)");
  GNAN_REQUIRE(a->s_stride >= a->W && (a->reduce_cr != 0 || a->y_stride >= a->W), "spmm: row stride smaller than W");
  if (a->s_dtype != GNAN_F32 && a->s_dtype != GNAN_BF16) return gnan::fail(GNAN_ERR_BAD_ARG, "spmm: unknown operand dtype %d", a->s_dtype);
  if (a->s_dtype == GNAN_BF16) {
    if (a->W % 8 != 0 || a->s_stride % 8 != 0 || reinterpret_cast<uintptr_t>(a->S) % 16 != 0 || a->rowptr == nullptr)
      return gnan::fail(GNAN_ERR_UNSUPPORTED, "spmm: bf16 operand rows need the CSR layout, W %% 8 == 0 and 16-B aligned rows");
    if (a->reduce_cr == 0 && (a->y_stride % 4 != 0 || reinterpret_cast<uintptr_t>(a->Y) % 16 != 0))
      return gnan::fail(GNAN_ERR_UNSUPPORTED, "spmm: bf16 operand rows need a 16-B aligned fp32 output");
    if (a->s_total && reinterpret_cast<uintptr_t>(a->s_total) % 16 != 0)
      return gnan::fail(GNAN_ERR_UNSUPPORTED, "spmm: s_total must be 16-B aligned");
  }
  GNAN_REQUIRE(!(a->weight_by_col && a->s_total), "spmm: weight_by_col excludes the rest-bucket term (add it outside)");
  GNAN_REQUIRE(!a->s_by_code || (a->rowptr != nullptr && a->s_total == nullptr && a->s_dtype == GNAN_F32),
               "spmm: s_by_code needs the CSR layout, fp32 rows and no rest-bucket term");
  if (a->s_by_code && a->W > 32)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "spmm: s_by_code covers operand rows of at most 32 columns (got W=%d)", a->W);
  GNAN_REQUIRE(!a->scatter_out || a->row_ids, "spmm: scatter_out needs row_ids");
  if (a->packed_index && (a->rowptr == nullptr || a->D > 4 || a->Cw != 1 || a->weight_by_col || a->minus_rest || a->s_by_code ||
                          a->n_cols > static_cast<int64_t>(kPackMask) + 1))
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "spmm: packed index entries need the CSR layout, D <= 4, one weight channel, plain "
                      "forward weights and n_cols <= 2^29");
  if (a->reduce_cr != 0) {
    const int cr = a->reduce_cr;
    if (!(cr == 1 || cr == 2 || cr == 4) || a->W % cr != 0)
      return gnan::fail(GNAN_ERR_UNSUPPORTED, "spmm: fused read-out needs reduce_cr in {1, 2, 4} dividing W (got %d, W=%d)",
                        cr, a->W);
    GNAN_REQUIRE(a->y_stride >= cr, "spmm: y_stride smaller than reduce_cr");
  }
  if (a->n_long > 0) {
    GNAN_REQUIRE(a->rowptr != nullptr || (a->n_long == a->n_rows && a->long_threshold == 0),
                 "spmm: a row plan for the dense layout must slice every row (n_long == n_rows, long_threshold == 0)");
    GNAN_REQUIRE(a->long_rows && a->long_slice_ptr && a->slice_edges > 0 && a->n_slices > 0,
                 "spmm: incomplete long-row plan");
  }
  return GNAN_OK;
}

}  // namespace

extern "C" size_t gnan_spmm_fwd_workspace_bytes(const gnan_spmm_args* a) {
  if (!a || a->n_long <= 0) return 0;
  return static_cast<size_t>(a->n_slices) * 2 * static_cast<size_t>(a->W) * sizeof(float);
}

namespace {

Params make_params(const gnan_spmm_args* a) {
  Params p;
  p.n_rows = a->n_rows; p.n_cols = a->n_cols; p.nnz = a->nnz > 0 ? a->nnz : 0;
  p.rowptr = a->rowptr; p.rowptr_is64 = a->rowptr_is64;
  p.col = a->col; p.code = a->code; p.row_ids = a->row_ids;
  p.S = a->S; p.W = a->W; p.s_stride = a->s_stride;
  p.lut = a->lut; p.lut_row_stride = a->lut_row_stride; p.D = a->D; p.Cw = a->Cw;
  p.cnt = a->cnt; p.cnt_stride = a->cnt_stride; p.s_total = a->s_total;
  p.weight_by_col = a->weight_by_col; p.minus_rest = a->minus_rest; p.reduce_cr = a->reduce_cr;
  p.scatter_out = a->scatter_out;
  p.s_by_code = a->s_by_code;
  p.packed = a->packed_index;
  p.Y = a->Y; p.y_stride = a->y_stride;
  p.long_threshold = a->n_long > 0 ? a->long_threshold : INT64_MAX;
  p.long_rows = a->long_rows; p.long_slice_ptr = a->long_slice_ptr;
  p.n_long = a->n_long > 0 ? a->n_long : 0;
  p.n_slices = a->n_long > 0 ? a->n_slices : 0;
  p.slice_edges = a->slice_edges;
  p.partial = static_cast<float*>(a->workspace);
  p.hot_lo = a->hot_lo; p.hot_n = a->hot_rows;
  p.shell_out = a->shell_out;
  return p;
}

// operand rows are read 16 B per lane when shape and alignment allow it, else 4 B per lane
void pick_tiling(const gnan_spmm_args* a, const float* out, int64_t out_stride, int* vec, int* lpr) {
  auto aligned = [](const void* ptr, size_t n) { return (reinterpret_cast<uintptr_t>(ptr) % n) == 0; };
  *vec = 1;
  if (a->W % 4 == 0 && a->s_stride % 4 == 0 && out_stride % 4 == 0 && aligned(a->S, 16) && aligned(out, 16) &&
      (!a->s_total || aligned(a->s_total, 16)))
    *vec = 4;
  *lpr = 1;
  while (*lpr * *vec < a->W && *lpr < kWave) *lpr <<= 1;
}

}  // namespace

namespace {

// ---------------------------------------------------------------------------------------------
// Gradient w.r.t. the weight table without materialising the per-shell sums (truncated-hop case: D <= 4, Cw == 1).
//   dwt[q, d] = inv(q, d) * sum_w dY[q, w] * T[q, d, w],   T[q, d, :] = sum of the operand rows of q's hop-d pairs,
//   T[q, rest, :] = total - sum_{d < rest} T[q, d, :]  (or the listed rest pairs when there is no rest bucket).
// Same traversal as the forward kernel (gather the operand rows once), four accumulators per lane instead of one,
// contracted with the row's dY in the epilogue; spmm_shell_sums_kernel + torch needed a [n, D, W] tensor and one
// global read-modify-write per listed pair for this.  Rows / hub slices / fix-up as in the forward.  With
// `reduce_rows` the rows' contributions meet in fixed-order float64 partials (workgroup, then grid): dlut[d].
// ---------------------------------------------------------------------------------------------
struct GradParams {
  const float* dY;       // [n_rows, dy_channels]; column w of the operand pairs with dY[q, w % dy_channels]
  int64_t dy_stride;
  int dy_channels;
  float* dwt;            // per-row mode: [n_rows, D]
  double* blk;           // reduce mode: [n_row_blocks + n_long, 4]
  float* slice_T;        // [n_slices, 4, W]
  int reduce_rows;
  int64_t n_row_blocks;
  // BWD mode (gnan_spmm_bwd_narrow): the traversal runs over the TRANSPOSED adjacency, p.S holds one row per (neighbour,
  // hop code), 2 * half wide: [ dY_i / cnt(i, d) | dY_i / cnt(i, rest) ]; row j's own operand row S_j is contracted with
  // the per-code sums for the table gradient and the same sums, weighted by the table, are its operand gradient
  const float* s_rows;   // [n_rows, w_real] operand rows of the OUTPUT rows
  int64_t s_rows_stride;
  int half, w_real;      // p.W == 2 * half (half a power of two >= w_real)
  float* dS;             // [n_rows, w_real]
  int64_t ds_stride;
  int with_rest;
  const float* ds_add;   // optional [w_real]: added to every row of dS (the rest bucket's column-sum term) ...
  const float* ds_scale; // ... times this device scalar when given (rho(0) = lut[rest]: the caller hands over the bare column sums)
  const float* rest_total;   // optional [w_real], with rest_q [w_real]: dlut[D - 1] += <rest_total, rest_q> in the final pass
  const float* rest_q;
  int hot_code_lo, hot_codes;   // spmm_bwd_hot_kernel: packed rows [hot_lo, hot_lo + hot_n) of these code blocks are served from LDS
};

// BWD epilogue of one row for this lane's VEC columns: lanes of the first half hold A_d = sum over the row's code-d pairs
// of dY / cnt(., d), their partners (half columns further) Q = sum over ALL pairs of dY / cnt(., rest).
//   dS_j = sum_{d < rest} lut[d] A_d - lut[rest] Q        dlut[d] += <S_j, A_d>      dlut[rest] -= <S_j, Q>
template <int VEC, int LPR>
__device__ __forceinline__ void bwd_finish(const Params& p, const GradParams& gp, int64_t oq, int cw,
                                           Vec<VEC> (&t)[4], float (&pd)[4]) {
  const int rest = p.D - 1;
  const bool listed_rest = !gp.with_rest;        // no rest bucket: code D-1 is an ordinary listed shell
  float all[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) all[v] = t[0].v[v] + t[1].v[v] + t[2].v[v] + t[3].v[v];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    // Q of this column: the partner lane's sum over all codes (same v) — or, with one lane per row (rows of 2 or 4
    // floats), the value `half` positions further in this lane's own vector
    float q;
    if constexpr (LPR == 1) q = all[(v + VEC / 2) % VEC];
    else q = __shfl_xor(all[v], LPR / 2);
    const int w = cw + v;
    if (w < gp.w_real) {
      const float sj = gp.s_rows[oq * gp.s_rows_stride + w];
      float ds = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        if (d < rest || (listed_rest && d == rest)) {
          ds = fmaf(p.lut[d], t[d].v[v], ds);
          pd[d] = fmaf(sj, t[d].v[v], pd[d]);
        }
      }
      if (gp.with_rest) {
        ds = fmaf(-p.lut[rest], q, ds);
        pd[rest & 3] = fmaf(-sj, q, pd[rest & 3]);
      }
      if (gp.ds_add) ds += gp.ds_scale ? __fmul_rn(*gp.ds_scale, gp.ds_add[w]) : gp.ds_add[w];
      gp.dS[oq * gp.ds_stride + w] = ds;
    }
  }
}

template <int VEC>
__device__ __forceinline__ void grad_finish(const Params& p, const GradParams& gp, int64_t i, int64_t oq, int cw,
                                            bool col_ok, Vec<VEC> (&t)[4], float (&pd)[4]) {
  // contract this lane's columns of T with dY and fold the rest bucket
  const int rest = p.D - 1;
  if (col_ok) {
    Vec<VEC> dy;
#pragma unroll
    for (int v = 0; v < VEC; ++v) dy.v[v] = gp.dY[oq * gp.dy_stride + (cw + v) % gp.dy_channels];
    if (p.s_total) {
      const Vec<VEC> tot = load_vec<VEC>(p.s_total + cw);
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        float lower = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) lower += d < rest ? t[d].v[v] : 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) t[d].v[v] = d == rest ? tot.v[v] - lower : t[d].v[v];
      }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int v = 0; v < VEC; ++v) pd[d] = fmaf(dy.v[v], t[d].v[v], pd[d]);
  }
}

__device__ __forceinline__ float grad_inv(const Params& p, int64_t i, int d) {
  if (!p.cnt || d >= p.D) return d < p.D ? 1.f : 0.f;
  const int c = p.cnt[i * p.cnt_stride + d];
  return 1.f / static_cast<float>(c > 1 ? c : 1);
}

template <int VEC, int LPR, bool BWD = false>
__global__ __launch_bounds__(256) void spmm_lut_grad_kernel(const Params p, const GradParams gp) {
  constexpr int G = kWave / LPR;
  constexpr int TILE = LPR * VEC;
  constexpr int NW = 4;
  __shared__ float red[NW][4][TILE];            // slice blocks: waves -> wave 0
  __shared__ float rowsum[NW * G][4];           // row blocks: the groups' dwt for the workgroup partial
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int sub = lane % LPR;
  const int slot = lane / LPR;
  const int rest = p.D - 1;

  if (static_cast<int>(blockIdx.x) < p.n_slices) {
    // ---- hub-row slice: per-shell sums of this slice -> slice_T[s] ------------------------------
    const int s = blockIdx.x;
    int a = 0, b = p.n_long;
    while (b - a > 1) {
      const int mid = (a + b) >> 1;
      if (p.long_slice_ptr[mid] <= s) a = mid; else b = mid;
    }
    const int64_t q = p.long_rows[a];
    const int64_t i = adj_row(p, q);
    const int64_t row_lo = load_rowptr(p, i), row_hi = load_rowptr(p, i + 1);
    const int64_t lo = row_lo + static_cast<int64_t>(s - p.long_slice_ptr[a]) * p.slice_edges;
    const int64_t hi = lo + p.slice_edges < row_hi ? lo + p.slice_edges : row_hi;
    for (int w0 = 0; w0 < p.W; w0 += TILE) {
      const int cw = w0 + sub * VEC;
      const bool col_ok = cw < p.W;
      Vec<VEC> t[4];
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int v = 0; v < VEC; ++v) t[d].v[v] = 0.f;
      for (int64_t base = lo + static_cast<int64_t>(wave) * kWave; base < hi; base += NW * kWave) {
        const int64_t e = base + lane;
        int colv = 0, codev = 0;
        if (e < hi) { colv = p.col[e]; codev = p.code[e]; }
        const int m = static_cast<int>(hi - base < kWave ? hi - base : kWave);
#pragma unroll 4
        for (int tt = 0; tt < LPR; ++tt) {
          const int j = slot + tt * G;
          const int c = __shfl(colv, j);
          int d = __shfl(codev, j);
          d = d < rest ? d : rest;
          if (j < m && col_ok) {
            const Vec<VEC> sv = load_operand<VEC>(p.S, BWD ? static_cast<int64_t>(d) * p.n_cols + c : static_cast<int64_t>(c), p.s_stride, cw);
#pragma unroll
            for (int dd = 0; dd < 4; ++dd)
#pragma unroll
              for (int v = 0; v < VEC; ++v) t[dd].v[v] += d == dd ? sv.v[v] : 0.f;
          }
        }
      }
#pragma unroll
      for (int off = LPR; off < kWave; off <<= 1)
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int v = 0; v < VEC; ++v) t[d].v[v] += __shfl_xor(t[d].v[v], off);
      __syncthreads();
      if (slot == 0)
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int v = 0; v < VEC; ++v) red[wave][d][sub * VEC + v] = t[d].v[v];
      __syncthreads();
      if (wave == 0 && slot == 0 && col_ok) {
        float* out = gp.slice_T + static_cast<int64_t>(s) * 4 * p.W;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            float x = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) x += red[w][d][sub * VEC + v];
            out[d * p.W + cw + v] = x;
          }
      }
    }
    return;
  }

  // ---- row block: one LPR-lane group per output row ------------------------------------------------
  const int64_t block_id = static_cast<int64_t>(blockIdx.x) - p.n_slices;
  const int64_t q = (block_id * NW + wave) * G + slot;
  float pd[4] = {0.f, 0.f, 0.f, 0.f};
  int64_t i = 0, oq = 0;
  bool live = q < p.n_rows;
  if (live) {
    i = adj_row(p, q);
    oq = out_row(p, q, i);
    const int64_t lo = load_rowptr(p, i), hi = load_rowptr(p, i + 1);
    live = hi - lo <= p.long_threshold;                 // hub rows: slices + fix-up
    if (live) {
      for (int w0 = 0; w0 < p.W; w0 += TILE) {
        const int cw = w0 + sub * VEC;
        const bool col_ok = cw < p.W;
        Vec<VEC> t[4];
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int v = 0; v < VEC; ++v) t[d].v[v] = 0.f;
        // as in the forward: IW index pairs per round (IPL per lane), so narrow rows still see 16 gathers between
        // two dependent index loads
        constexpr int IW = LPR >= 8 ? LPR : 16;
        constexpr int IPL = IW / LPR;
        for (int64_t base = lo; base < hi; base += IW) {
          int colv[IPL], codev[IPL];
          bool wide = false;
          if constexpr (IPL % 4 == 0) {
            const int64_t e0 = base + sub * IPL;
            wide = e0 + IPL <= p.nnz;
            if (wide) load_index_run<IPL>(p.col + e0, p.code + e0, colv, codev);
          }
          if (!wide) {
#pragma unroll
            for (int r = 0; r < IPL; ++r) {
              const int64_t e = base + sub * IPL + r;
              colv[r] = codev[r] = 0;
              if (e < hi) { colv[r] = p.col[e]; codev[r] = p.code[e]; }
            }
          }
          const int m = static_cast<int>(hi - base < IW ? hi - base : IW);
#pragma unroll(IPL > 1 ? IW / 4 : 1)
          for (int j0 = 0; j0 < (IPL > 1 ? IW : m); j0 += 4) {
            if (IPL > 1 && j0 >= m) break;
            Vec<VEC> sv[4];
            int d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int j = j0 + u;
              const int c = __shfl(colv[j % IPL], j / IPL, LPR);
              d[u] = __shfl(codev[j % IPL], j / IPL, LPR);
              d[u] = d[u] < rest ? d[u] : rest;
#pragma unroll
              for (int v = 0; v < VEC; ++v) sv[u].v[v] = 0.f;
              if (j < m && col_ok)
                sv[u] = load_operand<VEC>(p.S, BWD ? static_cast<int64_t>(d[u]) * p.n_cols + c : static_cast<int64_t>(c), p.s_stride, cw);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (j0 + u < m)
#pragma unroll
                for (int dd = 0; dd < 4; ++dd)
#pragma unroll
                  for (int v = 0; v < VEC; ++v) t[dd].v[v] += d[u] == dd ? sv[u].v[v] : 0.f;
          }
        }
        if constexpr (BWD) bwd_finish<VEC, LPR>(p, gp, oq, cw, t, pd);       // one pass: 2 * half == LPR * VEC
        else grad_finish<VEC>(p, gp, i, oq, cw, col_ok, t, pd);
      }
#pragma unroll
      for (int off = 1; off < LPR; off <<= 1)
#pragma unroll
        for (int d = 0; d < 4; ++d) pd[d] += __shfl_xor(pd[d], off);
      if constexpr (!BWD) {
#pragma unroll
        for (int d = 0; d < 4; ++d) pd[d] *= grad_inv(p, i, d);
      }
      if (!gp.reduce_rows && sub == 0)
        for (int d = 0; d < p.D; ++d) gp.dwt[oq * p.D + d] = pd[d];
    }
  }
  if (gp.reduce_rows) {
    if (sub == 0)
#pragma unroll
      for (int d = 0; d < 4; ++d) rowsum[wave * G + slot][d] = live ? pd[d] : 0.f;
    __syncthreads();
    if (threadIdx.x < 4) {
      double acc = 0.0;
      for (int r = 0; r < NW * G; ++r) acc += rowsum[r][threadIdx.x];
      gp.blk[block_id * 4 + threadIdx.x] = acc;
    }
  }
}

// hub rows: add the slices in order, contract with dY, scale; one workgroup per hub row
template <bool BWD = false>
__global__ __launch_bounds__(256) void spmm_lut_grad_fixup_kernel(const Params p, const GradParams gp) {
  // one wave per hub row (four rows per workgroup, no barriers), as in spmm_long_fixup_kernel: lane = column, operands
  // narrower than a wave put K = 64 / W' lanes on a column, lane k takes slices k, k + K, ...
  const int lane = threadIdx.x & (kWave - 1);
  const int r = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  if (r >= p.n_long) return;
  const int64_t q = p.long_rows[r];
  const int64_t i = adj_row(p, q);
  const int64_t oq = out_row(p, q, i);
  const int s0 = p.long_slice_ptr[r], s1 = p.long_slice_ptr[r + 1];
  const int rest = p.D - 1;
  int wp = 1;
  while (wp < p.W && wp < kWave) wp <<= 1;
  const int K = kWave / wp;
  const int k = lane / wp;
  double pd[4] = {0.0, 0.0, 0.0, 0.0};
  for (int w0 = 0; w0 < p.W; w0 += kWave) {
    const int w = w0 + lane % wp;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    if (w < p.W) {
      int s = s0 + k;
      for (; s + K < s1; s += 2 * K) {     // two slices = eight independent loads at a time
        float a[4], b[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          a[d] = gp.slice_T[(static_cast<int64_t>(s) * 4 + d) * p.W + w];
          b[d] = gp.slice_T[(static_cast<int64_t>(s + K) * 4 + d) * p.W + w];
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) t[d] = (t[d] + a[d]) + b[d];
      }
      for (; s < s1; s += K)
#pragma unroll
        for (int d = 0; d < 4; ++d) t[d] += gp.slice_T[(static_cast<int64_t>(s) * 4 + d) * p.W + w];
    }
    for (int off = wp; off < kWave; off <<= 1)   // slice lanes of a column: fixed butterfly, every lane ends with the sum
#pragma unroll
      for (int d = 0; d < 4; ++d) t[d] += __shfl_xor(t[d], off);
    if constexpr (BWD) {
      // one pass (2 * half <= 64 columns): the lane `half` columns further holds this column's Q
      const float all = t[0] + t[1] + t[2] + t[3];
      const float qv = __shfl_xor(all, gp.half);
      if (k == 0 && w < gp.w_real) {
        const float sj = gp.s_rows[oq * gp.s_rows_stride + w];
        float ds = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d)
          if (d < rest || (!gp.with_rest && d == rest)) {
            ds = fmaf(p.lut[d], t[d], ds);
            pd[d] += static_cast<double>(sj) * t[d];
          }
        if (gp.with_rest) {
          ds = fmaf(-p.lut[rest], qv, ds);
          pd[rest & 3] -= static_cast<double>(sj) * qv;
        }
        if (gp.ds_add) ds += gp.ds_scale ? __fmul_rn(*gp.ds_scale, gp.ds_add[w]) : gp.ds_add[w];
        gp.dS[oq * gp.ds_stride + w] = ds;
      }
    } else if (k == 0 && w < p.W) {
      if (p.s_total) {
        float lower = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) lower += d < rest ? t[d] : 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) t[d] = d == rest ? p.s_total[w] - lower : t[d];
      }
      const float dy = gp.dY[oq * gp.dy_stride + w % gp.dy_channels];
#pragma unroll
      for (int d = 0; d < 4; ++d) pd[d] += static_cast<double>(dy) * t[d];
    }
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1)
#pragma unroll
    for (int d = 0; d < 4; ++d) pd[d] += __shfl_xor(pd[d], off);
  if (lane < 4) {
    const int d = lane;
    double mine = pd[0];
    mine = d == 1 ? pd[1] : mine;
    mine = d == 2 ? pd[2] : mine;
    mine = d == 3 ? pd[3] : mine;
    const double v = BWD ? mine : mine * grad_inv(p, i, d);
    if (gp.reduce_rows) gp.blk[(gp.n_row_blocks + r) * 4 + d] = v;
    else if (d < p.D) gp.dwt[oq * p.D + d] = static_cast<float>(v);
  }
}

// dlut[d] = sum over the workgroup / hub-row partials, fixed order
// (+ <tot, q> over w floats on entry D - 1 when tot is given: the rest bucket's column-sum term, gnan_spmm_bwd_narrow)
__global__ __launch_bounds__(1024) void spmm_lut_grad_final_kernel(const double* __restrict__ blk, int64_t n, int D,
                                                                   float* __restrict__ out, const float* __restrict__ tot = nullptr,
                                                                   const float* __restrict__ q = nullptr, int w = 0) {
  // one 1024-thread workgroup: a thread adds whole [4] records (32 contiguous bytes), eight loads in flight; the partials
  // meet in a fixed tree.  (Four 256-thread workgroups walking 72k records of a 10M-node graph one by one took 95 us.)
  __shared__ double red[4][1024];
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  const double2* rec = reinterpret_cast<const double2*>(blk);
  int64_t b = threadIdx.x;
  for (; b + 7 * 1024 < n; b += 8 * 1024) {
    double2 lo[8], hi[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      lo[u] = rec[(b + u * 1024) * 2];
      hi[u] = rec[(b + u * 1024) * 2 + 1];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s[0] += lo[u].x; s[1] += lo[u].y; s[2] += hi[u].x; s[3] += hi[u].y;
    }
  }
  for (; b < n; b += 1024) {
    const double2 lo = rec[b * 2], hi = rec[b * 2 + 1];
    s[0] += lo.x; s[1] += lo.y; s[2] += hi.x; s[3] += hi.y;
  }
#pragma unroll
  for (int d = 0; d < 4; ++d) red[d][threadIdx.x] = s[d];
  __syncthreads();
  for (int st = 512; st > 0; st >>= 1) {
    if (static_cast<int>(threadIdx.x) < st)
#pragma unroll
      for (int d = 0; d < 4; ++d) red[d][threadIdx.x] += red[d][threadIdx.x + st];
    __syncthreads();
  }
  if (static_cast<int>(threadIdx.x) < D && threadIdx.x < 4) {
    double v = red[threadIdx.x][0];
    if (tot && static_cast<int>(threadIdx.x) == D - 1)
      for (int c = 0; c < w; ++c) v = fma(static_cast<double>(tot[c]), static_cast<double>(q[c]), v);
    out[threadIdx.x] = static_cast<float>(v);
  }
}

template <int VEC, int LPR, bool BWD = false>
int launch_lut_grad(const Params& p, GradParams gp, hipStream_t st, float* dlut) {
  constexpr int G = kWave / LPR;
  const int64_t row_blocks = (p.n_rows + 4 * G - 1) / (4 * G);
  gp.n_row_blocks = row_blocks;
  const int64_t blocks = row_blocks + p.n_slices;
  if (blocks > 0x7fffffffLL) return gnan::fail(GNAN_ERR_UNSUPPORTED, "lut_grad: too many rows for one launch");
  hipLaunchKernelGGL((spmm_lut_grad_kernel<VEC, LPR, BWD>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, p, gp);
  if (int rc = gnan::check_launch("spmm_lut_grad_kernel")) return rc;
  if (p.n_slices > 0) {
    hipLaunchKernelGGL(spmm_lut_grad_fixup_kernel<BWD>, dim3(static_cast<unsigned>((p.n_long + 3) / 4)), dim3(256), 0, st, p, gp);
    if (int rc = gnan::check_launch("spmm_lut_grad_fixup_kernel")) return rc;
  }
  if (gp.reduce_rows) {
    hipLaunchKernelGGL(spmm_lut_grad_final_kernel, dim3(1), dim3(1024), 0, st, gp.blk, row_blocks + p.n_long, p.D, dlut, gp.rest_total,
                       gp.rest_q, gp.w_real);
    return gnan::check_launch("spmm_lut_grad_final_kernel");
  }
  return GNAN_OK;
}

// ---------------------------------------------------------------------------------------------
// spmm_bwd_hot_kernel — spmm_lut_grad_kernel<2, 1, true> (one-channel operands: packed rows of 2 floats, one lane per row)
// the way spmm_hot_kernel runs the forward: the degree-sorted copy of the TRANSPOSED adjacency read as one packed index
// stream, persistent 1024-thread workgroups (two per CU), and the packed rows of the most listed nodes — [hot_lo,
// hot_lo + hot_n) of the code blocks [hot_code_lo, hot_code_lo + hot_codes) of V — served from a 64-KB LDS copy.
// Ordinary rows: the arithmetic of spmm_lut_grad_kernel pair by pair (dS bit-identical); the table gradient's partials
// are per WAVE — 64 rows at a time through a fixed float64 butterfly, added up over the wave's blocks: one record per
// wave of the grid, so the order is fixed for a given device (the grid is two workgroups per CU); hub-row slices are summed
// by one wave each (fixed butterfly) and finished by spmm_lut_grad_fixup_kernel<true>.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ Vec<2> bwd_hot_gather(const Params& p, const GradParams& gp, const float2* hot, int c, int d) {
  // branch-free, as hot_gather: both loads are issued, the hot lanes of the global load share one row
  const int r = c - static_cast<int>(p.hot_lo);
  const int dl = d - gp.hot_code_lo;
  const bool is_hot = static_cast<unsigned>(r) < static_cast<unsigned>(p.hot_n) &&
                      static_cast<unsigned>(dl) < static_cast<unsigned>(gp.hot_codes);
  const int64_t row = is_hot ? static_cast<int64_t>(gp.hot_code_lo) * p.n_cols + p.hot_lo
                             : static_cast<int64_t>(d) * p.n_cols + c;
  const float2 g = *reinterpret_cast<const float2*>(static_cast<const float*>(p.S) + row * 2);
  const float2 l = hot[is_hot ? dl * p.hot_n + r : 0];
  Vec<2> out;
  out.v[0] = is_hot ? l.x : g.x;
  out.v[1] = is_hot ? l.y : g.y;
  return out;
}

__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))
void spmm_bwd_hot_kernel(const Params p, const GradParams gp) {
  extern __shared__ __attribute__((aligned(16))) float2 hot2[];      // [hot_codes][hot_n]
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int sub = wave >> 2, w4 = wave & 3;
  for (int i = tid; i < gp.hot_codes * p.hot_n; i += 1024) {
    const int dl = i / p.hot_n, r = i - dl * p.hot_n;
    hot2[i] = *reinterpret_cast<const float2*>(static_cast<const float*>(p.S) +
                                               (static_cast<int64_t>(gp.hot_code_lo + dl) * p.n_cols + p.hot_lo + r) * 2);
  }
  __syncthreads();
  const int rest = p.D - 1;
  const int idle_c = static_cast<int>(p.hot_lo), idle_d = gp.hot_code_lo;   // what a pair past the end reads: no request

  // ---- hub-row slices: one wave per slice, a contiguous run of slices per wave (see spmm_hot_kernel) --------------------
  {
    const int n_waves = static_cast<int>(gridDim.x) * 16;
    const int per = (p.n_slices + n_waves - 1) / n_waves;
    const int gw = static_cast<int>(blockIdx.x) * 16 + wave;
    const int s_lo = gw * per, s_hi = s_lo + per < p.n_slices ? s_lo + per : p.n_slices;
    int a = 0;
    if (s_lo < s_hi) {
      int b = p.n_long;
      while (b - a > 1) {
        const int mid = (a + b) >> 1;
        if (p.long_slice_ptr[mid] <= s_lo) a = mid; else b = mid;
      }
    }
    for (int sidx = s_lo; sidx < s_hi; ++sidx) {
      while (p.long_slice_ptr[a + 1] <= sidx) ++a;
      const int64_t i = adj_row(p, p.long_rows[a]);
      const int64_t row_lo = load_rowptr(p, i), row_hi = load_rowptr(p, i + 1);
      const int64_t lo = row_lo + static_cast<int64_t>(sidx - p.long_slice_ptr[a]) * p.slice_edges;
      const int64_t hi = lo + p.slice_edges < row_hi ? lo + p.slice_edges : row_hi;
      Vec<2> t[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) t[d].v[0] = t[d].v[1] = 0.f;
      constexpr int SF = 4;                              // pairs in flight per lane
      for (int64_t base = lo + lane; base < hi; base += SF * kWave) {
        unsigned ce[SF];
#pragma unroll
        for (int k = 0; k < SF; ++k) {
          const int64_t e = base + static_cast<int64_t>(k) * kWave;
          ce[k] = e < hi ? static_cast<unsigned>(p.col[e]) : 0u;
        }
        Vec<2> sv[SF];
        int dk[SF];
#pragma unroll
        for (int k = 0; k < SF; ++k) {
          const bool ok = base + static_cast<int64_t>(k) * kWave < hi;
          int d = static_cast<int>(ce[k] >> kPackShift);
          d = d < rest ? d : rest;
          dk[k] = ok ? d : -1;
          sv[k] = bwd_hot_gather(p, gp, hot2, ok ? static_cast<int>(ce[k] & kPackMask) : idle_c, ok ? d : idle_d);
        }
#pragma unroll
        for (int k = 0; k < SF; ++k)
#pragma unroll
          for (int dd = 0; dd < 4; ++dd) {
            t[dd].v[0] += dk[k] == dd ? sv[k].v[0] : 0.f;
            t[dd].v[1] += dk[k] == dd ? sv[k].v[1] : 0.f;
          }
      }
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          t[d].v[0] += __shfl_xor(t[d].v[0], off);
          t[d].v[1] += __shfl_xor(t[d].v[1], off);
        }
      if (lane < 8) {                                    // slice_T[s][d][w], W == 2: lane = 2 d + w
        float x = t[0].v[0];
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int v = 0; v < 2; ++v) x = lane == 2 * d + v ? t[d].v[v] : x;
        gp.slice_T[static_cast<int64_t>(sidx) * 8 + lane] = x;
      }
    }
  }

  // ---- ordinary rows: one lane per row, virtual 256-row blocks as in spmm_hot_kernel --------------------------------------
  const int64_t n_vblocks = (p.n_rows + 255) / 256;
  double run = 0.0;                                      // lane d < 4: this wave's share of dlut[d], 64 rows at a time
  for (int64_t vb = static_cast<int64_t>(blockIdx.x) * 4 + sub; vb < n_vblocks; vb += static_cast<int64_t>(gridDim.x) * 4) {
    const int64_t q = (vb * 4 + w4) * kWave + lane;
    float pd[4] = {0.f, 0.f, 0.f, 0.f};
    bool live = q < p.n_rows;
    int64_t lo = 0, hi = 0, oq = 0;
    if (live) {
      const int64_t i = adj_row(p, q);
      oq = out_row(p, q, i);
      lo = load_rowptr(p, i);
      hi = load_rowptr(p, i + 1);
      live = hi - lo <= p.long_threshold;               // hub row: sliced above, finished by the fix-up kernel
    }
    if (live) {
      Vec<2> t[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) t[d].v[0] = t[d].v[1] = 0.f;
      constexpr int RUN = 8;                             // index entries per round (16 as in the forward: spills at 64 VGPRs)
      for (int64_t base = lo; base < hi; base += RUN) {
        int colv[RUN];
        if (base + RUN <= p.nnz) {
          load_col_run<RUN>(p.col + base, colv);
        } else {
#pragma unroll
          for (int r = 0; r < RUN; ++r) colv[r] = base + r < hi ? p.col[base + r] : 0;
        }
        const int m = static_cast<int>(hi - base < RUN ? hi - base : RUN);
        constexpr int FLY = 4;
#pragma unroll
        for (int j0 = 0; j0 < RUN; j0 += FLY) {
          if (j0 >= m) break;
          Vec<2> sv[FLY];
          int d[FLY];
#pragma unroll
          for (int u = 0; u < FLY; ++u) {
            const unsigned ce = static_cast<unsigned>(colv[j0 + u]);
            const bool ok = j0 + u < m;
            int dd = static_cast<int>(ce >> kPackShift);
            dd = dd < rest ? dd : rest;
            d[u] = ok ? dd : -1;
            sv[u] = bwd_hot_gather(p, gp, hot2, ok ? static_cast<int>(ce & kPackMask) : idle_c, ok ? dd : idle_d);
          }
#pragma unroll
          for (int u = 0; u < FLY; ++u)
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
              t[dd].v[0] += d[u] == dd ? sv[u].v[0] : 0.f;
              t[dd].v[1] += d[u] == dd ? sv[u].v[1] : 0.f;
            }
        }
      }
      bwd_finish<2, 1>(p, gp, oq, 0, t, pd);
    }
    // the table gradient's partial of these 64 rows: float64, fixed butterfly
    double s[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) s[d] = live ? static_cast<double>(pd[d]) : 0.0;
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1)
#pragma unroll
      for (int d = 0; d < 4; ++d) s[d] += __shfl_xor(s[d], off);
    double mine = s[0];
    mine = lane == 1 ? s[1] : mine;
    mine = lane == 2 ? s[2] : mine;
    mine = lane == 3 ? s[3] : mine;
    run += mine;
  }
  if (lane < 4) gp.blk[(static_cast<int64_t>(blockIdx.x) * 16 + wave) * 4 + lane] = run;
}

// does gnan_spmm_bwd_narrow run on spmm_bwd_hot_kernel?  (packed index stream: only that kernel reads it)
bool bwd_hot_applies(const gnan_spmm_args* a) { return a->packed_index && a->W == 2; }

int bwd_hot_grid() {                              // two workgroups per CU
  static int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  return cus * 2;
}

size_t bwd_hot_blk_entries(const gnan_spmm_args* a) {    // one record per wave of the grid + one per hub row
  return static_cast<size_t>(bwd_hot_grid()) * 16 + static_cast<size_t>(a->n_long > 0 ? a->n_long : 0);
}

int launch_bwd_hot(const Params& p, GradParams gp, hipStream_t st, float* dlut) {
  const int grid = bwd_hot_grid();
  size_t lds = static_cast<size_t>(gp.hot_codes) * p.hot_n * 2 * sizeof(float);
  lds = lds < 16 ? 16 : lds;                      // the idle LDS read of a launch without hot rows
  gp.n_row_blocks = static_cast<int64_t>(grid) * 16;
  hipLaunchKernelGGL(spmm_bwd_hot_kernel, dim3(static_cast<unsigned>(grid)), dim3(1024), lds, st, p, gp);
  if (int rc = gnan::check_launch("spmm_bwd_hot_kernel")) return rc;
  if (p.n_slices > 0) {
    hipLaunchKernelGGL(spmm_lut_grad_fixup_kernel<true>, dim3(static_cast<unsigned>((p.n_long + 3) / 4)), dim3(256), 0, st, p, gp);
    if (int rc = gnan::check_launch("spmm_lut_grad_fixup_kernel")) return rc;
  }
  hipLaunchKernelGGL(spmm_lut_grad_final_kernel, dim3(1), dim3(1024), 0, st, gp.blk, gp.n_row_blocks + p.n_long, p.D, dlut,
                     gp.rest_total, gp.rest_q, gp.w_real);
  return gnan::check_launch("spmm_lut_grad_final_kernel");
}

// ---------------------------------------------------------------------------------------------
// Table gradient on the DENSE layout (every pair listed, up to 256 hop codes; Cw == 1, global table):
//   dlut[d] = sum_q inv(q, d) * sum_{j : code(q, j) == d} < dY[q, :], S[j, :] >
// — what gnan_spmm_shell_sums + six framework launches computed through a [n, D, W] tensor of read-modify-writes in
// global memory (Cora-shaped: 0.59 ms of a 2.7-ms training step; a 30-node graph: 8 of its 34 launches).  One wave per
// row: lane l takes neighbours l, l + 64, ..., forms the W-term dot product and adds it to ITS column of the wave's
// [D][64] LDS bins (no atomics, no conflicts by construction); lane d then adds bin row d front to back, scales by
// 1 / count and keeps a float64 running sum over the wave's rows.  One record per wave, a fixed-order final pass:
// bit-reproducible.
// ---------------------------------------------------------------------------------------------
constexpr int kBinStride = kWave + 1;        // bin rows one bank apart: lane d's walk along row d does not collide with lane d + 1's

__global__ __launch_bounds__(256) void dense_lut_grad_kernel(const Params p, const GradParams gp, int waves_total) {
  extern __shared__ __attribute__((aligned(16))) float dense_bins[];    // [waves per block][D][65]
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int wpb = blockDim.x / kWave;
  float* mine = dense_bins + static_cast<size_t>(wave) * p.D * kBinStride;
  const int rest = p.D - 1;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};      // codes lane, lane + 64, lane + 128, lane + 192
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * wpb; base < p.n_rows; base += waves_total) {
    const int64_t q = base + wave;
    const bool live = q < p.n_rows;
    const int64_t i = live ? adj_row(p, q) : 0;
    for (int d = 0; d < p.D; ++d) mine[d * kBinStride + lane] = 0.f;
    if (live) {
      const float* dy = gp.dY + q * gp.dy_stride;
      const uint8_t* codes = p.code + i * p.n_cols;
      for (int64_t j = lane; j < p.n_cols; j += kWave) {
        int d = codes[j];
        d = d < rest ? d : rest;
        const float* srow = static_cast<const float*>(p.S) + j * p.s_stride;
        float dot = 0.f;
        for (int w = 0; w < p.W; ++w) dot = fmaf(dy[w % gp.dy_channels], srow[w], dot);
        mine[d * kBinStride + lane] += dot;
      }
    }
    __syncthreads();                           // (uniform trip count: every wave of the block sees the same `base`)
    if (live) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int d = lane + k * kWave;
        if (d < p.D) {
          float s = 0.f;
          for (int l = 0; l < kWave; ++l) s += mine[d * kBinStride + l];
          if (p.cnt) {
            const int c = p.cnt[i * p.cnt_stride + d];
            s *= 1.f / static_cast<float>(c > 1 ? c : 1);
          }
          acc[k] += static_cast<double>(s);
        }
      }
    }
    __syncthreads();
  }
  const int64_t gw = static_cast<int64_t>(blockIdx.x) * wpb + wave;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = lane + k * kWave;
    if (d < p.D && gw < waves_total) gp.blk[gw * p.D + d] = acc[k];
  }
}

// one workgroup per hop code: 256 threads stride over the waves' records, then a fixed tree
__global__ __launch_bounds__(256) void dense_lut_grad_final_kernel(const double* __restrict__ blk, int waves_total, int D,
                                                                   float* __restrict__ out) {
  __shared__ double red[256];
  const int d = blockIdx.x;
  double s = 0.0;
  for (int w = threadIdx.x; w < waves_total; w += 256) s += blk[static_cast<int64_t>(w) * D + d];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (static_cast<int>(threadIdx.x) < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[d] = static_cast<float>(red[0]);
}

bool dense_lut_grad_applies(const gnan_spmm_args* a, int32_t reduce_rows) {
  return a->rowptr == nullptr && a->Cw == 1 && a->D <= 256 && reduce_rows && a->s_dtype == GNAN_F32 && !a->weight_by_col &&
         a->s_total == nullptr && a->lut_row_stride == 0 && !a->s_by_code;
}

int dense_lut_grad_waves(const gnan_spmm_args* a) {      // one wave per row up to 2048 waves
  return static_cast<int>(a->n_rows < 2048 ? (a->n_rows < 1 ? 1 : a->n_rows) : 2048);
}

int launch_dense_lut_grad(const Params& p, GradParams gp, const gnan_spmm_args* a, hipStream_t st, float* dlut) {
  const size_t per_wave = static_cast<size_t>(p.D) * kBinStride * sizeof(float);
  int wpb = static_cast<int>((64 * 1024) / per_wave);
  wpb = wpb < 1 ? 1 : (wpb > 4 ? 4 : wpb);
  int waves = dense_lut_grad_waves(a);
  waves = (waves + wpb - 1) / wpb * wpb;                    // whole workgroups
  const size_t lds = per_wave * wpb;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dense_lut_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "lut_grad: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(dense_lut_grad_kernel, dim3(static_cast<unsigned>(waves / wpb)), dim3(wpb * kWave), lds, st, p, gp, waves);
  if (int rc = gnan::check_launch("dense_lut_grad_kernel")) return rc;
  hipLaunchKernelGGL(dense_lut_grad_final_kernel, dim3(static_cast<unsigned>(p.D)), dim3(256), 0, st, gp.blk, waves, p.D, dlut);
  return gnan::check_launch("dense_lut_grad_final_kernel");
}

__global__ void zero_floats_kernel(float* out, int n) {     // (a kernel: captured memsets replay wrongly on ROCm 7.2)
  for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = 0.f;
}

template <int VEC>
int launch_lut_grad_lpr(const Params& p, const GradParams& gp, int lpr, hipStream_t st, float* dlut) {
  switch (lpr) {
    case 1: return launch_lut_grad<VEC, 1>(p, gp, st, dlut);
    case 2: return launch_lut_grad<VEC, 2>(p, gp, st, dlut);
    case 4: return launch_lut_grad<VEC, 4>(p, gp, st, dlut);
    case 8: return launch_lut_grad<VEC, 8>(p, gp, st, dlut);
    case 16: return launch_lut_grad<VEC, 16>(p, gp, st, dlut);
    case 32: return launch_lut_grad<VEC, 32>(p, gp, st, dlut);
    default: return launch_lut_grad<VEC, 64>(p, gp, st, dlut);
  }
}

size_t lut_grad_blk_entries(const gnan_spmm_args* a, int vec, int lpr) {
  const int G = kWave / lpr;
  return static_cast<size_t>((a->n_rows + 4 * G - 1) / (4 * G)) + static_cast<size_t>(a->n_long > 0 ? a->n_long : 0);
}

}  // namespace

static size_t lut_grad_workspace_bytes(const gnan_spmm_args* a, int32_t reduce_rows) {
  if (!a || a->n_rows <= 0) return 0;
  if (dense_lut_grad_applies(a, reduce_rows))              // one [D] float64 record per wave (rounded up to whole workgroups)
    return (static_cast<size_t>(dense_lut_grad_waves(a)) + 4) * static_cast<size_t>(a->D) * sizeof(double);
  int vec, lpr;
  pick_tiling(a, static_cast<const float*>(a->S), a->s_stride, &vec, &lpr);
  size_t bytes = a->n_long > 0 ? static_cast<size_t>(a->n_slices) * 4 * static_cast<size_t>(a->W) * sizeof(float) : 0;
  bytes = (bytes + 15) / 16 * 16;
  if (reduce_rows) bytes += lut_grad_blk_entries(a, vec, lpr) * 4 * sizeof(double);
  return bytes;
}

extern "C" size_t gnan_spmm_lut_grad_workspace_bytes(const gnan_spmm_lut_grad_args* g) {
  return g ? lut_grad_workspace_bytes(&g->spmm, g->reduce_rows) : 0;
}

extern "C" int gnan_spmm_lut_grad(const gnan_spmm_lut_grad_args* g, gnan_stream_t stream) {
  GNAN_REQUIRE(g != nullptr, "lut_grad: null args");
  const gnan_spmm_args* a = &g->spmm;
  const float* dY = g->dY;
  const int64_t dy_stride = g->dy_stride;
  const int32_t dy_channels = g->dy_channels, reduce_rows = g->reduce_rows;
  float* dwt = g->dwt;
  void* workspace = g->workspace;
  const size_t workspace_bytes = g->workspace_bytes;
  if (int rc = validate(a)) return rc;
  GNAN_REQUIRE(!a->packed_index, "lut_grad: packed index entries are read by gnan_spmm_fwd only");
  GNAN_REQUIRE(dwt != nullptr, "lut_grad: null output");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a->n_rows == 0) {
    if (reduce_rows) {
      hipLaunchKernelGGL(zero_floats_kernel, dim3(1), dim3(64), 0, st, dwt, a->D);
      return gnan::check_launch("zero_floats_kernel");
    }
    return GNAN_OK;
  }
  GNAN_REQUIRE(dY != nullptr && dy_channels >= 1 && dy_stride >= dy_channels, "lut_grad: bad dY");
  if (dense_lut_grad_applies(a, reduce_rows)) {
    GNAN_REQUIRE(a->W % dy_channels == 0, "lut_grad: dy_channels must be a divisor of W");
    const size_t need = lut_grad_workspace_bytes(a, reduce_rows);
    if (workspace == nullptr || workspace_bytes < need)
      return gnan::fail(GNAN_ERR_WORKSPACE, "lut_grad: workspace %zu B < required %zu B", workspace_bytes, need);
    const Params p = make_params(a);
    GradParams gp{};
    gp.dY = dY; gp.dy_stride = dy_stride; gp.dy_channels = dy_channels; gp.dwt = dwt; gp.reduce_rows = 1;
    gp.blk = static_cast<double*>(workspace);
    return launch_dense_lut_grad(p, gp, a, st, dwt);
  }
  if (a->rowptr == nullptr || a->D > 4 || a->Cw != 1 || a->s_dtype != GNAN_F32 || a->weight_by_col)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "lut_grad: needs the CSR layout with D <= 4 — or the dense layout with a global table, "
                      "reduce_rows and no rest-bucket total —, one weight channel and fp32 operand rows");
  GNAN_REQUIRE(a->W % dy_channels == 0, "lut_grad: dy_channels must divide W");
  const size_t need = lut_grad_workspace_bytes(a, reduce_rows);
  if (need > 0 && (workspace == nullptr || workspace_bytes < need))
    return gnan::fail(GNAN_ERR_WORKSPACE, "lut_grad: workspace %zu B < required %zu B", workspace_bytes, need);
  const Params p = make_params(a);
  int vec, lpr;
  pick_tiling(a, static_cast<const float*>(a->S), a->s_stride, &vec, &lpr);
  GradParams gp;
  gp.dY = dY; gp.dy_stride = dy_stride; gp.dy_channels = dy_channels; gp.dwt = dwt; gp.reduce_rows = reduce_rows;
  gp.ds_add = nullptr; gp.ds_scale = nullptr; gp.rest_total = nullptr; gp.rest_q = nullptr; gp.w_real = 0;
  gp.slice_T = static_cast<float*>(workspace);
  size_t off = a->n_long > 0 ? static_cast<size_t>(a->n_slices) * 4 * static_cast<size_t>(a->W) * sizeof(float) : 0;
  off = (off + 15) / 16 * 16;   // the final reduction reads 16-byte halves of the [4] records
  gp.blk = reinterpret_cast<double*>(static_cast<char*>(workspace) + off);
  gp.n_row_blocks = 0;
  return vec == 4 ? launch_lut_grad_lpr<4>(p, gp, lpr, st, dwt) : launch_lut_grad_lpr<1>(p, gp, lpr, st, dwt);
}

namespace {
// partial[blockIdx.x] = the sum of `v` over the 256 threads of the workgroup (fixed tree, float64) — EVERY thread calls it.
__device__ __forceinline__ void block_sum_to(double v, double* __restrict__ partial) {
  __shared__ double red[256];
  red[threadIdx.x] = v;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (static_cast<int>(threadIdx.x) < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// ... and, with an arrival counter, the last workgroup of the pass adds the partials (pack_q_final_kernel's sum, same order)
__device__ __forceinline__ void block_sum_finish(double v, double* partial, unsigned* arrive, float* out) {
  block_sum_to(v, partial);
  if (arrive != nullptr && gnan::last_block(arrive)) {
    const double s_all = gnan::sum_partials_256(partial, static_cast<int64_t>(gridDim.x));
    if (threadIdx.x == 0) out[0] = static_cast<float>(s_all);
  }
}

// q_sum[0] = sum of the workgroups' partials (one workgroup, fixed order): sum_i dY_i / cnt(i, rest), what gnan_colsum over the
// packed rows' second halves returned — two launches and a strided 40-MB read on the 10M-node graph
__global__ __launch_bounds__(256) void pack_q_final_kernel(const double* __restrict__ partial, int64_t n_partial, float* __restrict__ q_sum) {
  double s = 0.0;
  for (int64_t b = threadIdx.x; b < n_partial; b += 256) s += partial[b];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (static_cast<int>(threadIdx.x) < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) q_sum[0] = static_cast<float>(red[0]);
}

// V[d * n + i, :] = [ dY_i / cnt(i, d) | dY_i / cnt(i, D-1) ]  (the packed operand of gnan_spmm_bwd_narrow), zero padded.
// Thread = node: its gradient row and counts are read once, its D packed rows are one contiguous run of the output.
__global__ __launch_bounds__(256) void pack_bwd_rows_kernel(const float* __restrict__ dY, int64_t dy_stride, int W,
                                                            const int32_t* __restrict__ cnt, int64_t cnt_stride, int D,
                                                            int64_t n, int with_rest, float* __restrict__ V, int half,
                                                            const int64_t* __restrict__ hot, int64_t n_hot, int64_t o_begin,
                                                            double* q_partial, unsigned* q_arrive, float* q_sum) {
  double qs = 0.0;                                    // q_partial (W == 1): this thread's sum of dY_i / cnt(i, rest) over REAL nodes
  for (int64_t o = o_begin + static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; o < n + n_hot; o += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t i = o < n ? o : hot[o - n];         // packed rows [n, n + n_hot): second copies of the nodes hot[]
    float r_rest = 1.f;
    if (cnt) {
      const int k = cnt[i * cnt_stride + D - 1];
      r_rest = static_cast<float>(k > 1 ? k : 1);
    }
    // code-major: V[d * (n + n_hot) + o, :] — the rows of ONE hop code are contiguous, so the lines a pass over the code-1
    // pairs fetches hold sixteen useful rows each (node-major (o, d) rows: a third of every line was the never-gathered
    // rest code and the once-per-node self code) and the hot block of a code is 2 MB instead of 6
    for (int d = 0; d < D; ++d) {
      float* out = V + (static_cast<int64_t>(d) * (n + n_hot) + o) * 2 * half;
      float r = 1.f;
      if (cnt) {
        const int k = cnt[i * cnt_stride + d];
        r = static_cast<float>(k > 1 ? k : 1);
      }
      for (int w = 0; w < half; ++w) {
        const float g = w < W ? dY[i * dy_stride + w] : 0.f;
        out[w] = g / r;
        out[half + w] = with_rest ? g / r_rest : 0.f;
      }
    }
    if (q_partial && with_rest && o < n) qs += static_cast<double>(dY[i * dy_stride] / r_rest);
  }
  if (q_partial) block_sum_finish(qs, q_partial, q_arrive, q_sum);
}

// One-channel gradients (half == 1: packed rows of two floats) with shell counts and at most four codes — the shape of every
// sum-first training step: thread = TWO consecutive nodes, so that a node pair's counts are three 8-byte loads, its gradients
// one, and its packed rows of a code ONE 16-byte store (the one-node form above moves the 10M-node graph's 400 MB at 2.9 TB/s:
// 4- and 8-byte accesses).  Covers the nodes [0, n_pairs * 2); the tail and the hot copies go through the kernel above.
template <int D>
__global__ __launch_bounds__(256) void pack_bwd_pairs_kernel(const float* __restrict__ dY, const int32_t* __restrict__ cnt,
                                                             int64_t n, int64_t n_pairs, int with_rest, float* __restrict__ V,
                                                             const int64_t* __restrict__ hot, int64_t n_hot, int pair_blocks,
                                                             double* q_partial, unsigned* q_arrive, float* q_sum) {
  const int64_t rows_per_code = n + n_hot;
  double qs = 0.0;
  if (static_cast<int>(blockIdx.x) >= pair_blocks) {
    // the odd last node and the second copies of the nodes hot[] — in the SAME launch, next to the streaming part (a launch of
    // their own: 35 us behind the pairs' 61 on the 10M-node graph)
    const int64_t o = 2 * n_pairs + (static_cast<int64_t>(blockIdx.x) - pair_blocks) * 256 + threadIdx.x;
    if (o < rows_per_code) {
      const int64_t i = o < n ? o : hot[o - n];
      const float g = dY[i];
      int k[D];
#pragma unroll
      for (int d = 0; d < D; ++d) k[d] = cnt[i * D + d];
      const float q = with_rest ? g / static_cast<float>(k[D - 1] > 1 ? k[D - 1] : 1) : 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d)
        *reinterpret_cast<float2*>(V + (static_cast<int64_t>(d) * rows_per_code + o) * 2) =
            make_float2(g / static_cast<float>(k[d] > 1 ? k[d] : 1), q);
      if (o < n) qs = static_cast<double>(q);
    }
    if (q_partial) block_sum_finish(qs, q_partial, q_arrive, q_sum);
    return;
  }
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; t < n_pairs; t += static_cast<int64_t>(pair_blocks) * 256) {
    const float2 g = *reinterpret_cast<const float2*>(dY + 2 * t);
    int k[2 * D];                                         // (the pair's 2 D counts start 8-byte aligned whatever D is)
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int2 c = *reinterpret_cast<const int2*>(cnt + 2 * D * t + 2 * u);
      k[2 * u] = c.x; k[2 * u + 1] = c.y;
    }
    float r[2][D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      r[0][d] = static_cast<float>(k[d] > 1 ? k[d] : 1);
      r[1][d] = static_cast<float>(k[D + d] > 1 ? k[D + d] : 1);
    }
    const float q0 = with_rest ? g.x / r[0][D - 1] : 0.f, q1 = with_rest ? g.y / r[1][D - 1] : 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d)
      *reinterpret_cast<float4*>(V + (static_cast<int64_t>(d) * rows_per_code + 2 * t) * 2) = make_float4(g.x / r[0][d], q0, g.y / r[1][d], q1);
    qs += static_cast<double>(q0) + static_cast<double>(q1);
  }
  if (q_partial) block_sum_finish(qs, q_partial, q_arrive, q_sum);
}
}  // namespace

// the launch gnan_spmm_pack_bwd_rows makes for these arguments: node pairs (pair_blocks > 0) or one node per thread
namespace {
struct PackGrid {
  bool pairs;
  int64_t n_pairs, pair_blocks, blocks;    // blocks: the whole grid
};
PackGrid pack_grid(const gnan_pack_bwd_rows_args* a) {
  PackGrid g{false, 0, 0, 0};
  const int64_t n = a->n, n_hot = a->n_hot;
  // node pairs (large graphs): needs the packed rows of every code to start 16-byte aligned ((n + n_hot) even) and dense inputs
  if (a->half == 1 && a->W == 1 && a->cnt != nullptr && a->cnt_stride == a->D && a->dy_stride == 1 && a->D >= 2 && a->D <= 4 &&
      n >= (int64_t(1) << 20) && (n + n_hot) % 2 == 0 && reinterpret_cast<uintptr_t>(a->dY) % 8 == 0 &&
      reinterpret_cast<uintptr_t>(a->cnt) % 8 == 0 && reinterpret_cast<uintptr_t>(a->V) % 16 == 0) {
    g.n_pairs = n / 2;
    g.pair_blocks = (g.n_pairs + 255) / 256;
    g.pair_blocks = g.pair_blocks > 65536 ? 65536 : g.pair_blocks;
    const int64_t tb = (n + n_hot - 2 * g.n_pairs + 255) / 256;
    if (tb < (int64_t(1) << 20)) {
      g.pairs = true;
      g.blocks = g.pair_blocks + tb;
      return g;
    }
  }
  g.blocks = (n + n_hot + 255) / 256;
  g.blocks = g.blocks > 65536 ? 65536 : g.blocks;
  return g;
}
}  // namespace

extern "C" size_t gnan_spmm_pack_bwd_rows_workspace_bytes(const gnan_pack_bwd_rows_args* a) {
  if (!a || a->q_sum == nullptr || a->n <= 0) return 0;
  return static_cast<size_t>(pack_grid(a).blocks) * sizeof(double);
}

extern "C" int gnan_spmm_pack_bwd_rows(const gnan_pack_bwd_rows_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "pack_bwd_rows: null args");
  const float* dY = a->dY;
  const int64_t dy_stride = a->dy_stride, cnt_stride = a->cnt_stride, n = a->n, n_hot = a->n_hot;
  const int32_t W = a->W, D = a->D, with_rest = a->with_rest, half = a->half;
  const int32_t* cnt = a->cnt;
  float* V = a->V;
  const int64_t* hot = a->hot;
  GNAN_REQUIRE(n >= 0 && W >= 1 && D >= 1 && half >= W && (half & (half - 1)) == 0, "pack_bwd_rows: bad sizes");
  GNAN_REQUIRE((dY && V) || n == 0, "pack_bwd_rows: null pointer");
  GNAN_REQUIRE(dy_stride >= W && (cnt == nullptr || cnt_stride >= D), "pack_bwd_rows: row stride smaller than the width");
  GNAN_REQUIRE(n_hot >= 0 && (n_hot == 0 || hot != nullptr), "pack_bwd_rows: n_hot without hot");
  GNAN_REQUIRE(a->q_sum == nullptr || (W == 1 && with_rest), "pack_bwd_rows: q_sum is the one-channel rest-bucket sum (W == 1, with_rest)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    if (a->q_sum) {
      hipLaunchKernelGGL(pack_q_final_kernel, dim3(1), dim3(256), 0, st, static_cast<const double*>(nullptr), static_cast<int64_t>(0), a->q_sum);
      return gnan::check_launch("pack_q_final_kernel");
    }
    return GNAN_OK;
  }
  const PackGrid pg = pack_grid(a);
  double* q_partial = nullptr;
  if (a->q_sum) {
    const size_t need = static_cast<size_t>(pg.blocks) * sizeof(double);
    if (a->q_workspace == nullptr || a->q_workspace_bytes < need)
      return gnan::fail(GNAN_ERR_WORKSPACE, "pack_bwd_rows: q workspace %zu B < required %zu B", a->q_workspace_bytes, need);
    GNAN_REQUIRE(reinterpret_cast<uintptr_t>(a->q_workspace) % 8 == 0, "pack_bwd_rows: q workspace must be 8-byte aligned");
    q_partial = static_cast<double*>(a->q_workspace);
  }
  // q_sum by the last workgroup of the packing launch where the caller lends an arrival counter (a fence per workgroup: small grids)
  unsigned* q_arrive = (q_partial && pg.blocks <= gnan::kMaxArriveBlocks) ? reinterpret_cast<unsigned*>(a->q_arrive) : nullptr;
  if (pg.pairs) {
    const dim3 grid(static_cast<unsigned>(pg.blocks)), block(256);
    const int pbi = static_cast<int>(pg.pair_blocks);
    if (D == 2) hipLaunchKernelGGL(pack_bwd_pairs_kernel<2>, grid, block, 0, st, dY, cnt, n, pg.n_pairs, with_rest, V, hot, n_hot, pbi, q_partial, q_arrive, a->q_sum);
    else if (D == 3) hipLaunchKernelGGL(pack_bwd_pairs_kernel<3>, grid, block, 0, st, dY, cnt, n, pg.n_pairs, with_rest, V, hot, n_hot, pbi, q_partial, q_arrive, a->q_sum);
    else hipLaunchKernelGGL(pack_bwd_pairs_kernel<4>, grid, block, 0, st, dY, cnt, n, pg.n_pairs, with_rest, V, hot, n_hot, pbi, q_partial, q_arrive, a->q_sum);
    if (int rc = gnan::check_launch("pack_bwd_pairs_kernel")) return rc;
  } else {
    hipLaunchKernelGGL(pack_bwd_rows_kernel, dim3(static_cast<unsigned>(pg.blocks)), dim3(256), 0, st,
                       dY, dy_stride, W, cnt, cnt_stride, D, n, with_rest, V, half, hot, n_hot, static_cast<int64_t>(0), q_partial,
                       q_arrive, a->q_sum);
    if (int rc = gnan::check_launch("pack_bwd_rows_kernel")) return rc;
  }
  if (q_partial && q_arrive == nullptr) {
    hipLaunchKernelGGL(pack_q_final_kernel, dim3(1), dim3(256), 0, st, q_partial, pg.blocks, a->q_sum);
    return gnan::check_launch("pack_q_final_kernel");
  }
  return GNAN_OK;
}

static size_t bwd_narrow_workspace_bytes(const gnan_spmm_args* a) {
  if (!a || a->n_rows <= 0) return 0;
  const int half = a->W / 2;
  const int vec = half <= 2 ? 2 * (half < 1 ? 1 : half) : 4;
  const int lpr = a->W / vec >= 1 ? a->W / vec : 1;
  size_t bytes = a->n_long > 0 ? static_cast<size_t>(a->n_slices) * 4 * static_cast<size_t>(a->W) * sizeof(float) : 0;
  bytes = (bytes + 15) / 16 * 16;
  const size_t entries = bwd_hot_applies(a) ? bwd_hot_blk_entries(a) : lut_grad_blk_entries(a, vec, lpr);
  return bytes + entries * 4 * sizeof(double);
}

extern "C" size_t gnan_spmm_bwd_narrow_workspace_bytes(const gnan_spmm_bwd_narrow_args* g) {
  return g ? bwd_narrow_workspace_bytes(&g->spmm) : 0;
}

extern "C" int gnan_spmm_bwd_narrow(const gnan_spmm_bwd_narrow_args* g, gnan_stream_t stream) {
  GNAN_REQUIRE(g != nullptr, "bwd_narrow: null args");
  const gnan_spmm_args* a = &g->spmm;
  const float* s_rows = g->s_rows;
  const int64_t s_rows_stride = g->s_rows_stride, ds_stride = g->ds_stride;
  const int32_t w_real = g->w_real, with_rest = g->with_rest;
  float* dS = g->dS;
  float* dlut = g->dlut;
  void* workspace = g->workspace;
  const size_t workspace_bytes = g->workspace_bytes;
  if (int rc = validate(a)) return rc;
  GNAN_REQUIRE(!a->packed_index || a->W == 2, "bwd_narrow: packed index entries are read for one-channel operands only (W == 2)");
  GNAN_REQUIRE(dS != nullptr && dlut != nullptr && (s_rows != nullptr || a->n_rows == 0), "bwd_narrow: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a->rowptr == nullptr || a->D > 4 || a->Cw != 1 || a->s_dtype != GNAN_F32 || a->lut_row_stride != 0 || a->cnt != nullptr)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "bwd_narrow: needs the CSR layout, D <= 4, one global weight channel, fp32 rows, no cnt");
  const int half = a->W / 2;
  if (a->W < 2 || a->W > 64 || (a->W & (a->W - 1)) != 0 || w_real < 1 || w_real > half || a->s_stride != a->W)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "bwd_narrow: operand rows must be 2 * half floats, half a power of two in [w_real, 32] (got W=%d, w_real=%d)", a->W, w_real);
  GNAN_REQUIRE(s_rows_stride >= w_real && ds_stride >= w_real, "bwd_narrow: row stride smaller than the width");
  GNAN_REQUIRE((g->rest_total == nullptr) == (g->rest_q == nullptr), "bwd_narrow: rest_total and rest_q come together");
  GNAN_REQUIRE(g->rest_total == nullptr || with_rest, "bwd_narrow: rest_total without a rest bucket");
  if (a->n_rows == 0) {
    hipLaunchKernelGGL(zero_floats_kernel, dim3(1), dim3(64), 0, st, dlut, a->D);
    return gnan::check_launch("zero_floats_kernel");
  }
  const size_t need = bwd_narrow_workspace_bytes(a);
  if (need > 0 && (workspace == nullptr || workspace_bytes < need))
    return gnan::fail(GNAN_ERR_WORKSPACE, "bwd_narrow: workspace %zu B < required %zu B", workspace_bytes, need);
  const Params p = make_params(a);
  // the two halves of a row must sit in different lanes, partner = lane + LPR / 2: VEC = min(4, half), LPR = 2 * half / VEC
  if (reinterpret_cast<uintptr_t>(a->S) % 16 != 0)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "bwd_narrow: operand rows must be 16-byte aligned");
  GradParams gp;
  gp.dY = nullptr; gp.dy_stride = 0; gp.dy_channels = 1; gp.dwt = dlut; gp.reduce_rows = 1;
  gp.slice_T = static_cast<float*>(workspace);
  size_t off = a->n_long > 0 ? static_cast<size_t>(a->n_slices) * 4 * static_cast<size_t>(a->W) * sizeof(float) : 0;
  off = (off + 15) / 16 * 16;
  gp.blk = reinterpret_cast<double*>(static_cast<char*>(workspace) + off);
  gp.n_row_blocks = 0;
  gp.s_rows = s_rows; gp.s_rows_stride = s_rows_stride; gp.half = half; gp.w_real = w_real;
  gp.dS = dS; gp.ds_stride = ds_stride; gp.with_rest = with_rest; gp.ds_add = g->ds_add;
  gp.ds_scale = g->ds_add ? g->ds_add_scale : nullptr;
  gp.rest_total = g->rest_total; gp.rest_q = g->rest_q;
  gp.hot_code_lo = 0; gp.hot_codes = 0;
  if (bwd_hot_applies(a)) {
    // one-channel operands over a packed index stream: the persistent kernel, with the head of the appended hot rows in LDS
    Params ph = p;
    if (a->hot_rows > 0) {
      GNAN_REQUIRE(g->hot_codes >= 1 && g->hot_code_lo >= 0 && g->hot_code_lo + g->hot_codes <= a->D,
                   "bwd_narrow: hot code blocks outside [0, D)");
      GNAN_REQUIRE(a->hot_lo >= 0 && a->hot_lo + a->hot_rows <= a->n_cols, "bwd_narrow: hot rows outside the packed rows");
      GNAN_REQUIRE(static_cast<int64_t>(a->hot_rows) * g->hot_codes * 2 <= kHotLdsFloats, "bwd_narrow: hot rows exceed 64 KB of LDS");
      gp.hot_code_lo = g->hot_code_lo; gp.hot_codes = g->hot_codes;
    } else {
      ph.hot_lo = 0; ph.hot_n = 0;
    }
    return launch_bwd_hot(ph, gp, st, dlut);
  }
  switch (half) {      // one lane per row while a row is one 8- or 16-byte load (64 rows per wavefront instead of 32)
    case 1: return launch_lut_grad<2, 1, true>(p, gp, st, dlut);
    case 2: return launch_lut_grad<4, 1, true>(p, gp, st, dlut);
    case 4: return launch_lut_grad<4, 2, true>(p, gp, st, dlut);
    case 8: return launch_lut_grad<4, 4, true>(p, gp, st, dlut);
    case 16: return launch_lut_grad<4, 8, true>(p, gp, st, dlut);
    default: return launch_lut_grad<4, 16, true>(p, gp, st, dlut);
  }
}

extern "C" int gnan_spmm_shell_sums(const gnan_spmm_args* a, gnan_stream_t stream) {
  if (int rc = validate(a)) return rc;
  GNAN_REQUIRE(!a->packed_index, "shell_sums: packed index entries are read by gnan_spmm_fwd only");
  if (a->n_rows == 0) return GNAN_OK;
  GNAN_REQUIRE(!a->weight_by_col, "shell_sums: weight_by_col has no meaning here");
  if (a->s_dtype != GNAN_F32) return gnan::fail(GNAN_ERR_UNSUPPORTED, "shell_sums: fp32 operand rows only (no backward for bf16 storage)");
  const Params p = make_params(a);
  int vec, lpr;
  pick_tiling(a, a->Y, a->W, &vec, &lpr);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool dense = a->rowptr == nullptr;
  return vec == 4 ? launch_shell_lpr<4>(p, lpr, dense, st) : launch_shell_lpr<1>(p, lpr, dense, st);
}

extern "C" int gnan_spmm_fwd(const gnan_spmm_args* a, gnan_stream_t stream) {
  if (int rc = validate(a)) return rc;
  if (a->n_rows == 0) return GNAN_OK;
  const size_t need = gnan_spmm_fwd_workspace_bytes(a);
  if (need > 0 && (a->workspace == nullptr || a->workspace_bytes < need))
    return gnan::fail(GNAN_ERR_WORKSPACE, "spmm: workspace %zu B < required %zu B", a->workspace_bytes, need);
  const Params p = make_params(a);

  const bool dense = a->rowptr == nullptr;
  const bool smalld = !dense && a->Cw == 1 && a->D <= 4 && !a->weight_by_col && !a->minus_rest;
  int vec, lpr;
  if (a->reduce_cr) {
    pick_tiling(a, static_cast<const float*>(a->S), a->s_stride, &vec, &lpr);  // narrow output: scalar stores
  } else {
    pick_tiling(a, a->Y, a->y_stride, &vec, &lpr);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a->s_dtype == GNAN_BF16) {
    lpr = 1;
    while (lpr * 8 < a->W && lpr < kWave) lpr <<= 1;
    return launch_lpr<8>(p, lpr, false, smalld, st);
  }
  if (a->shell_out != nullptr) {
    const bool ok = smalld && a->W == 1 && a->n_slices == 0 && a->reduce_cr == 0 && !a->s_by_code && a->s_dtype == GNAN_F32 &&
                    a->hot_rows == 0 && vec == 1 && lpr == 1;
    if (!ok)
      return gnan::fail(GNAN_ERR_UNSUPPORTED, "spmm: shell_out serves a one-column fp32 operand on the small-D CSR route without hub-row "
                                              "slices, fused read-out or hot rows");
  }
  if (smalld && hot_kernel_applies(a))                // narrow rows, hottest operand rows in LDS (persistent workgroups)
    return a->W == 1 ? launch_hot<1>(p, st) : (a->W == 2 ? launch_hot<2>(p, st) : launch_hot<4>(p, st));
  return vec == 4 ? launch_lpr<4>(p, lpr, dense, smalld, st) : launch_lpr<1>(p, lpr, dense, smalld, st);
}
