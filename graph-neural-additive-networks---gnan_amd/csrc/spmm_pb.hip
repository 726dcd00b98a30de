// Narrow rho-weighted neighbourhood sum (sum-first order, GNAN.py:157-170: S = f_sums is [N, C], C small) without a single
// per-pair memory request: "propagation blocking" through LDS on both sides (gfx950).
//
// The row-parallel kernels (spmm.hip: spmm_kernel<1,...>, spmm_hot_kernel) issue one 4-byte gather per listed pair.  On the
// 10M-node / 110M-pair R-MAT graph 30 % of those gathers miss the XCD's L2 and each miss drags a 128-byte line over the
// fabric (profiles/r06_pmc_spmm_hot.csv: 3.4e7 fabric requests, 4.3 GB fetched for 0.13 GB of useful operand bytes); the kernel
// sits on that request wall at 0.87 ms = 0.17 of the HBM roofline.  Even if every gather hit L2 the 1.1e8 L2 requests would
// cost ~0.4 ms.  Here NO pair touches L2 at random:
//
//   the pairs of the graph are bucketed ONCE per graph (HopGraph.pb_plan, static index work) into tiles
//   (row bin b, column block cb): a column block is as many operand rows as fit 64 KB of LDS, a row bin as many output
//   accumulators as fit 64 KB of LDS; entries are stored bin-major, 2 + 2 bytes per pair (column inside its block,
//   accumulator inside its bin).
//
//   phase 1  pb_expand_kernel:  a workgroup loads ONE column block of the operand into LDS (coalesced) and writes
//            E[q] = S[col(q)] for the entries of its tiles: 2 B read + 4 W B written per pair, all of it streamed.
//   phase 2  pb_reduce_kernel:  a workgroup owns ONE row bin: its entries are one contiguous range; every entry is added to
//            its row's accumulator in LDS with a 64-bit INTEGER atomic (fixed point scaled from max |S|: order-independent,
//            hence bit-reproducible, and exact to 2^-40 of max |S| — closer to the float64 sum than a float32 chain is);
//            the epilogue applies the row's weights  sum_d (wt(i,d) - wt(i,rest)) * T[i,d] + wt(i,rest) * total  and stores
//            the output row: 2 + 4 W B read per pair.
//
// Hub rows own several accumulators (one per 512 pairs, entries dealt round-robin) so that no LDS address is hit by more
// than a few lanes of an instruction; the epilogue adds them.  The self pair of every row (hop code 0: exactly one pair per
// row in a hop-coded graph) never enters the tiles: the epilogue reads S[self] directly.
// Algorithmic bytes per pair (SURVEY section 8d, W = 1): 4 (index) + 4 (operand) = 8; this scheme moves 2 + 4 + 2 + 4 = 12 B
// per pair, streamed, against ~39 B of fabric traffic per pair for the gathers.
#include "common.hpp"

#include <cstdint>

namespace {

using gnan::kWave;

constexpr int kChunk = 16;          // entries per chunk: tiles are padded to whole chunks (pads add to a dummy accumulator)
constexpr int kThreads = 1024;

struct PbParams {
  int64_t n_rows, n_cols;
  const float* S;
  int W;
  const float* lut;
  int D;
  const int32_t* cnt;
  int64_t cnt_stride;
  const float* s_total;
  float* Y;
  int64_t y_stride;
  int64_t n_entries;
  const uint16_t* src;
  const uint16_t* dst;
  int cb_width, n_cblocks, n_split;
  const int32_t* chunk_q;
  const int32_t* cb_chunk_ptr;
  int n_bins;
  const int32_t* bin_order;
  const int32_t* bin_entry_ptr;
  const int32_t* bin_row_ptr;
  const int32_t* slot_ptr;
  int n_acc, code_base;
  const int32_t* self_col;
  int acc_per_bin, headroom_bits, self_is_row;
  float* E;
  unsigned* absmax;     // bits of max |S| (non-negative floats order like their bit patterns; NaN sorts above +inf)
  // backward (pb_reduce_kernel<2, true>): the operand is the packed rows V[code_base] = [dY_i / cnt(i, d) | dY_i / cnt(i, rest)]
  const float* v_self;  // V[0]: the packed rows of hop code 0 (the self pairs self_col serves), or null
  const float* s_rows;  // [n_rows] forward operand (for the table gradient)
  int64_t s_rows_stride;
  int with_rest;
  float* dS;
  int64_t ds_stride;
  const float* ds_add;        // optional [1]: added to every dS row ...
  const float* ds_add_scale;  // ... times this [1] (optional)
  double* dlut_partial;       // [n_bins, 4]
  // forward extras (W == 1): the rows' raw shell sums out, a separate operand for the self pairs, a constant added to every row
  float* shell_out;
  const float* S_self;
  const float* out_add;
  const float* out_add_scale;
};

__global__ void pb_prep_kernel(unsigned* absmax) {
  if (threadIdx.x == 0) *absmax = 0u;
}

// ---------------------------------------------------------------------------------------------
// phase 1: E[q, :] = S[block start + src[q], :]
// ---------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void expand_store(const PbParams& p, const float* sblk, unsigned s, int q, bool ok) {
  if constexpr (W == 1) {
    const float v = sblk[s];
    if (ok) p.E[q] = v;
  } else if constexpr (W == 2) {
    const float2 v = *reinterpret_cast<const float2*>(sblk + 2 * s);
    if (ok) *reinterpret_cast<float2*>(p.E + 2 * static_cast<int64_t>(q)) = v;
  } else {
    const float4 v = *reinterpret_cast<const float4*>(sblk + 4 * s);
    if (ok) *reinterpret_cast<float4*>(p.E + 4 * static_cast<int64_t>(q)) = v;
  }
}

// BATCH: a wave takes 64 chunks at a time — lane l fetches chunk_q[k + l] with ONE coalesced load, the 16 wave-iterations of
// the batch (four chunks of 16 entries each) read their offsets from it by ds_bpermute, so that every iteration is
// "entry column -> LDS -> store" with nothing in front of it and the 16 column loads of a batch are in flight together.
// (Reading chunk_q inside every iteration put a dependent global load in front of each: 293 us per launch.)
template <int W, bool BATCH>
__global__ __launch_bounds__(kThreads) void pb_expand_kernel(const PbParams p) {
  extern __shared__ __attribute__((aligned(16))) float sblk[];          // [cb_width * W]
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int cb = static_cast<int>(blockIdx.x) / p.n_split, part = static_cast<int>(blockIdx.x) % p.n_split;
  const int64_t c0 = static_cast<int64_t>(cb) * p.cb_width;
  const int ncol = static_cast<int>(p.n_cols - c0 < p.cb_width ? p.n_cols - c0 : p.cb_width);
  {
    const float* src = p.S + c0 * W;                  // 16-byte aligned: c0 * W is a multiple of the block's floats
    const int nf = ncol * W;
    float m = 0.f;
    auto seen = [&m](float v) {
      const float a = fabsf(v);
      m = (a > m || v != v) ? a : m;                  // a NaN sticks (nothing compares greater than it afterwards)
    };
    for (int i = tid * 4; i + 3 < nf; i += kThreads * 4) {
      const float4 v = *reinterpret_cast<const float4*>(src + i);
      *reinterpret_cast<float4*>(sblk + i) = v;
      seen(v.x); seen(v.y); seen(v.z); seen(v.w);
    }
    for (int i = (nf & ~3) + tid; i < nf; i += kThreads) {
      const float v = src[i];
      sblk[i] = v;
      seen(v);
    }
    if (part == 0) {                                 // every column block is seen by exactly one `part == 0` workgroup
      unsigned bits = __float_as_uint(m);
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const unsigned o = __shfl_xor(bits, off);
        bits = o > bits ? o : bits;
      }
      if (lane == 0 && bits) atomicMax(p.absmax, bits);
    }
  }
  __syncthreads();
  const int k_lo = p.cb_chunk_ptr[cb], k_hi = p.cb_chunk_ptr[cb + 1];
  const int g = lane >> 4, l16 = lane & 15;
  if constexpr (BATCH) {
    int per = (k_hi - k_lo + p.n_split - 1) / p.n_split;
    per = (per + 63) & ~63;                           // whole batches
    const int lo = k_lo + part * per;
    const int hi = lo + per < k_hi ? lo + per : k_hi;
    for (int kb = lo + wave * kWave; kb < hi; kb += (kThreads / kWave) * kWave) {
      const int cq = kb + lane < hi ? p.chunk_q[kb + lane] : -1;
      int q[16];
      unsigned s[16];
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int q0 = __shfl(cq, it * 4 + g);
        q[it] = q0 < 0 ? -1 : q0 + l16;
      }
#pragma unroll
      for (int it = 0; it < 16; ++it) s[it] = p.src[q[it] < 0 ? 0 : q[it]];
#pragma unroll
      for (int it = 0; it < 16; ++it) expand_store<W>(p, sblk, s[it], q[it], q[it] >= 0);
    }
  } else {
    int per = (k_hi - k_lo + p.n_split - 1) / p.n_split;
    per = (per + 3) & ~3;                               // whole wave-iterations (four chunks of 16 entries)
    const int lo = k_lo + part * per;
    const int hi = lo + per < k_hi ? lo + per : k_hi;
    constexpr int U = 4;                                // wave-iterations in flight
    for (int kw = lo + wave * 4; kw < hi; kw += (kThreads / kWave) * 4 * U) {
      int q[U];
      unsigned s[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = kw + u * (kThreads / kWave) * 4 + g;
        ok[u] = k < hi;
        q[u] = p.chunk_q[ok[u] ? k : lo] + l16;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) s[u] = p.src[q[u]];
#pragma unroll
      for (int u = 0; u < U; ++u) expand_store<W>(p, sblk, s[u], q[u], ok[u]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// phase 2: one row bin per workgroup; 64-bit fixed-point accumulators in LDS
// ---------------------------------------------------------------------------------------------
// v * 2^shift as a 64-bit integer.  BITS: from the float's own mantissa and exponent (a 24-bit integer shifted into place:
// ~10 integer instructions, truncation toward zero below 2^-shift); otherwise through float64 (convert, multiply, and the
// compiler's multi-instruction double -> int64 sequence; round to nearest).
template <bool BITS>
__device__ __forceinline__ long long to_fixed(float v, double scale, int shift) {
  if constexpr (BITS) {
    const unsigned b = __float_as_uint(v);
    const int ex = static_cast<int>((b >> 23) & 0xffu);
    const long long m = static_cast<long long>((b & 0x7fffffu) | (ex ? 0x800000u : 0u));
    const int sh = (ex ? ex : 1) - 150 + shift;                    // v = +-m * 2^(ex - 150)
    // both shifts are always executed (one of the two amounts is zero): a `sh >= 0 ? << : >>` became a divergent branch per entry
    const int up = sh > 0 ? (sh > 62 ? 62 : sh) : 0, down = sh < 0 ? (sh < -63 ? 63 : -sh) : 0;
    const long long x = (m << up) >> down;
    return (b >> 31) ? -x : x;
  } else {
    return __double2ll_rn(static_cast<double>(v) * scale);
  }
}

template <bool BITS>
__device__ __forceinline__ void lds_add(long long* acc, int idx, float v, double scale, int shift) {
  const long long x = to_fixed<BITS>(v, scale, shift);
  atomicAdd(reinterpret_cast<unsigned long long*>(acc + idx), static_cast<unsigned long long>(x));
}

// The forward epilogue over the rows [r_lo, r_hi) of a bin: RB rows per thread at a time, every load of the batch issued before the
// first use (one row after the other — slot bounds -> accumulators, self column -> operand value — the epilogue took as long as
// streaming the entries).  No load sits behind a condition: a guarded load ends in s_waitcnt vmcnt(0) and the batch would run one
// load at a time; an absent count table / self column is read from some other valid array and the value dropped by a select.
template <int W, bool SINGLE, bool SELFROW>
__device__ __forceinline__ void fwd_rows(const PbParams& p, const long long* acc, int tid, int r_lo, int r_hi, int slot0,
                                         double inv_scale, bool bad) {
  constexpr int RB = 4;
  const int rest = p.D - 1;
  float l[4], tot[W];
#pragma unroll
  for (int d = 0; d < 4; ++d) l[d] = d < p.D ? p.lut[d] : 0.f;
#pragma unroll
  for (int w = 0; w < W; ++w) tot[w] = p.s_total ? p.s_total[w] : 0.f;
  const bool has_cnt = p.cnt != nullptr, has_self = SELFROW || p.self_col != nullptr;
  const int32_t* cnt_base = has_cnt ? p.cnt : reinterpret_cast<const int32_t*>(p.lut);
  const int64_t cnt_step = has_cnt ? p.cnt_stride : 0;
  const int32_t* self_base = p.self_col ? p.self_col : p.slot_ptr;
  const float* s_self = p.S_self ? p.S_self : p.S;
  const float add = p.out_add ? p.out_add[0] * p.out_add_scale[0] : 0.f;
  for (int i0 = r_lo + tid; i0 < r_hi; i0 += kThreads * RB) {
    int s_lo[RB], s_hi[RB], sc[RB], c[RB][4];
    float sv[RB][W];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int i = i0 + r * kThreads < r_hi ? i0 + r * kThreads : r_lo;      // rows past the end re-read the first row
      if constexpr (SINGLE) {
        s_lo[r] = i - r_lo;
        s_hi[r] = s_lo[r] + 1;
      } else {
        s_lo[r] = p.slot_ptr[i] - slot0;
        s_hi[r] = p.slot_ptr[i + 1] - slot0;
      }
      if constexpr (SELFROW) sc[r] = i;
      else sc[r] = self_base[i];
#pragma unroll
      for (int d = 0; d < 4; ++d) c[r][d] = cnt_base[static_cast<int64_t>(i) * cnt_step + (d < p.D ? d : 0)];
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      sc[r] = has_self ? sc[r] : -1;
#pragma unroll
      for (int d = 0; d < 4; ++d) c[r][d] = has_cnt ? c[r][d] : 1;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
#pragma unroll
      for (int w = 0; w < W; ++w) sv[r][w] = s_self[static_cast<int64_t>(sc[r] < 0 ? 0 : sc[r]) * W + w];
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int i = i0 + r * kThreads;
      if (i >= r_hi) break;
      float wt[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) wt[d] = l[d] / static_cast<float>(c[r][d] > 1 ? c[r][d] : 1);   // IEEE division, as torch.div (l[d] = 0 for d >= D)
      const float w_rest = p.s_total ? wt[rest] : 0.f;
#pragma unroll
      for (int w = 0; w < W; ++w) {
        float out = 0.f;
        if (sc[r] >= 0) out = fmaf(wt[0] - w_rest, sv[r][w], out);
        for (int a = 0; a < p.n_acc; ++a) {
          long long t = 0;
          for (int sl = s_lo[r]; sl < s_hi[r]; ++sl) t += acc[(sl * p.n_acc + a) * W + w];
          const float tf = static_cast<float>(static_cast<double>(t) * inv_scale);
          if constexpr (W == 1) {
            if (p.shell_out) p.shell_out[i] = bad ? __uint_as_float(0x7fc00000u) : tf;      // (n_acc == 1 with shell_out: validated)
          }
          const int d = p.code_base + a;
          const float wd = d == 0 ? wt[0] : (d == 1 ? wt[1] : (d == 2 ? wt[2] : wt[3]));
          out = fmaf(wd - w_rest, tf, out);
        }
        if (p.s_total) out = fmaf(w_rest, tot[w], out);
        out += add;
        if (bad) out = __uint_as_float(0x7fc00000u);
        p.Y[static_cast<int64_t>(i) * p.y_stride + w] = out;
      }
    }
  }
}

// The backward epilogue over the rows of a bin (see pb_reduce_kernel<2, true>): same batching as fwd_rows.
template <bool SINGLE, bool SELFROW>
__device__ __forceinline__ void bwd_rows(const PbParams& p, const long long* acc, int tid, int r_lo, int r_hi, int slot0,
                                         double inv_scale, bool bad, float l0, float l1, float lr, float add, double& g0, double& g1,
                                         double& gr) {
  constexpr int RB = 4;
  const int d1 = p.code_base;
  const bool selfs = SELFROW || (p.self_col && p.v_self);
  const int32_t* self_base = (p.self_col && p.v_self) ? p.self_col : p.slot_ptr;         // (unconditional loads, see fwd_rows)
  const float* vself_base = p.v_self ? p.v_self : p.S;
  for (int i0 = r_lo + tid; i0 < r_hi; i0 += kThreads * RB) {
    int s_lo[RB], s_hi[RB], sc[RB];
    float sj[RB];
    float2 v0[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int i = i0 + r * kThreads < r_hi ? i0 + r * kThreads : r_lo;
      if constexpr (SINGLE) {
        s_lo[r] = i - r_lo;
        s_hi[r] = s_lo[r] + 1;
      } else {
        s_lo[r] = p.slot_ptr[i] - slot0;
        s_hi[r] = p.slot_ptr[i + 1] - slot0;
      }
      if constexpr (SELFROW) sc[r] = i;
      else sc[r] = self_base[i];
      sj[r] = p.s_rows[static_cast<int64_t>(i) * p.s_rows_stride];
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      sc[r] = selfs ? sc[r] : -1;
      v0[r] = *reinterpret_cast<const float2*>(vself_base + 2 * static_cast<int64_t>(sc[r] < 0 ? 0 : sc[r]));
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int i = i0 + r * kThreads;
      if (i >= r_hi) break;
      long long t1i = 0, tri = 0;
      for (int sl = s_lo[r]; sl < s_hi[r]; ++sl) {
        t1i += acc[2 * sl];
        tri += acc[2 * sl + 1];
      }
      const float t1 = static_cast<float>(static_cast<double>(t1i) * inv_scale);
      float tr = static_cast<float>(static_cast<double>(tri) * inv_scale);
      float a0 = 0.f;
      if (sc[r] >= 0) {
        a0 = v0[r].x;
        tr += v0[r].y;
      }
      float ds = d1 == 0 ? 0.f : l0 * a0;
      ds = fmaf(l1, t1, ds);
      ds = fmaf(-lr, tr, ds);
      ds += add;
      if (bad) ds = __uint_as_float(0x7fc00000u);
      p.dS[static_cast<int64_t>(i) * p.ds_stride] = ds;
      const double s64 = static_cast<double>(sj[r]);
      g0 += s64 * a0;
      g1 += s64 * t1;
      gr += s64 * tr;
    }
  }
}

template <int W, bool BWD = false, bool BITS = false, int U = 2>
__global__ __launch_bounds__(kThreads) void pb_reduce_kernel(const PbParams p) {
  extern __shared__ __attribute__((aligned(16))) long long acc[];     // [acc_per_bin * W]
  const int tid = threadIdx.x;
  const int b = p.bin_order[blockIdx.x];
  for (int i = tid; i < p.acc_per_bin * W; i += kThreads) acc[i] = 0;
  // fixed point: |v| <= mx < 2^e; at most 2^headroom terms per output row => |sum * 2^shift| < 2^62
  const float mx = __uint_as_float(*p.absmax);
  const bool bad = !(mx <= 3.0e38f);                                    // inf or NaN somewhere in the operand
  int e = 0;
  if (!bad && mx > 0.f) (void)frexpf(mx, &e);
  const int shift = 62 - p.headroom_bits - e;
  const double scale = bad ? 0.0 : ldexp(1.0, shift), inv_scale = bad ? 0.0 : ldexp(1.0, -shift);
  __syncthreads();
  const int q_lo = p.bin_entry_ptr[b], q_hi = p.bin_entry_ptr[b + 1];   // multiples of kChunk
  for (int base = q_lo + tid * 4; base < q_hi; base += kThreads * 4 * U) {
    uint2 d[U];
    float4 v[U][W];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = base + u * kThreads * 4;
      ok[u] = q < q_hi;
      const int qs = ok[u] ? q : q_lo;
      d[u] = *reinterpret_cast<const uint2*>(p.dst + qs);
#pragma unroll
      for (int w = 0; w < W; ++w) v[u][w] = *reinterpret_cast<const float4*>(p.E + static_cast<int64_t>(qs) * W + 4 * w);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // (no branch around the atomics — the compiler sinks the second round's loads behind it: rounds past the end re-read
      //  the bin's first entries and add them to the pads' dummy accumulator)
      const int dummy = p.acc_per_bin - 1;
      const int i0 = ok[u] ? static_cast<int>(d[u].x & 0xffffu) : dummy, i1 = ok[u] ? static_cast<int>(d[u].x >> 16) : dummy;
      const int i2 = ok[u] ? static_cast<int>(d[u].y & 0xffffu) : dummy, i3 = ok[u] ? static_cast<int>(d[u].y >> 16) : dummy;
      if constexpr (W == 1) {
        lds_add<BITS>(acc, i0, v[u][0].x, scale, shift);
        lds_add<BITS>(acc, i1, v[u][0].y, scale, shift);
        lds_add<BITS>(acc, i2, v[u][0].z, scale, shift);
        lds_add<BITS>(acc, i3, v[u][0].w, scale, shift);
      } else if constexpr (W == 2) {
        lds_add<BITS>(acc, 2 * i0, v[u][0].x, scale, shift); lds_add<BITS>(acc, 2 * i0 + 1, v[u][0].y, scale, shift);
        lds_add<BITS>(acc, 2 * i1, v[u][0].z, scale, shift); lds_add<BITS>(acc, 2 * i1 + 1, v[u][0].w, scale, shift);
        lds_add<BITS>(acc, 2 * i2, v[u][1].x, scale, shift); lds_add<BITS>(acc, 2 * i2 + 1, v[u][1].y, scale, shift);
        lds_add<BITS>(acc, 2 * i3, v[u][1].z, scale, shift); lds_add<BITS>(acc, 2 * i3 + 1, v[u][1].w, scale, shift);
      } else {
        const int idx[4] = {i0, i1, i2, i3};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          lds_add<BITS>(acc, 4 * idx[j], v[u][j].x, scale, shift); lds_add<BITS>(acc, 4 * idx[j] + 1, v[u][j].y, scale, shift);
          lds_add<BITS>(acc, 4 * idx[j] + 2, v[u][j].z, scale, shift); lds_add<BITS>(acc, 4 * idx[j] + 3, v[u][j].w, scale, shift);
        }
      }
    }
  }
  __syncthreads();
  // ---- epilogue: the rows of the bin ---------------------------------------------------------
  const int r_lo = p.bin_row_ptr[b], r_hi = p.bin_row_ptr[b + 1];
  const int slot0 = p.slot_ptr[r_lo];
  const int rest = p.D - 1;
  if constexpr (!BWD) {
    // which of the per-row index arrays the bin needs at all (workgroup-uniform: whole-batch variants, no guarded loads):
    //   single   every row of the bin owns exactly one accumulator slot (all but the bins that hold a hub row): slot = row - first row
    //   selfrow  the self pair of row i lists operand row i (gnan_spmm_pb_args.self_is_row): no self_col read
    const bool single = p.slot_ptr[r_hi] - slot0 == r_hi - r_lo;
    const bool selfrow = p.self_is_row != 0;
    if (single && selfrow) fwd_rows<W, true, true>(p, acc, tid, r_lo, r_hi, slot0, inv_scale, bad);
    else if (single) fwd_rows<W, true, false>(p, acc, tid, r_lo, r_hi, slot0, inv_scale, bad);
    else if (selfrow) fwd_rows<W, false, true>(p, acc, tid, r_lo, r_hi, slot0, inv_scale, bad);
    else fwd_rows<W, false, false>(p, acc, tid, r_lo, r_hi, slot0, inv_scale, bad);
  } else {
    // Backward over the TRANSPOSED adjacency (W == 2, one accumulated code d1 = code_base): row j of the bin is operand node j,
    //   t1 = sum_{i lists j with code d1} dY_i / cnt(i, d1),   tr = sum_{the same i} dY_i / cnt(i, rest),
    //   (a0, b0) = the packed row of j's self pair (hop code 0), then
    //   dS_j = l_0 a0 + l_d1 t1 - l_rest (tr + b0) [+ ds_add],     dlut_d += S_j * (a0 | t1),   dlut_rest -= S_j (tr + b0).
    static_assert(!BWD || W == 2, "the backward's packed rows are two floats");
    const int d1 = p.code_base;
    const float l0 = p.lut[0], l1 = p.lut[d1], lr = p.with_rest ? p.lut[rest] : 0.f;
    float add = 0.f;
    if (p.ds_add) add = p.ds_add_scale ? p.ds_add[0] * p.ds_add_scale[0] : p.ds_add[0];
    double g0 = 0.0, g1 = 0.0, gr = 0.0;
    const bool single = p.slot_ptr[r_hi] - slot0 == r_hi - r_lo;
    const bool selfrow = p.self_is_row != 0 && p.v_self != nullptr;
    if (single && selfrow) bwd_rows<true, true>(p, acc, tid, r_lo, r_hi, slot0, inv_scale, bad, l0, l1, lr, add, g0, g1, gr);
    else if (single) bwd_rows<true, false>(p, acc, tid, r_lo, r_hi, slot0, inv_scale, bad, l0, l1, lr, add, g0, g1, gr);
    else if (selfrow) bwd_rows<false, true>(p, acc, tid, r_lo, r_hi, slot0, inv_scale, bad, l0, l1, lr, add, g0, g1, gr);
    else bwd_rows<false, false>(p, acc, tid, r_lo, r_hi, slot0, inv_scale, bad, l0, l1, lr, add, g0, g1, gr);
    // per-bin partials of the table gradient: lanes, then waves, in a fixed order; the bins are added by pb_dlut_final_kernel
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      g0 += __shfl_xor(g0, off);
      g1 += __shfl_xor(g1, off);
      gr += __shfl_xor(gr, off);
    }
    __syncthreads();                                   // every accumulator has been read: the LDS is free
    double* red = reinterpret_cast<double*>(acc);
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    if (lane == 0) {
      red[wave * 3] = g0;
      red[wave * 3 + 1] = g1;
      red[wave * 3 + 2] = gr;
    }
    __syncthreads();
    if (tid < 3) {
      double t = 0.0;
      for (int w = 0; w < kThreads / kWave; ++w) t += red[w * 3 + tid];
      p.dlut_partial[static_cast<int64_t>(b) * 4 + tid] = bad ? __longlong_as_double(0x7ff8000000000000LL) : t;
    }
  }
}

// dlut[d] from the bins' partials, in bin order (one workgroup; n_bins ~ 10^3); <rest_total, rest_q> goes into dlut[rest]
__global__ __launch_bounds__(256) void pb_dlut_final_kernel(const double* __restrict__ partial, int n_bins, int D, int d1, int with_rest,
                                                            const float* rest_total, const float* rest_q, float* __restrict__ dlut) {
  __shared__ double red[256][3];
  const int tid = threadIdx.x;
  double g[3] = {0.0, 0.0, 0.0};
  for (int b = tid; b < n_bins; b += 256) {
#pragma unroll
    for (int k = 0; k < 3; ++k) g[k] += partial[static_cast<int64_t>(b) * 4 + k];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) red[tid][k] = g[k];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
#pragma unroll
      for (int k = 0; k < 3; ++k) red[tid][k] += red[tid + off][k];
    }
    __syncthreads();
  }
  if (tid == 0) {
    for (int d = 0; d < D; ++d) dlut[d] = 0.f;
    if (d1 != 0) dlut[0] = static_cast<float>(red[0][0]);
    dlut[d1] = static_cast<float>(red[0][1]);
    if (with_rest) {
      double r = -red[0][2];
      if (rest_total && rest_q) r += static_cast<double>(rest_total[0]) * static_cast<double>(rest_q[0]);
      dlut[D - 1] = static_cast<float>(r);
    }
  }
}

// ---- the one-column backward's row pass (gnan_spmm_pb_pack1): c, e, and float64 partials of (q, g0, g1, gr) per workgroup ----------
struct Pack1Params {
  int64_t n;
  const float* dY;
  int64_t dy_stride;
  const int32_t* cnt;
  int64_t cnt_stride;
  int D, d1, with_rest, self_is_row;
  const int32_t* self_col;
  const float* lut;
  const float* S;
  const float* shell;
  float* c;
  float* e;
  double* partial;      // [blocks, 4]
};

constexpr int kPackRows = 4;      // rows per thread

__global__ __launch_bounds__(256) void pb_pack1_kernel(const Pack1Params p) {
  const int rest = p.D - 1;
  const float l0 = p.lut[0], l1 = p.lut[p.d1], lr = p.with_rest ? p.lut[rest] : 0.f;
  const bool has_cnt = p.cnt != nullptr;
  const int32_t* cnt_base = has_cnt ? p.cnt : reinterpret_cast<const int32_t*>(p.lut);       // (unconditional loads, see fwd_rows)
  const int64_t cnt_step = has_cnt ? p.cnt_stride : 0;
  const bool col_given = p.self_col != nullptr;
  const int32_t* self_base = col_given ? p.self_col : reinterpret_cast<const int32_t*>(p.lut);
  double q = 0.0, g0 = 0.0, g1 = 0.0, gr = 0.0;
  const int64_t first = (static_cast<int64_t>(blockIdx.x) * kPackRows) * 256 + threadIdx.x;
  float dy[kPackRows], sh[kPackRows], sv[kPackRows];
  int c0[kPackRows], c1[kPackRows], cr[kPackRows], sc[kPackRows];
#pragma unroll
  for (int r = 0; r < kPackRows; ++r) {
    const int64_t i = first + r * 256 < p.n ? first + r * 256 : 0;
    dy[r] = p.dY[i * p.dy_stride];
    sh[r] = p.shell[i];
    c0[r] = cnt_base[i * cnt_step];
    c1[r] = cnt_base[i * cnt_step + (has_cnt ? p.d1 : 0)];
    cr[r] = cnt_base[i * cnt_step + (has_cnt ? rest : 0)];
    sc[r] = self_base[col_given ? i : 0];
  }
#pragma unroll
  for (int r = 0; r < kPackRows; ++r) {
    const int64_t i = first + r * 256 < p.n ? first + r * 256 : 0;
    sc[r] = p.self_is_row ? static_cast<int>(i) : (col_given ? sc[r] : -1);
    sv[r] = p.S[sc[r] < 0 ? 0 : sc[r]];
  }
#pragma unroll
  for (int r = 0; r < kPackRows; ++r) {
    const int64_t i = first + r * 256;
    if (i >= p.n) break;
    const float d0 = has_cnt ? static_cast<float>(c0[r] > 1 ? c0[r] : 1) : 1.f;
    const float d1f = has_cnt ? static_cast<float>(c1[r] > 1 ? c1[r] : 1) : 1.f;
    const float drf = has_cnt ? static_cast<float>(cr[r] > 1 ? cr[r] : 1) : 1.f;
    const float a0 = dy[r] / d0, a1 = dy[r] / d1f, ar = p.with_rest ? dy[r] / drf : 0.f;      // IEEE division, as the packed rows
    const bool self = sc[r] >= 0;
    const float sself = self ? sv[r] : 0.f;
    p.c[i] = fmaf(l1, a1, -(lr * ar));
    p.e[i] = self ? fmaf(l0, a0, -(lr * ar)) : 0.f;
    q += static_cast<double>(ar);
    g0 += static_cast<double>(sself) * static_cast<double>(a0);
    g1 += static_cast<double>(sh[r]) * static_cast<double>(a1);
    gr += (static_cast<double>(sself) + static_cast<double>(sh[r])) * static_cast<double>(ar);
  }
  __shared__ double red[256][4];
  red[threadIdx.x][0] = q; red[threadIdx.x][1] = g0; red[threadIdx.x][2] = g1; red[threadIdx.x][3] = gr;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (static_cast<int>(threadIdx.x) < off) {
#pragma unroll
      for (int k = 0; k < 4; ++k) red[threadIdx.x][k] += red[threadIdx.x + off][k];
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) p.partial[static_cast<int64_t>(blockIdx.x) * 4 + threadIdx.x] = red[0][threadIdx.x];
}

// q and dlut from the workgroups' partials: one 1024-thread workgroup, records of 32 bytes, fixed order (~10^4 records on 10M rows)
__global__ __launch_bounds__(1024) void pb_pack1_final_kernel(const double* __restrict__ partial, int64_t n_blocks, int D, int d1,
                                                              int with_rest, const float* s_total, float* __restrict__ q_out,
                                                              float* __restrict__ dlut) {
  __shared__ double red[1024][4];
  double g[4] = {0.0, 0.0, 0.0, 0.0};
  const double2* rec = reinterpret_cast<const double2*>(partial);
  for (int64_t b = threadIdx.x; b < n_blocks; b += 1024) {
    const double2 lo = rec[b * 2], hi = rec[b * 2 + 1];
    g[0] += lo.x; g[1] += lo.y; g[2] += hi.x; g[3] += hi.y;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[threadIdx.x][k] = g[k];
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (static_cast<int>(threadIdx.x) < off) {
#pragma unroll
      for (int k = 0; k < 4; ++k) red[threadIdx.x][k] += red[threadIdx.x + off][k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double q = red[0][0];
    q_out[0] = static_cast<float>(q);
    for (int d = 0; d < D; ++d) dlut[d] = 0.f;
    dlut[0] = static_cast<float>(red[0][1]);
    dlut[d1] = static_cast<float>(red[0][2]);
    if (with_rest) dlut[D - 1] = static_cast<float>(-red[0][3] + (s_total ? static_cast<double>(s_total[0]) * static_cast<double>(q_out[0]) : 0.0));
  }
}

int validate(const gnan_spmm_pb_args* a) {
  GNAN_REQUIRE(a != nullptr, "gnan_spmm_pb_fwd: null arguments");
  GNAN_REQUIRE(a->W == 1 || a->W == 2 || a->W == 4, "gnan_spmm_pb_fwd: W must be 1, 2 or 4 (got %d)", a->W);
  GNAN_REQUIRE(a->n_rows >= 0 && a->n_cols > 0, "gnan_spmm_pb_fwd: bad sizes");
  GNAN_REQUIRE(a->D >= 2 && a->D <= 4, "gnan_spmm_pb_fwd: D must be in [2, 4] (got %d)", a->D);
  GNAN_REQUIRE(a->n_acc >= 1 && a->code_base >= 0 && a->code_base + a->n_acc <= a->D - 1,
               "gnan_spmm_pb_fwd: the accumulators cover hop codes [%d, %d) of %d listed ones", a->code_base,
               a->code_base + a->n_acc, a->D - 1);
  GNAN_REQUIRE(a->s_stride == a->W, "gnan_spmm_pb_fwd: operand rows must be contiguous (s_stride == W)");
  GNAN_REQUIRE(a->S && a->lut && a->Y && a->src && a->dst && a->chunk_q && a->cb_chunk_ptr && a->bin_order && a->bin_entry_ptr &&
                   a->bin_row_ptr && a->slot_ptr,
               "gnan_spmm_pb_fwd: null pointer");
  GNAN_REQUIRE(a->n_entries >= 0 && a->n_entries % kChunk == 0 && a->n_entries < (int64_t{1} << 31),
               "gnan_spmm_pb_fwd: n_entries must be a multiple of %d below 2^31", kChunk);
  GNAN_REQUIRE(a->cb_width > 0 && static_cast<int64_t>(a->cb_width) * a->W * 4 <= 65536 && a->cb_width <= 65536,
               "gnan_spmm_pb_fwd: a column block must fit 64 KB of LDS");
  GNAN_REQUIRE(static_cast<int64_t>(a->n_cblocks) * a->cb_width >= a->n_cols && a->n_cblocks > 0, "gnan_spmm_pb_fwd: column blocks do not cover n_cols");
  GNAN_REQUIRE(a->acc_per_bin > 0 && static_cast<int64_t>(a->acc_per_bin) * a->W * 8 <= 65536 && a->acc_per_bin <= 65536,
               "gnan_spmm_pb_fwd: a bin's accumulators must fit 64 KB of LDS");
  GNAN_REQUIRE(a->n_bins > 0 && a->headroom_bits >= 0 && a->headroom_bits <= 40, "gnan_spmm_pb_fwd: bad plan");
  GNAN_REQUIRE(a->y_stride >= a->W, "gnan_spmm_pb_fwd: y_stride < W");
  GNAN_REQUIRE(!a->self_is_row || a->n_cols >= a->n_rows, "gnan_spmm_pb_fwd: self_is_row needs an operand row per output row");
  GNAN_REQUIRE((reinterpret_cast<uintptr_t>(a->dst) % 8) == 0 && (reinterpret_cast<uintptr_t>(a->src) % 2) == 0,
               "gnan_spmm_pb_fwd: dst must be 8-byte aligned");
  GNAN_REQUIRE((a->shell_out == nullptr && a->S_self == nullptr) || a->W == 1, "gnan_spmm_pb_fwd: shell_out / S_self serve W == 1");
  GNAN_REQUIRE(a->shell_out == nullptr || a->n_acc == 1, "gnan_spmm_pb_fwd: shell_out is the sum of ONE accumulated hop code");
  GNAN_REQUIRE((a->out_add == nullptr) == (a->out_add_scale == nullptr), "gnan_spmm_pb_fwd: out_add and out_add_scale go together");
  return GNAN_OK;
}

size_t pb_bytes(const gnan_spmm_pb_args* a) {
  return 256 + static_cast<size_t>(a->n_entries) * static_cast<size_t>(a->W) * sizeof(float);
}

template <int W, bool BWD = false>
int launch(const PbParams& p, hipStream_t st, int flags) {
  hipLaunchKernelGGL(pb_prep_kernel, dim3(1), dim3(64), 0, st, p.absmax);
  if (int rc = gnan::check_launch("pb_prep_kernel")) return rc;
  const size_t lds1 = static_cast<size_t>(p.cb_width) * W * sizeof(float);
  const dim3 grid1(static_cast<unsigned>(p.n_cblocks) * p.n_split);
  if (flags & GNAN_PB_EXPAND_PER_ITERATION)
    hipLaunchKernelGGL((pb_expand_kernel<W, false>), grid1, dim3(kThreads), lds1, st, p);
  else
    hipLaunchKernelGGL((pb_expand_kernel<W, true>), grid1, dim3(kThreads), lds1, st, p);
  if (int rc = gnan::check_launch("pb_expand_kernel")) return rc;
  const size_t lds2 = static_cast<size_t>(p.acc_per_bin) * W * sizeof(long long);
  const dim3 grid2(static_cast<unsigned>(p.n_bins));
  const bool dbl = (flags & GNAN_PB_FIXED_VIA_DOUBLE) != 0, u4 = (flags & GNAN_PB_REDUCE_UNROLL4) != 0;
  if (dbl && u4) hipLaunchKernelGGL((pb_reduce_kernel<W, BWD, false, 4>), grid2, dim3(kThreads), lds2, st, p);
  else if (dbl) hipLaunchKernelGGL((pb_reduce_kernel<W, BWD, false, 2>), grid2, dim3(kThreads), lds2, st, p);
  else if (u4) hipLaunchKernelGGL((pb_reduce_kernel<W, BWD, true, 4>), grid2, dim3(kThreads), lds2, st, p);
  else hipLaunchKernelGGL((pb_reduce_kernel<W, BWD, true, 2>), grid2, dim3(kThreads), lds2, st, p);
  return gnan::check_launch("pb_reduce_kernel");
}

PbParams make_params(const gnan_spmm_pb_args* a) {
  PbParams p{};
  p.n_rows = a->n_rows; p.n_cols = a->n_cols; p.S = a->S; p.W = a->W; p.lut = a->lut; p.D = a->D;
  p.cnt = a->cnt; p.cnt_stride = a->cnt_stride; p.s_total = a->s_total; p.Y = a->Y; p.y_stride = a->y_stride;
  p.n_entries = a->n_entries; p.src = a->src; p.dst = a->dst; p.cb_width = a->cb_width; p.n_cblocks = a->n_cblocks;
  p.chunk_q = a->chunk_q; p.cb_chunk_ptr = a->cb_chunk_ptr; p.n_bins = a->n_bins; p.bin_order = a->bin_order;
  p.bin_entry_ptr = a->bin_entry_ptr; p.bin_row_ptr = a->bin_row_ptr; p.slot_ptr = a->slot_ptr; p.n_acc = a->n_acc;
  p.code_base = a->code_base; p.self_col = a->self_col; p.acc_per_bin = a->acc_per_bin; p.headroom_bits = a->headroom_bits;
  p.self_is_row = a->self_is_row;
  p.shell_out = a->shell_out; p.S_self = a->S_self; p.out_add = a->out_add; p.out_add_scale = a->out_add_scale;
  p.absmax = static_cast<unsigned*>(a->workspace);
  p.E = reinterpret_cast<float*>(static_cast<char*>(a->workspace) + 256);
  // enough workgroups per column block that the launch is >> the resident ones (two per CU) whatever the block count
  int split = (2048 + a->n_cblocks - 1) / a->n_cblocks;
  if ((a->flags >> 8) & 0xff) split = (a->flags >> 8) & 0xff;          // A/B: workgroups per column block
  p.n_split = split < 1 ? 1 : (split > 64 ? 64 : split);
  return p;
}

size_t pb_bwd_bytes(const gnan_spmm_pb_bwd_args* g) {
  return (pb_bytes(&g->pb) + 15) / 16 * 16 + static_cast<size_t>(g->pb.n_bins) * 4 * sizeof(double);
}

}  // namespace

extern "C" size_t gnan_spmm_pb_workspace_bytes(const gnan_spmm_pb_args* a) {
  if (!a || a->n_entries < 0 || a->W <= 0) return 0;
  return pb_bytes(a);
}

extern "C" int gnan_spmm_pb_fwd(const gnan_spmm_pb_args* a, gnan_stream_t stream) {
  if (int rc = validate(a)) return rc;
  GNAN_REQUIRE(a->workspace && a->workspace_bytes >= pb_bytes(a), "gnan_spmm_pb_fwd: workspace too small (%zu < %zu)",
               a->workspace_bytes, pb_bytes(a));
  GNAN_REQUIRE((reinterpret_cast<uintptr_t>(a->workspace) % 16) == 0, "gnan_spmm_pb_fwd: workspace must be 16-byte aligned");
  if (a->n_rows == 0) return GNAN_OK;
  const PbParams p = make_params(a);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (a->W) {
    case 1: return launch<1>(p, st, a->flags);
    case 2: return launch<2>(p, st, a->flags);
    default: return launch<4>(p, st, a->flags);
  }
}

extern "C" size_t gnan_spmm_pb_bwd_workspace_bytes(const gnan_spmm_pb_bwd_args* g) {
  if (!g || g->pb.n_entries < 0 || g->pb.W != 2 || g->pb.n_bins <= 0) return 0;
  return pb_bwd_bytes(g);
}

extern "C" int gnan_spmm_pb_bwd(const gnan_spmm_pb_bwd_args* g, gnan_stream_t stream) {
  GNAN_REQUIRE(g != nullptr, "gnan_spmm_pb_bwd: null arguments");
  gnan_spmm_pb_args a = g->pb;
  a.shell_out = nullptr; a.S_self = nullptr; a.out_add = nullptr; a.out_add_scale = nullptr;       // (forward-only fields)
  a.Y = g->dS;                       // (validate() wants an output; the backward epilogue writes dS)
  a.y_stride = g->ds_stride < 2 ? 2 : g->ds_stride;
  if (int rc = validate(&a)) return rc;
  GNAN_REQUIRE(a.W == 2 && a.n_acc == 1, "gnan_spmm_pb_bwd: packed rows of two floats, one accumulated hop code (W %d, n_acc %d)", a.W,
               a.n_acc);
  GNAN_REQUIRE(g->dS && g->dlut && g->s_rows && g->ds_stride >= 1, "gnan_spmm_pb_bwd: null pointer");
  GNAN_REQUIRE(a.code_base == 0 || g->v_self || !a.self_col, "gnan_spmm_pb_bwd: self pairs are left out of the entries but v_self is null");
  GNAN_REQUIRE(a.workspace && a.workspace_bytes >= pb_bwd_bytes(g), "gnan_spmm_pb_bwd: workspace too small (%zu < %zu)",
               a.workspace_bytes, pb_bwd_bytes(g));
  GNAN_REQUIRE((reinterpret_cast<uintptr_t>(a.workspace) % 16) == 0, "gnan_spmm_pb_bwd: workspace must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  PbParams p = make_params(&a);
  p.v_self = g->v_self; p.s_rows = g->s_rows; p.s_rows_stride = g->s_rows_stride; p.with_rest = g->with_rest;
  p.dS = g->dS; p.ds_stride = g->ds_stride; p.ds_add = g->ds_add; p.ds_add_scale = g->ds_add_scale;
  p.dlut_partial = reinterpret_cast<double*>(static_cast<char*>(a.workspace) + (pb_bytes(&a) + 15) / 16 * 16);
  if (a.n_rows > 0) {
    if (int rc = launch<2, true>(p, st, a.flags)) return rc;
  }
  hipLaunchKernelGGL(pb_dlut_final_kernel, dim3(1), dim3(256), 0, st, p.dlut_partial, a.n_rows > 0 ? a.n_bins : 0, a.D, a.code_base,
                     g->with_rest, g->rest_total, g->rest_q, g->dlut);
  return gnan::check_launch("pb_dlut_final_kernel");
}

extern "C" size_t gnan_spmm_pb_pack1_workspace_bytes(int64_t n) {
  if (n <= 0) return 32;
  return static_cast<size_t>((n + 256 * kPackRows - 1) / (256 * kPackRows)) * 4 * sizeof(double);
}

extern "C" int gnan_spmm_pb_pack1(const gnan_pb_pack1_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr && a->n >= 0 && a->D >= 2 && a->D <= 4, "gnan_spmm_pb_pack1: bad sizes");
  GNAN_REQUIRE(a->d1 >= 1 && a->d1 < a->D - (a->with_rest ? 1 : 0), "gnan_spmm_pb_pack1: d1 = %d is not a listed hop code behind the self code", a->d1);
  GNAN_REQUIRE(a->q && a->dlut && a->lut && (a->n == 0 || (a->dY && a->S && a->shell && a->c && a->e)), "gnan_spmm_pb_pack1: null pointer");
  GNAN_REQUIRE(a->dy_stride >= 1 && (a->cnt == nullptr || a->cnt_stride >= a->D), "gnan_spmm_pb_pack1: row stride smaller than the width");
  GNAN_REQUIRE(a->workspace && a->workspace_bytes >= gnan_spmm_pb_pack1_workspace_bytes(a->n) &&
                   reinterpret_cast<uintptr_t>(a->workspace) % 16 == 0,
               "gnan_spmm_pb_pack1: workspace of %zu bytes, 16-byte aligned", gnan_spmm_pb_pack1_workspace_bytes(a->n));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t blocks = a->n > 0 ? (a->n + 256 * kPackRows - 1) / (256 * kPackRows) : 0;
  GNAN_REQUIRE(blocks < (int64_t{1} << 31), "gnan_spmm_pb_pack1: too many rows for one launch");
  double* partial = static_cast<double*>(a->workspace);
  if (blocks > 0) {
    Pack1Params p;
    p.n = a->n; p.dY = a->dY; p.dy_stride = a->dy_stride; p.cnt = a->cnt; p.cnt_stride = a->cnt_stride; p.D = a->D; p.d1 = a->d1;
    p.with_rest = a->with_rest; p.self_is_row = a->self_is_row; p.self_col = a->self_col; p.lut = a->lut; p.S = a->S;
    p.shell = a->shell; p.c = a->c; p.e = a->e; p.partial = partial;
    hipLaunchKernelGGL(pb_pack1_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, p);
    if (int rc = gnan::check_launch("pb_pack1_kernel")) return rc;
  }
  hipLaunchKernelGGL(pb_pack1_final_kernel, dim3(1), dim3(1024), 0, st, partial, blocks, a->D, a->d1, a->with_rest, a->s_total, a->q,
                     a->dlut);
  return gnan::check_launch("pb_pack1_final_kernel");
}
