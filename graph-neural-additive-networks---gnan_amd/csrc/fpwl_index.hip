// Shape functions by DIRECT-INDEX table look-up (gfx950): the piece of x is found by arithmetic, not by a search.
//
// fpwl_fast_kernel (csrc/fpwl.hip) walks an 8-level search tree per look-up: ~10 dependent LDS round trips and ~38 vector
// instructions, which — not HBM — set its time (SQ counters, profiles/r04_sq_fpwl.csv).  Here every feature gets a
// uniform grid of B buckets over the range the data actually takes (`range`, one column min / max pass per feature
// matrix: a HINT — any x is looked up exactly, see below), and a table entry per bucket:
//     key(x)   = (int) med3(fma(x, ks, ko), 0, B - 1)                 monotone non-decreasing in x
//     entry[k] = 4 * #{ breakpoints a_j : key(a_j) < k }  |  code(#{ a_j : key(a_j) == k }) << 14
// key is monotone, so every breakpoint with key(a_j) < key(x) is <= x and every one with key(a_j) > key(x) is > x:
// the piece of x is entry[key(x)] / 4 plus the number of the breakpoints INSIDE its bucket that are <= x — found by
// comparing x with the next anchors of the sorted array.  The builder computes key(a_j) with the very same
// instructions, so the look-up is exact for every float x whatever the range hint was.  code 0: at most one breakpoint
// in the bucket — one ds_read2_b32 (the piece's anchor and the next) and one comparison, the straight-line path; code 1:
// two or three (the pair a, nextafter(a) that stands behind a kink some input sits on EXACTLY, csrc/pwl_build.hip, is
// the common case: zero biases and one-hot features put most inputs there) — a second ds_read2_b32 and two more
// comparisons under a wave-level branch; code 3: more (typically the two end buckets, which take everything outside
// the hinted range) — a plain binary search over the sorted anchors, rare.
// Per look-up: 3 dependent LDS reads (entry, anchors, (val, slope)) and ~16 vector instructions; same formula
// val + slope * (x - anchor) on the same piece as fpwl_fast_kernel, hence bit-identical results.
// The kernel is bound by the LDS pipe (round 4 SQ counters: LDS busy 64 % of the CU cycles, 58 % of them bank conflicts —
// three RANDOM gathers per look-up, and a random gather of 32 lanes over the banks costs ~3.5 cycles per lane group whatever
// its width up to 8 bytes).  Round 5: the anchors sit in LDS as PAIRS (a_i, a_{i+1}) per piece, 8-byte aligned, so the two
// anchors a look-up compares with are ONE ds_read_b64 (one gather: ~7 LDS cycles) instead of a ds_read2_b32 (two: ~14);
// the code-1 path's next two anchors are the pair two pieces on.  28 -> 21 LDS cycles per wave look-up.
//
// Mapping: workgroup = (node block, group of FGX features), thread = (node, 4 features), one 16-B load of x and one 16-B
// store per node and thread.  FGX = 32 where F allows: a node's 32 features are ONE full 128-byte line of x and of the
// output, so a wave reads and writes whole lines and no line is shared between workgroups — with the look-up itself this
// cheap the kernel is bound by the copy pattern, and tools/lookup_ceiling.hip (y = 2x with these mappings, no look-up)
// gives 0.98-1.18 ms for half lines (FGX = 16, what fpwl_fast_kernel does) against 0.90 ms for full lines on the
// 10M x 64 matrix (a framework element-wise kernel: 0.86).  The x rows of the next round are requested before the
// current round's look-ups.  Column sums / feature sum / kept pieces in the epilogue as in fpwl_fast_kernel.
#include "common.hpp"
#include "fpwl_bucket.hpp"

#include <cstdint>
#include <type_traits>

namespace {

constexpr int FPT = 4;

struct IndexParams {
  const float* x;
  int64_t n, x_stride;
  int F;
  const int32_t* off;
  const float* anchor;
  const float* val;
  const float* slope;
  const uint16_t* table;   // [F][B]
  const float* key;        // [F][2]: ks, ko
  int n_groups, nodes_per_block;
  int tot_cap;             // pieces the LDS image holds per feature group
  int step0;               // slow path: largest power of two <= max breakpoints per feature
  int sum_features;
  float* out;
  int64_t out_stride;
  double* col_partial;
  int64_t total_rows;
  uint8_t* piece_out;
  float* part;             // feature sum, split over the groups: [n_groups][n] partial sums (else NULL: the groups are walked inside the workgroup)
};

typedef __attribute__((address_space(3))) const float lds_cfloat;
typedef __attribute__((address_space(3))) const uint16_t lds_cu16;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const f32x2_t lds_cfloat2;
__device__ __forceinline__ float lds_f32(int addr) { return *reinterpret_cast<lds_cfloat*>(static_cast<uintptr_t>(static_cast<unsigned>(addr))); }
__device__ __forceinline__ int lds_u16(int addr) { return *reinterpret_cast<lds_cu16*>(static_cast<uintptr_t>(static_cast<unsigned>(addr))); }
__device__ __forceinline__ float2 lds_f32x2(int addr) {
  const f32x2_t v = *reinterpret_cast<lds_cfloat2*>(static_cast<uintptr_t>(static_cast<unsigned>(addr)));
  return make_float2(v.x, v.y);
}

__device__ __forceinline__ unsigned bf16_bits(float f) {      // round-to-nearest-even, as torch.bfloat16
  const unsigned u = __float_as_uint(f);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

using gnan_index::bucket_of;      // (csrc/fpwl_bucket.hpp: ONE definition for the look-up and for the builders)

// ---------------------------------------------------------------------------------------------
// column minima / maxima of the feature matrix: the range hint.  Order-preserving integer images of the floats and
// integer atomics; NaNs are skipped, an all-NaN / empty column ends as (+inf, -inf) and the builder copes.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int ordered(float f) {
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float unordered(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ __launch_bounds__(256) void range_init_kernel(int* __restrict__ lohi, int F) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < F) { lohi[2 * i] = ordered(INFINITY); lohi[2 * i + 1] = ordered(-INFINITY); }
}

__global__ __launch_bounds__(256) void range_kernel(const float* __restrict__ x, int64_t n, int64_t stride, int F,
                                                    int64_t rows_per_block, int* __restrict__ lohi) {
  // thread = column (blockIdx.y covers columns in chunks of 256), rows of this block in order: coalesced along the row
  const int c = blockIdx.y * 256 + threadIdx.x;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  if (c >= F) return;
  float lo = INFINITY, hi = -INFINITY;
  for (int64_t r = r0; r < r1; ++r) {
    const float v = x[r * stride + c];
    lo = fminf(lo, v);                       // fminf / fmaxf drop NaN operands
    hi = fmaxf(hi, v);
  }
  atomicMin(&lohi[2 * c], ordered(lo));
  atomicMax(&lohi[2 * c + 1], ordered(hi));
}

__global__ __launch_bounds__(256) void range_finish_kernel(const int* __restrict__ lohi, int F, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 2 * F) out[i] = unordered(lohi[i]);
}

// ---------------------------------------------------------------------------------------------
// the bucket tables of all features: one workgroup per feature
// ---------------------------------------------------------------------------------------------
template <int LOGB>
__global__ __launch_bounds__(256) void index_build_kernel(const int32_t* __restrict__ off, const float* __restrict__ anchor,
                                                          const float* __restrict__ range,
                                                          uint16_t* __restrict__ table, float* __restrict__ key,
                                                          int32_t* __restrict__ stats) {
  const int f = blockIdx.x, base = off[f];
  gnan_index::build_bucket_index<LOGB>(f, anchor + base, off[f + 1] - base - 1, range, table, key, stats);
}

// ---------------------------------------------------------------------------------------------
// the look-up
// LDS image of a feature group: [FG][B] uint16 entries | anchor PAIRS (a_i, a_{i+1}), feature f's run at s_off[f] + f * KF
// (KF NaN pairs behind every feature, and NaN as the last piece's "next": the comparisons never see the next feature's
// anchors) | [tot] (val, slope) pairs | s_off[FG + 1]
// ---------------------------------------------------------------------------------------------
template <int FG, bool SUM, bool OUT16, int LOGB, int BS>
__global__ __launch_bounds__(BS) void fpwl_index_kernel(const IndexParams p) {
  constexpr int KF = 2;                             // NaN pairs behind every feature's run: what the comparisons may read
  static_assert(FG == 16 || FG == 32, "feature groups of 16 (half lines) or 32 (full lines)");
  constexpr int TPN = FG / FPT, B = 1 << LOGB, NODES = BS / TPN;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)smem));
  constexpr int kTableBytes = FG * B * 2;
  const int a_pairs = p.tot_cap + FG * KF;                       // anchor pairs (a_i, a_{i+1}), 8 bytes each
  float2* an_l = reinterpret_cast<float2*>(smem + kTableBytes / 4);
  float2* vs_l = an_l + a_pairs;
  int* s_off = reinterpret_cast<int*>(vs_l + p.tot_cap);
  const int a_base = static_cast<int>(lds_base) + kTableBytes;
  const int vs_base = a_base + 8 * a_pairs;
  const int tid = threadIdx.x;
  const int q = tid % TPN;
  const int nl = tid / TPN;
  int64_t nb = blockIdx.x;
  int g_first = 0;
  // feature sum on a medium batch: a workgroup per (node block, group) too — its partial sums go to p.part and are added in
  // group order by sum_groups_kernel (walking the nine groups of a 144-feature model inside each of 662 small workgroups paid
  // nine LDS images per 256 nodes and left most of the chip idle: 94 us for 24M look-ups)
  const bool split = SUM && p.part != nullptr;
  if (!SUM || split) {                              // (id % 8) = XCD, groups of a node block adjacent inside it (FG = 16: shared lines)
    const int64_t id = blockIdx.x;
    g_first = static_cast<int>((id >> 3) % p.n_groups);
    nb = ((id >> 3) / p.n_groups) * 8 + (id & 7);
  }
  const int64_t n_lo = nb * p.nodes_per_block;
  if (n_lo >= p.n) return;
  const int64_t n_hi = n_lo + p.nodes_per_block < p.n ? n_lo + p.nodes_per_block : p.n;
  const int g_lo = (SUM && !split) ? 0 : g_first;
  const int g_hi = (SUM && !split) ? p.n_groups : g_lo + 1;
  const float top = static_cast<float>(B - 1);

  for (int g = g_lo; g < g_hi; ++g) {
    const int k0 = g * FG;
    const int base = p.off[k0];
    const int tot = p.off[k0 + FG] - base;
    __syncthreads();                                // previous group's look-ups are done with the LDS image
    if (tid <= FG) s_off[tid] = p.off[k0 + tid] - base;
    {                                               // bucket entries of the group: one contiguous run, 16 bytes per lane
      const uint4* src = reinterpret_cast<const uint4*>(p.table + static_cast<int64_t>(k0) * B);
      uint4* dst = reinterpret_cast<uint4*>(smem);
      for (int i = tid; i < kTableBytes / 16; i += BS) dst[i] = src[i];
    }
    __syncthreads();
    {                                               // x >= NaN is false for every x, +inf included
      const float nanv = __builtin_nanf("");
      for (int i = tid; i < tot + FG * KF; i += BS) an_l[i] = make_float2(nanv, nanv);
    }
    __syncthreads();
    for (int i = tid; i < tot; i += BS) {
      // feature of global piece i: the largest f with s_off[f] <= i (a 4- or 5-step search on the LDS offsets)
      int f = 0;
#pragma unroll
      for (int st = FG / 2; st > 0; st >>= 1) f += (s_off[f + st] <= i) ? st : 0;
      const float next = i + 1 < s_off[f + 1] ? p.anchor[base + i + 1] : __builtin_nanf("");   // the feature's last piece: no next anchor
      an_l[i + f * KF] = make_float2(p.anchor[base + i], next);
      vs_l[i] = make_float2(p.val[base + i], p.slope[base + i]);
    }
    __syncthreads();

    float ks[FPT], ko[FPT];
    int tb, a0[FPT], cvs[FPT];
    tb = static_cast<int>(lds_base) + q * (FPT * B * 2);
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
      const int fg = q * FPT + f;
      ks[f] = p.key[2 * (k0 + fg)];
      ko[f] = p.key[2 * (k0 + fg) + 1];
      a0[f] = a_base + 8 * (s_off[fg] + fg * KF);              // LDS byte address of the feature's pair 0
      cvs[f] = vs_base - a_base - 8 * fg * KF;                 // (val, slope) address = pair address + cvs
    }

    float ps[FPT] = {0.f, 0.f, 0.f, 0.f};           // column sums of the output (per-feature mode)
    auto look_up = [&](const int64_t n, const float4 t, const float before) {
      const float xv[FPT] = {t.x, t.y, t.z, t.w};
      int e[FPT];
#pragma unroll
      for (int f = 0; f < FPT; ++f) {
        const int k = bucket_of(xv[f], ks[f], ko[f], top);
        e[f] = lds_u16(tb + 2 * k + f * (B * 2));
      }
      int pa[FPT];                                  // LDS byte address of the piece's anchor
      float an[FPT];
#pragma unroll
      for (int f = 0; f < FPT; ++f) {
        const int a = a0[f] + 2 * (e[f] & 0x3fff);             // (entries hold 4 * piece)
        const float2 A01 = lds_f32x2(a);                       // (a_i, a_{i+1}): one 8-byte read
        pa[f] = a + (xv[f] >= A01.y ? 8 : 0);
        an[f] = xv[f] >= A01.y ? A01.y : A01.x;
      }
      if ((e[0] | e[1] | e[2] | e[3]) >> 14) {      // some bucket holds more than one breakpoint
#pragma unroll
        for (int f = 0; f < FPT; ++f) {
          if (e[f] >> 14) {                         // two or three: compare with the next two anchors as well
            const int a = a0[f] + 2 * (e[f] & 0x3fff);
            const float2 A23 = lds_f32x2(a + 16);              // pair i + 2 = (a_{i+2}, a_{i+3})
            pa[f] += (xv[f] >= A23.x ? 8 : 0) + (xv[f] >= A23.y ? 8 : 0);
            an[f] = xv[f] >= A23.y ? A23.y : (xv[f] >= A23.x ? A23.x : an[f]);
          }
        }
        if (((e[0] & e[0] >> 1) | (e[1] & e[1] >> 1) | (e[2] & e[2] >> 1) | (e[3] & e[3] >> 1)) >> 14) {   // code 3 somewhere
#pragma unroll
          for (int f = 0; f < FPT; ++f) {
            if ((e[f] >> 14) == 3) {                // more: search the feature's sorted anchors
              const int fg = q * FPT + f;
              const float2* A = an_l + s_off[fg] + fg * KF;
              const int pn = s_off[fg + 1] - s_off[fg] - 1;
              int idx = 0;
              for (int step = p.step0; step > 0; step >>= 1) {
                const int j = idx + step;
                const int jj = j <= pn ? j : 0;
                idx = (j <= pn && A[jj].x <= xv[f]) ? j : idx;
              }
              pa[f] = a0[f] + 8 * idx;
              an[f] = A[idx].x;
            }
          }
        }
      }
      float y[FPT];
#pragma unroll
      for (int f = 0; f < FPT; ++f) {
        const float2 vs = lds_f32x2(pa[f] + cvs[f]);
        y[f] = fmaf(vs.y, xv[f] - an[f], vs.x);
      }
      if constexpr (SUM) {
        if (p.piece_out) {                          // (uniform) keep the pieces for the backward pass, group-major
          unsigned packed = 0u;
#pragma unroll
          for (int f = 0; f < FPT; ++f) packed |= static_cast<unsigned>((pa[f] - a0[f]) >> 3) << (8 * f);
          // (16-feature groups whatever FG is: the layout gnan_fpwl_moments_fixed reads)
          constexpr int G16 = FG / 16;
          *reinterpret_cast<unsigned*>(p.piece_out + (static_cast<int64_t>(g * G16 + q / 4) * p.n + n) * 16 + (q % 4) * FPT) = packed;
        }
        float acc = ((y[0] + y[1]) + y[2]) + y[3];          // fpwl_fast_kernel's association: bit-identical sums
#pragma unroll
        for (int o = 1; o < TPN; o <<= 1) acc += __shfl_xor(acc, o);
        if (q == 0) {
          if (split) p.part[static_cast<int64_t>(g) * p.n + n] = acc;
          else p.out[n * p.out_stride] = before + acc;       // (`before`: the earlier groups' sum, 0 for the first group)
        }
      } else {
        float4 r = make_float4(y[0], y[1], y[2], y[3]);
        if constexpr (OUT16) {
          const unsigned b0 = bf16_bits(r.x), b1 = bf16_bits(r.y), b2 = bf16_bits(r.z), b3 = bf16_bits(r.w);
          uint16_t* o16 = reinterpret_cast<uint16_t*>(p.out) + n * p.out_stride + (k0 + q * FPT);
          *reinterpret_cast<uint2*>(o16) = make_uint2(b0 | (b1 << 16), b2 | (b3 << 16));
          r = make_float4(__uint_as_float(b0 << 16), __uint_as_float(b1 << 16), __uint_as_float(b2 << 16),
                          __uint_as_float(b3 << 16));
        } else {
          *reinterpret_cast<float4*>(p.out + n * p.out_stride + k0 + q * FPT) = r;
        }
        if (n < p.total_rows) { ps[0] += r.x; ps[1] += r.y; ps[2] += r.z; ps[3] += r.w; }
      }
    };
    // U nodes per thread and round, the x rows of the NEXT round requested before this round's look-ups start: a look-up
    // is ~16 vector instructions and 3 LDS round trips, so without loads in flight across rounds a wave would spend its
    // time waiting for one 16-byte load at a time (16 waves per CU x 1 KB = 4 MB in flight chip-wide: < 3 TB/s)
    constexpr int U = 2;                     // (three rows in flight in the feature-sum mode — 120 registers — measured the same: round 5)
    // (unconditional loads from a clamped address: a guarded load would hide the number of loads in flight from the
    //  compiler, which then waits for ALL of them — the prefetch included — before the first look-up)
    const float* xq = p.x + k0 + q * FPT;
    const float* xlast = xq + (n_hi - 1) * p.x_stride;
    const int64_t xstep = static_cast<int64_t>(NODES) * p.x_stride;
    const float* xp = xq + (n_lo + nl) * p.x_stride;
    auto row = [&](const float* ptr) { return *reinterpret_cast<const float4*>(ptr <= xlast ? ptr : xlast); };
    const bool add_before = SUM && !split && g > g_lo;   // feature sum over several groups inside the workgroup: read-modify-write of the output
    // Two buffers of U rows take turns (A: even rounds, B: odd ones), each reloaded in place right after its look-ups: a
    // "current = next" hand-over at the end of the loop body made every iteration wait for the rows it had just requested.
    // The earlier groups' sums of a round's nodes are requested one half-iteration ahead, BEFORE the rows that follow, and
    // unconditionally (from the row's own x when there is nothing to add): loads retire in order, so a look-up that waits for
    // its sum waits for everything requested before it and for nothing requested after it — behind the next rows (or behind a
    // guard the compiler cannot count through) the sums made the look-ups wait for the very prefetch they should overlap with.
    if constexpr (SUM) {
      float4 bufA[U], bufB[U];
      float bvA[U], bvB[U];
      auto rows = [&](float4 (&buf)[U], const float* from) {
#pragma unroll
        for (int u = 0; u < U; ++u) buf[u] = row(from + u * xstep);
      };
      auto befores = [&](float (&bv)[U], const int64_t nn) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          bv[u] = 0.f;
          if constexpr (SUM) {                         // (per-feature rows: nothing is ever added to)
            const int64_t m = nn + u * NODES < n_hi ? nn + u * NODES : n_hi - 1;
            bv[u] = *(add_before ? p.out + m * p.out_stride : xq + m * p.x_stride);
          }
        }
      };
      auto round = [&](const int64_t nn, const float4 (&buf)[U], const float (&bv)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (nn + u * NODES < n_hi) look_up(nn + u * NODES, buf[u], add_before ? bv[u] : 0.f);
      };
      int64_t n = n_lo + nl;
      rows(bufA, xp);
      befores(bvA, n);
      rows(bufB, xp + U * xstep);
      for (; n < n_hi; n += 2 * U * NODES) {
        befores(bvB, n + U * NODES);
        round(n, bufA, bvA);
        rows(bufA, xp + 2 * U * xstep);
        befores(bvA, n + 2 * U * NODES);
        round(n + U * NODES, bufB, bvB);
        rows(bufB, xp + 3 * U * xstep);
        xp += 2 * U * xstep;
      }
    } else {
      // per-feature rows: the plain hand-over loop.  This mode sits at its copy pattern's ceiling (0.97 ms on C4) and the
      // two-buffer form measured SLOWER here (tools/lookup_ab.py: rows 1.03 -> 1.07 ms, bf16 rows 0.875 -> 0.95 — twice the
      // code per iteration, nothing to gain from more loads in flight)
      float4 cur[U], nxt[U];
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = row(xp + u * xstep);
      for (int64_t n = n_lo + nl; n < n_hi; n += U * NODES) {
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[u] = row(xp + (U + u) * xstep);
        xp += U * xstep;
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (n + u * NODES < n_hi) look_up(n + u * NODES, cur[u], 0.f);
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
      }
    }
    if constexpr (!SUM) {
      if (p.col_partial) {   // fixed-order workgroup reduction: NODES node slots per feature, float64 (as fpwl_fast_kernel)
        float* red = smem;
        __syncthreads();
#pragma unroll
        for (int f = 0; f < FPT; ++f) red[tid * FPT + f] = ps[f];
        __syncthreads();
        if (tid < FG) {
          const int qq = tid / FPT, ff = tid % FPT;
          double acc = 0.0;
          for (int s2 = 0; s2 < NODES; ++s2) acc += red[(s2 * TPN + qq) * FPT + ff];
          p.col_partial[nb * p.F + k0 + tid] = acc;
        }
      }
    }
  }
}

// out[n] = ((part[0][n] + part[1][n]) + part[2][n]) + ... : the association of the in-workgroup walk (bit-identical sums)
// `col_partial` (optional): col_partial[blockIdx.x] = the workgroup's sum of out over the rows below total_rows (float64, fixed
// tree) — the rest bucket's column sum out of this pass instead of a gnan_colsum over the result (gnan_fpwl_args.sum_total).
__global__ __launch_bounds__(256) void sum_groups_kernel(const float* __restrict__ part, int n_groups, int64_t n,
                                                         float* __restrict__ out, int64_t out_stride,
                                                         double* col_partial, int64_t total_rows, unsigned* arrive, float* total) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  float s = 0.f;
  if (i < n) {
    s = part[i];
    for (int g = 1; g < n_groups; ++g) s = s + part[static_cast<int64_t>(g) * n + i];
    out[i * out_stride] = s;
  }
  if (col_partial == nullptr) return;
  __shared__ double red[256];
  red[threadIdx.x] = i < total_rows ? static_cast<double>(s) : 0.0;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (static_cast<int>(threadIdx.x) < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) col_partial[blockIdx.x] = red[0];
  if (arrive != nullptr && gnan::last_block(arrive)) {      // (sum_total_final_kernel's sum, by the last workgroup of this pass)
    const double s_all = gnan::sum_partials_256(col_partial, static_cast<int64_t>(gridDim.x));
    if (threadIdx.x == 0) total[0] = static_cast<float>(s_all);
  }
}

__global__ __launch_bounds__(256) void sum_total_final_kernel(const double* __restrict__ partial, int64_t n_partial, float* __restrict__ total) {
  double s = 0.0;
  for (int64_t b = threadIdx.x; b < n_partial; b += 256) s += partial[b];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (static_cast<int>(threadIdx.x) < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) total[0] = static_cast<float>(red[0]);
}

int cu_count_() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  return cus;
}

struct IndexPlan {
  bool ok;
  bool split;              // feature sum with a workgroup per (node block, group) + sum_groups_kernel
  int fg, logb, bs, tot_cap, npb;
  size_t lds;
};

constexpr int64_t kSplitSumMaxNodes = 1 << 19;     // beyond, node blocks alone fill the chip and the groups share a workgroup's x lines

size_t image_bytes(int fg, int logb, int tot_cap) {          // bucket entries | anchor pairs (+ 2 NaN pairs per feature) | (val, slope) | offsets
  const size_t a_pairs = static_cast<size_t>(tot_cap) + fg * 2;
  return (static_cast<size_t>(fg) << logb) * 2 + a_pairs * 8 + static_cast<size_t>(tot_cap) * 8 + (fg + 1) * 4;
}

// Which instance serves these arguments, if any (one channel, whole feature groups, 16-byte aligned rows), and its node
// block: the LDS image (bucket entries + tables, 50-100 KB) is worth ~500 look-up rows, so blocks should be large, and
// the grid should be whole rounds of the resident workgroups.
IndexPlan index_plan(const gnan_fpwl_args* a) {
  IndexPlan pl{};
  auto aligned = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) % 16) == 0; };
  if (!a->index_table || !a->index_key || a->C != 1 || a->F % 16 != 0 || a->x_stride % 4 != 0 || !aligned(a->x)) return pl;
  const int B = a->index_buckets;
  if (B != 256 && B != 512 && B != 1024 && B != 2048) return pl;
  if (!aligned(a->index_table) || a->max_pieces > 4096) return pl;          // entries hold 4 * piece + flag in 16 bits
  if (!a->sum_features && (a->out_stride % 4 != 0 || !aligned(a->out))) return pl;
  if (a->sum_features && a->out_dtype != GNAN_F32) return pl;
  if (a->piece_out && (!a->sum_features || a->max_pieces > 256)) return pl;
  pl.logb = 6;
  while ((1 << pl.logb) < B) ++pl.logb;
  // full lines where the feature count allows and the image of a 32-feature group fits one workgroup per CU
  pl.fg = (a->F % 32 == 0 && !(a->flags & GNAN_FPWL_INDEX_HALF_LINES)) ? 32 : 16;
  pl.tot_cap = a->max_group_pieces * (pl.fg / 16);      // (max_group_pieces: the largest run of 16 consecutive features)
  pl.lds = image_bytes(pl.fg, pl.logb, pl.tot_cap);
  if (pl.fg == 32 && pl.lds > 150 * 1024) {
    pl.fg = 16;
    pl.tot_cap = a->max_group_pieces;
    pl.lds = image_bytes(16, pl.logb, pl.tot_cap);
  }
  if (pl.lds > 150 * 1024) return pl;
  // measured on the 10M x 64 matrix (tools/lookup_ab.py): per-feature rows 0.99 ms / bf16 rows 0.82 ms with 512 threads
  // against 1.01 / 0.95 with 1024; the feature sum the other way round (0.97 against 0.69: its 8-lane sums and the
  // read-modify-write of the output want the 16 waves)
  pl.bs = 512;
  if (pl.fg == 32 && a->sum_features) pl.bs = 1024;
  if (pl.fg == 32 && (a->flags & GNAN_FPWL_INDEX_BS512)) pl.bs = 512;
  if (pl.fg == 32 && (a->flags & GNAN_FPWL_INDEX_BS1024)) pl.bs = 1024;
  const int n_groups = a->F / pl.fg;
  pl.split = a->sum_features && n_groups > 1 && a->n >= 16384 && a->n < kSplitSumMaxNodes && a->sum_workspace != nullptr &&
             a->sum_workspace_bytes >= static_cast<size_t>(n_groups) * static_cast<size_t>(a->n) * sizeof(float);
  if (pl.split) {
    pl.bs = 512;
    int per_cu = static_cast<int>((160 * 1024) / pl.lds);
    if (per_cu > 2048 / pl.bs) per_cu = 2048 / pl.bs;
    if (per_cu < 1) per_cu = 1;
    const int64_t resident = static_cast<int64_t>(cu_count_()) * per_cu;
    const int unit = 64, overhead = 500;
    int64_t best_cost = -1;
    int best = 2048;
    for (int npb = 256; npb <= 16384; npb += unit) {
      const int64_t wgs = ((a->n + npb - 1) / npb + 7) / 8 * 8 * n_groups;
      const int64_t rounds = (wgs + resident - 1) / resident;
      const int64_t cost = rounds * (npb + overhead);
      if (best_cost < 0 || cost < best_cost || (cost == best_cost && npb > best)) { best_cost = cost; best = npb; }
    }
    pl.npb = best;
  } else if (a->n < 262144) {
    const int64_t npb = (a->n / 1024 + 255) / 256 * 256;
    pl.npb = static_cast<int>(npb < 256 ? 256 : (npb > 4096 ? 4096 : npb));
  } else {
    int per_cu = static_cast<int>((160 * 1024) / pl.lds);
    if (per_cu > 2048 / pl.bs) per_cu = 2048 / pl.bs;
    if (per_cu < 1) per_cu = 1;
    const int64_t resident = static_cast<int64_t>(cu_count_()) * per_cu;
    const int64_t groups = a->sum_features ? 1 : a->F / pl.fg;
    // (with one or two workgroups per CU the grid is only a handful of rounds: blocks up to 16384 nodes so that the last round
    //  can be a full one — 10M rows, two groups: 9 rounds of 8704 nodes instead of 9.5 of 8192)
    const int unit = 64, overhead = 500;
    int64_t best_cost = -1;
    int best = 4096;
    for (int npb = 2048; npb <= 16384; npb += unit) {
      const int64_t wgs = (a->n + npb - 1) / npb * groups;
      const int64_t rounds = (wgs + resident - 1) / resident;
      const int64_t cost = rounds * (npb + overhead);
      if (best_cost < 0 || cost < best_cost || (cost == best_cost && npb > best)) { best_cost = cost; best = npb; }
    }
    pl.npb = best;
  }
  pl.ok = true;
  return pl;
}

}  // namespace

// (called by gnan_fpwl_fwd / gnan_fpwl_total_workspace_bytes, csrc/fpwl.hip)
bool gnan_index_applies(const gnan_fpwl_args* a) { return index_plan(a).ok; }
int gnan_index_nodes_per_block(const gnan_fpwl_args* a) { return index_plan(a).npb; }

// col_partial: the column-sum partials ([node blocks][F] doubles) or NULL
int gnan_index_fwd(const gnan_fpwl_args* a, double* col_partial, hipStream_t st) {
  const IndexPlan pl = index_plan(a);
  if (!pl.ok) return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl_index: arguments outside what the direct-index kernel serves");
  // everything that can be refused is refused BEFORE the first launch (a caller that retries another kernel on an error code —
  // or a capture — must not find a look-up already queued)
  if (a->sum_total) {
    if (!pl.split)
      return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl_index: sum_total is written by the group-split feature sum only (sum_workspace)");
    const size_t need = static_cast<size_t>((a->n + 255) / 256) * sizeof(double);
    if (a->sum_total_workspace == nullptr || a->sum_total_workspace_bytes < need || reinterpret_cast<uintptr_t>(a->sum_total_workspace) % 8 != 0)
      return gnan::fail(GNAN_ERR_WORKSPACE, "fpwl_index: sum_total workspace %zu B < required %zu B (8-byte aligned)",
                        a->sum_total_workspace_bytes, need);
  }
  IndexParams p;
  p.x = a->x; p.n = a->n; p.x_stride = a->x_stride; p.F = a->F;
  p.off = a->off; p.anchor = a->anchor; p.val = a->val; p.slope = a->slope;
  p.table = a->index_table; p.key = a->index_key;
  p.n_groups = a->F / pl.fg;
  p.nodes_per_block = pl.npb;
  p.tot_cap = pl.tot_cap;
  int step0 = 0;
  while ((step0 ? step0 * 2 : 1) <= a->max_pieces - 1) step0 = step0 ? step0 * 2 : 1;
  p.step0 = step0;
  p.sum_features = a->sum_features;
  p.out = static_cast<float*>(a->out); p.out_stride = a->out_stride;
  p.col_partial = col_partial;
  p.total_rows = (a->total_rows > 0 && a->total_rows < a->n) ? a->total_rows : a->n;
  p.piece_out = a->piece_out;
  p.part = pl.split ? static_cast<float*>(a->sum_workspace) : nullptr;
  size_t lds = pl.lds;
  if (col_partial && lds < static_cast<size_t>(pl.bs) * 4 * sizeof(float)) lds = static_cast<size_t>(pl.bs) * 4 * sizeof(float);
  const int64_t bx = (p.n + p.nodes_per_block - 1) / p.nodes_per_block;
  const int64_t wgs = (a->sum_features && !pl.split) ? bx : (bx + 7) / 8 * 8 * p.n_groups;
  if (wgs > 0x7fffffffLL) return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl: too many nodes for one launch");
  auto go = [&](auto kernel) {
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(lds));
      if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "fpwl_index: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(wgs)), dim3(pl.bs), lds, st, p);
    return gnan::check_launch("fpwl_index_kernel");
  };
  auto by_mode = [&](auto fg_c, auto bs_c, auto lb_c) {
    constexpr int G = decltype(fg_c)::value, S = decltype(bs_c)::value, LB = decltype(lb_c)::value;
    if (a->out_dtype == GNAN_BF16) return go(fpwl_index_kernel<G, false, true, LB, S>);
    return a->sum_features ? go(fpwl_index_kernel<G, true, false, LB, S>) : go(fpwl_index_kernel<G, false, false, LB, S>);
  };
  auto by_kf = [&](auto fg_c, auto bs_c) {
    switch (pl.logb) {
      case 8: return by_mode(fg_c, bs_c, std::integral_constant<int, 8>{});
      case 9: return by_mode(fg_c, bs_c, std::integral_constant<int, 9>{});
      case 10: return by_mode(fg_c, bs_c, std::integral_constant<int, 10>{});
      default: return by_mode(fg_c, bs_c, std::integral_constant<int, 11>{});
    }
  };
  using I16 = std::integral_constant<int, 16>;
  using I32 = std::integral_constant<int, 32>;
  using S512 = std::integral_constant<int, 512>;
  using S1024 = std::integral_constant<int, 1024>;
  int rc;
  if (pl.fg == 32) rc = pl.bs == 1024 ? by_kf(I32{}, S1024{}) : by_kf(I32{}, S512{});
  else rc = by_kf(I16{}, S512{});
  if (rc != GNAN_OK) return rc;
  if (!pl.split) return rc;
  const int64_t sb = (a->n + 255) / 256;
  double* tot_partial = a->sum_total ? static_cast<double*>(a->sum_total_workspace) : nullptr;      // (checked above)
  unsigned* arrive = (tot_partial && sb <= gnan::kMaxArriveBlocks) ? reinterpret_cast<unsigned*>(a->sum_total_arrive) : nullptr;
  hipLaunchKernelGGL(sum_groups_kernel, dim3(static_cast<unsigned>(sb)), dim3(256), 0, st, p.part, p.n_groups, p.n,
                     p.out, p.out_stride, tot_partial, p.total_rows, arrive, a->sum_total);
  if (int rc2 = gnan::check_launch("sum_groups_kernel")) return rc2;
  if (tot_partial && arrive == nullptr) {
    hipLaunchKernelGGL(sum_total_final_kernel, dim3(1), dim3(256), 0, st, tot_partial, sb, a->sum_total);
    return gnan::check_launch("sum_total_final_kernel");
  }
  return GNAN_OK;
}

// bytes of gnan_fpwl_args.sum_workspace this call would use (0: none): what index_plan's split needs
size_t gnan_index_sum_workspace_bytes(const gnan_fpwl_args* a) {
  if (!a || !a->sum_features || a->n < 16384 || a->n >= kSplitSumMaxNodes) return 0;
  gnan_fpwl_args probe = *a;
  probe.sum_workspace = nullptr;
  probe.sum_workspace_bytes = 0;
  const IndexPlan pl = index_plan(&probe);
  if (!pl.ok) return 0;
  const int n_groups = a->F / pl.fg;
  return n_groups > 1 ? static_cast<size_t>(n_groups) * static_cast<size_t>(a->n) * sizeof(float) : 0;
}

extern "C" int gnan_feature_range(const float* x, int64_t n, int64_t x_stride, int32_t F, float* range, void* workspace,
                                  size_t workspace_bytes, gnan_stream_t stream) {
  GNAN_REQUIRE(n >= 0 && F >= 1, "feature_range: bad sizes");
  GNAN_REQUIRE(range && workspace && (n == 0 || x), "feature_range: null pointer");
  GNAN_REQUIRE(x_stride >= F, "feature_range: x row stride smaller than F");
  if (workspace_bytes < static_cast<size_t>(F) * 2 * sizeof(int))
    return gnan::fail(GNAN_ERR_WORKSPACE, "feature_range: workspace %zu B < required %zu B", workspace_bytes,
                      static_cast<size_t>(F) * 2 * sizeof(int));
  hipStream_t st = static_cast<hipStream_t>(stream);
  int* lohi = static_cast<int*>(workspace);
  hipLaunchKernelGGL(range_init_kernel, dim3((F + 255) / 256), dim3(256), 0, st, lohi, F);
  if (n > 0) {
    const int col_blocks = (F + 255) / 256;
    int64_t row_blocks = 4096 / col_blocks;
    if (row_blocks > n) row_blocks = n;
    if (row_blocks < 1) row_blocks = 1;
    const int64_t rpb = (n + row_blocks - 1) / row_blocks;
    row_blocks = (n + rpb - 1) / rpb;
    hipLaunchKernelGGL(range_kernel, dim3(static_cast<unsigned>(row_blocks), col_blocks), dim3(256), 0, st, x, n, x_stride, F,
                       rpb, lohi);
  }
  hipLaunchKernelGGL(range_finish_kernel, dim3((2 * F + 255) / 256), dim3(256), 0, st, lohi, F, range);
  return gnan::check_launch("range_kernel");
}

extern "C" int gnan_fpwl_index_build(const gnan_fpwl_index_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "fpwl_index_build: null args");
  GNAN_REQUIRE(a->F >= 1 && a->off && a->anchor && a->range && a->table && a->key, "fpwl_index_build: null pointer / bad sizes");
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto go = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(a->F), dim3(256), 0, st, a->off, a->anchor, a->range, a->table, a->key, a->stats);
    return gnan::check_launch("index_build_kernel");
  };
  switch (a->buckets) {
    case 256: return go(index_build_kernel<8>);
    case 512: return go(index_build_kernel<9>);
    case 1024: return go(index_build_kernel<10>);
    case 2048: return go(index_build_kernel<11>);
    default: return gnan::fail(GNAN_ERR_BAD_ARG, "fpwl_index_build: buckets must be 256, 512, 1024 or 2048");
  }
}
