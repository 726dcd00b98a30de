// Shape functions by exact piecewise-linear table look-up (gfx950).
//
// f_k is a ReLU MLP of a scalar, i.e. exactly piecewise linear; gnan_amd/pwl.py tabulates it per forward
// (anchor / value / slope per piece, ~130 pieces for H = 64, L = 3).  This kernel evaluates
//     f_k(x) = val[i] + slope[i] * (x - anchor[i]),   i = #{ breakpoints of f_k <= x }
// for every (node, feature): it replaces the F x L addmm/relu launches of GNAN.py:57-62 by N*F binary
// searches in LDS.  HBM-bound (x in, fx out), no matrix work at all.
//
// Mapping: a workgroup owns a contiguous block of nodes and a group of <= FG consecutive features whose
// tables it copies to LDS once (amortised over the node block).  Thread = (node, 4 features): one 16-B load,
// 4 searches in lock-step (independent LDS reads per step hide the LDS latency), one 16-B store; the 4 threads
// of a node cover one 64-B sector of x and of fx.
#include "common.hpp"

#include <cstdlib>
#include <type_traits>

namespace {

struct Params {
  const float* x;
  int64_t n, x_stride;
  int F, C;
  const int32_t* off;
  const float* anchor;
  const float* val;
  const float* slope;
  int step0;        // largest power of two <= max breakpoints per feature (0 if none)
  int n_groups;
  int nodes_per_block;
  int sum_features;
  int vec_x, vec_out;
  float* out;
  int64_t out_stride;
  double* col_partial;   // optional [gridDim.x, F]: per-workgroup column sums of the output (FAST, per-feature mode)
  int out_bf16;          // per-feature output stored as bf16 rows (FAST path only)
  uint8_t* piece_out;    // optional [n, F] (fast feature-sum kernel): the piece of every (node, feature) within its feature,
                         // kept for gnan_fpwl_moments_fixed(piece_in) — the backward pass then skips the search
  const uint8_t* piece_in;
  int64_t total_rows;    // column sums cover nodes [0, total_rows) only
  int max_pieces;        // largest piece count of one feature
  int max_group_pieces;  // largest piece count of one feature group
  int piece_stride;      // fpwl_moments_c1_kernel over kept pieces: bins of piece j of the group's feature f at [j * FG + f] (even, >= max_pieces)
  int soff_offset;       // fpwl_fast_kernel: float offset of its group-offset array in dynamic LDS
  int acc_offset;        // > 0 (sum over features, C > 1): float offset in dynamic LDS of the [C][NODES] accumulators;
                         // the workgroup then owns ONE pass of nodes and walks all feature groups for them
};

// LDS row strides of the per-piece tables: an odd number of words per piece for C > 1.  The threads of a wavefront
// read channel c of DIFFERENT pieces (thread = node), i.e. addresses `piece * stride + c`: with stride = C = 40 those
// fall on 4 of the 32 banks (16-way conflicts; C = 32: one bank), and the 64-bit bins of the moment kernels, stride 2 C
// quad-words, on a single bank pair.  arxiv-shaped C = 40: look-up 2.33 -> 2.30 ms, moments 4.5 -> 4.0 ms (the structure, not the conflicts, was the bound: fpwl_rows.hip).
__host__ __device__ __forceinline__ int table_stride(int C) { return C > 1 ? (C | 1) : 1; }          // floats per (val | slope) row
__host__ __device__ __forceinline__ int bin_stride(int C) { return C > 1 ? 2 * C + 1 : 2; }          // bins per piece: [2][C] (+ 1)

// Thread = (node, feature quad): FPT = min(FG, 4) features per thread, TPN = FG / FPT threads per node, so a
// node's FG x values are one contiguous 16-B load per thread (a 64-B sector per node for FG = 16), the
// per-thread state is a handful of registers (8 waves/SIMD), and every thread runs FPT independent searches.
template <int FG, int BS = 256>
struct Map {
  static constexpr int FPT = FG < 4 ? FG : 4;
  static constexpr int TPN = FG / FPT;
  static constexpr int NODES = BS / TPN;  // nodes per workgroup pass
};

// Piece of feature f (group-relative, LDS tables): i = #{ j in 1..pn : anchor[po + j] <= x }, searched for
// the thread's FPT features in lock-step (FPT independent LDS reads per step).
template <int FPT>
__device__ __forceinline__ void search(const float* anchor_l, const int (&po)[FPT], const int (&pn)[FPT],
                                       const float (&xv)[FPT], int step0, int (&idx)[FPT]) {
#pragma unroll
  for (int f = 0; f < FPT; ++f) idx[f] = 0;
  for (int step = step0; step > 0; step >>= 1) {
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
      const int j = idx[f] + step;
      const int jj = j <= pn[f] ? j : 0;  // out of range -> harmless in-range read
      const float a = anchor_l[po[f] + jj];
      idx[f] = (j <= pn[f] && a <= xv[f]) ? j : idx[f];
    }
  }
#pragma unroll
  for (int f = 0; f < FPT; ++f) idx[f] += po[f];
}

__device__ __forceinline__ unsigned bf16_bits(float f) {      // round-to-nearest-even, as torch.bfloat16
  const unsigned u = __float_as_uint(f);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// SUM: out[n, c] = sum over features (f_sums, GNAN.py:157); FAST: C == 1, full groups, 16-B aligned rows;
// OUT16 (FAST, per-feature output only): store bf16 rows — the operand format of the bf16-storage aggregation.
// BS threads per workgroup: the LDS image of the tables is shared by BS / 64 waves, so a large workgroup is what buys
// occupancy here (25 KB of tables per 4 waves would cap a CU at 20 waves; per 8 waves it reaches the full 32).
template <int FG, bool SUM, bool FAST, bool OUT16 = false, int BS = 256>
__global__ __launch_bounds__(BS) void fpwl_kernel(const Params p) {
  constexpr int FPT = Map<FG, BS>::FPT, TPN = Map<FG, BS>::TPN, NODES = Map<FG, BS>::NODES;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_off[FG + 1];
  const int tid = threadIdx.x;
  const int q = tid % TPN;            // which feature quad of the group
  const int nl = tid / TPN;           // node slot inside a pass
  const int C = p.C;
  // Per-feature mode: the n_groups workgroups that share a node block read different 64-B sectors of the same x rows.
  // Linear workgroup ids go round-robin over the 8 XCDs, so (id % 8) picks the XCD and, inside it, the groups of one
  // node block are adjacent in time: the rows' 128-B lines are fetched into that XCD's L2 once.
  int64_t nb = blockIdx.x;
  int g_first = 0;
  if (!SUM) {
    const int64_t id = blockIdx.x;
    g_first = static_cast<int>((id >> 3) % p.n_groups);
    nb = ((id >> 3) / p.n_groups) * 8 + (id & 7);
  }
  const int64_t n_lo = nb * p.nodes_per_block;
  if (n_lo >= p.n) return;
  const int64_t n_hi = n_lo + p.nodes_per_block < p.n ? n_lo + p.nodes_per_block : p.n;
  const int g_lo = SUM ? 0 : g_first;
  const int g_hi = SUM ? p.n_groups : g_lo + 1;
  float* acc_l = smem + p.acc_offset;
  if (SUM && p.acc_offset) {
    for (int i = tid; i < C * NODES; i += BS) acc_l[i] = 0.f;
  }

  for (int g = g_lo; g < g_hi; ++g) {
    const int k0 = g * FG;
    const int nf = FAST ? FG : (p.F - k0 < FG ? p.F - k0 : FG);
    const int base = p.off[k0];
    const int tot = p.off[k0 + nf] - base;
    const int Cs = table_stride(C);
    float* anchor_l = smem;
    float* val_l = smem + tot;
    float* slope_l = val_l + static_cast<int64_t>(tot) * Cs;
    __syncthreads();  // previous group's searches are done with the LDS tables
    for (int i = tid; i < tot; i += BS) anchor_l[i] = p.anchor[base + i];
    if (Cs == C) {
      for (int i = tid; i < tot * C; i += BS) {
        val_l[i] = p.val[static_cast<int64_t>(base) * C + i];
        slope_l[i] = p.slope[static_cast<int64_t>(base) * C + i];
      }
    } else {
      for (int i = tid; i < tot * C; i += BS) {
        const int r = i / C, j = r * Cs + (i - r * C);
        val_l[j] = p.val[static_cast<int64_t>(base) * C + i];
        slope_l[j] = p.slope[static_cast<int64_t>(base) * C + i];
      }
    }
    if (tid <= nf) s_off[tid] = p.off[k0 + tid] - base;
    __syncthreads();
    int po[FPT], pn[FPT];
    bool live[FPT];
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
      const int fg = q * FPT + f;
      live[f] = fg < nf;
      po[f] = live[f] ? s_off[fg] : 0;
      pn[f] = live[f] ? s_off[fg + 1] - s_off[fg] - 1 : 0;
    }

    float ps[FPT];  // this thread's share of the column sums of fx (rest-bucket total of the aggregation)
#pragma unroll
    for (int f = 0; f < FPT; ++f) ps[f] = 0.f;
    for (int64_t n = n_lo + nl; n < n_hi; n += NODES) {
      float xv[FPT];
      const float* xr = p.x + n * p.x_stride + k0 + q * FPT;
      if constexpr (FAST && FPT == 4) {
        const float4 t = *reinterpret_cast<const float4*>(xr);
        xv[0] = t.x; xv[1 % FPT] = t.y; xv[2 % FPT] = t.z; xv[3 % FPT] = t.w;
      } else {
#pragma unroll
        for (int f = 0; f < FPT; ++f) xv[f] = live[f] ? xr[f] : 0.f;
      }
      int idx[FPT];
      search<FPT>(anchor_l, po, pn, xv, p.step0, idx);
      float d[FPT];
#pragma unroll
      for (int f = 0; f < FPT; ++f) d[f] = xv[f] - anchor_l[idx[f]];

      if constexpr (SUM) {
        float* o = p.out + n * p.out_stride;
        // channels in chunks of CU: the run-time loop over C is not unrolled by the compiler, and one channel per iteration
        // is a chain of four dependent LDS round trips (table reads, accumulator read, write) — at one workgroup per CU
        // (C = 40: 129 KB of LDS) that chain, not bandwidth, set the kernel's time (arxiv-shaped C = 40: 2.3 ms)
        constexpr int CU = FAST ? 1 : 8;
        for (int c0 = 0; c0 < (FAST ? 1 : C); c0 += CU) {
          float a[CU];
#pragma unroll
          for (int u = 0; u < CU; ++u) {
            const int c = c0 + u < C ? c0 + u : C - 1;       // tail: an in-range read whose result is dropped
            a[u] = 0.f;
#pragma unroll
            for (int f = 0; f < FPT; ++f)
              if (live[f]) a[u] += fmaf(slope_l[idx[f] * Cs + c], d[f], val_l[idx[f] * Cs + c]);
          }
#pragma unroll
          for (int off = 1; off < TPN; off <<= 1)            // the node's TPN threads
#pragma unroll
            for (int u = 0; u < CU; ++u) a[u] += __shfl_xor(a[u], off);
          // groups run one after the other inside the workgroup and a node keeps its thread: no race
          if (q == 0) {
            if (p.acc_offset) {                              // many channels: accumulate on chip, store once
              float old[CU];
#pragma unroll
              for (int u = 0; u < CU; ++u) old[u] = acc_l[(c0 + u < C ? c0 + u : C - 1) * NODES + nl];
#pragma unroll
              for (int u = 0; u < CU; ++u)
                if (c0 + u < C) acc_l[(c0 + u) * NODES + nl] = old[u] + a[u];
            } else {
#pragma unroll
              for (int u = 0; u < CU; ++u)
                if (c0 + u < C) o[c0 + u] = g == 0 ? a[u] : o[c0 + u] + a[u];
            }
          }
        }
      } else {
        float* o = p.out + n * p.out_stride + static_cast<int64_t>(k0 + q * FPT) * C;
        if constexpr (FAST && FPT == 4) {
          float4 t;
          t.x = fmaf(slope_l[idx[0]], d[0], val_l[idx[0]]);
          t.y = fmaf(slope_l[idx[1 % FPT]], d[1 % FPT], val_l[idx[1 % FPT]]);
          t.z = fmaf(slope_l[idx[2 % FPT]], d[2 % FPT], val_l[idx[2 % FPT]]);
          t.w = fmaf(slope_l[idx[3 % FPT]], d[3 % FPT], val_l[idx[3 % FPT]]);
          if constexpr (OUT16) {
            const unsigned b0 = bf16_bits(t.x), b1 = bf16_bits(t.y), b2 = bf16_bits(t.z), b3 = bf16_bits(t.w);
            uint16_t* o16 = reinterpret_cast<uint16_t*>(p.out) + n * p.out_stride + (k0 + q * FPT);
            *reinterpret_cast<uint2*>(o16) = make_uint2(b0 | (b1 << 16), b2 | (b3 << 16));
            // the column sums must describe the operand the aggregation will actually read: the rounded values
            t = make_float4(__uint_as_float(b0 << 16), __uint_as_float(b1 << 16), __uint_as_float(b2 << 16),
                            __uint_as_float(b3 << 16));
          } else {
            *reinterpret_cast<float4*>(o) = t;
          }
          if (n < p.total_rows) { ps[0] += t.x; ps[1 % FPT] += t.y; ps[2 % FPT] += t.z; ps[3 % FPT] += t.w; }
        } else {
#pragma unroll
          for (int f = 0; f < FPT; ++f)
            if (live[f])
              for (int c = 0; c < C; ++c) o[f * C + c] = fmaf(slope_l[idx[f] * Cs + c], d[f], val_l[idx[f] * Cs + c]);
        }
      }
    }
    if constexpr (FAST && !SUM && FPT == 4) {
      if (p.col_partial) {   // fixed-order workgroup reduction: 64 node slots per feature, float64
        float* red = smem + tot * 3;               // FAST: C == 1, tables take 3 floats per piece
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
  if (SUM && p.acc_offset) {
    __syncthreads();
    const int64_t n = n_lo + nl;
    if (n < n_hi)
      for (int c = q; c < C; c += TPN) p.out[n * p.out_stride + c] = acc_l[c * NODES + nl];
  }
}

// ---------------------------------------------------------------------------------------------
// The C == 1 work-horse.  Two things bound the plain kernel above: vector instructions (68 per look-up, counted with
// SQ_INSTS_VALU) and LDS bank conflicts of the binary search (level s of a search over a sorted array probes 2^s
// addresses that are 2^(NSTEP-s) words apart: with 32 banks for ds_read_b32 that is one or two banks for every level
// but the last two, i.e. up to 8-way conflicts).  This variant
//   * keeps every feature's anchors in LDS as a complete search TREE in breadth-first (Eytzinger) order, padded with
//     +inf to P2 = 2^NSTEP - 1 nodes: the 2^s nodes of level s are 2^s consecutive words, i.e. 2^s distinct banks
//     up to level 5 and at most 2^(s-5) addresses per bank below, and there is no bounds test in the loop;
//   * unrolls the loop completely and carries the node as an LDS *byte address* a = Q + 4k (Q: tree of the thread's
//     first feature; the other three trees are reached through the immediate offset of the ds_read), so a step is
//       t = lds[a];  a = 2a + (t <= x ? 4 - Q : -Q)            (compare, select, shift-add: 3 vector instructions)
//   * after NSTEP steps k - P2 is the piece: its anchor sits in a compact array at a + dA[f] and (val, slope) is one
//     8-byte read at 2 (a + dA[f]) + const.
// Same piece and same arithmetic per look-up as fpwl_kernel (val + slope * (x - anchor)), hence bit-identical results.
// ---------------------------------------------------------------------------------------------
// reads at raw LDS byte addresses (see fpwl_fast_kernel)
typedef __attribute__((address_space(3))) const float lds_cfloat;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const f32x2_t lds_cfloat2;
__device__ __forceinline__ float lds_f32(int addr) { return *reinterpret_cast<lds_cfloat*>(static_cast<uintptr_t>(static_cast<unsigned>(addr))); }
__device__ __forceinline__ float2 lds_f32x2(int addr) {
  const f32x2_t v = *reinterpret_cast<lds_cfloat2*>(static_cast<uintptr_t>(static_cast<unsigned>(addr)));
  return make_float2(v.x, v.y);
}

// Sorted index of node k (1 <= k < 2^NSTEP) of the complete breadth-first tree over the sorted entries 1 .. 2^NSTEP - 1.
template <int NSTEP>
__device__ __forceinline__ int tree_sorted_index(int k) {
  const int l = 31 - __clz(k);                       // level of the node, root = 0
  return (2 * (k - (1 << l)) + 1) << (NSTEP - 1 - l);
}

// Words of skew between the trees of consecutive feature QUADS.  One ds_read of the search serves the four quads of a
// wave (lane = (node, quad)), i.e. four different trees; were the trees aligned alike, their level-s nodes would share
// banks and every level would be a 4-way conflict (SQ_LDS_BANK_CONFLICT: 68 % of the kernel's LDS cycles).  With 8 words
// of skew the nodes of levels 0-3 of the four trees fall on 32 distinct banks and level 4 on each bank twice.
constexpr int kTreeSkew = 8;

// LDS image of one feature group: [FG][P2] tree words (+ kTreeSkew words per quad), [tot] (val, slope) pairs, [tot] anchors
// in piece order.
template <int FG, int NSTEP, int BS>
__device__ __forceinline__ void load_tree_tables(const Params& p, float* smem, const int* s_off, int base, int tot,
                                                 bool with_vs) {
  constexpr int P2 = 1 << NSTEP;
  const int tid = threadIdx.x;
  for (int i = tid; i < FG * P2; i += BS) {
    const int f = i >> NSTEP, k = i & (P2 - 1);
    float v = __builtin_nanf("");          // padding: NaN <= x is false for EVERY x (+inf <= +inf would be true)
    if (k) {
      const int j = tree_sorted_index<NSTEP>(k);
      if (j < s_off[f + 1] - s_off[f]) v = p.anchor[base + s_off[f] + j];
    }
    smem[i + (f / 4) * kTreeSkew] = v;
  }
  constexpr int kTreeWords = FG * P2 + (FG / 4) * kTreeSkew;
  float2* vs_l = reinterpret_cast<float2*>(smem + kTreeWords);
  float* an_l = smem + kTreeWords + 2 * tot;
  for (int i = tid; i < tot; i += BS) {
    if (with_vs) vs_l[i] = make_float2(p.val[base + i], p.slope[base + i]);
    an_l[i] = p.anchor[base + i];
  }
}

// RAGGED (feature-sum mode only): F need not be a multiple of FG and the rows of x need not be 16-byte aligned — the
// reference's inputs are raw features + a ones column (129, 1434: pre_process_datasets.py:108), which sent every real
// feature matrix below the padding threshold to the general kernel above (arxiv-shaped: 0.109 ms against ~0.05 ms).
// The last group is partial: its missing features get empty (+inf) trees, their x is not read and their terms are dropped.
template <int FG, bool SUM, bool OUT16, int NSTEP, int BS, bool RAGGED = false>
__global__ __launch_bounds__(BS) void fpwl_fast_kernel(const Params p) {
  static_assert(FG % 4 == 0, "feature quads");
  static_assert(!RAGGED || (SUM && !OUT16), "ragged feature counts: feature-sum mode");
  constexpr int FPT = 4, TPN = FG / FPT, NODES = BS / TPN, P2 = 1 << NSTEP;
  // addresses are raw LDS byte addresses (LDS base folded in once per group): addressing through the `smem` symbol
  // costs a fourth instruction per step, because its base is only resolved at link time
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* s_off = reinterpret_cast<int*>(smem) + p.soff_offset;       // [FG + 1], behind the tables
  const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) float*)smem));
  const int tid = threadIdx.x;
  const int q = tid % TPN;
  const int nl = tid / TPN;
  int64_t nb = blockIdx.x;
  int g_first = 0;
  if (!SUM) {                                       // (id % 8) = XCD, groups of a node block adjacent inside it
    const int64_t id = blockIdx.x;
    g_first = static_cast<int>((id >> 3) % p.n_groups);
    nb = ((id >> 3) / p.n_groups) * 8 + (id & 7);
  }
  const int64_t n_lo = nb * p.nodes_per_block;
  if (n_lo >= p.n) return;
  const int64_t n_hi = n_lo + p.nodes_per_block < p.n ? n_lo + p.nodes_per_block : p.n;
  const int g_lo = SUM ? 0 : g_first;
  const int g_hi = SUM ? p.n_groups : g_lo + 1;
  constexpr int kTreeBytes = (FG * P2 + TPN * kTreeSkew) * 4;   // (val, slope) pairs start behind the trees, the anchors behind them

  for (int g = g_lo; g < g_hi; ++g) {
    const int k0 = g * FG;
    const int nf = RAGGED ? (p.F - k0 < FG ? p.F - k0 : FG) : FG;     // live features of the group
    const int base = p.off[k0];
    const int tot = p.off[k0 + nf] - base;
    __syncthreads();                                // previous group's look-ups are done with the LDS tables
    if (tid <= FG) s_off[tid] = p.off[k0 + (tid < nf ? tid : nf)] - base;
    __syncthreads();
    load_tree_tables<FG, NSTEP, BS>(p, smem, s_off, base, tot, true);
    __syncthreads();
    const int Q = static_cast<int>(lds_base) + q * ((FPT * P2 + kTreeSkew) * 4);   // tree of the thread's first feature
    int nQ = -Q, nQ4 = 4 - Q;
    asm volatile("" : "+v"(nQ), "+v"(nQ4));         // two opaque registers: keeps the step at compare, select, shift-add
    const int vs_minus_2an = static_cast<int>(lds_base) + kTreeBytes - 2 * (static_cast<int>(lds_base) + kTreeBytes + 8 * tot);
    int dA[FPT];
#pragma unroll
    for (int f = 0; f < FPT; ++f)
      dA[f] = static_cast<int>(lds_base) + kTreeBytes + 8 * tot + 4 * s_off[q * FPT + f] - 4 * P2 - Q;

    float ps[FPT] = {0.f, 0.f, 0.f, 0.f};           // column sums of the output (per-feature mode)
    bool live[FPT];
#pragma unroll
    for (int f = 0; f < FPT; ++f) live[f] = !RAGGED || q * FPT + f < nf;
    for (int64_t n = n_lo + nl; n < n_hi; n += NODES) {
      float xv[FPT];
      if constexpr (RAGGED) {
        const float* xr = p.x + n * p.x_stride + k0 + q * FPT;
#pragma unroll
        for (int f = 0; f < FPT; ++f) xv[f] = live[f] ? xr[f] : 0.f;
      } else {
        const float4 t = *reinterpret_cast<const float4*>(p.x + n * p.x_stride + k0 + q * FPT);
        xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
      }
      int a[FPT];
#pragma unroll
      for (int f = 0; f < FPT; ++f) a[f] = Q + 4;   // node 1 = root
#pragma unroll
      for (int step = 0; step < NSTEP; ++step) {
#pragma unroll
        for (int f = 0; f < FPT; ++f) {
          const float e = lds_f32(a[f] + f * (P2 * 4));
          a[f] = (a[f] << 1) + (e <= xv[f] ? nQ4 : nQ);
        }
      }
      float y[FPT];
#pragma unroll
      for (int f = 0; f < FPT; ++f) {
        const int pa = a[f] + dA[f];
        const float an = lds_f32(pa);
        const float2 vs = lds_f32x2(2 * pa + vs_minus_2an);
        y[f] = fmaf(vs.y, xv[f] - an, vs.x);
      }
      if constexpr (SUM) {
        if (p.piece_out) {                          // (uniform) keep the pieces for the backward pass: one byte per look-up
          // group-major [n_groups][n][FG]: the pass over a feature group writes one contiguous FG-byte run per node (node-
          // major [n, F] rows got a quarter of every line per pass: the stores doubled the kernel's time)
          unsigned packed = 0u;
#pragma unroll
          for (int f = 0; f < FPT; ++f) packed |= static_cast<unsigned>(((a[f] - Q) >> 2) - P2) << (8 * f);
          *reinterpret_cast<unsigned*>(p.piece_out + (static_cast<int64_t>(g) * p.n + n) * FG + q * FPT) = packed;
        }
        float acc = 0.f;
#pragma unroll
        for (int f = 0; f < FPT; ++f) acc += (!RAGGED || live[f]) ? y[f] : 0.f;
#pragma unroll
        for (int off = 1; off < TPN; off <<= 1) acc += __shfl_xor(acc, off);   // the node's TPN threads
        if (q == 0) {                               // groups run one after the other and a node keeps its thread
          float* o = p.out + n * p.out_stride;
          o[0] = g == 0 ? acc : o[0] + acc;
        }
      } else {
        float4 r = make_float4(y[0], y[1], y[2], y[3]);
        if constexpr (OUT16) {
          const unsigned b0 = bf16_bits(r.x), b1 = bf16_bits(r.y), b2 = bf16_bits(r.z), b3 = bf16_bits(r.w);
          uint16_t* o16 = reinterpret_cast<uint16_t*>(p.out) + n * p.out_stride + (k0 + q * FPT);
          *reinterpret_cast<uint2*>(o16) = make_uint2(b0 | (b1 << 16), b2 | (b3 << 16));
          // the column sums must describe the operand the aggregation will actually read: the rounded values
          r = make_float4(__uint_as_float(b0 << 16), __uint_as_float(b1 << 16), __uint_as_float(b2 << 16),
                          __uint_as_float(b3 << 16));
        } else {
          *reinterpret_cast<float4*>(p.out + n * p.out_stride + k0 + q * FPT) = r;
        }
        if (n < p.total_rows) { ps[0] += r.x; ps[1] += r.y; ps[2] += r.z; ps[3] += r.w; }
      }
    }
    if constexpr (!SUM) {
      if (p.col_partial) {   // fixed-order workgroup reduction: NODES node slots per feature, float64
        float* red = smem;   // the tables are dead by now: the scratch takes their place (the host sizes LDS for both)
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

// One workgroup per column: 256 threads stride over the per-workgroup partials, then a fixed-order tree.
__global__ __launch_bounds__(256) void fpwl_total_kernel(const double* __restrict__ partial, int blocks, int W,
                                                         float* __restrict__ total) {
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
  if (threadIdx.x == 0) total[w] = static_cast<float>(red[0]);
}

// ---------------------------------------------------------------------------------------------
// backward: per-piece moments of the upstream gradient
//   M[t, 0, c] = sum_{(n,k) in piece t} g[n,k,c]            M[t, 1, c] = sum g[n,k,c] * (x[n,k] - anchor[t])
// On a piece f_k is affine in x AND so is d f_k(x) / d theta (fixed activation pattern), hence the parameter
// gradient  sum_n g_n * df_k(x_n)/dtheta  depends on the nodes of a piece only through these two moments;
// gnan_amd/pwl.py turns them into exact parameter gradients by back-propagating through the tiny MLP at
// two points per piece.  Same traversal as the forward; bins live in LDS (ds_add_f32) and are flushed with
// one global atomic per bin per workgroup.
// ---------------------------------------------------------------------------------------------
int tuned_moment_block(int64_t n, int n_groups, size_t lds, int bs);

struct MomentParams {
  Params f;            // x, tables, grouping as in the forward (val / slope / out unused)
  const float* g;      // upstream gradient: [n, F*C] (per feature) or [n, C] (sum_features)
  int64_t g_stride;
  float* M;            // [T, 2, C], zeroed by the caller
  // fixed-point mode (FIXED): LDS float atomics run at ~200 G/s on gfx950 (a compare-and-swap loop), integer ones at
  // ~2500 G/s (tools/lds_atomic_rate.hip), so the bins hold round(v * 2^e) in 64-bit integers: 12x cheaper to update,
  // and — integer addition being associative — bit-reproducible.  scales = {2^e0, 2^e1} for M0 / M1 terms.
  const double* scales;
  unsigned long long* Mi;   // [T, 2, C] two's-complement sums, zeroed by the caller
  int vec_g;                // gradient rows may be read as 16-byte quads (C == 1, per-feature gradient)
  int general_only;         // GNAN_FPWL_MOMENTS_GENERAL: keep the general kernel where the one-channel kernel applies
};

// round(v * s) as a two's-complement 64-bit integer for a power of two s and |v * s| < 2^51: one fused multiply-add onto
// 1.5 * 2^52 leaves the integer in the low mantissa bits (round to nearest even, like __double2ll_rn, whose library
// routine costs ~10 float64 instructions — v_rndne, v_ldexp, v_floor, v_fma, two conversions).  gnan_fpwl_moment_scales
// caps its exponents so that every term stays below 2^50.
__device__ __forceinline__ unsigned long long fixed_bits(float v, double s) {
  const double d = fma(static_cast<double>(v), s, 6755399441055744.0);
  return static_cast<unsigned long long>(__double_as_longlong(d)) - 0x4338000000000000ull;
}

__device__ __forceinline__ unsigned long long fixed_bits_d(double v, double s) {
  const double d = fma(v, s, 6755399441055744.0);
  return static_cast<unsigned long long>(__double_as_longlong(d)) - 0x4338000000000000ull;
}

template <int FG, int BS, bool FIXED>
__global__ __launch_bounds__(BS) void fpwl_moments_kernel(const MomentParams mp) {
  using bin_t = std::conditional_t<FIXED, unsigned long long, float>;
  constexpr int FPT = Map<FG, BS>::FPT, TPN = Map<FG, BS>::TPN, NODES = Map<FG, BS>::NODES;
  const Params& p = mp.f;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_off[FG + 1];
  const int tid = threadIdx.x;
  const int q = tid % TPN, nl = tid / TPN;
  const int C = p.C;
  const int64_t n_lo = static_cast<int64_t>(blockIdx.x) * p.nodes_per_block;
  const int64_t n_hi = n_lo + p.nodes_per_block < p.n ? n_lo + p.nodes_per_block : p.n;
  const int k0 = blockIdx.y * FG;
  const int nf = p.F - k0 < FG ? p.F - k0 : FG;
  const int base = p.off[k0];
  const int tot = p.off[k0 + nf] - base;
  float* anchor_l = smem;
  const int Rb = bin_stride(C);
  bin_t* bins = reinterpret_cast<bin_t*>(smem + (tot + 1) / 2 * 2);   // [tot][2][C] (+ 1 bin of padding per piece), 8-byte aligned
  for (int i = tid; i < tot; i += BS) anchor_l[i] = p.anchor[base + i];
  for (int i = tid; i < tot * Rb; i += BS) bins[i] = bin_t(0);
  double s0 = 1.0, s1 = 1.0;
  if constexpr (FIXED) { s0 = mp.scales[0]; s1 = mp.scales[1]; }
  if (tid <= nf) s_off[tid] = p.off[k0 + tid] - base;
  __syncthreads();
  int po[FPT], pn[FPT];
  bool live[FPT];
#pragma unroll
  for (int f = 0; f < FPT; ++f) {
    const int fg = q * FPT + f;
    live[f] = fg < nf;
    po[f] = live[f] ? s_off[fg] : 0;
    pn[f] = live[f] ? s_off[fg + 1] - s_off[fg] - 1 : 0;
  }
  for (int64_t n = n_lo + nl; n < n_hi; n += NODES) {
    float xv[FPT];
    const float* xr = p.x + n * p.x_stride + k0 + q * FPT;
#pragma unroll
    for (int f = 0; f < FPT; ++f) xv[f] = live[f] ? xr[f] : 0.f;
    int idx[FPT];
    search<FPT>(anchor_l, po, pn, xv, p.step0, idx);
    const float* gr = mp.g + n * mp.g_stride + (p.sum_features ? 0 : static_cast<int64_t>(k0 + q * FPT) * C);
    float d[FPT];
    bin_t* b[FPT];
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
      d[f] = xv[f] - anchor_l[idx[f]];
      b[f] = bins + static_cast<int64_t>(idx[f]) * Rb;
    }
    // channel-major: in feature-sum mode a channel's gradient is read (and converted) once for the thread's features;
    // channels in chunks of 8 so that the (node-strided, i.e. uncoalesced) gradient loads of a chunk are in flight
    // together — one channel per iteration of the run-time loop waited a full memory latency per channel
    for (int c0 = 0; c0 < C; c0 += 8) {
      float gs[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) gs[u] = (p.sum_features && c0 + u < C) ? gr[c0 + u] : 0.f;
#pragma unroll
      for (int f = 0; f < FPT; ++f) {
        if (live[f]) {
          float gf[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) gf[u] = p.sum_features ? gs[u] : (c0 + u < C ? gr[f * C + c0 + u] : 0.f);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int c = c0 + u;
            if (c < C) {
              if constexpr (FIXED) {
                atomicAdd(b[f] + c, fixed_bits(gf[u], s0));
                atomicAdd(b[f] + C + c, fixed_bits(gf[u] * d[f], s1));
              } else {
                atomicAdd(b[f] + c, gf[u]);
                atomicAdd(b[f] + C + c, gf[u] * d[f]);
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();
  if constexpr (FIXED) {
    unsigned long long* out = mp.Mi + static_cast<int64_t>(base) * 2 * C;
    for (int i = tid; i < tot * 2 * C; i += BS) {
      const unsigned long long v = bins[(i / (2 * C)) * Rb + i % (2 * C)];
      if (v != 0ull) atomicAdd(out + i, v);
    }
  } else {
    float* out = mp.M + static_cast<int64_t>(base) * 2 * C;
    for (int i = tid; i < tot * 2 * C; i += BS) {
      const float v = bins[(i / (2 * C)) * Rb + i % (2 * C)];
      if (v != 0.f) atomicAdd(out + i, v);
    }
  }
}

// Fixed-point moments with the search of fpwl_fast_kernel (skewed breadth-first trees, 3 instructions per step; whole feature
// groups, <= 1023 pieces per feature).  With padded SORTED anchors, aligned alike for the four quads of a wave, 75 % of this
// kernel's LDS cycles were bank conflicts (SQ_LDS_BANK_CONFLICT 7.9e8 of SQ_LDS_IDX_ACTIVE 1.06e9 on C4).
// LDS = trees [FG][2^NSTEP] (+ skew) | 64-bit bins [tot][2][C] | anchors in piece order [tot] | group offsets.
template <int FG, int NSTEP, int BS>
__global__ __launch_bounds__(BS) void fpwl_moments_fast_kernel(const MomentParams mp) {
  static_assert(FG % 4 == 0, "feature quads");
  constexpr int FPT = 4, TPN = FG / FPT, NODES = BS / TPN, P2 = 1 << NSTEP;
  constexpr int kTreeWords = FG * P2 + TPN * kTreeSkew;                 // even: the bins behind it are 8-byte aligned
  const Params& p = mp.f;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int q = tid % TPN, nl = tid / TPN;
  const int C = p.C;
  const int64_t n_lo = static_cast<int64_t>(blockIdx.x) * p.nodes_per_block;
  const int64_t n_hi = n_lo + p.nodes_per_block < p.n ? n_lo + p.nodes_per_block : p.n;
  const int k0 = blockIdx.y * FG;
  const int base = p.off[k0];
  const int tot = p.off[k0 + FG] - base;
  const int Rb = bin_stride(C);
  unsigned long long* bins = reinterpret_cast<unsigned long long*>(smem + kTreeWords);
  float* an_l = reinterpret_cast<float*>(bins + static_cast<int64_t>(tot) * Rb);
  int* s_off = reinterpret_cast<int*>(smem) + p.soff_offset;
  const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) float*)smem));
  if (tid <= FG) s_off[tid] = p.off[k0 + tid] - base;
  for (int i = tid; i < tot * Rb; i += BS) bins[i] = 0ull;
  for (int i = tid; i < tot; i += BS) an_l[i] = p.anchor[base + i];
  __syncthreads();
  for (int i = tid; i < FG * P2; i += BS) {
    const int f = i >> NSTEP, k = i & (P2 - 1);
    float v = __builtin_nanf("");          // padding: NaN <= x is false for EVERY x (+inf <= +inf would be true)
    if (k) {
      const int j = tree_sorted_index<NSTEP>(k);
      if (j < s_off[f + 1] - s_off[f]) v = p.anchor[base + s_off[f] + j];
    }
    smem[i + (f / FPT) * kTreeSkew] = v;
  }
  __syncthreads();
  const int Q = static_cast<int>(lds_base) + q * ((FPT * P2 + kTreeSkew) * 4);   // tree of the thread's first feature
  int nQ = -Q, nQ4 = 4 - Q;
  asm volatile("" : "+v"(nQ), "+v"(nQ4));           // two opaque registers: compare, select, shift-add per step
  int binoff[FPT];
#pragma unroll
  for (int f = 0; f < FPT; ++f) binoff[f] = s_off[q * FPT + f];
  const double s0 = mp.scales[0], s1 = mp.scales[1];
  for (int64_t n = n_lo + nl; n < n_hi; n += NODES) {
    const float* xr = p.x + n * p.x_stride + k0 + q * FPT;
    float xv[FPT];
    if (p.vec_x) {
      const float4 t = *reinterpret_cast<const float4*>(xr);
      xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
    } else {
#pragma unroll
      for (int f = 0; f < FPT; ++f) xv[f] = xr[f];
    }
    int a[FPT];
#pragma unroll
    for (int f = 0; f < FPT; ++f) a[f] = Q + 4;     // node 1 = root
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
#pragma unroll
      for (int f = 0; f < FPT; ++f) {
        const float e = lds_f32(a[f] + f * (P2 * 4));
        a[f] = (a[f] << 1) + (e <= xv[f] ? nQ4 : nQ);
      }
    }
    const float* gr = mp.g + n * mp.g_stride + (p.sum_features ? 0 : static_cast<int64_t>(k0 + q * FPT) * C);
    float d[FPT];
    unsigned long long* b[FPT];
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
      const int piece = binoff[f] + (((a[f] - Q) >> 2) - P2);
      d[f] = xv[f] - an_l[piece];
      b[f] = bins + static_cast<int64_t>(piece) * Rb;
    }
    for (int c0 = 0; c0 < C; c0 += 8) {             // channel-major, chunks of 8 (see fpwl_moments_kernel)
      float gs[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) gs[u] = (p.sum_features && c0 + u < C) ? gr[c0 + u] : 0.f;
#pragma unroll
      for (int f = 0; f < FPT; ++f) {
        float gf[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) gf[u] = p.sum_features ? gs[u] : (c0 + u < C ? gr[f * C + c0 + u] : 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int c = c0 + u;
          if (c < C) {
            atomicAdd(b[f] + c, fixed_bits(gf[u], s0));
            atomicAdd(b[f] + C + c, fixed_bits(gf[u] * d[f], s1));
          }
        }
      }
    }
  }
  __syncthreads();
  unsigned long long* out = mp.Mi + static_cast<int64_t>(base) * 2 * C;
  for (int i = tid; i < tot * 2 * C; i += BS) {
    const unsigned long long v = bins[(i / (2 * C)) * Rb + i % (2 * C)];
    if (v != 0ull) atomicAdd(out + i, v);
  }
}

// C == 1 specialisation of the fixed-point moments (round 2).  The general kernel above spends, per (node, feature) and
// AFTER the search, a dependent global load of the gradient (the channel loop is a run-time loop), two library
// float64 -> int64 conversions and a random LDS read of the piece's anchor: five dependent memory latencies per node and
// 4 waves per SIMD (58 KB of LDS per 512 threads).  Here the gradient travels with the x row (one request each, issued
// together), the anchor of the piece is the last tree entry the search stepped right at (one v_cndmask per step, no
// anchor array in LDS: 50 KB per workgroup, three workgroups = 6 waves per SIMD), the bins are two arrays [2][tot]
// (8-byte stride: the 64-bit atomics of a wave spread over all banks; interleaved (M0, M1) pairs used every other
// bank pair; over the forward's kept pieces: piece-major, round 5), a conversion is fixed_bits() and the workgroup map is the forward's (the groups of a node block run back
// to back on one XCD and share the 128-B lines of x in its L2).  SUMF: the gradient is [n, 1] (feature sum) and its M0
// term is converted once per node; otherwise [n, F] and read as one 16-byte load next to x.
template <int FG, int NSTEP, int BS, bool SUMF, bool RAGGED = false>
__global__ __launch_bounds__(BS) void fpwl_moments_c1_kernel(const MomentParams mp) {
  static_assert(FG % 4 == 0, "feature quads");
  constexpr int FPT = 4, TPN = FG / FPT, NODES = BS / TPN, P2 = 1 << NSTEP;
  constexpr int kTreeWords = FG * P2 + TPN * kTreeSkew;                 // even: the bins behind it are 8-byte aligned
  const Params& p = mp.f;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int q = tid % TPN, nl = tid / TPN;
  const int64_t id = blockIdx.x;                                        // (id % 8) = XCD, groups of a node block adjacent inside it
  const int grp = static_cast<int>((id >> 3) % p.n_groups);
  const int64_t nb = ((id >> 3) / p.n_groups) * 8 + (id & 7);
  const int64_t n_lo = nb * p.nodes_per_block;
  if (n_lo >= p.n) return;
  const int64_t n_hi = n_lo + p.nodes_per_block < p.n ? n_lo + p.nodes_per_block : p.n;
  const int k0 = grp * FG;
  const int nf = RAGGED ? (p.F - k0 < FG ? p.F - k0 : FG) : FG;       // live features of the group (see fpwl_fast_kernel)
  const int base = p.off[k0];
  const int tot = p.off[k0 + nf] - base;
  const bool saved = p.piece_in != nullptr;          // (uniform) the forward pass kept the pieces: no trees, no search
  const bool kept = !RAGGED && saved && (SUMF || mp.vec_g);   // ... and the loop below bins them piece-major, from offset 0
  const int PM = p.piece_stride;
  unsigned long long* bins = reinterpret_cast<unsigned long long*>(smem + (kept ? 0 : kTreeWords));   // [2][tot] | [2][PM][FG]
  int* s_off = reinterpret_cast<int*>(smem) + p.soff_offset;
  const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) float*)smem));
  if (tid <= FG) s_off[tid] = p.off[k0 + (tid < nf ? tid : nf)] - base;
  for (int i = tid; i < (kept ? 2 * FG * PM : 2 * tot); i += BS) bins[i] = 0ull;
  __syncthreads();
  if (kept) {
    // nothing to stage: the anchors are read once per piece at the flush, from memory
  } else if (saved) {
    for (int i = tid; i < tot; i += BS) smem[i] = p.anchor[base + i];       // anchors in piece order (tot <= FG * P2)
  } else {
    for (int i = tid; i < FG * P2; i += BS) {
      const int f = i >> NSTEP, k = i & (P2 - 1);
      float v = __builtin_nanf("");          // padding: NaN <= x is false for EVERY x (+inf <= +inf would be true)
      if (k) {
        const int j = tree_sorted_index<NSTEP>(k);
        if (j < s_off[f + 1] - s_off[f]) v = p.anchor[base + s_off[f] + j];
      }
      smem[i + (f / FPT) * kTreeSkew] = v;
    }
  }
  __syncthreads();
  const int Q = static_cast<int>(lds_base) + q * ((FPT * P2 + kTreeSkew) * 4);   // tree of the thread's first feature
  int nQ = -Q, nQ4 = 4 - Q;
  asm volatile("" : "+v"(nQ), "+v"(nQ4));           // two opaque registers: compare, select, shift-add per step
  int binoff[FPT];
  float a0[FPT];                                    // anchor of piece 0 (what the search keeps when it never steps right)
  bool live[FPT];
#pragma unroll
  for (int f = 0; f < FPT; ++f) {
    live[f] = !RAGGED || q * FPT + f < nf;
    binoff[f] = s_off[q * FPT + f] - P2;
    a0[f] = live[f] ? p.anchor[base + s_off[q * FPT + f]] : 0.f;
  }
  const double s0 = mp.scales[0], s1 = mp.scales[1];
  if constexpr (!RAGGED) {
    if (saved && (SUMF || mp.vec_g)) {
      // The forward kept the pieces: no search.  Binned here are sum g and sum g * x (the product exact in float64) — two LDS
      // atomics per look-up and NOTHING the wave has to wait for (reading the piece's anchor first put an LDS round trip in
      // front of every pair of atomics: the LDS pipe sat idle 40 % of the time, profiles/r04_sq_train.txt); the anchor enters
      // once per piece and workgroup, when the bins are flushed: M1 = sum g (x - a) = sum g x - a sum g.  The x / gradient /
      // pieces of the next TWO rounds are in flight while a round is binned (unconditional loads from a clamped address — a
      // guarded load hides the loads in flight from the compiler, which then waits for all of them).  In this branch the bins
      // are piece-major, [2][piece_stride][FG] from LDS offset 0 (no trees, no anchors in LDS): see the loop.
      int64_t n = n_lo + nl;
      const int rot = nl & 3;
      if (n < n_hi) {
        const int64_t last = n_hi - 1;
        const uint8_t* pin = p.piece_in + static_cast<int64_t>(grp) * p.n * FG + q * FPT;
        const float* xin = p.x + k0 + q * FPT;
        const float* gin = SUMF ? mp.g : mp.g + k0 + q * FPT;
        struct Round { float4 x; unsigned pc; float4 g; };
        // unconditional loads from a clamped node (see above); three rounds are in flight, each in registers of its own: a
        // "next = far" hand-over at the end of the loop body made every iteration wait for the loads it had just issued
        auto fetch = [&](int64_t m) {
          m = m < n_hi ? m : last;
          Round rd;
          rd.x = *reinterpret_cast<const float4*>(xin + m * p.x_stride);
          rd.pc = *reinterpret_cast<const unsigned*>(pin + m * FG);
          if constexpr (SUMF) { rd.g.x = gin[m * mp.g_stride]; rd.g.y = rd.g.z = rd.g.w = rd.g.x; }
          else rd.g = *reinterpret_cast<const float4*>(gin + m * mp.g_stride);
          return rd;
        };
        auto bin = [&](const Round& rd) {
          const float4 cx = rd.x, cg = rd.g;
          const unsigned cp = rd.pc;
          const float xv[FPT] = {cx.x, cx.y, cx.z, cx.w};
          const float gv[FPT] = {cg.x, cg.y, cg.z, cg.w};
          // Two things keep the lanes of one wave-wide atomic apart (72 % of the LDS-active cycles were bank conflicts,
          // profiles/r04f_sq_train.txt; real data sits in a handful of pieces per feature, and atomics that meet in a bank pair —
          // on one address or not — are served one after the other):
          //  * issue slot s takes feature r = (s + node) % 4 of the thread's quad, so an instruction spreads over the bins of all
          //    FG features, 64 / FG nodes each, instead of four features x sixteen nodes (4.36 -> 4.16 ms per C4 training step);
          //  * the bins are piece-major, [piece][feature]: a feature owns the bank pairs f and f + 16 whatever pieces its nodes
          //    fall into, so lanes of different features never meet (conflict cycles 1.7e8 -> 9e6 per launch).
          // Integer adds: the sums are the same bit for bit.  With the conflicts gone the loop is bound by its vector
          // instructions: the rotation selects x (32 bits) before the product, the piece by a variable bit-field extract,
          // and g * 2^e1 is formed once per node (a power of two: the fused multiply-add rounds the same real number).
          double gs1 = 0.0;
          unsigned long long t0 = 0ull;
          if constexpr (SUMF) {
            gs1 = static_cast<double>(gv[0]) * s1;
            t0 = fixed_bits(gv[0], s0);
          }
#pragma unroll
          for (int s = 0; s < FPT; ++s) {
            const int r = (s + rot) & 3;
            const bool odd = (r & 1) != 0, up = (r & 2) != 0;      // (two selects by the bits of r: a chain of r == k tests
            const float xlo = odd ? xv[1] : xv[0], xhi = odd ? xv[3] : xv[2];   //  is lowered to a switch with branches)
            const float xr = up ? xhi : xlo;
            const int ix = static_cast<int>(__builtin_amdgcn_ubfe(cp, 8u * static_cast<unsigned>(r), 8u)) * FG + (q * FPT + r);
            unsigned long long v0 = t0;
            double gr = gs1;
            if constexpr (!SUMF) {
              const float glo = odd ? gv[1] : gv[0], ghi = odd ? gv[3] : gv[2];
              const float gsel = up ? ghi : glo;
              v0 = fixed_bits(gsel, s0);
              gr = static_cast<double>(gsel) * s1;
            }
            const double d = fma(gr, static_cast<double>(xr), 6755399441055744.0);
            const unsigned long long v1 = static_cast<unsigned long long>(__double_as_longlong(d)) - 0x4338000000000000ull;
            atomicAdd(bins + ix, v0);
            atomicAdd(bins + FG * PM + ix, v1);
          }
        };
        Round ra = fetch(n), rb = fetch(n + NODES), rc = fetch(n + 2 * NODES);
        for (; n < n_hi; n += 3 * NODES) {
          bin(ra);
          ra = fetch(n + 3 * NODES);
          if (n + NODES < n_hi) bin(rb);
          rb = fetch(n + 4 * NODES);
          if (n + 2 * NODES < n_hi) bin(rc);
          rc = fetch(n + 5 * NODES);
        }
      }
      __syncthreads();
      unsigned long long* out = mp.Mi + static_cast<int64_t>(base) * 2;      // [T][2]
      const double ratio = s1 / s0;
      for (int i = tid; i < tot; i += BS) {
        int f = 0;
#pragma unroll
        for (int k = 1; k < FG; ++k) f += s_off[k] <= i ? 1 : 0;              // the feature of piece i (s_off ascending)
        const int b = (i - s_off[f]) * FG + f;
        const long long m0 = static_cast<long long>(bins[b]), m1x = static_cast<long long>(bins[FG * PM + b]);
        if (m0 != 0 || m1x != 0) {
          // (float64: relative error 2^-53 of |a sum g| per workgroup and piece — the same in every run: the node blocks are fixed)
          const long long m1 = m1x - __double2ll_rn(static_cast<double>(p.anchor[base + i]) * static_cast<double>(m0) * ratio);
          if (m0 != 0) atomicAdd(out + 2 * i, static_cast<unsigned long long>(m0));
          if (m1 != 0) atomicAdd(out + 2 * i + 1, static_cast<unsigned long long>(m1));
        }
      }
      return;
    }
  }
  for (int64_t n = n_lo + nl; n < n_hi; n += NODES) {
    float xv[FPT];
    if constexpr (RAGGED) {
      const float* xr = p.x + n * p.x_stride + k0 + q * FPT;
#pragma unroll
      for (int f = 0; f < FPT; ++f) xv[f] = live[f] ? xr[f] : 0.f;
    } else {
      const float4 t = *reinterpret_cast<const float4*>(p.x + n * p.x_stride + k0 + q * FPT);
      xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
    }
    float gv[FPT];
    if constexpr (SUMF) {
      gv[0] = gv[1] = gv[2] = gv[3] = mp.g[n * mp.g_stride];
    } else {
      const float* gr = mp.g + n * mp.g_stride + k0 + q * FPT;
      if (!RAGGED && mp.vec_g) {
        const float4 g4 = *reinterpret_cast<const float4*>(gr);
        gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
      } else {
#pragma unroll
        for (int f = 0; f < FPT; ++f) gv[f] = live[f] ? gr[f] : 0.f;
      }
    }
    int a[FPT];
    float last[FPT];
    if (saved) {
      const unsigned packed = *reinterpret_cast<const unsigned*>(p.piece_in + (static_cast<int64_t>(grp) * p.n + n) * FG + q * FPT);
#pragma unroll
      for (int f = 0; f < FPT; ++f) {
        const int pc = static_cast<int>((packed >> (8 * f)) & 0xffu);
        a[f] = Q + ((pc + P2) << 2);                // the tree address the search would have ended at
        last[f] = live[f] ? smem[binoff[f] + P2 + pc] : 0.f;
      }
    } else {
#pragma unroll
      for (int f = 0; f < FPT; ++f) {
        a[f] = Q + 4;                               // node 1 = root
        last[f] = a0[f];
      }
#pragma unroll
      for (int step = 0; step < NSTEP; ++step) {
#pragma unroll
        for (int f = 0; f < FPT; ++f) {
          const float e = lds_f32(a[f] + f * (P2 * 4));
          const bool right = e <= xv[f];
          a[f] = (a[f] << 1) + (right ? nQ4 : nQ);
          last[f] = right ? e : last[f];
        }
      }
    }
    int idx[FPT];
    unsigned long long t0[FPT], t1[FPT];
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
      idx[f] = (!RAGGED || live[f]) ? binoff[f] + ((a[f] - Q) >> 2) : -1;
      t0[f] = fixed_bits(gv[SUMF ? 0 : f], s0);
      t1[f] = fixed_bits(gv[f] * (xv[f] - last[f]), s1);
    }
    const int rot = nl & 3;                            // (issue slots rotated over the quad's features: see the kept-pieces loop)
#pragma unroll
    for (int s = 0; s < FPT; ++s) {
      const int r = (s + rot) & 3;
      const int ix = r == 0 ? idx[0] : (r == 1 ? idx[1] : (r == 2 ? idx[2] : idx[3]));
      const unsigned long long v0 = SUMF ? t0[0] : (r == 0 ? t0[0] : (r == 1 ? t0[1] : (r == 2 ? t0[2] : t0[3])));
      const unsigned long long v1 = r == 0 ? t1[0] : (r == 1 ? t1[1] : (r == 2 ? t1[2] : t1[3]));
      if (!RAGGED || ix >= 0) {
        atomicAdd(bins + ix, v0);
        atomicAdd(bins + tot + ix, v1);
      }
    }
  }
  __syncthreads();
  unsigned long long* out = mp.Mi + static_cast<int64_t>(base) * 2;      // [T][2]
  for (int i = tid; i < 2 * tot; i += BS) {
    const unsigned long long v = bins[i];
    const int m = i >= tot ? 1 : 0;
    if (v != 0ull) atomicAdd(out + 2 * (i - m * tot) + m, v);
  }
}

template <int FG, int NSTEP, int BS, bool RAGGED = false>
int launch_moments_c1(MomentParams mp, hipStream_t st) {
  Params& p = mp.f;
  const size_t pieces = static_cast<size_t>(p.max_group_pieces);
  size_t lds = ((static_cast<size_t>(FG) << NSTEP) + (FG / 4) * kTreeSkew) * sizeof(float) + pieces * 2 * sizeof(unsigned long long);
  p.piece_stride = (p.max_pieces + 1) & ~1;
  if (!RAGGED && p.piece_in != nullptr && (p.sum_features || mp.vec_g))    // kept pieces: piece-major bins from offset 0, no trees
    lds = static_cast<size_t>(2) * FG * p.piece_stride * sizeof(unsigned long long);
  p.soff_offset = static_cast<int>(lds / sizeof(float));
  lds += (FG + 1) * sizeof(int);
  if (lds > 150 * 1024) return -1;                       // caller falls back to the general kernels
  const void* fn = p.sum_features ? reinterpret_cast<const void*>(&fpwl_moments_c1_kernel<FG, NSTEP, BS, true, RAGGED>)
                                  : reinterpret_cast<const void*>(&fpwl_moments_c1_kernel<FG, NSTEP, BS, false, RAGGED>);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "fpwl: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  p.nodes_per_block = tuned_moment_block(p.n, p.n_groups, lds, BS);
  const int64_t bx = ((p.n + p.nodes_per_block - 1) / p.nodes_per_block + 7) / 8 * 8;   // whole rounds of the 8 XCDs
  if (bx * p.n_groups > 0x7fffffffLL) return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl: too many nodes for one launch");
  const dim3 grid(static_cast<unsigned>(bx * p.n_groups));
  if (p.sum_features) {
    hipLaunchKernelGGL((fpwl_moments_c1_kernel<FG, NSTEP, BS, true, RAGGED>), grid, dim3(BS), lds, st, mp);
  } else {
    hipLaunchKernelGGL((fpwl_moments_c1_kernel<FG, NSTEP, BS, false, RAGGED>), grid, dim3(BS), lds, st, mp);
  }
  return gnan::check_launch("fpwl_moments_c1_kernel");
}

template <int FG, int NSTEP, int BS>
int launch_moments_fast(MomentParams mp, hipStream_t st) {
  Params& p = mp.f;
  const size_t pieces = static_cast<size_t>(p.max_group_pieces);
  size_t lds = ((static_cast<size_t>(FG) << NSTEP) + (FG / 4) * kTreeSkew) * sizeof(float) +
               pieces * static_cast<size_t>(bin_stride(p.C)) * sizeof(unsigned long long) + pieces * sizeof(float);
  p.soff_offset = static_cast<int>(lds / sizeof(float));
  lds += (FG + 1) * sizeof(int);
  if (lds > 150 * 1024) return -1;                       // caller falls back to the plain kernel
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fpwl_moments_fast_kernel<FG, NSTEP, BS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "fpwl: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const int64_t bx = (p.n + p.nodes_per_block - 1) / p.nodes_per_block;
  if (bx > 0x7fffffffLL) return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl: too many nodes for one launch");
  hipLaunchKernelGGL((fpwl_moments_fast_kernel<FG, NSTEP, BS>), dim3(static_cast<unsigned>(bx), static_cast<unsigned>(p.n_groups)),
                     dim3(BS), lds, st, mp);
  return gnan::check_launch("fpwl_moments_fast_kernel");
}

template <int FG, int BS>
int launch_moments(const MomentParams& mp_in, size_t lds, hipStream_t st) {
  MomentParams mp = mp_in;
  if (mp.f.max_pieces > 256) mp.f.piece_in = nullptr;      // one byte per look-up: searched again instead
  const bool fixed = mp.Mi != nullptr;
  if constexpr (FG % 4 == 0) {
    const bool whole = mp.f.F % FG == 0 && mp.f.vec_x;
    if (fixed && mp.f.C == 1 && !whole) {               // ragged feature count / unaligned rows: the C == 1 kernel's RAGGED variant
      int nstep = 6;
      while ((1 << nstep) < mp.f.max_pieces) ++nstep;
      int rc = -1;
      if (!mp.general_only) {
        switch (nstep) {
          case 6: rc = launch_moments_c1<FG, 6, BS, true>(mp, st); break;
          case 7: rc = launch_moments_c1<FG, 7, BS, true>(mp, st); break;
          case 8: rc = launch_moments_c1<FG, 8, BS, true>(mp, st); break;
          case 9: rc = launch_moments_c1<FG, 9, BS, true>(mp, st); break;
          case 10: rc = launch_moments_c1<FG, 10, BS, true>(mp, st); break;
          default: break;
        }
        if (rc != -1) return rc;
      }
    }
    if (fixed && mp.f.F % FG == 0) {
      int nstep = 6;
      while ((1 << nstep) < mp.f.max_pieces) ++nstep;
      int rc = -1;
      if (mp.f.C == 1 && mp.f.vec_x) {
        if (!mp.general_only) {                              // (GNAN_FPWL_MOMENTS_GENERAL: A/B aid, tests)
          // 512 threads = three workgroups (24 waves) per CU; 640 (30 waves): +13 %/+6 %, 1024 (32 waves): +-0 on C4
          switch (nstep) {
            case 6: rc = launch_moments_c1<FG, 6, BS>(mp, st); break;
            case 7: rc = launch_moments_c1<FG, 7, BS>(mp, st); break;
            case 8: rc = launch_moments_c1<FG, 8, BS>(mp, st); break;
            case 9: rc = launch_moments_c1<FG, 9, BS>(mp, st); break;
            case 10: rc = launch_moments_c1<FG, 10, BS>(mp, st); break;
            default: break;
          }
          if (rc != -1) return rc;
        }
      }
      switch (nstep) {
        case 6: rc = launch_moments_fast<FG, 6, BS>(mp, st); break;
        case 7: rc = launch_moments_fast<FG, 7, BS>(mp, st); break;
        case 8: rc = launch_moments_fast<FG, 8, BS>(mp, st); break;
        case 9: rc = launch_moments_fast<FG, 9, BS>(mp, st); break;
        case 10: rc = launch_moments_fast<FG, 10, BS>(mp, st); break;
        default: break;
      }
      if (rc != -1) return rc;
    }
  }
  const void* fn = fixed ? reinterpret_cast<const void*>(&fpwl_moments_kernel<FG, BS, true>)
                         : reinterpret_cast<const void*>(&fpwl_moments_kernel<FG, BS, false>);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "fpwl: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const Params& p = mp.f;
  const int64_t bx = (p.n + p.nodes_per_block - 1) / p.nodes_per_block;
  if (bx > 0x7fffffffLL) return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl: too many nodes for one launch");
  const dim3 grid(static_cast<unsigned>(bx), static_cast<unsigned>(p.n_groups));
  if (fixed) {
    hipLaunchKernelGGL((fpwl_moments_kernel<FG, BS, true>), grid, dim3(BS), lds, st, mp);
  } else {
    hipLaunchKernelGGL((fpwl_moments_kernel<FG, BS, false>), grid, dim3(BS), lds, st, mp);
  }
  return gnan::check_launch("fpwl_moments_kernel");
}

template <int FG, int BS>
int launch(Params p, size_t lds, hipStream_t st, float* total_out) {
  const size_t table_lds = lds;            // (1 + 2C) floats per piece of the largest group
  if (p.sum_features && p.C > 1) {        // accumulate the feature sum in LDS: one pass of nodes per workgroup
    if (lds + static_cast<size_t>(p.C) * Map<FG, BS>::NODES * sizeof(float) <= 150 * 1024) {
      p.acc_offset = static_cast<int>(lds / sizeof(float));
      p.nodes_per_block = Map<FG, BS>::NODES;
    }
  }
  const int64_t bx = (p.n + p.nodes_per_block - 1) / p.nodes_per_block;
  const int64_t wgs = p.sum_features ? bx : (bx + 7) / 8 * 8 * p.n_groups;     // see the id -> (node block, group) map
  if (wgs > 0x7fffffffLL) return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl: too many nodes for one launch");
  const dim3 grid(static_cast<unsigned>(wgs));
  if (p.acc_offset) lds += static_cast<size_t>(p.C) * Map<FG, BS>::NODES * sizeof(float);
  // FAST: one output channel, whole groups only, 16-B aligned x (and fx) rows
  const bool whole = p.F % FG == 0 && p.vec_x;                       // whole groups, 16-B aligned rows of x
  const bool ragged = FG % 4 == 0 && p.C == 1 && p.sum_features && !whole && !p.out_bf16 && !p.col_partial;
  const bool fast = FG % 4 == 0 && p.C == 1 && ((whole && (p.sum_features || p.vec_out)) || ragged);
  if (p.col_partial) {
    if (!fast || p.sum_features)
      return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl: fused column sums need C == 1, F %% %d == 0, 16-B aligned rows, per-feature output", FG);
    lds += BS * 4 * sizeof(float);
  }
  if constexpr (FG % 4 == 0) {
    int nstep = 6;
    while ((1 << nstep) < p.max_pieces) ++nstep;
    // trees + (val, slope) pairs + anchors in piece order (C == 1: 3 floats per piece); the column-sum scratch re-uses it
    size_t lds_fast = ((static_cast<size_t>(FG) << nstep) + (FG / 4) * kTreeSkew) * sizeof(float) + table_lds;
    if (p.col_partial && lds_fast < BS * 4 * sizeof(float)) lds_fast = BS * 4 * sizeof(float);
    p.soff_offset = static_cast<int>(lds_fast / sizeof(float));
    lds_fast += (FG + 1) * sizeof(int);
    if (p.piece_out && !(fast && p.sum_features && nstep <= 8 && lds_fast <= 150 * 1024))
      return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl: piece_out needs the fast feature-sum kernel (C == 1, feature quads) and at most 256 pieces per feature");
    if (fast && nstep <= 10 && lds_fast <= 150 * 1024) {
      auto fgo = [&](auto kernel) {
        if (lds_fast > 64 * 1024) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_fast));
          if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "fpwl: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(kernel, grid, dim3(BS), lds_fast, st, p);
        return gnan::check_launch("fpwl_fast_kernel");
      };
      auto by_mode = [&](auto ns) {
        constexpr int NS = decltype(ns)::value;
        if (p.out_bf16) return p.sum_features ? gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl: bf16 output is per-feature only")
                                              : fgo(fpwl_fast_kernel<FG, false, true, NS, BS>);
        if (ragged) return fgo(fpwl_fast_kernel<FG, true, false, NS, BS, true>);
        return p.sum_features ? fgo(fpwl_fast_kernel<FG, true, false, NS, BS>)
                              : fgo(fpwl_fast_kernel<FG, false, false, NS, BS>);
      };
      int rc;
      switch (nstep) {
        case 6: rc = by_mode(std::integral_constant<int, 6>{}); break;
        case 7: rc = by_mode(std::integral_constant<int, 7>{}); break;
        case 8: rc = by_mode(std::integral_constant<int, 8>{}); break;
        case 9: rc = by_mode(std::integral_constant<int, 9>{}); break;
        default: rc = by_mode(std::integral_constant<int, 10>{}); break;
      }
      if (rc) return rc;
      if (p.col_partial) {
        hipLaunchKernelGGL(fpwl_total_kernel, dim3(p.F), dim3(256), 0, st, p.col_partial, static_cast<int>(bx), p.F,
                           total_out);
        return gnan::check_launch("fpwl_total_kernel");
      }
      return GNAN_OK;
    }
  }
  if (p.piece_out) return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl: piece_out needs the fast feature-sum kernel");
  auto go = [&](auto kernel) {
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "fpwl: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kernel, grid, dim3(BS), lds, st, p);
    return gnan::check_launch("fpwl_kernel");
  };
  if (p.out_bf16) {
    if constexpr (FG % 4 == 0) {
      if (!fast || p.sum_features)
        return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl: bf16 output needs C == 1, whole feature groups, aligned rows, per-feature mode");
      if (int rc = go(fpwl_kernel<FG, false, true, true, BS>)) return rc;
    } else {
      return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl: bf16 output needs feature groups of >= 4");
    }
  } else {
    if (p.sum_features) return fast ? go(fpwl_kernel<FG, true, true, false, BS>) : go(fpwl_kernel<FG, true, false, false, BS>);
    if (int rc = fast ? go(fpwl_kernel<FG, false, true, false, BS>) : go(fpwl_kernel<FG, false, false, false, BS>)) return rc;
  }
  if (p.col_partial) {
    hipLaunchKernelGGL(fpwl_total_kernel, dim3(p.F), dim3(256), 0, st, p.col_partial, static_cast<int>(bx), p.F,
                       total_out);
    return gnan::check_launch("fpwl_total_kernel");
  }
  return GNAN_OK;
}

}  // namespace

namespace {
int common_checks(const gnan_fpwl_args* a) {
  GNAN_REQUIRE(a != nullptr, "fpwl: null args");
  GNAN_REQUIRE(a->n >= 0 && a->F >= 1 && a->C >= 1, "fpwl: bad sizes");
  if (a->n == 0) return GNAN_OK;
  GNAN_REQUIRE(a->x && a->off && a->anchor, "fpwl: null pointer");
  GNAN_REQUIRE(a->x_stride >= a->F, "fpwl: x row stride smaller than F");
  GNAN_REQUIRE(a->max_pieces >= 1 && a->max_group_pieces >= 1, "fpwl: max_pieces / max_group_pieces must be >= 1");
  const int fg = a->features_per_group;
  GNAN_REQUIRE(fg == 1 || fg == 2 || fg == 4 || fg == 8 || fg == 16, "fpwl: features_per_group must be 1, 2, 4, 8 or 16");
  const size_t lds = static_cast<size_t>(a->max_group_pieces) * (1 + 2 * static_cast<size_t>(table_stride(a->C))) * sizeof(float);
  if (lds > 150 * 1024)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl: %zu B of tables per feature group exceed LDS; use fewer features per group",
                      lds);
  return GNAN_OK;
}

// Nodes per workgroup along the node axis.  A workgroup first loads the LDS image of its feature group (worth ~300
// look-up rows of a 16-feature group), so blocks should be large; the L2 sharing of x lines between the groups of a node
// block wants them <= 4096 (measured on C4: 2048 +6 %, 8192 +2 %, 16384 +7 %); and the grid should be a whole number
// of rounds of the workgroups the chip holds at once (LDS-limited: 3 per CU for 16-feature groups), which matters for
// shares of a few million rows (2.7M rows: 5.04 rounds of 2816-node blocks -> 4 rounds of 3584-node blocks, -15 %).
int cu_count() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  return cus;
}

// The same trade-off for fpwl_moments_c1_kernel: its image (trees + 64-bit bins) is zeroed / built at the start and
// flushed with one global atomic per touched bin at the end — worth ~600 rows — and x lines are shared between the
// groups of a node block in L2, so blocks of 1024..8192 nodes in whole rounds of the resident workgroups.
int tuned_moment_block(int64_t n, int n_groups, size_t lds, int bs) {
  if (n < 16384) {
    const int64_t npb = (n / 1024 + 255) / 256 * 256;
    return static_cast<int>(npb < 256 ? 256 : (npb > 4096 ? 4096 : npb));
  }
  // (round 5: the same model from 16k nodes on — the arxiv-shaped graph, 169k nodes x 9 groups, ran 5958 workgroups of 256
  //  nodes, each paying the ~600 rows of image and flush: 70 % overhead and 25M global atomics onto 37k bins; one round of
  //  ~750 workgroups of 2048 nodes instead)
  int per_cu = static_cast<int>((160 * 1024) / lds);
  if (per_cu > 2048 / bs) per_cu = 2048 / bs;
  if (per_cu < 1) per_cu = 1;
  const int64_t resident = static_cast<int64_t>(cu_count()) * per_cu;
  const int unit = 128, overhead = 600;
  int64_t best_cost = -1;
  int best = 4096;
  for (int npb = n < 262144 ? 256 : 1024; npb <= 8192; npb += unit) {
    const int64_t wgs = (n + npb - 1) / npb * n_groups;
    const int64_t rounds = (wgs + resident - 1) / resident;
    const int64_t cost = rounds * (npb + overhead);
    if (best_cost < 0 || cost < best_cost || (cost == best_cost && npb > best)) { best_cost = cost; best = npb; }
  }
  return best;
}

int tuned_nodes_per_block(const gnan_fpwl_args* a, int n_groups) {
  const int cus = cu_count();
  const int fg = a->features_per_group;
  if (a->C != 1 || fg < 4 || a->n < 262144) {              // general kernel / small inputs: ~1024 blocks along the node axis
    const int64_t npb = (a->n / 1024 + 255) / 256 * 256;
    return static_cast<int>(npb < 256 ? 256 : (npb > 4096 ? 4096 : npb));
  }
  int nstep = 6;
  while ((1 << nstep) < a->max_pieces) ++nstep;
  const size_t lds = (static_cast<size_t>(fg) << nstep) * 4 + static_cast<size_t>(a->max_group_pieces) * 12 + 128;
  const int bs = fg >= 8 ? 512 : 256;
  int per_cu = static_cast<int>((160 * 1024) / lds);
  if (per_cu > 2048 / bs) per_cu = 2048 / bs;
  if (per_cu < 1) per_cu = 1;
  const int64_t resident = static_cast<int64_t>(cus) * per_cu;
  const int64_t groups = a->sum_features ? 1 : n_groups;   // feature-sum mode walks its groups inside one workgroup
  const int unit = 128, overhead = 300;
  int64_t best_cost = -1;
  int best = 4096;
  for (int npb = 1024; npb <= 4096; npb += unit) {
    const int64_t wgs = (a->n + npb - 1) / npb * groups;
    const int64_t rounds = (wgs + resident - 1) / resident;
    const int64_t cost = rounds * (npb + overhead);
    if (best_cost < 0 || cost < best_cost || (cost == best_cost && npb > best)) { best_cost = cost; best = npb; }
  }
  return best;
}

Params base_params(const gnan_fpwl_args* a) {
  Params p;
  p.x = a->x; p.n = a->n; p.x_stride = a->x_stride; p.F = a->F; p.C = a->C;
  p.off = a->off; p.anchor = a->anchor; p.val = a->val; p.slope = a->slope;
  int step0 = 0;
  while ((step0 ? step0 * 2 : 1) <= a->max_pieces - 1) step0 = step0 ? step0 * 2 : 1;
  p.step0 = step0;
  p.n_groups = (a->F + a->features_per_group - 1) / a->features_per_group;
  p.nodes_per_block = tuned_nodes_per_block(a, p.n_groups);
  p.sum_features = a->sum_features;
  p.vec_x = p.vec_out = 0;
  p.out = static_cast<float*>(a->out); p.out_stride = a->out_stride;
  p.col_partial = nullptr;
  p.max_pieces = a->max_pieces;
  p.max_group_pieces = a->max_group_pieces;
  p.soff_offset = 0;
  p.total_rows = (a->total_rows > 0 && a->total_rows < a->n) ? a->total_rows : a->n;
  p.acc_offset = 0;
  p.out_bf16 = a->out_dtype == GNAN_BF16;
  p.piece_out = a->piece_out; p.piece_in = a->piece_in;
  return p;
}
}  // namespace

namespace {
// ---------------------------------------------------------------------------------------------
// Phase 1 of the two-phase look-up for several channels (csrc/fpwl_rows.hip) with this file's search: the piece of every
// (node, feature) and dx = x - anchor, [n, F] each.  Same structure as fpwl_moments_c1_kernel — skewed breadth-first trees
// of a 16-feature group in LDS, thread = (node, 4 features), the piece's anchor tracked by the search, partial last
// group and unaligned rows allowed, XCD-aware workgroup map — and 3x the rate of the sorted-array search it replaces
// (10M nodes x 64 features: 3.4 -> 1.6 ms; used from 262 144 nodes).
// ---------------------------------------------------------------------------------------------
struct LocateTreeParams {
  Params f;            // x, off, anchor, grouping (n_groups of FG features), nodes_per_block, soff_offset
  int32_t* piece;      // [n, F]
  float* dx;           // [n, F]
  int vec_x, vec_out;  // 16-byte loads of x / stores of piece and dx
};

template <int FG, int NSTEP, int BS>
__global__ __launch_bounds__(BS) void fpwl_locate_tree_kernel(const LocateTreeParams lp) {
  static_assert(FG % 4 == 0, "feature quads");
  constexpr int FPT = 4, TPN = FG / FPT, NODES = BS / TPN, P2 = 1 << NSTEP;
  const Params& p = lp.f;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int q = tid % TPN, nl = tid / TPN;
  const int64_t id = blockIdx.x;                                        // (id % 8) = XCD, groups of a node block adjacent inside it
  const int grp = static_cast<int>((id >> 3) % p.n_groups);
  const int64_t nb = ((id >> 3) / p.n_groups) * 8 + (id & 7);
  const int64_t n_lo = nb * p.nodes_per_block;
  if (n_lo >= p.n) return;
  const int64_t n_hi = n_lo + p.nodes_per_block < p.n ? n_lo + p.nodes_per_block : p.n;
  const int k0 = grp * FG;
  const int nf = p.F - k0 < FG ? p.F - k0 : FG;                         // live features of the group
  const int base = p.off[k0];
  int* s_off = reinterpret_cast<int*>(smem) + p.soff_offset;
  const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) float*)smem));
  if (tid <= FG) s_off[tid] = p.off[k0 + (tid < nf ? tid : nf)] - base;
  __syncthreads();
  for (int i = tid; i < FG * P2; i += BS) {
    const int f = i >> NSTEP, k = i & (P2 - 1);
    float v = __builtin_nanf("");          // padding: NaN <= x is false for EVERY x (+inf <= +inf would be true)
    if (k) {
      const int j = tree_sorted_index<NSTEP>(k);
      if (j < s_off[f + 1] - s_off[f]) v = p.anchor[base + s_off[f] + j];
    }
    smem[i + (f / FPT) * kTreeSkew] = v;
  }
  __syncthreads();
  const int Q = static_cast<int>(lds_base) + q * ((FPT * P2 + kTreeSkew) * 4);   // tree of the thread's first feature
  int nQ = -Q, nQ4 = 4 - Q;
  asm volatile("" : "+v"(nQ), "+v"(nQ4));           // two opaque registers: compare, select, shift-add per step
  int row0[FPT];                                    // global table row of the feature's piece 0, minus P2
  float a0[FPT];
  bool live[FPT];
#pragma unroll
  for (int f = 0; f < FPT; ++f) {
    live[f] = q * FPT + f < nf;
    row0[f] = base + s_off[q * FPT + f] - P2;
    a0[f] = live[f] ? p.anchor[base + s_off[q * FPT + f]] : 0.f;
  }
  for (int64_t n = n_lo + nl; n < n_hi; n += NODES) {
    float xv[FPT];
    const float* xr = p.x + n * p.x_stride + k0 + q * FPT;
    if (lp.vec_x && nf == FG) {
      const float4 t = *reinterpret_cast<const float4*>(xr);
      xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
    } else {
#pragma unroll
      for (int f = 0; f < FPT; ++f) xv[f] = live[f] ? xr[f] : 0.f;
    }
    int a[FPT];
    float last[FPT];
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
      a[f] = Q + 4;                                 // node 1 = root
      last[f] = a0[f];
    }
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
#pragma unroll
      for (int f = 0; f < FPT; ++f) {
        const float e = lds_f32(a[f] + f * (P2 * 4));
        const bool right = e <= xv[f];
        a[f] = (a[f] << 1) + (right ? nQ4 : nQ);
        last[f] = right ? e : last[f];
      }
    }
    const int64_t o = n * p.F + k0 + q * FPT;
    int pr[FPT];
    float dd[FPT];
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
      pr[f] = row0[f] + ((a[f] - Q) >> 2);
      dd[f] = xv[f] - last[f];
    }
    if (lp.vec_out && nf == FG) {
      *reinterpret_cast<int4*>(lp.piece + o) = make_int4(pr[0], pr[1], pr[2], pr[3]);
      *reinterpret_cast<float4*>(lp.dx + o) = make_float4(dd[0], dd[1], dd[2], dd[3]);
    } else {
#pragma unroll
      for (int f = 0; f < FPT; ++f)
        if (live[f]) { lp.piece[o + f] = pr[f]; lp.dx[o + f] = dd[f]; }
    }
  }
}

template <int NSTEP>
int launch_locate_tree(LocateTreeParams lp, hipStream_t st) {
  constexpr int FG = 16, BS = 512;
  Params& p = lp.f;
  size_t lds = ((static_cast<size_t>(FG) << NSTEP) + (FG / 4) * kTreeSkew) * sizeof(float);
  p.soff_offset = static_cast<int>(lds / sizeof(float));
  lds += (FG + 1) * sizeof(int);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fpwl_locate_tree_kernel<FG, NSTEP, BS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "fpwl_locate: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  p.nodes_per_block = tuned_moment_block(p.n, p.n_groups, lds, BS);
  const int64_t bx = ((p.n + p.nodes_per_block - 1) / p.nodes_per_block + 7) / 8 * 8;   // whole rounds of the 8 XCDs
  if (bx * p.n_groups > 0x7fffffffLL) return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl_locate: too many nodes for one launch");
  hipLaunchKernelGGL((fpwl_locate_tree_kernel<FG, NSTEP, BS>), dim3(static_cast<unsigned>(bx * p.n_groups)), dim3(BS), lds, st, lp);
  return gnan::check_launch("fpwl_locate_tree_kernel");
}
}  // namespace

// (called by gnan_fpwl_locate, csrc/fpwl_rows.hip; -1: not applicable, use the sorted-array kernel there)
int gnan_locate_tree(const gnan_fpwl_args* a, int32_t* piece, float* dx, hipStream_t st) {
  // (small batches: the sorted-array kernel's few large workgroups win — arxiv-shaped 0.102 against 0.141 ms; 10M nodes
  //  x 64 features: 3.4 ms against 1.6 ms the other way)
  if (a->max_pieces > 1024 || a->n < 262144 || (a->flags & GNAN_FPWL_LOCATE_SORTED)) return -1;
  LocateTreeParams lp;
  gnan_fpwl_args b = *a;
  b.features_per_group = 16;                        // the search has its own grouping, whatever the tables were planned for
  b.C = 1;
  lp.f = base_params(&b);
  lp.f.n_groups = (a->F + 15) / 16;
  lp.piece = piece; lp.dx = dx;
  lp.vec_x = a->F % 4 == 0 && a->x_stride % 4 == 0 && reinterpret_cast<uintptr_t>(a->x) % 16 == 0;
  lp.vec_out = a->F % 4 == 0 && reinterpret_cast<uintptr_t>(piece) % 16 == 0 && reinterpret_cast<uintptr_t>(dx) % 16 == 0;
  int nstep = 6;
  while ((1 << nstep) < a->max_pieces) ++nstep;
  switch (nstep) {
    case 6: return launch_locate_tree<6>(lp, st);
    case 7: return launch_locate_tree<7>(lp, st);
    case 8: return launch_locate_tree<8>(lp, st);
    case 9: return launch_locate_tree<9>(lp, st);
    default: return launch_locate_tree<10>(lp, st);
  }
}

namespace {
int moments_common(const gnan_fpwl_args* a, const float* grad, int64_t grad_stride, float* moments,
                   const double* scales, int64_t* moments_fixed, gnan_stream_t stream) {
  if (int rc = common_checks(a)) return rc;
  if (a->n == 0) return GNAN_OK;
  const int64_t gw = a->sum_features ? a->C : static_cast<int64_t>(a->F) * a->C;
  GNAN_REQUIRE(grad_stride >= gw, "fpwl_moments: grad row stride smaller than its width");
  MomentParams mp;
  mp.f = base_params(a);
  mp.g = grad; mp.g_stride = grad_stride; mp.M = moments;
  mp.scales = scales; mp.Mi = reinterpret_cast<unsigned long long*>(moments_fixed);
  mp.general_only = (a->flags & GNAN_FPWL_MOMENTS_GENERAL) != 0;
  mp.f.vec_x = a->features_per_group % 4 == 0 && a->F % 4 == 0 && a->x_stride % 4 == 0 &&
               reinterpret_cast<uintptr_t>(a->x) % 16 == 0;
  mp.vec_g = !a->sum_features && a->C == 1 && grad_stride % 4 == 0 && reinterpret_cast<uintptr_t>(grad) % 16 == 0;
  const size_t bin = moments_fixed ? sizeof(unsigned long long) : sizeof(float);
  const size_t pieces = static_cast<size_t>(a->max_group_pieces);
  const size_t lds = (pieces + 1) / 2 * 2 * sizeof(float) + pieces * static_cast<size_t>(bin_stride(a->C)) * bin;
  if (lds > 150 * 1024)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl_moments: %zu B of bins per feature group exceed LDS", lds);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // few features per group means many channels: tens of KB of bins per workgroup, so only a wide workgroup puts enough
  // waves on a CU (arxiv-shaped, C = 40: 91 KB of bins per feature -> one workgroup per CU)
  const bool wide = lds > 40 * 1024;
  switch (a->features_per_group) {
    case 1: return wide ? launch_moments<1, 1024>(mp, lds, st) : launch_moments<1, 256>(mp, lds, st);
    case 2: return wide ? launch_moments<2, 1024>(mp, lds, st) : launch_moments<2, 256>(mp, lds, st);
    case 4: return wide ? launch_moments<4, 1024>(mp, lds, st) : launch_moments<4, 256>(mp, lds, st);
    case 8: return launch_moments<8, 512>(mp, lds, st);
    default: return launch_moments<16, 512>(mp, lds, st);
  }
}
}  // namespace

extern "C" int gnan_fpwl_moments(const gnan_fpwl_args* a, const float* grad, int64_t grad_stride, float* moments,
                                 gnan_stream_t stream) {
  GNAN_REQUIRE(a == nullptr || a->n == 0 || (grad && moments), "fpwl_moments: null grad / moments");
  return moments_common(a, grad, grad_stride, moments, nullptr, nullptr, stream);
}

extern "C" int gnan_fpwl_moments_fixed(const gnan_fpwl_args* a, const float* grad, int64_t grad_stride,
                                       const double* scales, int64_t* moments, gnan_stream_t stream) {
  GNAN_REQUIRE(a == nullptr || a->n == 0 || (grad && moments && scales), "fpwl_moments_fixed: null grad / scales / moments");
  return moments_common(a, grad, grad_stride, nullptr, scales, moments, stream);
}

// csrc/fpwl_index.hip: the direct-index look-up (one channel, whole 16-feature groups, aligned rows)
bool gnan_index_applies(const gnan_fpwl_args* a);
int gnan_index_nodes_per_block(const gnan_fpwl_args* a);
int gnan_index_fwd(const gnan_fpwl_args* a, double* col_partial, hipStream_t st);
size_t gnan_index_sum_workspace_bytes(const gnan_fpwl_args* a);

extern "C" int gnan_fpwl_fwd(const gnan_fpwl_args* a, gnan_stream_t stream) {
  if (int rc = common_checks(a)) return rc;
  if (a->n == 0) return GNAN_OK;
  GNAN_REQUIRE(a->out && a->val && a->slope, "fpwl: null pointer");
  const int fg = a->features_per_group;
  const int64_t ow = a->sum_features ? a->C : static_cast<int64_t>(a->F) * a->C;
  GNAN_REQUIRE(a->out_stride >= ow, "fpwl: out row stride smaller than the output width");
  GNAN_REQUIRE(a->out_dtype == GNAN_F32 || a->out_dtype == GNAN_BF16, "fpwl: unknown out_dtype");
  if (fg == 16 && !(a->total && a->sum_features) && gnan_index_applies(a)) {
    hipStream_t ist = static_cast<hipStream_t>(stream);
    double* partial = nullptr;
    const int64_t ibx = (a->n + gnan_index_nodes_per_block(a) - 1) / gnan_index_nodes_per_block(a);
    if (a->total) {
      const size_t need = static_cast<size_t>(ibx) * a->F * sizeof(double);
      if (a->total_workspace == nullptr || a->total_workspace_bytes < need)
        return gnan::fail(GNAN_ERR_WORKSPACE, "fpwl: total workspace %zu B < required %zu B", a->total_workspace_bytes, need);
      partial = static_cast<double*>(a->total_workspace);
    }
    if (int rc = gnan_index_fwd(a, partial, ist)) return rc;
    if (partial) {
      hipLaunchKernelGGL(fpwl_total_kernel, dim3(a->F), dim3(256), 0, ist, partial, static_cast<int>(ibx), a->F, a->total);
      return gnan::check_launch("fpwl_total_kernel");
    }
    return GNAN_OK;
  }
  if (a->sum_total)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl: sum_total is written by the group-split direct-index feature sum only");
  const size_t lds = static_cast<size_t>(a->max_group_pieces) * (1 + 2 * static_cast<size_t>(table_stride(a->C))) * sizeof(float);
  Params p = base_params(a);
  auto aligned = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) % 16) == 0; };
  p.vec_x = fg % 4 == 0 && a->F % 4 == 0 && a->x_stride % 4 == 0 && aligned(a->x);
  p.vec_out = fg % 4 == 0 && a->F % 4 == 0 && a->out_stride % 4 == 0 && aligned(a->out);
  GNAN_REQUIRE(a->out_dtype == GNAN_F32 || a->out_dtype == GNAN_BF16, "fpwl: unknown out_dtype");
  if (a->total) {
    const int64_t bx = (p.n + p.nodes_per_block - 1) / p.nodes_per_block;
    const size_t need = static_cast<size_t>(bx) * a->F * sizeof(double);
    if (a->total_workspace == nullptr || a->total_workspace_bytes < need)
      return gnan::fail(GNAN_ERR_WORKSPACE, "fpwl: total workspace %zu B < required %zu B", a->total_workspace_bytes, need);
    p.col_partial = static_cast<double*>(a->total_workspace);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  // wide groups: 512-thread workgroups share one LDS image of the tables (A/B on C4: 1.92 -> 1.42 ms; 1024: 1.50 ms)
  // feature sum with many channels: the tables of a group are reloaded per pass of nodes, so the widest workgroup whose
  // [C][nodes] accumulators still fit LDS wins (arxiv-shaped, C = 40: one feature per group, 46 KB of tables per pass)
  if (a->sum_features && a->C > 1 && fg <= 4) {
    const int tpn = fg < 4 ? 1 : fg / 4;
    auto fits = [&](int bs) { return lds + static_cast<size_t>(a->C) * (bs / tpn) * sizeof(float) <= 150 * 1024; };
    const int bs = fits(1024) ? 1024 : fits(512) ? 512 : 256;
    switch (fg * 10000 + bs) {
      case 11024: return launch<1, 1024>(p, lds, st, a->total);
      case 10512: return launch<1, 512>(p, lds, st, a->total);
      case 21024: return launch<2, 1024>(p, lds, st, a->total);
      case 20512: return launch<2, 512>(p, lds, st, a->total);
      case 41024: return launch<4, 1024>(p, lds, st, a->total);
      case 40512: return launch<4, 512>(p, lds, st, a->total);
      default: break;
    }
  }
  switch (fg) {
    case 1: return launch<1, 256>(p, lds, st, a->total);
    case 2: return launch<2, 256>(p, lds, st, a->total);
    case 4: return launch<4, 256>(p, lds, st, a->total);
    case 8: return launch<8, 512>(p, lds, st, a->total);
    default: return launch<16, 512>(p, lds, st, a->total);
  }
}

extern "C" size_t gnan_fpwl_sum_workspace_bytes(const gnan_fpwl_args* a) {
  if (!a || a->n <= 0 || a->features_per_group != 16 || (a->total && a->sum_features)) return 0;
  return gnan_index_sum_workspace_bytes(a);
}

extern "C" size_t gnan_fpwl_total_workspace_bytes(const gnan_fpwl_args* a) {
  if (!a || a->n <= 0) return 0;
  const Params p = base_params(a);
  int64_t bx = (p.n + p.nodes_per_block - 1) / p.nodes_per_block;
  if (a->features_per_group == 16 && gnan_index_applies(a)) {       // whichever kernel serves the call: room for both
    const int64_t ibx = (a->n + gnan_index_nodes_per_block(a) - 1) / gnan_index_nodes_per_block(a);
    if (ibx > bx) bx = ibx;
  }
  return static_cast<size_t>(bx) * a->F * sizeof(double);
}
