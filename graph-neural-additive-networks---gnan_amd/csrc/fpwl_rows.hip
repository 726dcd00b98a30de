// Shape functions with SEVERAL output channels by table look-up, in two phases (gfx950).
//
// f_k : R -> R^C is piecewise linear with the same kinks for all C channels (the kinks are zero crossings of the hidden
// pre-activations, gnan_amd/pwl.py), so a look-up has a scalar part — which piece does x[n, k] fall into — and a vector part —
// add val[piece, :] + slope[piece, :] * (x - anchor[piece]) to the node's C accumulators (backward: add the node's C
// upstream gradients to the piece's moments).  fpwl.hip does both with thread = node: C > 1 then means C serial LDS round
// trips per look-up behind one search, node-strided (uncoalesced) reads of x and of the gradient, a table image that is
// reloaded per feature behind two barriers, and one workgroup per CU (arxiv-shaped, C = 40: look-up 2.3 ms, moments
// 4.5 ms for 21.8M look-ups).  Here
//   1. gnan_fpwl_locate        thread = (node, feature): piece[n, k] (row of the stacked tables) and dx[n, k] = x - anchor;
//                              coalesced along the features, anchors of a feature chunk in LDS;
//   2. gnan_fpwl_rows_fwd      lane = channel: a wavefront owns 64 / C' nodes (C' = C rounded up to a power of two), reads
//                              their (piece, dx) coalesced, broadcasts them inside the node's lanes and gathers the table
//                              rows from the (L2-resident) tables with coalesced C-float loads, 8 look-ups in flight;
//                              no LDS, no barriers, full occupancy;
//      gnan_fpwl_rows_moments  lane = channel as well: workgroup = (feature, node block), 64-bit fixed-point bins of that
//                              feature in LDS; a wavefront reads the upstream-gradient rows of its nodes coalesced and adds
//                              them to bins[piece][:] — the C lanes hit C consecutive bins: conflict-free.
// Same arithmetic per term as fpwl.hip (fmaf(slope, dx, val), added in feature order; fixed_bits() terms), replaces
// GNAN.py:57-62 / its autograd for models with several output channels (node classification: C = classes).
#include "common.hpp"

#include <cstdlib>

namespace {

using gnan::kWave;

// global-memory pointer that keeps its address space through an empty asm statement (a generic pointer would turn the
// loads into flat loads, which also count against the LDS counter and serialise with the LDS atomics)
typedef __attribute__((address_space(1))) const float gfloat;
__device__ __forceinline__ gfloat* as_global(const float* ptr) { return (gfloat*)ptr; }

// (same as fpwl.hip)
__device__ __forceinline__ unsigned long long fixed_bits(float v, double s) {
  const double d = fma(static_cast<double>(v), s, 6755399441055744.0);
  return static_cast<unsigned long long>(__double_as_longlong(d)) - 0x4338000000000000ull;
}

// ---------------------------------------------------------------------------------------------
// phase 1: piece[n, k] = off[k] + #{ j >= 1 : anchor[off[k] + j] <= x[n, k] },  dx[n, k] = x[n, k] - anchor[piece[n, k]]
// ---------------------------------------------------------------------------------------------
struct LocateParams {
  const float* x;
  int64_t n, x_stride;
  int F;
  const int32_t* off;
  const float* anchor;
  int step0;             // largest power of two <= max breakpoints per feature (0 if none)
  int chunk;             // features per chunk (their anchors fit the LDS image)
  int nodes_per_block;
  int32_t* piece;        // [n, F]
  float* dx;             // [n, F]
};

__global__ __launch_bounds__(1024) void fpwl_locate_kernel(const LocateParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int k0 = blockIdx.y * p.chunk;
  const int nf = p.F - k0 < p.chunk ? p.F - k0 : p.chunk;
  const int base = p.off[k0];
  const int tot = p.off[k0 + nf] - base;
  int* s_off = reinterpret_cast<int*>(smem);           // [nf + 1]
  float* anchor_l = smem + ((nf + 1 + 3) & ~3);
  for (int i = tid; i <= nf; i += 1024) s_off[i] = p.off[k0 + i] - base;
  for (int i = tid; i < tot; i += 1024) anchor_l[i] = p.anchor[base + i];
  __syncthreads();
  const int64_t n_lo = static_cast<int64_t>(blockIdx.x) * p.nodes_per_block;
  const int64_t n_hi = n_lo + p.nodes_per_block < p.n ? n_lo + p.nodes_per_block : p.n;
  // a wavefront walks the chunk's features of NB nodes at a time (coalesced reads of x, coalesced stores of piece / dx):
  // the NB loads are in flight together and the NB searches interleave — one node at a time paid a full memory latency
  // per node (0.16 ms for 21.8M look-ups)
  constexpr int NB = 4;
  for (int64_t n = n_lo + static_cast<int64_t>(wave) * NB; n < n_hi; n += 16 * NB) {   // 16 waves share one image
    for (int f = lane; f < nf; f += kWave) {
      float xv[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) xv[u] = n + u < n_hi ? p.x[(n + u) * p.x_stride + k0 + f] : 0.f;
      const int po = s_off[f], pn = s_off[f + 1] - po - 1;
      int idx[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) idx[u] = 0;
      for (int step = p.step0; step > 0; step >>= 1) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int j = idx[u] + step;
          const int jj = j <= pn ? j : 0;              // out of range -> harmless in-range read
          const float a = anchor_l[po + jj];
          idx[u] = (j <= pn && a <= xv[u]) ? j : idx[u];
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        if (n + u < n_hi) {
          const int64_t o = (n + u) * p.F + k0 + f;
          p.piece[o] = base + po + idx[u];
          p.dx[o] = xv[u] - anchor_l[po + idx[u]];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// phase 2, forward:  out[n, c] = sum_k val[piece[n,k], c] + slope[piece[n,k], c] * dx[n,k]       (sum_features)
//                    out[n, k*C + c] = val[...] + slope[...] * dx[n,k]                            (per feature)
// ---------------------------------------------------------------------------------------------
struct RowsParams {
  int64_t n;
  int F, C;
  const int32_t* piece;
  const float* dx;
  const float* val;
  const float* slope;
  int sum_features;
  float* out;
  int64_t out_stride;
};

// One node per wavefront (C > 32): the (piece, dx) pair of look-up j is wave-uniform, so it is moved to scalar registers
// (v_readlane with a constant lane) and the row address arithmetic runs on the scalar unit — with __shfl every lane
// repeated it (~16 vector instructions per look-up: the kernel was bound by vector issue, 0.61 ms for 21.8M look-ups).
template <int CP2, bool FULL>
struct Bcast {
  static __device__ __forceinline__ int i(int v, int slot, int j) { return __shfl(v, slot * CP2 + j); }
  static __device__ __forceinline__ float f(float v, int slot, int j) { return __shfl(v, slot * CP2 + j); }
};
template <>
struct Bcast<64, true> {
  static __device__ __forceinline__ int i(int v, int, int j) { return __builtin_amdgcn_readlane(v, j); }
  static __device__ __forceinline__ float f(float v, int, int j) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
  }
};

template <int CP2, bool FULL, bool SUMF>
__device__ __forceinline__ void rows_chunk(const RowsParams& p, int k0, int m, int pv, float dv, int slot, int c, int cs,
                                           bool live, float& acc, float* orow) {
  // no branch around the loads: every lane reads a valid address (column cs = min(c, C - 1) of a valid row), so the
  // compiler keeps the loads of the unrolled look-ups in flight together; only stores are predicated
#pragma unroll(CP2 < 8 ? CP2 : (CP2 == 64 && FULL ? 16 : 8))
  for (int j = 0; j < CP2; ++j) {
    if (!FULL && j >= m) break;
    const int t = Bcast<CP2, FULL>::i(pv, slot, j);
    const float dd = Bcast<CP2, FULL>::f(dv, slot, j);
    float y;
    if constexpr (CP2 == 64 && FULL) {
      gfloat* vr = as_global(p.val + static_cast<int64_t>(t) * p.C);
      gfloat* sr = as_global(p.slope + static_cast<int64_t>(t) * p.C);
      asm volatile("" : "+s"(vr), "+s"(sr));             // row bases stay scalar: saddr + lane offset
      y = fmaf(sr[cs], dd, vr[cs]);
    } else {
      // several nodes per wavefront: the piece differs between the node slots, but the element index fits 32 bits
      // (gnan_fpwl_rows_fwd checks F * max_pieces * C), so it is one multiply-add on top of the scalar table bases
      const unsigned o = static_cast<unsigned>(t) * static_cast<unsigned>(p.C) + static_cast<unsigned>(cs);
      y = fmaf(as_global(p.slope)[o], dd, as_global(p.val)[o]);
    }
    if constexpr (SUMF) acc += y;
    else if (live) orow[static_cast<int64_t>(k0 + j) * p.C + c] = y;
  }
}

template <int CP2, bool SUMF>
__global__ __launch_bounds__(256) void fpwl_rows_fwd_kernel(const RowsParams p) {
  constexpr int NPW = kWave / CP2;                     // nodes per wavefront
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int slot = lane / CP2;
  const int64_t n = (static_cast<int64_t>(blockIdx.x) * 4 + wave) * NPW + slot;
  const bool node_ok = n < p.n;
  if (CP2 == kWave && !node_ok) return;                // one node per wavefront: nothing to do (wave-uniform)
  const int64_t row = (node_ok ? n : 0) * p.F;
  float* orow = p.out + (node_ok ? n : 0) * p.out_stride;
  // more than 64 channels (CP2 == 64 only): the wavefront walks the node's look-ups once per chunk of 64 channels
  auto pass = [&](const int cb) {
    const int c = cb + lane % CP2;
    const bool live = node_ok && c < p.C;
    const int cs = c < p.C ? c : p.C - 1;
    float acc = 0.f;
    for (int k0 = 0; k0 < p.F; k0 += CP2) {
      // the node's CP2 lanes read its next CP2 (piece, dx) pairs with one coalesced load each and broadcast them in turn
      const int kk = k0 + lane % CP2;
      int pv = 0;
      float dv = 0.f;
      if (node_ok && kk < p.F) { pv = p.piece[row + kk]; dv = p.dx[row + kk]; }
      const int m = p.F - k0 < CP2 ? p.F - k0 : CP2;
      if (m == CP2) rows_chunk<CP2, true, SUMF>(p, k0, m, pv, dv, slot, c, cs, live, acc, orow);
      else rows_chunk<CP2, false, SUMF>(p, k0, m, pv, dv, slot, c, cs, live, acc, orow);
    }
    if (SUMF && live) orow[c] = acc;
  };
  if (p.C <= kWave) {
    pass(0);                                            // (the common case keeps its constant channel offset)
  } else {
    for (int cb = 0; cb < p.C; cb += kWave) pass(cb);
  }
}

// ---------------------------------------------------------------------------------------------
// phase 2, backward:  M[t, 0, c] += g[n, c]     M[t, 1, c] += g[n, c] * dx[n, k]      for t = piece[n, k]
// (g = grad[n, c] with sum_features, grad[n, k*C + c] without), in 64-bit fixed point (fpwl.hip)
// ---------------------------------------------------------------------------------------------
struct RowsMomentParams {
  int64_t n;
  int F, C;
  const int32_t* off;
  const int32_t* piece;
  const float* dx;
  const float* g;
  int64_t g_stride;
  int sum_features;
  const double* scales;
  unsigned long long* Mi;   // [T, 2, C]
  int nodes_per_block;
  int cc;                   // channels per workgroup (a chunk of the C channels whose bins fit LDS; <= 64)
  int64_t wgs_per_chunk;    // workgroups of one channel chunk
};

template <int CP2>
__global__ __launch_bounds__(1024) void fpwl_rows_moments_kernel(const RowsMomentParams p) {
  constexpr int NPW = kWave / CP2;
  constexpr int NWAVES = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned long long bins[];   // [pieces of feature k][2 C + 1]
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);              // (wave-uniform: lets node offsets live in SGPRs)
  const int c = lane % CP2, slot = lane / CP2;
  // (id % 8) = XCD; inside an XCD the features of ONE node block run back to back, so the block's gradient rows (read once
  // per feature: F times) stay in that XCD's L2 instead of costing a fabric request per row and feature
  // (32-bit unsigned arithmetic on wave-uniform values, pinned to scalar registers: the row addresses below build on them)
  const unsigned wpc = static_cast<unsigned>(p.wgs_per_chunk);
  const unsigned chunk = __builtin_amdgcn_readfirstlane(blockIdx.x / wpc);
  const unsigned id = __builtin_amdgcn_readfirstlane(blockIdx.x - chunk * wpc);
  const int k = __builtin_amdgcn_readfirstlane(static_cast<int>((id >> 3) % static_cast<unsigned>(p.F)));
  const int64_t nb = static_cast<int64_t>(__builtin_amdgcn_readfirstlane(((id >> 3) / static_cast<unsigned>(p.F)) * 8 + (id & 7)));
  const int c_lo = static_cast<int>(chunk) * p.cc;     // this workgroup's channels: [c_lo, c_lo + C) of the p.C
  const int C = p.C - c_lo < p.cc ? p.C - c_lo : p.cc;
  const int Rb = 2 * p.cc + 1;                         // odd stride: rows of consecutive pieces start on different banks
  const int base = p.off[k];
  const int tot = p.off[k + 1] - base;
  const int64_t n_lo = nb * p.nodes_per_block;
  if (n_lo >= p.n) return;                             // (uniform: before any barrier)
  for (int i = tid; i < tot * Rb; i += 1024) bins[i] = 0ull;
  __syncthreads();
  const double s0 = p.scales[0], s1 = p.scales[1];
  const int64_t n_hi = n_lo + p.nodes_per_block < p.n ? n_lo + p.nodes_per_block : p.n;
  const int64_t gbase = (p.sum_features ? 0 : static_cast<int64_t>(k) * p.C) + c_lo;
  const int cs = c < C ? c : C - 1;
  // a wavefront takes 64 consecutive nodes at a time: lane j reads (piece, dx) of node n0 + j (feature k), then the lanes
  // of a slot walk the nodes NPW at a time, lane = channel: coalesced gradient rows, consecutive bins
  for (int64_t n0 = n_lo + static_cast<int64_t>(wave) * kWave; n0 < n_hi; n0 += NWAVES * kWave) {
    int pv = 0;
    float dv = 0.f;
    if (n0 + lane < n_hi) {
      pv = p.piece[(n0 + lane) * p.F + k] - base;
      dv = p.dx[(n0 + lane) * p.F + k];
    }
    const int m = static_cast<int>(n_hi - n0 < kWave ? n_hi - n0 : kWave);
    constexpr int U = NPW >= 8 ? 8 : 16;               // gradient rows in flight per lane (the loop is latency-bound otherwise)
    if constexpr (NPW == 1) {
      // one node per step: its (piece, dx) pair is wave-uniform -> scalar registers (v_readlane), row addresses on the
      // scalar unit; the vector unit is left with two conversions and two LDS atomics per step (it was the bound)
#pragma unroll
      for (int i0 = 0; i0 < kWave; i0 += U) {
        if (i0 >= m) break;
        float gv[U], dd[U];
        int t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u < m ? i0 + u : m - 1;   // tail: a valid node, its terms are dropped below
          t[u] = __builtin_amdgcn_readlane(pv, i0 + u);
          dd[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dv), i0 + u));
          gfloat* gr = as_global(p.g + (n0 + i) * p.g_stride + gbase);
          asm volatile("" : "+s"(gr));
          gv[u] = gr[cs];
        }
        if (c < C) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (i0 + u < m) {
              unsigned long long* b = bins + static_cast<int64_t>(t[u]) * Rb + c;
              atomicAdd(b, fixed_bits(gv[u], s0));
              atomicAdd(b + p.cc, fixed_bits(gv[u] * dd[u], s1));
            }
          }
        }
      }
    } else {
      for (int i0 = 0; i0 < m; i0 += U * NPW) {
        float gv[U], dd[U];
        int t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * NPW + slot;           // < 64: kWave is a multiple of U * NPW
          t[u] = __shfl(pv, i);
          dd[u] = __shfl(dv, i);
          gv[u] = p.g[(n0 + (i < m ? i : m - 1)) * p.g_stride + gbase + cs];
        }
        if (c < C) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (i0 + u * NPW + slot < m) {
              unsigned long long* b = bins + static_cast<int64_t>(t[u]) * Rb + c;
              atomicAdd(b, fixed_bits(gv[u], s0));
              atomicAdd(b + p.cc, fixed_bits(gv[u] * dd[u], s1));
            }
          }
        }
      }
    }
  }
  __syncthreads();
  unsigned long long* out = p.Mi + static_cast<int64_t>(base) * 2 * p.C;
  for (int i = tid; i < tot * 2 * C; i += 1024) {
    const int t = i / (2 * C), r = i - t * 2 * C, m = r / C, c2 = r - m * C;
    const unsigned long long v = bins[t * Rb + m * p.cc + c2];
    if (v != 0ull) atomicAdd(out + static_cast<int64_t>(t) * 2 * p.C + m * p.C + c_lo + c2, v);
  }
}

// 34..42 channels (ogbn-arxiv: 40 classes): a lane per channel leaves 22-30 of a wavefront's 64 lanes idle in a kernel bound by
// vector issue.  Here a lane takes a PAIR of channels and a wavefront three nodes per step (3 x ceil(C / 2) <= 63 lanes busy):
// one 8-byte gradient load, four conversions, four LDS atomics per lane and step.  Same terms, same integer bins: the same bits
// as fpwl_rows_moments_kernel<64>.
__global__ __launch_bounds__(1024) void fpwl_rows_moments_pairs_kernel(const RowsMomentParams p) {
  constexpr int NWAVES = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned long long bins[];   // [pieces of feature k][2 C + 1]
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int GS = (p.cc + 1) / 2;                       // lanes per node
  const int slot = lane / GS, c0 = (lane - slot * GS) * 2;
  const bool lane_on = slot < 3;
  const unsigned wpc = static_cast<unsigned>(p.wgs_per_chunk);
  const unsigned chunk = __builtin_amdgcn_readfirstlane(blockIdx.x / wpc);
  const unsigned id = __builtin_amdgcn_readfirstlane(blockIdx.x - chunk * wpc);
  const int k = __builtin_amdgcn_readfirstlane(static_cast<int>((id >> 3) % static_cast<unsigned>(p.F)));
  const int64_t nb = static_cast<int64_t>(__builtin_amdgcn_readfirstlane(((id >> 3) / static_cast<unsigned>(p.F)) * 8 + (id & 7)));
  const int c_lo = static_cast<int>(chunk) * p.cc;
  const int C = p.C - c_lo < p.cc ? p.C - c_lo : p.cc;
  const int Rb = 2 * p.cc + 1;
  const int base = p.off[k];
  const int tot = p.off[k + 1] - base;
  const int64_t n_lo = nb * p.nodes_per_block;
  if (n_lo >= p.n) return;
  for (int i = tid; i < tot * Rb; i += 1024) bins[i] = 0ull;
  __syncthreads();
  const double s0 = p.scales[0], s1 = p.scales[1];
  const int64_t n_hi = n_lo + p.nodes_per_block < p.n ? n_lo + p.nodes_per_block : p.n;
  const int64_t gbase = (p.sum_features ? 0 : static_cast<int64_t>(k) * p.C) + c_lo;
  const bool has0 = lane_on && c0 < C, has1 = lane_on && c0 + 1 < C;
  const int ca = has0 ? c0 : 0, cb = has1 ? c0 + 1 : ca;
  // rows whose channel pairs are 8-byte aligned and whole (C even): float2 loads
  const bool pair_load = (C & 1) == 0 && (p.g_stride & 1) == 0 && (gbase & 1) == 0 && (reinterpret_cast<uintptr_t>(p.g) & 7) == 0;
  constexpr int U = 8;
  for (int64_t n0 = n_lo + static_cast<int64_t>(wave) * kWave; n0 < n_hi; n0 += NWAVES * kWave) {
    int pv = 0;
    float dv = 0.f;
    if (n0 + lane < n_hi) {
      pv = p.piece[(n0 + lane) * p.F + k] - base;
      dv = p.dx[(n0 + lane) * p.F + k];
    }
    const int m = static_cast<int>(n_hi - n0 < kWave ? n_hi - n0 : kWave);
    for (int i0 = 0; i0 < m; i0 += U * 3) {
      float ga[U], gb[U], dd[U];
      int t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * 3 + (lane_on ? slot : 0);
        const int ic = i < m ? i : m - 1;              // (past the end: a valid node, its terms are dropped below)
        t[u] = __shfl(pv, ic);
        dd[u] = __shfl(dv, ic);
        const float* row = p.g + (n0 + ic) * p.g_stride + gbase;
        if (pair_load) {                                 // (uniform) one 8-byte load per lane and node
          const float2 v = *reinterpret_cast<const float2*>(row + ca);
          ga[u] = v.x;
          gb[u] = v.y;
        } else {
          ga[u] = row[ca];
          gb[u] = row[cb];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (i0 + u * 3 + slot < m) {
          unsigned long long* b = bins + static_cast<int64_t>(t[u]) * Rb + c0;
          if (has0) {
            atomicAdd(b, fixed_bits(ga[u], s0));
            atomicAdd(b + p.cc, fixed_bits(ga[u] * dd[u], s1));
          }
          if (has1) {
            atomicAdd(b + 1, fixed_bits(gb[u], s0));
            atomicAdd(b + p.cc + 1, fixed_bits(gb[u] * dd[u], s1));
          }
        }
      }
    }
  }
  __syncthreads();
  unsigned long long* out = p.Mi + static_cast<int64_t>(base) * 2 * p.C;
  for (int i = tid; i < tot * 2 * C; i += 1024) {
    const int t = i / (2 * C), r = i - t * 2 * C, m = r / C, c2 = r - m * C;
    const unsigned long long v = bins[t * Rb + m * p.cc + c2];
    if (v != 0ull) atomicAdd(out + static_cast<int64_t>(t) * 2 * p.C + m * p.C + c_lo + c2, v);
  }
}

int cp2_of(int C) {
  int cp2 = 8;
  while (cp2 < C) cp2 <<= 1;
  return cp2;
}

int rows_checks(const gnan_fpwl_args* a, const int32_t* piece, const float* dx, const char* who) {
  GNAN_REQUIRE(a != nullptr, "%s: null args", who);
  GNAN_REQUIRE(a->n >= 0 && a->F >= 1 && a->C >= 1, "%s: bad sizes", who);
  if (a->C > 4096) return gnan::fail(GNAN_ERR_UNSUPPORTED, "%s: at most 4096 output channels (got %d)", who, a->C);
  if (a->n == 0) return GNAN_OK;
  GNAN_REQUIRE(piece && dx && a->off, "%s: null pointer", who);
  GNAN_REQUIRE(a->n * static_cast<int64_t>(a->F) < (1LL << 40), "%s: n * F too large", who);
  return GNAN_OK;
}

}  // namespace

extern "C" size_t gnan_fpwl_locate_bytes(const gnan_fpwl_args* a) {
  if (!a || a->n <= 0 || a->F <= 0) return 0;
  return static_cast<size_t>(a->n) * static_cast<size_t>(a->F) * 4;   // bytes of EACH of piece and dx
}

int gnan_locate_tree(const gnan_fpwl_args* a, int32_t* piece, float* dx, hipStream_t st);   // csrc/fpwl.hip

extern "C" int gnan_fpwl_locate(const gnan_fpwl_args* a, int32_t* piece, float* dx, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "fpwl_locate: null args");
  GNAN_REQUIRE(a->n >= 0 && a->F >= 1, "fpwl_locate: bad sizes");
  if (a->n == 0) return GNAN_OK;
  GNAN_REQUIRE(a->x && a->off && a->anchor && piece && dx, "fpwl_locate: null pointer");
  GNAN_REQUIRE(a->x_stride >= a->F, "fpwl_locate: x row stride smaller than F");
  GNAN_REQUIRE(a->max_pieces >= 1, "fpwl_locate: max_pieces must be >= 1");
  // the tree search of the C == 1 kernels (skewed breadth-first trees, thread = (node, 4 features)) where it applies
  if (int rc = gnan_locate_tree(a, piece, dx, static_cast<hipStream_t>(stream)); rc != -1) return rc;
  LocateParams p;
  p.x = a->x; p.n = a->n; p.x_stride = a->x_stride; p.F = a->F; p.off = a->off; p.anchor = a->anchor;
  int step0 = 0;
  while ((step0 ? step0 * 2 : 1) <= a->max_pieces - 1) step0 = step0 ? step0 * 2 : 1;
  p.step0 = step0;
  // features per chunk: their anchors (at most max_pieces each) fit a 64-KiB image
  int chunk = (16 * 1024 - 8) / a->max_pieces;
  if (chunk < 1) return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl_locate: %d pieces per feature exceed the LDS image", a->max_pieces);
  if (chunk > a->F) chunk = a->F;
  if (chunk > 4096) chunk = 4096;
  p.chunk = chunk;
  const int n_chunks = (a->F + chunk - 1) / chunk;
  // blocks of 256..4096 nodes: enough workgroups to fill the chip, few enough that the image load is amortised
  int64_t npb = (a->n * n_chunks + 1023) / 1024;
  npb = npb < 1024 ? 1024 : (npb > 8192 ? 8192 : (npb + 15) / 16 * 16);
  p.nodes_per_block = static_cast<int>(npb);
  p.piece = piece; p.dx = dx;
  const int64_t bx = (a->n + npb - 1) / npb;
  if (bx > 0x7fffffffLL || n_chunks > 65535) return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl_locate: grid too large");
  const size_t lds = (static_cast<size_t>((chunk + 1 + 3) & ~3) + static_cast<size_t>(chunk) * a->max_pieces) * sizeof(float);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fpwl_locate_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "fpwl_locate: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(fpwl_locate_kernel, dim3(static_cast<unsigned>(bx), static_cast<unsigned>(n_chunks)), dim3(1024), lds,
                     static_cast<hipStream_t>(stream), p);
  return gnan::check_launch("fpwl_locate_kernel");
}

namespace {
template <int CP2>
int launch_rows_fwd(const RowsParams& p, hipStream_t st) {
  constexpr int NPW = kWave / CP2;
  const int64_t blocks = (p.n + 4 * NPW - 1) / (4 * NPW);
  if (blocks > 0x7fffffffLL) return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl_rows_fwd: too many nodes for one launch");
  if (p.sum_features) {
    hipLaunchKernelGGL((fpwl_rows_fwd_kernel<CP2, true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL((fpwl_rows_fwd_kernel<CP2, false>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, p);
  }
  return gnan::check_launch("fpwl_rows_fwd_kernel");
}

template <int CP2>
int launch_rows_moments(const RowsMomentParams& p, size_t lds, dim3 grid, hipStream_t st) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fpwl_rows_moments_kernel<CP2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "fpwl_rows_moments: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL((fpwl_rows_moments_kernel<CP2>), grid, dim3(1024), lds, st, p);
  return gnan::check_launch("fpwl_rows_moments_kernel");
}
}  // namespace

extern "C" int gnan_fpwl_rows_fwd(const gnan_fpwl_args* a, const int32_t* piece, const float* dx, gnan_stream_t stream) {
  if (int rc = rows_checks(a, piece, dx, "fpwl_rows_fwd")) return rc;
  if (a->n == 0) return GNAN_OK;
  GNAN_REQUIRE(a->val && a->slope && a->out, "fpwl_rows_fwd: null pointer");
  GNAN_REQUIRE(a->out_dtype == GNAN_F32, "fpwl_rows_fwd: fp32 output only");
  GNAN_REQUIRE(a->max_pieces >= 1, "fpwl_rows_fwd: max_pieces must be >= 1");
  if (static_cast<int64_t>(a->F) * a->max_pieces * a->C >= (1LL << 32))
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl_rows_fwd: tables of F * max_pieces * C >= 2^32 floats");
  const int64_t ow = a->sum_features ? a->C : static_cast<int64_t>(a->F) * a->C;
  GNAN_REQUIRE(a->out_stride >= ow, "fpwl_rows_fwd: out row stride smaller than the output width");
  RowsParams p;
  p.n = a->n; p.F = a->F; p.C = a->C; p.piece = piece; p.dx = dx; p.val = a->val; p.slope = a->slope;
  p.sum_features = a->sum_features; p.out = static_cast<float*>(a->out); p.out_stride = a->out_stride;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // (one pass over all features: launching the features in chunks whose table rows stay resident in a 4-MiB L2 — and
  // carrying the accumulators through memory between the launches — measured 0.46 ms against 0.42 ms: the kernel is bound
  // by vector issue, not by the table rows' L2 misses)
  switch (cp2_of(a->C)) {
    case 8: return launch_rows_fwd<8>(p, st);
    case 16: return launch_rows_fwd<16>(p, st);
    case 32: return launch_rows_fwd<32>(p, st);
    default: return launch_rows_fwd<64>(p, st);
  }
}

extern "C" int gnan_fpwl_rows_moments_fixed(const gnan_fpwl_args* a, const int32_t* piece, const float* dx, const float* grad,
                                            int64_t grad_stride, const double* scales, int64_t* moments,
                                            gnan_stream_t stream) {
  if (int rc = rows_checks(a, piece, dx, "fpwl_rows_moments")) return rc;
  if (a->n == 0) return GNAN_OK;
  GNAN_REQUIRE(grad && scales && moments, "fpwl_rows_moments: null grad / scales / moments");
  const int64_t gw = a->sum_features ? a->C : static_cast<int64_t>(a->F) * a->C;
  GNAN_REQUIRE(grad_stride >= gw, "fpwl_rows_moments: grad row stride smaller than its width");
  GNAN_REQUIRE(a->max_pieces >= 1, "fpwl_rows_moments: max_pieces must be >= 1");
  // pieces of one feature: at most max_pieces, and at most max_group_pieces when the caller states it (> 0)
  const int per_feature = (a->max_group_pieces > 0 && a->max_group_pieces < a->max_pieces) ? a->max_group_pieces : a->max_pieces;
  // channels per workgroup: all of them if they fit a wavefront and their bins fit LDS, else equal chunks that do
  int cc = a->C < kWave ? a->C : kWave;
  auto bins_bytes = [&](int ch) { return static_cast<size_t>(per_feature) * (2 * static_cast<size_t>(ch) + 1) * sizeof(unsigned long long); };
  while (cc > 1 && bins_bytes(cc) > 150 * 1024) --cc;
  const int n_chunks = (a->C + cc - 1) / cc;
  cc = (a->C + n_chunks - 1) / n_chunks;               // equal chunks
  const size_t lds = bins_bytes(cc);
  if (lds > 150 * 1024)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl_rows_moments: %zu B of bins per feature and channel exceed LDS", lds);
  RowsMomentParams p;
  p.n = a->n; p.F = a->F; p.C = a->C; p.off = a->off; p.piece = piece; p.dx = dx; p.g = grad; p.g_stride = grad_stride;
  p.sum_features = a->sum_features; p.scales = scales; p.Mi = reinterpret_cast<unsigned long long*>(moments);
  // node blocks: each (feature, block) workgroup zeroes and flushes the feature's bins, so blocks should be large; their
  // gradient rows should fit an XCD's L2 with room to spare (<= 2 MiB); and F * blocks workgroups should fill the chip
  int64_t npb = a->n * static_cast<int64_t>(a->F) / 2048;
  const int64_t l2_rows = (2 << 20) / (static_cast<int64_t>(a->C) * 4);
  if (npb > l2_rows) npb = l2_rows;
  npb = npb < 1024 ? 1024 : (npb > 16384 ? 16384 : (npb + 1023) / 1024 * 1024);
  p.nodes_per_block = static_cast<int>(npb);
  const int64_t bx = ((a->n + npb - 1) / npb + 7) / 8 * 8;          // whole rounds of the 8 XCDs
  p.cc = cc;
  p.wgs_per_chunk = bx * a->F;
  if (p.wgs_per_chunk * n_chunks > 0x7fffffffLL) return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl_rows_moments: too many nodes for one launch");
  const dim3 grid(static_cast<unsigned>(p.wgs_per_chunk * n_chunks));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (cc > 32 && 3 * ((cc + 1) / 2) <= kWave && !(a->flags & GNAN_FPWL_ROWS_MOMENTS_LANE_PER_CHANNEL)) {
    // 33..42 channels: a pair of channels per lane, three nodes per step (see fpwl_rows_moments_pairs_kernel)
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fpwl_rows_moments_pairs_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "fpwl_rows_moments: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(fpwl_rows_moments_pairs_kernel, grid, dim3(1024), lds, st, p);
    return gnan::check_launch("fpwl_rows_moments_pairs_kernel");
  }
  switch (cp2_of(cc)) {
    case 8: return launch_rows_moments<8>(p, lds, grid, st);
    case 16: return launch_rows_moments<16>(p, lds, grid, st);
    case 32: return launch_rows_moments<32>(p, lds, grid, st);
    default: return launch_rows_moments<64>(p, lds, grid, st);
  }
}
