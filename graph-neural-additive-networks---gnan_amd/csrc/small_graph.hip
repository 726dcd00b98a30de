// The whole forward of a SMALL dense-coded graph in ONE launch (GNAN.py:146-172 / models.py:358-384, post-rho
// normalisation): shape functions of all features summed per node, rho on the D distinct distances, the rho-weighted
// shell-normalised aggregation and the graph read-out.  Graph-level tasks (trainer.py:23-86 with batch_size = 1) see a
// 30-node graph per step: the general path spends seven launches on it (two weight packs, two matrix-core MLP launches,
// a chunk sum, the aggregation, the read-out sum) and every one of them is shorter than the gap between two launches.
//
//   workgroup k < F   : f_k on all nodes.  Four waves share a block of 64 nodes (lane = node): every wave computes a
//                       quarter of each hidden layer's units into LDS columns, weights are wave-uniform (scalar loads);
//                       the C outputs of the block go to part[k, node, :] in the workspace.
//   workgroup F       : lut[d, :] = rho(u_d), u_d = float32(1 / (1 + d)) for d < D - 1, u_{D-1} = 0 (lane = d).
//   the LAST workgroup to finish (a counter in the workspace; __threadfence before the increment) adds the features in
//   order, S[j, :] = sum_k part[k, j, :], aggregates Y[i, c] = sum_j lut[code(i, j), c] / max(cnt(i, code), 1) * S[j, c]
//   (a thread per row and channel over LDS-resident tables, neighbours in ascending order) and sums the rows for the graph
//   read-out.  Fixed orders throughout: bit-reproducible.
// S and lut are written out as well — the backward pass of the general path takes them from there.
#include "small_graph_body.hpp"

namespace {

using namespace gnan_small;

struct SmallParams {
  const float* x;
  int64_t x_stride;
  int n, F;
  Mlp f, r;
  const uint8_t* code;
  int D;
  const int32_t* cnt;
  int64_t cnt_stride;
  float* S;
  float* lut;
  float* Y;
  float* Ysum;
  float* part;          // [F, n, C]
  unsigned* counter;    // zero before the first launch; the last workgroup zeroes it again
  int pre_rho;          // GNAN.py:65-67: rho at u_d / max(cnt[i, d], 1) — n * D arguments, lut is the rows' table [n, D]; workgroups
                        // F .. F + ceil(n D / 64) - 1 take 64 arguments each
  int rho_raw;          // rho's inputs are the raw hop counts d (batched_pyg_main.py:151) instead of 1 / (1 + d)
  int rest_zero;        // pairs beyond the last listed hop carry weight 0 (the -1 mask, batched_pyg_main.py:155-156) instead of rho(0)
};

// Many small graphs in one launch (batched_pyg_main.py:133-184): blockIdx.y = graph; nodes, codes, node sums, outputs, partial
// sums and arrival counters of graph g are runs of the batch's arrays.
struct BatchParams {
  SmallParams base;              // pointers of graph 0; n unused
  const int32_t* node_off;       // [G + 1]
  const int64_t* code_off;       // [G + 1] byte offsets of the graphs' [n_g][n_g] code blocks
};


// NB: node blocks of 64 the graph may have (1: n <= 64, the tables fit the activation columns as before; 2: n <= 128, the
// aggregation's tables need 64 KB).  Config 2's graphs have 30 nodes on average and 3 % of them
// more than 64; 128 covers 99.98 %.
template <int NB>
__device__ __forceinline__ void small_graph_body(const SmallParams& p, float* cols) {
  constexpr int kNodes = 64 * NB;
  // cols [small_cols_floats(NB)]: activation columns; the aggregation's tables later — node sums, outputs, rho table, row
  // weights [n][64], hop codes [n][n]
  __shared__ __attribute__((aligned(16))) float weights[kWeightFloats];
  __shared__ unsigned s_last;
  float* col_a = cols;
  float* col_b = cols + kMaxH * kWave;
  float* s_S = cols;                                 // [n, C]      (phase 3: the columns are dead by then)
  float* s_Y = cols + kNodes * kMaxC;                // [n, C]
  float* s_lut = cols + 2 * kNodes * kMaxC;          // [D, rho C]
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int k = blockIdx.x;
  float out[kMaxC];
  const MlpLds wl_ = carve(weights);
  // What the aggregation will read from memory that does not depend on the other workgroups — hop codes and shell sizes — is
  // requested NOW, by every workgroup (any of them may be the last), into registers: the loads fly while the MLP runs.  A cold
  // load costs more than a microsecond on this part and the kernel is a chain of them; this takes two links out.
  constexpr int kWordsPer = kNodes * kNodes / 4 / (kWaves * kWave);            // 4 code words per thread (NB = 2: 16)
  constexpr int kCntPer = kNodes * kWave / (kWaves * kWave);                    // 16 shell sizes per thread (D <= 64; NB = 2: 32)
  const bool fast = p.D <= kWave && p.r.C == 1;      // per-row weights lut[d] / max(cnt[i, d], 1) fit LDS
  const int code_words = p.n * p.n / 4;
  uint32_t pre_code[kWordsPer];
  int pre_cnt[kCntPer];
  {
    const uint32_t* cw = reinterpret_cast<const uint32_t*>(p.code);
#pragma unroll
    for (int t = 0; t < kWordsPer; ++t) {
      const int i = threadIdx.x + t * 256;
      pre_code[t] = i < code_words ? cw[i] : 0u;
    }
#pragma unroll
    for (int t = 0; t < kCntPer; ++t) {
      const int e = threadIdx.x + t * 256;
      pre_cnt[t] = (fast && !p.pre_rho && p.cnt && e < p.n * p.D) ? p.cnt[(e / p.D) * p.cnt_stride + e % p.D] : 1;
    }
  }
  if (k < p.F) {
    stage_weights(p.f, k, wl_);
    float xv[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {                   // (all node blocks' inputs requested before the first MLP)
      const int j = b * kWave + lane;
      xv[b] = j < p.n ? p.x[static_cast<int64_t>(j) * p.x_stride + k] : 0.f;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int j = b * kWave + lane;
      if (b * kWave < p.n) {                         // (uniform)
        mlp_block(p.f, wl_, xv[b], col_a, col_b, lane, wave, out);
        if (wave == 0 && j < p.n)
          for (int c = 0; c < p.f.C; ++c) p.part[(static_cast<int64_t>(k) * p.n + j) * p.f.C + c] = out[c];
      }
    }
  } else if (p.pre_rho) {
    // rho at the n * D normalised distances (GNAN.py:66: torch.div(node_distances, normalization_matrix), IEEE division): this
    // workgroup's 64 of them
    stage_weights(p.r, 0, wl_);
    const int e = (k - p.F) * kWave + lane;
    const bool valid = e < p.n * p.D;
    const int i = valid ? e / p.D : 0, d = valid ? e % p.D : 0;
    const int q = valid ? p.cnt[i * p.cnt_stride + d] : 1;
    const float ud = d < p.D - 1 ? 1.0f / (static_cast<float>(d) + 1.0f) : 0.f;
    mlp_block(p.r, wl_, ud / static_cast<float>(q > 1 ? q : 1), col_a, col_b, lane, wave, out);
    if (wave == 0 && valid) p.lut[e] = out[0];
  } else {
    stage_weights(p.r, 0, wl_);
    for (int d0 = 0; d0 < p.D; d0 += kWave) {
      const int d = d0 + lane;
      const float u = d < p.D - 1 ? (p.rho_raw ? static_cast<float>(d) : 1.0f / (static_cast<float>(d) + 1.0f)) : 0.f;     // graph.hop_inputs
      mlp_block(p.r, wl_, u, col_a, col_b, lane, wave, out);
      if (wave == 0 && d < p.D)
        for (int c = 0; c < p.r.C; ++c) p.lut[d * p.r.C + c] = (p.rest_zero && d == p.D - 1) ? 0.f : out[c];
    }
  }
  // ---- join: the last workgroup to arrive finishes the graph --------------------------------------------------------------
  if (!last_to_arrive(p.counter, gridDim.x, &s_last)) return;
  const int C = p.f.C, Cr = p.r.C, n = p.n;
  // (the fence above made the other workgroups' results visible to this one: plain loads from here on.  Every table the
  // aggregation reads — node sums, rho table, hop codes, the rows' weights — is staged in LDS; the per-row loop never waits
  // for memory)
  uint8_t* s_code = reinterpret_cast<uint8_t*>(cols + 2 * kNodes * kMaxC + 256 * kMaxC + kNodes * kWave);         // [n, n]
  float* s_w = cols + 2 * kNodes * kMaxC + 256 * kMaxC;                                                            // [n, 64]
  // the other workgroups' results: node sums (eight feature terms in flight at a time) and the rho table — one more batch
  float lut_v[kMaxC * 256 / (kWaves * kWave)];       // 8 table entries per thread at most
#pragma unroll
  for (int t = 0; t < kMaxC; ++t) {
    const int e = threadIdx.x + t * 256;
    lut_v[t] = e < p.D * Cr ? p.lut[e] : 0.f;
  }
  if (p.pre_rho) {                                   // (uniform) the rows' weights are the rho workgroups' outputs themselves
#pragma unroll
    for (int t = 0; t < kCntPer; ++t) {
      const int e = threadIdx.x + t * 256;
      pre_cnt[t] = e < n * p.D ? __float_as_int(p.lut[e]) : 0;
    }
  }
  // node sums: every (feature, node, channel) term is fetched by a thread of its own — all loads in flight at once — into the
  // (now free) weight area, then a thread per (node, channel) adds the features in order.  (A thread per (node, channel)
  // walking its F terms itself left 30 of 256 threads with two rounds of loads: 2 us of a 20-us launch.)
  {
    const int nc = n * C;
    // staging: the part of the columns behind s_S / s_Y (the aggregation's tables move in there only after this phase): 7 168
    // floats for one node block, 14 336 for two — 14 features per pass at 128 nodes x 8 channels (the weight image held 4)
    float* stage = cols + 2 * kNodes * kMaxC;
    constexpr int kStage = small_cols_floats(NB) - 2 * kNodes * kMaxC;
    const int per_chunk = kStage / nc;
    for (int e = threadIdx.x; e < nc; e += blockDim.x) s_S[e] = 0.f;
    for (int k0 = 0; k0 < p.F; k0 += per_chunk) {
      const int kn = p.F - k0 < per_chunk ? p.F - k0 : per_chunk;
      __syncthreads();
      // (a 30-node, one-channel graph has 450 terms: two per thread.  A 100-node graph of the batched variant's 8 channels has
      //  12 000: sixteen loads in flight per thread, not four — each batch of loads is one cold round trip)
      for (int e0 = 0; e0 < kn * nc; e0 += 16 * 256) {
        float v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int e = e0 + t * 256 + threadIdx.x;
          v[t] = e < kn * nc ? p.part[static_cast<int64_t>(k0) * nc + e] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int e = e0 + t * 256 + threadIdx.x;
          if (e < kn * nc) stage[e] = v[t];
        }
      }
      __syncthreads();
      for (int e = threadIdx.x; e < nc; e += blockDim.x) {
        float sum = s_S[e];
#pragma unroll 8
        for (int kk = 0; kk < kn; ++kk) sum += stage[kk * nc + e];
        s_S[e] = sum;
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nc; e += blockDim.x) p.S[e] = s_S[e];
  }
#pragma unroll
  for (int t = 0; t < kMaxC; ++t) {
    const int e = threadIdx.x + t * 256;
    if (e < p.D * Cr) s_lut[e] = lut_v[t];
  }
  {
    uint32_t* dw = reinterpret_cast<uint32_t*>(s_code);
#pragma unroll
    for (int t = 0; t < kWordsPer; ++t) {
      const int i = threadIdx.x + t * 256;
      if (i < code_words) dw[i] = pre_code[t];
    }
    for (int i = code_words * 4 + threadIdx.x; i < n * n; i += blockDim.x) s_code[i] = p.code[i];
  }
  __syncthreads();
  if (fast) {
#pragma unroll
    for (int t = 0; t < kCntPer; ++t) {
      const int e = threadIdx.x + t * 256;
      if (e < n * p.D) {
        const float l = s_lut[e % p.D];
        s_w[(e / p.D) * kWave + e % p.D] = p.pre_rho ? __int_as_float(pre_cnt[t])
                                                     : (p.cnt ? l / static_cast<float>(pre_cnt[t] > 1 ? pre_cnt[t] : 1) : l);
      }
    }
    __syncthreads();
  }
  // one thread per (row, channel), neighbours in ascending order
  for (int e = threadIdx.x; e < n * C; e += blockDim.x) {
    const int i = e / C, c = e % C;
    const int cr = Cr == 1 ? 0 : c;
    const uint8_t* codes = s_code + i * n;
    float acc = 0.f;
    // (three loops, not one with the case distinction inside: a branch per neighbour kept the compiler from overlapping the
    //  neighbours' LDS reads — 40 of a 128-node, 8-channel graph's 59 us)
    if (fast) {
      const float* wrow = s_w + i * kWave;
#pragma unroll 8
      for (int j = 0; j < n; ++j) {                    // (unrolled: the code -> weight reads of eight neighbours overlap)
        int d = codes[j];
        d = d < p.D - 1 ? d : p.D - 1;
        acc = fmaf(wrow[d], s_S[j * C + c], acc);
      }
    } else if (p.cnt == nullptr) {
      const float* lcol = s_lut + cr;
#pragma unroll 8
      for (int j = 0; j < n; ++j) {
        int d = codes[j];
        d = d < p.D - 1 ? d : p.D - 1;
        acc = fmaf(lcol[d * Cr], s_S[j * C + c], acc);
      }
    } else {
      for (int j = 0; j < n; ++j) {
        int d = codes[j];
        d = d < p.D - 1 ? d : p.D - 1;
        const int q = p.cnt[i * p.cnt_stride + d];
        acc = fmaf(s_lut[d * Cr + cr] / static_cast<float>(q > 1 ? q : 1), s_S[j * C + c], acc);
      }
    }
    s_Y[e] = acc;
    if (p.Y) p.Y[e] = acc;
  }
  __syncthreads();
  if (p.Ysum && static_cast<int>(threadIdx.x) < C) {
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < n; ++i) s += s_Y[i * C + threadIdx.x];
    p.Ysum[threadIdx.x] = s;
  }
  if (threadIdx.x == 0) *p.counter = 0u;
}

template <int NB>
__global__ __launch_bounds__(256) void small_graph_kernel(const SmallParams p) {
  extern __shared__ __attribute__((aligned(16))) float cols[];               // small_cols_floats(NB) floats
  small_graph_body<NB>(p, cols);
}

template <int NB>
__global__ __launch_bounds__(256) void small_graph_batch_kernel(const BatchParams bp) {
  extern __shared__ __attribute__((aligned(16))) float cols[];
  const int g = blockIdx.y;
  SmallParams p = bp.base;
  const int64_t o = bp.node_off[g];
  p.n = static_cast<int>(bp.node_off[g + 1] - o);
  p.x += o * p.x_stride;
  p.code += bp.code_off[g];
  p.S += o * p.f.C;
  if (p.cnt) p.cnt += o * p.cnt_stride;
  if (p.Y) p.Y += o * p.f.C;
  if (p.Ysum) p.Ysum += static_cast<int64_t>(g) * p.f.C;
  p.lut += static_cast<int64_t>(g) * p.D * p.r.C;
  p.part += o * p.F * p.f.C;
  p.counter += 32 * g;                                // a 128-byte line of its own per graph (arrivals of graphs sharing a line queue up behind each other)
  small_graph_body<NB>(p, cols);
}

// ---------------------------------------------------------------------------------------------
// The backward pass of the same small graph in ONE launch: gradients of every parameter of f and rho from the gradient of the
// node outputs (or of their sum).  No workgroup waits for another: workgroup k < F forms the operand gradient
//   dS[j, c] = sum_i lut[code(i, j)] / max(cnt(i, code), 1) * dY[i, c]
// itself (n^2 multiply-adds: cheaper than a launch that would share it) and then runs gnan_fmlp_bwd's body on feature k with
// dS read from LDS; workgroup F forms the table gradient
//   dlut[d] = sum_i 1 / max(cnt(i, d), 1) * sum_{j : code(i, j) == d} < dY[i, :], S[j, :] >
// (a row per wave at a time, the neighbours' dot products binned by hop code in lane-private LDS columns, float64 across
// rows — dense_lut_grad_kernel's scheme) and runs the same body on rho with the D distances as its inputs.
// Covers what the default models produce: one rho channel, D <= 64 shells, n <= 64 nodes.  Fixed orders: bit-reproducible.
// ---------------------------------------------------------------------------------------------
struct SmallBwdParams {
  const float* x;
  int64_t x_stride;
  int n, F;
  gnan_bwd::Weights f, r;
  int f_mid, r_mid;          // L == 3
  const uint8_t* code;
  int D;
  const int32_t* cnt;
  int64_t cnt_stride;
  const float* S;
  const float* lut;
  const float* dY;
  const float* dYsum;
  int pre_rho;               // lut is the rows' table [n, D] = rho(u_d / max(cnt[i, d], 1)) (GNAN.py:65-67)
  // pre-rho: rho's n * D arguments are split by rows over rho_groups workgroups (blockIdx F .. F + rho_groups - 1), rows_per rows
  // each; with more than one group every group leaves its gradients in its slot of part (kRhoSlot floats: W2's 64 x 64, then
  // five vectors of 64) and the last to arrive adds them in group order
  int rho_groups, rows_per;
  float* part;
  unsigned* counter;
  int rho_raw;               // batched variant: rho's inputs are the raw hop counts (batched_pyg_main.py:151)
  int rest_zero;             // ... and pairs beyond the last listed hop carry weight 0: no gradient through lut[D - 1]
  int rho_c;                 // rho's channels: 1, or f.C (one per output channel: the batched variant's rho, batched_pyg_main.py:
                             // 125-131; lut is [D, C] then; no shell normalisation, not with pre_rho)
};

// Many small graphs in one launch (the backward of small_graph_batch_kernel): blockIdx.y = graph.  Graph g's workgroups leave the
// gradients of f and rho that ITS nodes contribute in slab g of `grads` (the twelve tensors back to back, f's then rho's);
// batch_grad_reduce_kernel adds the slabs in graph order.
struct BatchBwdParams {
  SmallBwdParams base;           // pointers of graph 0; n unused; gradient pointers = slab 0
  const int32_t* node_off;
  const int64_t* code_off;
  int64_t slab;                  // floats per graph
  int dy_per_graph;              // dYsum is [G, C] (graph read-out) rather than dY [N, C]
};

constexpr int kBinStride = kWave + 1;
constexpr int kRhoSlot = kMaxH * kMaxH + 5 * kMaxH, kRhoGroupsMax = 12;       // 12 slots = 207 KB: inside a captured step's 256-KB scratch

// NB as in small_graph_kernel (2: n <= 128 — hop codes and the row-weight / bin area take 50 KB: dynamic LDS).
template <int C, int NB>
__device__ __forceinline__ void small_graph_bwd_body(const SmallBwdParams& p, float* dyn) {
  constexpr int kNodes = 64 * NB;
  constexpr int kUFloats = kNodes * kWave > 2 * kWave * kBinStride ? kNodes * kWave : 2 * kWave * kBinStride;
  __shared__ gnan_bwd::RedBuffer red;                                           // dyn: s_u | s_code
  float* s_u = dyn;                                   // row weights [n][64] (features) | bins [waves][D][n | 1] (rho)
  uint8_t* s_code = reinterpret_cast<uint8_t*>(dyn + kUFloats);                 // [n][n]
  float* s_dl = dyn + kUFloats + kNodes * kNodes / 4;                           // pre-rho only: table gradient [n][D]
  __shared__ float s_dY[kNodes * kMaxC];
  __shared__ float s_S[kNodes * kMaxC];
  __shared__ float s_g[kNodes * kMaxC];              // dS [n, C]  |  dlut [D]
  __shared__ double s_part[kWaves][kWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int k = blockIdx.x, n = p.n;
  const bool is_rho = k >= p.F;
  // ---- everything from memory in one batch ----------------------------------------------------------------------------------
  {
    constexpr int kWordsPer = kNodes * kNodes / 4 / 256;
    constexpr int kRowsPer = kNodes * kMaxC / 256;     // (node, channel) entries per thread
    constexpr int kCntPer = kNodes * kWave / 256;      // shell sizes per thread
    const int words = n * n / 4;
    const uint32_t* cw = reinterpret_cast<const uint32_t*>(p.code);
    uint32_t v[kWordsPer];
#pragma unroll
    for (int t = 0; t < kWordsPer; ++t) {
      const int i = threadIdx.x + t * 256;
      v[t] = i < words ? cw[i] : 0u;
    }
    float gy[kRowsPer], sv[kRowsPer];
    int q[kCntPer];
#pragma unroll
    for (int t = 0; t < kRowsPer; ++t) {
      const int e = threadIdx.x + t * 256;
      gy[t] = e < n * C ? (p.dY ? p.dY[e] : p.dYsum[e % C]) : 0.f;
      sv[t] = (is_rho && e < n * C) ? p.S[e] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < kCntPer; ++t) {
      const int e = threadIdx.x + t * 256;
      if (p.pre_rho) q[t] = (!is_rho && e < n * p.D) ? __float_as_int(p.lut[e]) : 0;       // the row's weight itself
      else q[t] = (!is_rho && p.cnt && e < n * p.D) ? p.cnt[(e / p.D) * p.cnt_stride + e % p.D] : 1;
    }
    const float l = lane < p.D ? p.lut[lane] : 0.f;       // (e % D below: re-read per entry from this register is not possible; see s_g)
    uint32_t* dw = reinterpret_cast<uint32_t*>(s_code);
#pragma unroll
    for (int t = 0; t < kWordsPer; ++t) {
      const int i = threadIdx.x + t * 256;
      if (i < words) dw[i] = v[t];
    }
    for (int i = words * 4 + threadIdx.x; i < n * n; i += 256) s_code[i] = p.code[i];
#pragma unroll
    for (int t = 0; t < kRowsPer; ++t) {
      const int e = threadIdx.x + t * 256;
      if (e < n * C) { s_dY[e] = gy[t]; s_S[e] = sv[t]; }
    }
    __syncthreads();                                       // (s_g doubles as dS / dlut below: nobody reads it yet)
    if (wave == 0 && lane < p.D) s_g[lane] = l;            // the rho table, for a moment
    __syncthreads();
    if (p.rho_c > 1) {                                     // (uniform) one table column per channel: [D][C] in place of the rows' weights
      if (!is_rho)
        for (int e = threadIdx.x; e < p.D * C; e += 256) s_u[e] = p.lut[e];
    } else if (!is_rho) {
#pragma unroll
      for (int t = 0; t < kCntPer; ++t) {
        const int e = threadIdx.x + t * 256;
        if (e < n * p.D) {
          const float lv = s_g[e % p.D];
          s_u[(e / p.D) * kWave + e % p.D] = p.pre_rho ? __int_as_float(q[t])
                                                       : (p.cnt ? lv / static_cast<float>(q[t] > 1 ? q[t] : 1) : lv);
        }
      }
    }
    __syncthreads();
  }
  const gnan_bwd::Drop nodrop{0u, 1.f, 0ull};
  if (!is_rho) {
    // ---- operand gradient of every node, then this feature's parameter gradients ---------------------------------------------
    for (int e = threadIdx.x; e < n * C; e += 256) {
      const int j = e / C, c = e % C;
      float acc = 0.f;
      if (p.rho_c > 1) {
#pragma unroll 8
        for (int i = 0; i < n; ++i) {
          int d = s_code[i * n + j];
          d = d < p.D - 1 ? d : p.D - 1;
          acc = fmaf(s_u[d * C + c], s_dY[i * C + c], acc);
        }
      } else {
#pragma unroll 8
        for (int i = 0; i < n; ++i) {
          int d = s_code[i * n + j];
          d = d < p.D - 1 ? d : p.D - 1;
          acc = fmaf(s_u[i * kWave + d], s_dY[i * C + c], acc);
        }
      }
      s_g[e] = acc;
    }
    __syncthreads();
    auto x_of = [&](int64_t node) { return p.x[node * p.x_stride + k]; };
    auto g_of = [&](int64_t node, int c) { return s_g[node * C + c]; };
    if (p.f_mid) gnan_bwd::feature_grads<C, true>(p.f, k, 0, n, 0, nodrop, x_of, g_of, red);
    else gnan_bwd::feature_grads<C, false>(p.f, k, 0, n, 0, nodrop, x_of, g_of, red);
    return;
  }
  if (p.rho_c > 1) {
    // ---- a rho of one channel per output channel (the batched variant): dlut[d, c] = sum over the pairs (i, j) of hop code d of
    // dY[i, c] S[j, c]  (the binning of the one-channel path would need C passes of its two barriers per row).
    // Rows in chunks of R: thread (row of the chunk, channel) bins the node sums of its row's neighbours by hop code into a
    // column of its own (no other thread touches it), then thread (d, c) adds dY[i, c] * bin over the chunk's rows in order:
    // n + R steps per thread and chunk.  (A thread per (d, c) walking ALL pairs was n^2 steps: 0.36 ms for one 128-node graph.)
    const int DC = p.D * C;
    const int Dp = p.D | 1;                                // odd stride: the threads' columns start in different banks
    if (p.dY == nullptr) {
      // the graph read-out's gradient is the same for every row: dlut[d, c] = g[c] * sum_j S[j, c] * #{ i : code(i, j) == d }.
      // A thread per column j counts its rows by hop code into a column of its own (n steps), a thread per (d, c) adds over j.
      int* cc = reinterpret_cast<int*>(s_u);               // [n][Dp]
      for (int j = threadIdx.x; j < n; j += 256) {
        int* col = cc + j * Dp;
        for (int d = 0; d < p.D; ++d) col[d] = 0;
        for (int i = 0; i < n; ++i) {
          int du = s_code[i * n + j];
          du = du < p.D - 1 ? du : p.D - 1;
          col[du] += 1;
        }
      }
      __syncthreads();
      for (int e = threadIdx.x; e < DC; e += 256) {
        const int d = e / C, c = e % C;
        float a = 0.f;
#pragma unroll 8
        for (int j = 0; j < n; ++j) a = fmaf(s_S[j * C + c], static_cast<float>(cc[j * Dp + d]), a);
        s_g[e] = (p.rest_zero && d == p.D - 1) ? 0.f : a * s_dY[c];
      }
      __syncthreads();
    } else {
    int R = kUFloats / (C * Dp);
    R = R < 256 / C ? R : 256 / C;
    R = R < 1 ? 1 : R;
    float* bins = s_u;                                     // [R * C][Dp]
    const int tr = static_cast<int>(threadIdx.x) / C, tc = static_cast<int>(threadIdx.x) % C;
    float acc[(kWave * kMaxC + 255) / 256];                // this thread's entries e = tid, tid + 256, ... of dlut [D][C]
#pragma unroll
    for (int u = 0; u < (kWave * kMaxC + 255) / 256; ++u) acc[u] = 0.f;
    for (int i0 = 0; i0 < n; i0 += R) {
      const int i = i0 + tr;
      if (tr < R && i < n) {
        float* col = bins + (tr * C + tc) * Dp;
        for (int d = 0; d < p.D; ++d) col[d] = 0.f;
        const uint8_t* codes = s_code + i * n;
        int j = 0;
        for (; j + 8 <= n; j += 8) {
          int dd[8];
          float sv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { dd[u] = codes[j + u]; sv[u] = s_S[(j + u) * C + tc]; }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int du = dd[u] < p.D - 1 ? dd[u] : p.D - 1;
            col[du] += sv[u];
          }
        }
        for (; j < n; ++j) {
          int du = codes[j];
          du = du < p.D - 1 ? du : p.D - 1;
          col[du] += s_S[j * C + tc];
        }
      }
      __syncthreads();
      const int rows = n - i0 < R ? n - i0 : R;
#pragma unroll
      for (int u = 0; u < (kWave * kMaxC + 255) / 256; ++u) {
        const int e = static_cast<int>(threadIdx.x) + u * 256;
        if (e < DC) {
          const int d = e / C, c = e % C;
          float a = acc[u];
#pragma unroll 4
          for (int r = 0; r < rows; ++r) a = fmaf(s_dY[(i0 + r) * C + c], bins[(r * C + c) * Dp + d], a);
          acc[u] = a;
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < (kWave * kMaxC + 255) / 256; ++u) {
      const int e = static_cast<int>(threadIdx.x) + u * 256;
      if (e < DC) s_g[e] = (p.rest_zero && e / C == p.D - 1) ? 0.f : acc[u];
    }
    __syncthreads();
    }
    auto uc_of = [&](int64_t node) {
      return node < p.D - 1 ? (p.rho_raw ? static_cast<float>(node) : 1.0f / (static_cast<float>(node) + 1.0f)) : 0.f;
    };
    auto gc_of = [&](int64_t node, int c) { return s_g[node * C + c]; };
    if (p.r_mid) gnan_bwd::feature_grads<C, true>(p.r, 0, 0, p.D, 0, nodrop, uc_of, gc_of, red);
    else gnan_bwd::feature_grads<C, false>(p.r, 0, 0, p.D, 0, nodrop, uc_of, gc_of, red);
    return;
  }
  // ---- table gradient, then rho's parameter gradients ---------------------------------------------------------------------------
  // (this workgroup is the kernel's critical path: a row per wave and round, all four waves where their bins fit — they do
  // for the graphs this kernel is for — and bin rows only as long as there are neighbours)
  const int stride = n | 1;                                // odd: lane d's walk along bin row d does not collide with lane d + 1's
  int nw = 1;                                              // waves that bin: as many as have room for their [D][stride] bins
  while (nw < 4 && 2 * nw * p.D * stride <= kUFloats) nw *= 2;
  float* bins = s_u + wave * p.D * stride;
  double acc = 0.0;                                        // lane d: dlut[d] over this wave's rows
  const int i_lo = p.pre_rho ? (k - p.F) * p.rows_per : 0;         // this workgroup's rows (all of them unless pre-rho splits)
  const int i_hi = p.pre_rho ? (i_lo + p.rows_per < n ? i_lo + p.rows_per : n) : n;
  for (int r = 0; i_lo + nw * r < i_hi; ++r) {
    const int i = i_lo + nw * r + wave;
    const bool live = wave < nw && i < i_hi;
    if (live) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int j = b * kWave + lane;                   // the neighbour this lane takes in this block
        if (j < n) {
          for (int d = 0; d < p.D; ++d) bins[d * stride + j] = 0.f;
          int d = s_code[i * n + j];
          d = d < p.D - 1 ? d : p.D - 1;
          float v = 0.f;
#pragma unroll
          for (int c = 0; c < C; ++c) v = fmaf(s_dY[i * C + c], s_S[j * C + c], v);
          bins[d * stride + j] = v;                       // (a lane owns its columns: no other lane wrote them)
        }
      }
    }
    __syncthreads();
    if (live && lane < p.D) {
      float sum = 0.f;
#pragma unroll 8
      for (int l = 0; l < n; ++l) sum += bins[lane * stride + l];        // (unrolled: eight reads in flight, not one)
      if (p.pre_rho) {
        s_dl[(i - i_lo) * p.D + lane] = sum;               // every (row, shell) is an argument of rho of its own
      } else {
        if (p.cnt) {
          const int c = p.cnt[i * p.cnt_stride + lane];
          sum *= 1.f / static_cast<float>(c > 1 ? c : 1);
        }
        acc += static_cast<double>(sum);
      }
    }
    __syncthreads();
  }
  if (p.pre_rho) {
    // rho's parameter gradients over this workgroup's arguments u_d / max(cnt[i, d], 1) (the bins are dead: the arguments take
    // their place)
    const int count = (i_hi - i_lo) * p.D;
    for (int e = threadIdx.x; e < count; e += 256) {
      const int d = e % p.D;
      const int c = p.cnt[(i_lo + e / p.D) * p.cnt_stride + d];
      const float ud = d < p.D - 1 ? 1.0f / (static_cast<float>(d) + 1.0f) : 0.f;
      s_u[e] = ud / static_cast<float>(c > 1 ? c : 1);
    }
    __syncthreads();
    auto ua_of = [&](int64_t e) { return s_u[e]; };
    auto ga_of = [&](int64_t e, int) { return s_dl[e]; };
    gnan_bwd::Weights w = p.r;
    if (p.rho_groups > 1) {                                // partial gradients into this group's slot
      float* slot = p.part + static_cast<int64_t>(k - p.F) * kRhoSlot;
      w.d_w_mid = slot;
      w.d_w_first = slot + kMaxH * kMaxH; w.d_b_first = p.r.d_b_first ? slot + kMaxH * kMaxH + kMaxH : nullptr;
      w.d_b_mid = p.r.d_b_mid ? slot + kMaxH * kMaxH + 2 * kMaxH : nullptr; w.d_w_last = slot + kMaxH * kMaxH + 3 * kMaxH;
      w.d_b_last = p.r.d_b_last ? slot + kMaxH * kMaxH + 4 * kMaxH : nullptr;
    }
    if (p.r_mid) gnan_bwd::feature_grads<1, true>(w, 0, 0, count, 0, nodrop, ua_of, ga_of, red);
    else gnan_bwd::feature_grads<1, false>(w, 0, 0, count, 0, nodrop, ua_of, ga_of, red);
    if (p.rho_groups == 1) return;
    // ---- join of the rho groups: the last to arrive adds the partial gradients in group order ------------------------------------
    __shared__ unsigned s_last_rho;
    if (!last_to_arrive(p.counter, static_cast<unsigned>(p.rho_groups), &s_last_rho)) return;
    // (a thread's loads of all groups' terms are issued together, 16 bytes each: a load -> add loop over the groups would
    //  wait for every load in turn — 160 cold round trips per thread, longer than the whole rest of the launch)
    const int H = p.r.H;
    float* const vec[5] = {p.r.d_w_first, p.r.d_b_first, p.r_mid ? p.r.d_b_mid : nullptr, p.r.d_w_last, p.r.d_b_last};
    float* const dW2 = p.r_mid ? p.r.d_w_mid : nullptr;
    for (int e4 = threadIdx.x; e4 < kRhoSlot / 4; e4 += 256) {
      float4 v[kRhoGroupsMax];
#pragma unroll
      for (int g = 0; g < kRhoGroupsMax; ++g) {
        const int gg = g < p.rho_groups ? g : 0;
        v[g] = *reinterpret_cast<const float4*>(p.part + static_cast<int64_t>(gg) * kRhoSlot + 4 * e4);
      }
      float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < kRhoGroupsMax; ++g) {
        if (g < p.rho_groups) { sum[0] += v[g].x; sum[1] += v[g].y; sum[2] += v[g].z; sum[3] += v[g].w; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = 4 * e4 + u;
        if (idx < kMaxH * kMaxH) {
          if (dW2 && idx < H * H) dW2[idx] = sum[u];
        } else {
          const int t = (idx - kMaxH * kMaxH) / kMaxH, e = (idx - kMaxH * kMaxH) % kMaxH;
          if (vec[t] && e < (t == 4 ? 1 : H)) vec[t][e] = sum[u];
        }
      }
    }
    if (threadIdx.x == 0) *p.counter = 0u;
    return;
  }
  s_part[wave][lane] = wave < nw ? acc : 0.0;
  __syncthreads();
  if (wave == 0 && lane < p.D)
    s_g[lane] = (p.rest_zero && lane == p.D - 1) ? 0.f
                : static_cast<float>(((s_part[0][lane] + s_part[1][lane]) + s_part[2][lane]) + s_part[3][lane]);
  __syncthreads();
  auto u_of = [&](int64_t node) {
    return node < p.D - 1 ? (p.rho_raw ? static_cast<float>(node) : 1.0f / (static_cast<float>(node) + 1.0f)) : 0.f;
  };
  auto gl_of = [&](int64_t node, int) { return s_g[node]; };
  if (p.r_mid) gnan_bwd::feature_grads<1, true>(p.r, 0, 0, p.D, 0, nodrop, u_of, gl_of, red);
  else gnan_bwd::feature_grads<1, false>(p.r, 0, 0, p.D, 0, nodrop, u_of, gl_of, red);
}

template <int C, int NB>
__global__ __launch_bounds__(256) void small_graph_bwd_kernel(const SmallBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  small_graph_bwd_body<C, NB>(p, dyn);
}

template <int C, int NB>
__global__ __launch_bounds__(256) void small_graph_batch_bwd_kernel(const BatchBwdParams bp) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  const int g = blockIdx.y;
  SmallBwdParams p = bp.base;
  const int64_t o = bp.node_off[g];
  p.n = static_cast<int>(bp.node_off[g + 1] - o);
  p.x += o * p.x_stride;
  p.code += bp.code_off[g];
  p.S += o * C;
  if (p.cnt) p.cnt += o * p.cnt_stride;
  p.lut += static_cast<int64_t>(g) * p.D * p.rho_c;
  if (bp.dy_per_graph) p.dYsum += static_cast<int64_t>(g) * C;
  else p.dY += o * C;
  // graph g's slab of gradients: the same tensor layout, bp.slab floats further on
  const int64_t by = static_cast<int64_t>(g) * bp.slab;
  gnan_bwd::Weights* ws[2] = {&p.f, &p.r};
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    gnan_bwd::Weights& w = *ws[t];
    w.d_w_first += by; w.d_w_last += by;
    if (w.d_b_first) w.d_b_first += by;
    if (w.d_w_mid) w.d_w_mid += by;
    if (w.d_b_mid) w.d_b_mid += by;
    if (w.d_b_last) w.d_b_last += by;
  }
  small_graph_bwd_body<C, NB>(p, dyn);
}

// out[e] = sum_g slabs[g][e], graphs in order, eight graphs' terms requested together; the slab is the twelve gradient tensors
// back to back and `dst` lists where each lives (offsets ascending, a NULL destination = a tensor that does not exist)
struct ReduceSeg { int64_t at[13]; float* dst[12]; };
__global__ __launch_bounds__(256) void batch_grad_reduce_kernel(const float* __restrict__ slabs, int64_t slab, int n_graphs,
                                                              const ReduceSeg seg) {
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < slab; e += static_cast<int64_t>(gridDim.x) * 256) {
    float sum = 0.f;
    for (int g0 = 0; g0 < n_graphs; g0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = g0 + u < n_graphs ? slabs[static_cast<int64_t>(g0 + u) * slab + e] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v[u];
    }
    int t = 0;
#pragma unroll
    for (int u = 1; u < 12; ++u) t += e >= seg.at[u] ? 1 : 0;
    if (seg.dst[t]) seg.dst[t][e - seg.at[t]] = sum;
  }
}

template <int NB>
constexpr size_t small_bwd_dyn_bytes(bool pre_rho) {
  constexpr int nodes = 64 * NB;
  constexpr int u = nodes * kWave > 2 * kWave * kBinStride ? nodes * kWave : 2 * kWave * kBinStride;
  return static_cast<size_t>(u) * sizeof(float) + static_cast<size_t>(nodes) * nodes +
         (pre_rho ? static_cast<size_t>(nodes) * kWave * sizeof(float) : 0);               // + the table gradient [n][D]
}

template <int C>
int launch_small_bwd(const SmallBwdParams& p, hipStream_t st) {
  if (p.n <= 64) {
    hipLaunchKernelGGL((small_graph_bwd_kernel<C, 1>), dim3(static_cast<unsigned>(p.F + p.rho_groups)), dim3(256),
                       small_bwd_dyn_bytes<1>(p.pre_rho != 0), st, p);
  } else {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&small_graph_bwd_kernel<C, 2>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       static_cast<int>(small_bwd_dyn_bytes<2>(true)));
    if (attr != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "small_graph_bwd: hipFuncSetAttribute: %s", hipGetErrorString(attr));
    hipLaunchKernelGGL((small_graph_bwd_kernel<C, 2>), dim3(static_cast<unsigned>(p.F + p.rho_groups)), dim3(256),
                       small_bwd_dyn_bytes<2>(p.pre_rho != 0), st, p);
  }
  return gnan::check_launch("small_graph_bwd_kernel");
}

}  // namespace

extern "C" size_t gnan_small_graph_workspace_bytes(int32_t n, int32_t F, int32_t C) {
  return 16 + static_cast<size_t>(F) * n * C * sizeof(float);      // counter (own 16 bytes) | part [F, n, C]
}

extern "C" int gnan_small_graph_fwd(const gnan_small_graph_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "small_graph: null args");
  GNAN_REQUIRE(a->n >= 1 && a->F >= 1 && a->D >= 1, "small_graph: bad sizes n=%d F=%d D=%d", a->n, a->F, a->D);
  if (a->n > kMaxNodes || a->D > 256 || !mlp_ok(&a->f, kMaxC) || !mlp_ok(&a->rho, kMaxC) ||
      (a->rho.C != 1 && a->rho.C != a->f.C))
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "small_graph: covers n <= %d nodes, D <= 256 shells, L in {2, 3}, H <= %d, C <= %d and a "
                      "rho of one channel or one per output channel (got n=%d D=%d L=%d/%d H=%d/%d C=%d/%d)", kMaxNodes, kMaxH,
                      kMaxC, a->n, a->D, a->f.L, a->rho.L, a->f.H, a->rho.H, a->f.C, a->rho.C);
  GNAN_REQUIRE(a->x && a->code && a->S && a->lut && (a->Y || a->Ysum), "small_graph: null x / code / S / lut / outputs");
  GNAN_REQUIRE(a->x_stride >= a->F && (a->cnt == nullptr || a->cnt_stride >= a->D), "small_graph: row stride smaller than the width");
  if (a->pre_rho && (a->cnt == nullptr || a->rho.C != 1 || a->D > kWave))
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "small_graph: pre-rho normalisation needs the shell sizes, a one-channel rho and D <= %d "
                      "(got cnt=%p rho.C=%d D=%d)", kWave, static_cast<const void*>(a->cnt), a->rho.C, a->D);
  const size_t need = gnan_small_graph_workspace_bytes(a->n, a->F, a->f.C);
  if (a->workspace == nullptr || a->workspace_bytes < need)
    return gnan::fail(GNAN_ERR_WORKSPACE, "small_graph: workspace %zu B < required %zu B", a->workspace_bytes, need);
  SmallParams p;
  p.x = a->x; p.x_stride = a->x_stride; p.n = a->n; p.F = a->F;
  p.f = to_mlp(&a->f); p.r = to_mlp(&a->rho);
  p.code = a->code; p.D = a->D; p.cnt = a->cnt; p.cnt_stride = a->cnt_stride;
  p.S = a->S; p.lut = a->lut; p.Y = a->Y; p.Ysum = a->Ysum;
  p.counter = static_cast<unsigned*>(a->workspace);
  p.part = reinterpret_cast<float*>(static_cast<char*>(a->workspace) + 16);
  p.rho_raw = 0; p.rest_zero = 0; p.pre_rho = a->pre_rho != 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned rho_groups = p.pre_rho ? static_cast<unsigned>((a->n * a->D + kWave - 1) / kWave) : 1u;
  if (a->n <= 64) {
    hipLaunchKernelGGL(small_graph_kernel<1>, dim3(static_cast<unsigned>(a->F) + rho_groups), dim3(kWaves * kWave),
                       small_cols_floats(1) * sizeof(float), st, p);
  } else {
    constexpr size_t lds = small_cols_floats(2) * sizeof(float);          // 64 KB of tables + the static weight image
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&small_graph_kernel<2>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (attr != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "small_graph: hipFuncSetAttribute: %s", hipGetErrorString(attr));
    hipLaunchKernelGGL(small_graph_kernel<2>, dim3(static_cast<unsigned>(a->F) + rho_groups), dim3(kWaves * kWave), lds, st, p);
  }
  return gnan::check_launch("small_graph_kernel");
}

extern "C" size_t gnan_small_graph_bwd_workspace_bytes(int32_t n, int32_t D) {
  (void)n; (void)D;
  return 16 + static_cast<size_t>(kRhoGroupsMax) * kRhoSlot * sizeof(float);   // counter (own 16 bytes) | part [<= 12][kRhoSlot]
}

extern "C" int gnan_small_graph_bwd(const gnan_small_graph_bwd_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "small_graph_bwd: null args");
  GNAN_REQUIRE(a->n >= 1 && a->F >= 1 && a->D >= 1, "small_graph_bwd: bad sizes n=%d F=%d D=%d", a->n, a->F, a->D);
  if (a->n > kMaxNodes || a->D > kWave || !mlp_ok(&a->f, kMaxC) || !mlp_ok(&a->rho, 1))
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "small_graph_bwd: covers n <= %d nodes, D <= %d shells, L in {2, 3}, H <= %d, C <= %d and a "
                      "one-channel rho (got n=%d D=%d L=%d/%d H=%d/%d C=%d/%d)", kMaxNodes, kWave, kMaxH, kMaxC, a->n, a->D,
                      a->f.L, a->rho.L, a->f.H, a->rho.H, a->f.C, a->rho.C);
  GNAN_REQUIRE(a->x && a->code && a->S && a->lut && (a->dY || a->dYsum), "small_graph_bwd: null x / code / S / lut / output gradient");
  GNAN_REQUIRE(a->x_stride >= a->F && (a->cnt == nullptr || a->cnt_stride >= a->D), "small_graph_bwd: row stride smaller than the width");
  GNAN_REQUIRE(grads_ok(&a->f, &a->df) && grads_ok(&a->rho, &a->drho),
               "small_graph_bwd: a gradient pointer for every weight, and for a bias exactly where there is one");
  SmallBwdParams p;
  p.x = a->x; p.x_stride = a->x_stride; p.n = a->n; p.F = a->F;
  p.f = to_weights(&a->f, &a->df); p.r = to_weights(&a->rho, &a->drho);
  p.f_mid = a->f.L == 3; p.r_mid = a->rho.L == 3;
  p.code = a->code; p.D = a->D; p.cnt = a->cnt; p.cnt_stride = a->cnt_stride;
  p.S = a->S; p.lut = a->lut; p.dY = a->dY; p.dYsum = a->dYsum; p.pre_rho = a->pre_rho != 0;
  GNAN_REQUIRE(!p.pre_rho || a->cnt != nullptr, "small_graph_bwd: pre-rho normalisation needs the shell sizes");
  p.rho_groups = 1; p.rows_per = a->n; p.part = nullptr; p.counter = nullptr; p.rho_raw = 0; p.rest_zero = 0; p.rho_c = 1;
  if (p.pre_rho) {
    // rho's n * D arguments by rows over up to 12 workgroups — about as many arguments each as a feature workgroup has nodes —
    // when the caller lent the room for their partial gradients
    int groups = a->D < kRhoGroupsMax ? a->D : kRhoGroupsMax;
    groups = groups < a->n ? groups : a->n;
    const size_t need = gnan_small_graph_bwd_workspace_bytes(a->n, a->D);
    if (groups > 1 && a->workspace != nullptr && a->workspace_bytes >= need) {
      p.rows_per = (a->n + groups - 1) / groups;
      p.rho_groups = (a->n + p.rows_per - 1) / p.rows_per;
      p.counter = static_cast<unsigned*>(a->workspace);
      p.part = reinterpret_cast<float*>(static_cast<char*>(a->workspace) + 16);
    }
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (a->f.C) {
    case 1: return launch_small_bwd<1>(p, st);
    case 2: return launch_small_bwd<2>(p, st);
    case 3: return launch_small_bwd<3>(p, st);
    case 4: return launch_small_bwd<4>(p, st);
    case 5: return launch_small_bwd<5>(p, st);
    case 6: return launch_small_bwd<6>(p, st);
    case 7: return launch_small_bwd<7>(p, st);
    default: return launch_small_bwd<8>(p, st);
  }
}


extern "C" size_t gnan_small_batch_workspace_bytes(int32_t n_graphs, int64_t total_nodes, int32_t F, int32_t C) {
  return static_cast<size_t>(n_graphs) * 128 + static_cast<size_t>(total_nodes) * F * C * sizeof(float);    // counters (a line each) | part
}

extern "C" int gnan_small_batch_fwd(const gnan_small_batch_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "small_batch: null args");
  GNAN_REQUIRE(a->n_graphs >= 0 && a->F >= 1 && a->D >= 1 && a->total_nodes >= 0, "small_batch: bad sizes");
  if (a->n_graphs == 0) return GNAN_OK;
  if (a->max_nodes < 1 || a->max_nodes > kMaxNodes || a->D > 256 || !mlp_ok(&a->f, kMaxC) || !mlp_ok(&a->rho, kMaxC) ||
      (a->rho.C != 1 && a->rho.C != a->f.C))
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "small_batch: covers graphs of <= %d nodes, D <= 256 shells, L in {2, 3}, H <= %d, C <= %d "
                      "and a rho of one channel or one per output channel (got max n=%d D=%d L=%d/%d H=%d/%d C=%d/%d)", kMaxNodes,
                      kMaxH, kMaxC, a->max_nodes, a->D, a->f.L, a->rho.L, a->f.H, a->rho.H, a->f.C, a->rho.C);
  GNAN_REQUIRE(a->x && a->code && a->node_off && a->code_off && a->S && a->lut && (a->Y || a->Ysum),
               "small_batch: null x / code / offsets / S / lut / outputs");
  GNAN_REQUIRE(a->x_stride >= a->F, "small_batch: row stride smaller than the width");
  GNAN_REQUIRE(a->n_graphs <= 65535, "small_batch: at most 65535 graphs per launch");
  GNAN_REQUIRE(a->cnt == nullptr || a->cnt_stride >= a->D, "small_batch: cnt row stride smaller than D");
  if (a->cnt && (a->D > kWave || a->rho.C != 1))
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "small_batch: shell sizes need D <= %d and a one-channel rho (got D=%d rho.C=%d)", kWave,
                      a->D, a->rho.C);
  const size_t need = gnan_small_batch_workspace_bytes(a->n_graphs, a->total_nodes, a->F, a->f.C);
  if (a->workspace == nullptr || a->workspace_bytes < need)
    return gnan::fail(GNAN_ERR_WORKSPACE, "small_batch: workspace %zu B < required %zu B", a->workspace_bytes, need);
  BatchParams bp;
  SmallParams& p = bp.base;
  p.x = a->x; p.x_stride = a->x_stride; p.n = 0; p.F = a->F;
  p.f = to_mlp(&a->f); p.r = to_mlp(&a->rho);
  p.code = a->code; p.D = a->D; p.cnt = a->cnt; p.cnt_stride = a->cnt_stride;
  p.S = a->S; p.lut = a->lut; p.Y = a->Y; p.Ysum = a->Ysum;
  p.counter = static_cast<unsigned*>(a->workspace);
  p.part = reinterpret_cast<float*>(static_cast<char*>(a->workspace) + static_cast<size_t>(a->n_graphs) * 128);
  p.rho_raw = a->rho_raw_hops != 0; p.rest_zero = a->rest_zero != 0; p.pre_rho = 0;
  bp.node_off = a->node_off; bp.code_off = a->code_off;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(a->F) + 1, static_cast<unsigned>(a->n_graphs));
  if (a->max_nodes <= 64) {
    hipLaunchKernelGGL(small_graph_batch_kernel<1>, grid, dim3(kWaves * kWave), small_cols_floats(1) * sizeof(float), st, bp);
  } else {
    constexpr size_t lds = small_cols_floats(2) * sizeof(float);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&small_graph_batch_kernel<2>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (attr != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "small_batch: hipFuncSetAttribute: %s", hipGetErrorString(attr));
    hipLaunchKernelGGL(small_graph_batch_kernel<2>, grid, dim3(kWaves * kWave), lds, st, bp);
  }
  return gnan::check_launch("small_graph_batch_kernel");
}

namespace {
// the twelve gradient tensors of (f, rho) back to back: offsets (floats) of each inside a slab, 0-sized where a tensor is absent
int64_t slab_layout(const gnan_small_mlp* f, int F, const gnan_small_mlp* r, int64_t at[13]) {
  int64_t o = 0;
  const gnan_small_mlp* m[2] = {f, r};
  const int Fs[2] = {F, 1};
  for (int t = 0; t < 2; ++t) {
    const int64_t H = m[t]->H, C = m[t]->C, Fk = Fs[t];            // (rho: C = its own channel count)
    const bool mid = m[t]->L == 3;
    const int64_t size[6] = {Fk * H, m[t]->b_first ? Fk * H : 0, mid ? Fk * H * H : 0, (mid && m[t]->b_mid) ? Fk * H : 0,
                             Fk * C * H, m[t]->b_last ? Fk * C : 0};
    for (int u = 0; u < 6; ++u) { at[6 * t + u] = o; o += size[u]; }
  }
  at[12] = o;
  return o;
}

template <int C>
int launch_small_batch_bwd(const BatchBwdParams& bp, int n_graphs, int max_nodes, hipStream_t st) {
  const dim3 grid(static_cast<unsigned>(bp.base.F) + 1, static_cast<unsigned>(n_graphs));
  if (max_nodes <= 64) {
    hipLaunchKernelGGL((small_graph_batch_bwd_kernel<C, 1>), grid, dim3(256), small_bwd_dyn_bytes<1>(false), st, bp);
  } else {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&small_graph_batch_bwd_kernel<C, 2>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       static_cast<int>(small_bwd_dyn_bytes<2>(false)));
    if (attr != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "small_batch_bwd: hipFuncSetAttribute: %s", hipGetErrorString(attr));
    hipLaunchKernelGGL((small_graph_batch_bwd_kernel<C, 2>), grid, dim3(256), small_bwd_dyn_bytes<2>(false), st, bp);
  }
  return gnan::check_launch("small_graph_batch_bwd_kernel");
}
}  // namespace

extern "C" size_t gnan_small_batch_bwd_workspace_bytes(const gnan_small_batch_bwd_args* a) {
  if (a == nullptr) return 0;
  int64_t at[13];
  return static_cast<size_t>(a->n_graphs) * static_cast<size_t>(slab_layout(&a->f, a->F, &a->rho, at)) * sizeof(float);
}

extern "C" int gnan_small_batch_bwd(const gnan_small_batch_bwd_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "small_batch_bwd: null args");
  GNAN_REQUIRE(a->n_graphs >= 0 && a->F >= 1 && a->D >= 1 && a->total_nodes >= 0, "small_batch_bwd: bad sizes");
  GNAN_REQUIRE(grads_ok(&a->f, &a->df) && grads_ok(&a->rho, &a->drho),
               "small_batch_bwd: a gradient pointer for every weight, and for a bias exactly where there is one");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t at[13];
  const int64_t slab = slab_layout(&a->f, a->F, &a->rho, at);
  float* const dst[12] = {a->df.w_first, a->df.b_first, a->f.L == 3 ? a->df.w_mid : nullptr, a->f.L == 3 ? a->df.b_mid : nullptr,
                          a->df.w_last, a->df.b_last, a->drho.w_first, a->drho.b_first, a->rho.L == 3 ? a->drho.w_mid : nullptr,
                          a->rho.L == 3 ? a->drho.b_mid : nullptr, a->drho.w_last, a->drho.b_last};
  if (a->n_graphs == 0) {                              // nothing contributes: the gradients are zero
    for (int t = 0; t < 12; ++t)
      if (dst[t] && at[t + 1] > at[t]) {
        hipError_t e = hipMemsetAsync(dst[t], 0, static_cast<size_t>(at[t + 1] - at[t]) * sizeof(float), st);
        if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "small_batch_bwd: hipMemsetAsync: %s", hipGetErrorString(e));
      }
    return GNAN_OK;
  }
  if (a->max_nodes < 1 || a->max_nodes > kMaxNodes || a->D > kWave || !mlp_ok(&a->f, kMaxC) || !mlp_ok(&a->rho, kMaxC) ||
      (a->rho.C != 1 && a->rho.C != a->f.C))
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "small_batch_bwd: covers graphs of <= %d nodes, D <= %d shells, L in {2, 3}, H <= %d, C <= %d "
                      "and a rho of one channel or one per output channel (got max n=%d D=%d L=%d/%d H=%d/%d C=%d/%d)", kMaxNodes,
                      kWave, kMaxH, kMaxC, a->max_nodes, a->D, a->f.L, a->rho.L, a->f.H, a->rho.H, a->f.C, a->rho.C);
  GNAN_REQUIRE(a->x && a->code && a->node_off && a->code_off && a->S && a->lut && (a->dY || a->dYsum),
               "small_batch_bwd: null x / code / offsets / S / lut / output gradient");
  GNAN_REQUIRE(a->x_stride >= a->F, "small_batch_bwd: row stride smaller than the width");
  GNAN_REQUIRE(a->n_graphs <= 65535, "small_batch_bwd: at most 65535 graphs per launch");
  GNAN_REQUIRE(a->cnt == nullptr || a->cnt_stride >= a->D, "small_batch_bwd: cnt row stride smaller than D");
  if (a->cnt && a->rho.C != 1)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "small_batch_bwd: shell sizes need a one-channel rho (got rho.C=%d)", a->rho.C);
  // ONE graph (a batch-size-1 step through graph slots): its share IS the gradient — the kernel writes the caller's tensors,
  // no slab, no reduction launch (the kernel shifts graph g's pointers by g * slab = 0)
  const bool direct = a->n_graphs == 1;
  const size_t need = direct ? 0 : static_cast<size_t>(a->n_graphs) * static_cast<size_t>(slab) * sizeof(float);
  if (!direct && (a->workspace == nullptr || a->workspace_bytes < need))
    return gnan::fail(GNAN_ERR_WORKSPACE, "small_batch_bwd: workspace %zu B < required %zu B", a->workspace_bytes, need);
  float* slabs = static_cast<float*>(a->workspace);
  // graph 0's slab as the kernels' gradient tensors (the kernel shifts them by g * slab)
  gnan_small_mlp_grads gf = {slabs + at[0], a->f.b_first ? slabs + at[1] : nullptr, slabs + at[2],
                             (a->f.L == 3 && a->f.b_mid) ? slabs + at[3] : nullptr, slabs + at[4], a->f.b_last ? slabs + at[5] : nullptr};
  gnan_small_mlp_grads gr = {slabs + at[6], a->rho.b_first ? slabs + at[7] : nullptr, slabs + at[8],
                             (a->rho.L == 3 && a->rho.b_mid) ? slabs + at[9] : nullptr, slabs + at[10],
                             a->rho.b_last ? slabs + at[11] : nullptr};
  if (direct) { gf = a->df; gr = a->drho; }
  BatchBwdParams bp;
  SmallBwdParams& p = bp.base;
  p.x = a->x; p.x_stride = a->x_stride; p.n = 0; p.F = a->F;
  p.f = to_weights(&a->f, &gf); p.r = to_weights(&a->rho, &gr);
  if (a->f.L != 3) { p.f.d_w_mid = nullptr; p.f.d_b_mid = nullptr; }
  if (a->rho.L != 3) { p.r.d_w_mid = nullptr; p.r.d_b_mid = nullptr; }
  p.f_mid = a->f.L == 3; p.r_mid = a->rho.L == 3;
  p.code = a->code; p.D = a->D; p.cnt = a->cnt; p.cnt_stride = a->cnt_stride;
  p.S = a->S; p.lut = a->lut; p.dY = a->dYsum ? nullptr : a->dY; p.dYsum = a->dYsum; p.pre_rho = 0;
  p.rho_groups = 1; p.rows_per = 0; p.part = nullptr; p.counter = nullptr;
  p.rho_raw = a->rho_raw_hops != 0; p.rest_zero = a->rest_zero != 0; p.rho_c = a->rho.C;
  bp.node_off = a->node_off; bp.code_off = a->code_off; bp.slab = slab; bp.dy_per_graph = a->dYsum != nullptr;
  int rc;
  switch (a->f.C) {
    case 1: rc = launch_small_batch_bwd<1>(bp, a->n_graphs, a->max_nodes, st); break;
    case 2: rc = launch_small_batch_bwd<2>(bp, a->n_graphs, a->max_nodes, st); break;
    case 3: rc = launch_small_batch_bwd<3>(bp, a->n_graphs, a->max_nodes, st); break;
    case 4: rc = launch_small_batch_bwd<4>(bp, a->n_graphs, a->max_nodes, st); break;
    case 5: rc = launch_small_batch_bwd<5>(bp, a->n_graphs, a->max_nodes, st); break;
    case 6: rc = launch_small_batch_bwd<6>(bp, a->n_graphs, a->max_nodes, st); break;
    case 7: rc = launch_small_batch_bwd<7>(bp, a->n_graphs, a->max_nodes, st); break;
    default: rc = launch_small_batch_bwd<8>(bp, a->n_graphs, a->max_nodes, st); break;
  }
  if (rc != GNAN_OK || direct) return rc;
  ReduceSeg seg;
  for (int t = 0; t < 13; ++t) seg.at[t] = at[t];
  for (int t = 0; t < 12; ++t) seg.dst[t] = (at[t + 1] > at[t]) ? dst[t] : nullptr;
  int64_t blocks = (slab + 255) / 256;
  blocks = blocks > 4096 ? 4096 : blocks;
  hipLaunchKernelGGL(batch_grad_reduce_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, slabs, slab, a->n_graphs, seg);
  return gnan::check_launch("batch_grad_reduce_kernel");
}

// Per-graph hop matrices (batched_pyg_main.py:19-48: float hop counts, -1 = unreachable) -> the packed uint8 codes the
// batched kernel reads: code = hop for 0 <= hop <= 254, 255 for negative entries.  status[0] |= 1 for a value that is not
// such an integer; status[1] = the largest hop seen.  From the packed concatenation of the blocks (src_stride == 0, count
// values) or cut out of the dense block-diagonal matrix of the reference's collate function (batched_pyg_main.py:54-91:
// [N, N] with row stride src_stride; status[0] |= 2 if an entry OUTSIDE the diagonal blocks is listed).
namespace {
__global__ __launch_bounds__(256) void hops_to_code_kernel(const float* __restrict__ src, int64_t count, uint8_t* __restrict__ code,
                                                           int* __restrict__ status) {
  int bad = 0, top = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < count; i += static_cast<int64_t>(gridDim.x) * 256) {
    const float h = src[i];
    int c = 255;
    if (h >= 0.f) {
      const float r = rintf(h);
      if (r != h || r > 254.f) bad = 1;
      c = static_cast<int>(fminf(r, 254.f));
      top = c > top ? c : top;
    } else if (!(h < 0.f)) {
      bad = 1;                                       // NaN
    }
    code[i] = static_cast<uint8_t>(c);
  }
  if (bad) atomicOr(&status[0], 1);
  if (top) atomicMax(&status[1], top);
}

// one workgroup per row of the dense [N, N] matrix: the row's block entries go to the packed codes, the rest must be < 0
__global__ __launch_bounds__(256) void dense_blocks_to_code_kernel(const float* __restrict__ dist, int64_t stride, int64_t n,
                                                                   const int32_t* __restrict__ graph_of, const int32_t* __restrict__ node_off,
                                                                   const int64_t* __restrict__ code_off, uint8_t* __restrict__ code,
                                                                   int* __restrict__ status) {
  const int64_t i = blockIdx.x;
  const int g = graph_of[i];
  const int64_t lo = node_off[g], hi = node_off[g + 1];
  const float* row = dist + i * stride;
  uint8_t* out = code + code_off[g] + (i - lo) * (hi - lo);
  int bad = 0, top = 0;
  for (int64_t j = threadIdx.x; j < n; j += 256) {
    const float h = row[j];
    if (j >= lo && j < hi) {
      int c = 255;
      if (h >= 0.f) {
        const float r = rintf(h);
        if (r != h || r > 254.f) bad |= 1;
        c = static_cast<int>(fminf(r, 254.f));
        top = c > top ? c : top;
      } else if (!(h < 0.f)) {
        bad |= 1;
      }
      out[j - lo] = static_cast<uint8_t>(c);
    } else if (!(h < 0.f)) {
      bad |= 2;                                      // a listed (or NaN) pair across two graphs: not block-diagonal
    }
  }
  if (bad) atomicOr(&status[0], bad);
  if (top) atomicMax(&status[1], top);
}
}  // namespace

extern "C" int gnan_hops_to_code(const float* hops, int64_t count, uint8_t* code, int32_t* status, gnan_stream_t stream) {
  GNAN_REQUIRE(count >= 0 && status, "hops_to_code: bad arguments");
  if (count == 0) return GNAN_OK;
  GNAN_REQUIRE(hops && code, "hops_to_code: null pointer");
  int64_t blocks = (count + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(hops_to_code_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), hops,
                     count, code, status);
  return gnan::check_launch("hops_to_code_kernel");
}

extern "C" int gnan_dense_blocks_to_code(const float* dist, int64_t stride, int64_t n, const int32_t* graph_of,
                                         const int32_t* node_off, const int64_t* code_off, uint8_t* code, int32_t* status,
                                         gnan_stream_t stream) {
  GNAN_REQUIRE(n >= 0 && stride >= n && status, "dense_blocks_to_code: bad arguments");
  if (n == 0) return GNAN_OK;
  GNAN_REQUIRE(dist && graph_of && node_off && code_off && code, "dense_blocks_to_code: null pointer");
  GNAN_REQUIRE(n <= 0x7fffffffLL, "dense_blocks_to_code: too many rows for one launch");
  hipLaunchKernelGGL(dense_blocks_to_code_kernel, dim3(static_cast<unsigned>(n)), dim3(256), 0, static_cast<hipStream_t>(stream), dist,
                     stride, n, graph_of, node_off, code_off, code, status);
  return gnan::check_launch("dense_blocks_to_code_kernel");
}
