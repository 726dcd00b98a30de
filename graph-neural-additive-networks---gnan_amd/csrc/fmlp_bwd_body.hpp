// Parameter gradients of ONE scalar MLP ("feature" k of a stack) over a range of inputs, by one 256-thread workgroup — the
// body of gnan_fmlp_bwd's kernels (csrc/fmlp_bwd.hip), shared with the one-launch backward of small graphs
// (csrc/small_graph.hip), which feeds it gradients and inputs that live in LDS.
//
// A wave owns an input ("node") at a time; lane j is hidden unit j of both hidden layers.  The lane keeps row j and column j
// of W2 and row j of dW2 in registers (192 VGPRs: one wave per SIMD), and the activations and deltas of the other units
// reach it through v_readlane with constant lane numbers — scalar operands of the fmas, no LDS in the node loop.  Per node and
// wave: 3 H^2 fmas (z2, dW2 += dz2 x h1, dh1 = W2^T dz2).  The four waves' partial sums meet in LDS in wave order at the end.
#pragma once
#include "common.hpp"
#include "dropout.hpp"

namespace gnan_bwd {

using gnan::kWave;

constexpr int kH = 64;      // lanes = hidden units (H <= 64: the rest idle with zero weights)
constexpr int kCmax = 8;
constexpr int kWaves = 4;

struct Weights {            // feature k's slices are taken inside (stacked layout of gnan_fmlp_args)
  int H;
  const float *w_first, *b_first, *w_mid, *b_mid, *w_last, *b_last;
  float *d_w_first, *d_b_first, *d_w_mid, *d_b_mid, *d_w_last, *d_b_last;
};

struct Drop {               // training-mode Dropout of the forward pass (csrc/dropout.hpp); thresh 0 = none
  uint32_t thresh;
  float scale;
  uint64_t seed;
};

typedef float RedBuffer[kWaves][kH][17];     // chunked reduction of the waves' partial sums (16 values + pad)

__device__ __forceinline__ float lane_value(float v, int lane) {      // lane must be a compile-time constant
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// x_of(node) -> the MLP's input; g_of(node, c) -> dLoss / d output c at that input; nodes n_lo <= node < n_hi; outputs at
// offset `so` of the gradient tensors (0, or a split's block of partial gradients).  MID: L == 3 (a hidden-to-hidden matrix).
// DX: dLoss / d input of every node goes to dx[node - n_lo] as well (sum_j dz1_j w1_j, a fixed butterfly over the lanes).
template <int C, bool MID, bool DX = false, typename XFn, typename GFn>
__device__ __forceinline__ void feature_grads(const Weights& p, int k, int64_t n_lo, int64_t n_hi, int64_t so, const Drop& drop,
                                              XFn x_of, GFn g_of, RedBuffer& red, float* dx = nullptr) {
  const int j = threadIdx.x & (kWave - 1);        // hidden unit
  const int wv = threadIdx.x / kWave;             // node slot
  const int H = p.H;
  const bool unit = j < H;
  const float w1 = unit ? p.w_first[static_cast<int64_t>(k) * H + j] : 0.f;
  const float b1 = unit && p.b_first ? p.b_first[static_cast<int64_t>(k) * H + j] : 0.f;
  float b2 = 0.f;
  float w2row[MID ? kH : 1], w2col[MID ? kH : 1], dw2[MID ? kH : 1];
  if constexpr (MID) {
    b2 = unit && p.b_mid ? p.b_mid[static_cast<int64_t>(k) * H + j] : 0.f;
    const float* W2 = p.w_mid + static_cast<int64_t>(k) * H * H;
    // W2 through LDS: read straight from memory, lane j's ROW is 64 loads that each touch 64 different lines per wave (16k
    // line requests per workgroup, 5 us of an 18-us launch on a 30-node graph); staged coalesced into the reduction buffer
    // ([64][65]: rows and columns both conflict-free) it is 16 loads per thread, all in flight together
    static_assert(sizeof(RedBuffer) >= kH * (kH + 1) * sizeof(float), "W2 tile fits the reduction buffer");
    float* tile = &red[0][0][0];
    {
      constexpr int kPer = kH * kH / (kWaves * kWave);
      float v[kPer];
#pragma unroll
      for (int t = 0; t < kPer; ++t) {
        const int e = static_cast<int>(threadIdx.x) + t * kWaves * kWave;
        v[t] = e < H * H ? W2[e] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < kPer; ++t) {
        const int e = static_cast<int>(threadIdx.x) + t * kWaves * kWave;
        if (e < H * H) tile[(e / H) * (kH + 1) + e % H] = v[t];
      }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kH; ++t) {
      w2row[t] = unit && t < H ? tile[j * (kH + 1) + t] : 0.f;       // W2[j, t]
      w2col[t] = unit && t < H ? tile[t * (kH + 1) + j] : 0.f;       // W2[t, j]
      dw2[t] = 0.f;
    }
    __syncthreads();                                                // (the buffer goes back to the reductions)
  }
  float w3[C], dw3[C], db3[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    w3[c] = unit ? p.w_last[(static_cast<int64_t>(k) * C + c) * H + j] : 0.f;
    dw3[c] = db3[c] = 0.f;
  }
  float dw1 = 0.f, db1 = 0.f, db2 = 0.f;

  for (int64_t node = n_lo + wv; node < n_hi; node += kWaves) {
    const float x = x_of(node);
    float gv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) gv[c] = g_of(node, c);
    // forward (m1, m2: this unit's Dropout factors of the two hidden layers, 0 or 1 / (1 - p); 1 without Dropout)
    float m1 = 1.f, m2 = 1.f;
    if (drop.thresh != 0u) {
      const uint32_t dbase = gnan::drop_base(drop.seed, node, k);
      m1 = gnan::drop_keep(dbase, 0, j, drop.thresh) ? drop.scale : 0.f;
      if constexpr (MID) m2 = gnan::drop_keep(dbase, 1, j, drop.thresh) ? drop.scale : 0.f;
    }
    const float a1 = fmaf(w1, x, b1);
    const float h1 = unit && a1 > 0.f ? a1 * m1 : 0.f;
    if constexpr (MID) {
      float z2 = b2;
#pragma unroll
      for (int t = 0; t < kH; ++t) z2 = fmaf(w2row[t], lane_value(h1, t), z2);
      const float h2 = unit && z2 > 0.f ? z2 * m2 : 0.f;
      // backward through the output layer
      float dh2 = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        dh2 = fmaf(gv[c], w3[c], dh2);
        dw3[c] = fmaf(gv[c], h2, dw3[c]);
        db3[c] += gv[c];
      }
      const float dz2 = unit && z2 > 0.f ? dh2 * m2 : 0.f;
      db2 += dz2;
      // dW2[j, t] += dz2_j h1_t;   dh1_j = sum_t W2[t, j] dz2_t
      float dh1 = 0.f;
#pragma unroll
      for (int t = 0; t < kH; ++t) {
        dw2[t] = fmaf(dz2, lane_value(h1, t), dw2[t]);
        dh1 = fmaf(w2col[t], lane_value(dz2, t), dh1);
      }
      const float dz1 = unit && a1 > 0.f ? dh1 * m1 : 0.f;
      dw1 = fmaf(dz1, x, dw1);
      db1 += dz1;
      if constexpr (DX) {
        float v = dz1 * w1;
#pragma unroll
        for (int off = kWave / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
        if (j == 0) dx[node - n_lo] = v;
      }
    } else {
      float dh1 = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        dh1 = fmaf(gv[c], w3[c], dh1);
        dw3[c] = fmaf(gv[c], h1, dw3[c]);
        db3[c] += gv[c];
      }
      const float dz1 = unit && a1 > 0.f ? dh1 * m1 : 0.f;
      dw1 = fmaf(dz1, x, dw1);
      db1 += dz1;
      if constexpr (DX) {
        float v = dz1 * w1;
#pragma unroll
        for (int off = kWave / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
        if (j == 0) dx[node - n_lo] = v;
      }
    }
  }

  // waves -> wave 0, in wave order: dW2 sixteen columns at a time, then the vectors
  if constexpr (MID) {
    float* dW2 = p.d_w_mid + so + static_cast<int64_t>(k) * H * H;
    for (int t0 = 0; t0 < kH; t0 += 16) {
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 16; ++t) red[wv][j][t] = dw2[t0 + t];
      __syncthreads();
      if (wv == 0 && unit) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          if (t0 + t < H) {
            float s = red[0][j][t];
            for (int w = 1; w < kWaves; ++w) s += red[w][j][t];
            dW2[j * H + t0 + t] = s;
          }
        }
      }
    }
  }
  __syncthreads();
  red[wv][j][0] = dw1; red[wv][j][1] = db1; red[wv][j][2] = db2;
#pragma unroll
  for (int c = 0; c < C; ++c) { red[wv][j][3 + c] = dw3[c]; }
  __syncthreads();
  if (wv == 0 && unit) {
    float s[3 + C];
#pragma unroll
    for (int t = 0; t < 3 + C; ++t) {
      s[t] = red[0][j][t];
      for (int w = 1; w < kWaves; ++w) s[t] += red[w][j][t];
    }
    p.d_w_first[so + static_cast<int64_t>(k) * H + j] = s[0];
    if (p.d_b_first) p.d_b_first[so + static_cast<int64_t>(k) * H + j] = s[1];
    if (MID && p.d_b_mid) p.d_b_mid[so + static_cast<int64_t>(k) * H + j] = s[2];
#pragma unroll
    for (int c = 0; c < C; ++c) p.d_w_last[so + (static_cast<int64_t>(k) * C + c) * H + j] = s[3 + c];
  }
  if (p.d_b_last) {       // db3 is the same in every lane of a wave: lane 0 of each wave, then wave order
    __syncthreads();
    if (j == 0)
#pragma unroll
      for (int c = 0; c < C; ++c) red[wv][0][c] = db3[c];
    __syncthreads();
    if (threadIdx.x < C) {
      float s = red[0][0][threadIdx.x];
      for (int w = 1; w < kWaves; ++w) s += red[w][0][threadIdx.x];
      p.d_b_last[so + static_cast<int64_t>(k) * C + threadIdx.x] = s;
    }
  }
}

}  // namespace gnan_bwd
