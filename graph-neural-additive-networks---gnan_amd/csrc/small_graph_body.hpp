// Pieces the one-launch small-graph kernels share (csrc/small_graph.hip, csrc/small_graph_nam.hip): the LDS image of one
// scalar MLP's weights, its evaluation on a block of 64 inputs by four waves, and the host-side conversions of the C ABI's
// gnan_small_mlp.
#pragma once
#include "common.hpp"
#include "fmlp_bwd_body.hpp"

namespace gnan_small {

using gnan::kWave;

struct Mlp {
  int L, H, C;
  const float *w_first, *b_first, *w_mid, *b_mid, *w_last, *b_last;
};

constexpr int kMaxH = 64, kMaxC = 8, kWaves = 4, kMaxNodes = 128;     // nodes: NB blocks of 64 (kernels are built for NB = 1 and 2)

// LDS image of one scalar MLP's weights (scalar loads straight from memory made every inner-loop step wait ~150 cycles for
// its weights: 33 us for a 30-node graph; read as LDS broadcasts the same loop is bound by LDS issue)
struct MlpLds {
  float* w1;   // [H]
  float* b1;   // [H]
  float* w2;   // [H, H]
  float* b2;   // [H]
  float* w3;   // [C, H]
  float* b3;   // [C]
};

__device__ __forceinline__ MlpLds carve(float* base) {
  MlpLds w;
  w.w2 = base;
  w.w1 = base + kMaxH * kMaxH;
  w.b1 = w.w1 + kMaxH;
  w.b2 = w.b1 + kMaxH;
  w.w3 = w.b2 + kMaxH;
  w.b3 = w.w3 + kMaxC * kMaxH;
  return w;
}
constexpr int kWeightFloats = kMaxH * kMaxH + 3 * kMaxH + kMaxC * kMaxH + kMaxC;

// (every load is issued before the first LDS store: a load -> store loop keeps ONE load in flight, and a cold load costs
// more than a microsecond — sixteen of them in a row were most of the kernel's time)
__device__ __forceinline__ void stage_weights(const Mlp& m, int k, const MlpLds& w) {
  const int H = m.H, C = m.C, tid = threadIdx.x;
  constexpr int kPer = kMaxH * kMaxH / (kWaves * kWave);           // 16 hidden-to-hidden weights per thread
  float v2[kPer], v3[2];
  const float* w2 = m.L == 3 ? m.w_mid + static_cast<int64_t>(k) * H * H : nullptr;
#pragma unroll
  for (int t = 0; t < kPer; ++t) {
    const int i = tid + t * (kWaves * kWave);
    v2[t] = (w2 && i < H * H) ? w2[i] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int i = tid + t * (kWaves * kWave);
    v3[t] = i < C * H ? m.w_last[static_cast<int64_t>(k) * C * H + i] : 0.f;
  }
  const bool hid = tid < H;
  const float a1 = hid ? m.w_first[k * H + tid] : 0.f;
  const float a2 = (hid && m.b_first) ? m.b_first[k * H + tid] : 0.f;
  const float a3 = (hid && m.L == 3 && m.b_mid) ? m.b_mid[k * H + tid] : 0.f;
  const float a4 = (tid < C && m.b_last) ? m.b_last[k * C + tid] : 0.f;
#pragma unroll
  for (int t = 0; t < kPer; ++t) {
    const int i = tid + t * (kWaves * kWave);
    if (i < H * H) w.w2[i] = v2[t];
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int i = tid + t * (kWaves * kWave);
    if (i < C * H) w.w3[i] = v3[t];
  }
  if (hid) { w.w1[tid] = a1; w.b1[tid] = a2; w.b2[tid] = a3; }
  if (tid < C) w.b3[tid] = a4;
  __syncthreads();
}

// One scalar MLP on the 64 inputs of this node block: lane = input, the four waves split every hidden layer's units;
// activations live in LDS columns a / b ([H][64]).  Wave 0 ends with the C outputs of its lane in out[].
__device__ __forceinline__ void mlp_block(const Mlp& m, const MlpLds& w, float xv, float* a, float* b, int lane, int wave,
                                          float (&out)[kMaxC]) {
  const int H = m.H, C = m.C;
  for (int j = wave; j < H; j += kWaves) a[j * kWave + lane] = fmaxf(fmaf(xv, w.w1[j], w.b1[j]), 0.f);
  __syncthreads();
  float* cur = a;
  float* nxt = b;
  if (m.L == 3) {                                    // the one hidden-to-hidden layer: rows of w2 = output units
    const int per = (H + kWaves - 1) / kWaves;       // a contiguous run of output units per wave, four at a time
    const int o_lo = wave * per, o_hi = o_lo + per < H ? o_lo + per : H;
    for (int o0 = o_lo; o0 < o_hi; o0 += 4) {
      float acc[4];
      const float* row[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int o = o0 + t < o_hi ? o0 + t : o_hi - 1;             // clamp: uniform, keeps the reads in range
        acc[t] = w.b2[o];
        row[t] = w.w2 + o * H;
      }
      if ((H & 3) == 0) {                              // four inputs per step: the weights as 16-byte LDS broadcasts
#pragma unroll 4
        for (int i = 0; i < H; i += 4) {
          float h[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) h[u] = cur[(i + u) * kWave + lane];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float4 w4 = *reinterpret_cast<const float4*>(row[t] + i);
            acc[t] = fmaf(w4.x, h[0], acc[t]);
            acc[t] = fmaf(w4.y, h[1], acc[t]);
            acc[t] = fmaf(w4.z, h[2], acc[t]);
            acc[t] = fmaf(w4.w, h[3], acc[t]);
          }
        }
      } else {
#pragma unroll 16
        for (int i = 0; i < H; ++i) {
          const float h = cur[i * kWave + lane];
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = fmaf(row[t][i], h, acc[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (o0 + t < o_hi) nxt[(o0 + t) * kWave + lane] = fmaxf(acc[t], 0.f);
    }
    __syncthreads();
    float* t = cur; cur = nxt; nxt = t;
  }
  if (wave == 0) {
    for (int c = 0; c < C; ++c) {
      float acc = w.b3[c];
#pragma unroll 16
      for (int i = 0; i < H; ++i) acc = fmaf(w.w3[c * H + i], cur[i * kWave + lane], acc);
      out[c] = acc;
    }
  }
  __syncthreads();                                   // the columns are free for the next node block
}

constexpr int small_cols_floats(int nb) {
  const int nodes = 64 * nb;
  const int tables = 2 * nodes * kMaxC + 256 * kMaxC + nodes * kWave + nodes * nodes / 4;
  return tables > 2 * kMaxH * kWave ? tables : 2 * kMaxH * kWave;
}

// Is this workgroup the last of `expected` to arrive at `counter`?  ONE thread releases the workgroup's results (the barrier before
// it orders the other threads' stores before its fence) and, in the last workgroup, acquires the others'.  An agent-scope fence
// is an L2 write-back / invalidate on this multi-die part: executed by every wave — 4 waves x 2 fences x the 512 workgroups of a
// 32-graph batch — they queued up behind each other for 50 of the launch's 68 us.
__device__ __forceinline__ bool last_to_arrive(unsigned* counter, unsigned expected, unsigned* s_slot) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    *s_slot = atomicAdd(counter, 1u);
  }
  __syncthreads();
  if (*s_slot != expected - 1) return false;
  if (threadIdx.x == 0) __threadfence();
  __syncthreads();
  return true;
}

inline gnan_bwd::Weights to_weights(const gnan_small_mlp* m, const gnan_small_mlp_grads* g) {
  gnan_bwd::Weights w;
  w.H = m->H;
  w.w_first = m->w_first; w.b_first = m->b_first; w.w_mid = m->L == 3 ? m->w_mid : nullptr; w.b_mid = m->L == 3 ? m->b_mid : nullptr;
  w.w_last = m->w_last; w.b_last = m->b_last;
  w.d_w_first = g->w_first; w.d_b_first = g->b_first; w.d_w_mid = g->w_mid; w.d_b_mid = g->b_mid; w.d_w_last = g->w_last;
  w.d_b_last = g->b_last;
  return w;
}

inline bool grads_ok(const gnan_small_mlp* m, const gnan_small_mlp_grads* g) {
  return g->w_first && g->w_last && (m->L == 2 || g->w_mid) && ((m->b_first == nullptr) == (g->b_first == nullptr)) &&
         (m->L == 2 || ((m->b_mid == nullptr) == (g->b_mid == nullptr))) && ((m->b_last == nullptr) == (g->b_last == nullptr));
}

inline bool mlp_ok(const gnan_small_mlp* m, int max_c) {
  return (m->L == 2 || m->L == 3) && m->H >= 1 && m->H <= kMaxH && m->C >= 1 && m->C <= max_c && m->w_first && m->w_last &&
         (m->L == 2 || m->w_mid);
}

inline Mlp to_mlp(const gnan_small_mlp* m) {
  Mlp r;
  r.L = m->L; r.H = m->H; r.C = m->C;
  r.w_first = m->w_first; r.b_first = m->b_first; r.w_mid = m->w_mid; r.b_mid = m->b_mid; r.w_last = m->w_last; r.b_last = m->b_last;
  return r;
}

}  // namespace gnan_small
