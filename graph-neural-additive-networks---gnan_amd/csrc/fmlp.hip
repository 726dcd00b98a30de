// Per-feature shape functions f_k : R -> R^C as one batched kernel (gfx950).
//
// Replaces the Python loop GNAN.py:57-62 (= GNAN.py:150-155, models.py:360-365, models.py:292-297):
// F x (L addmm + (L-1) relu + strided copy_) launches become one launch over (node tile, all features).
// See include/gnan_hip.h for the weight stacking.
//
// fmlp_lane_kernel (any H, L, C): lane = node, wave = 64 nodes walking all F features.  The
// activations of a lane live in a wave-private LDS column [H][64] (a lane only ever touches its own
// column: no barriers), weights are wave-uniform so they come through the scalar cache, outputs
// are register-blocked JB at a time so one LDS read feeds JB FMAs.
//
// fmlp_mfma_kernel (H <= 64, 3 <= L <= 4, C <= 8 — the shapes the reference's defaults produce): the
// hidden layers are a grouped GEMM (F groups of [nodes, H] x [H, H]) and run on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32, exact fp32).  Formulated transposed, D[j', node] = W[j', j] * h[j, node], so that
//   * layer 1 (rank-1: relu(x*w1 + b1)) is computed straight into the MFMA B operand, never stored;
//   * the accumulator layout (hidden unit over registers, node over lanes) IS the next layer's B operand
//     once the K order is permuted accordingly — the permutation is baked into the packed weights, so
//     activations never leave registers between layers;
//   * the last layer (H -> C) is C dot products over a lane's own registers + one cross-half add.
// Per feature the packed weights (16 KiB for H = 64) reach LDS by LDS-DMA (global_load_lds_dwordx4) once per
// workgroup, double-buffered against the MFMA work; [N, F, H] activations are never materialised.
#include "common.hpp"
#include "dropout.hpp"

namespace {

using gnan::kWave;

struct Params {
  const float* x;
  int64_t n, x_stride;
  int F, L, H, C;
  const float* w_first;
  const float* b_first;
  const float* w_mid;
  const float* b_mid;
  const float* w_last;
  const float* b_last;
  int sum_features;
  float* out;
  int64_t out_stride;
  // matrix-core kernel only: features are cut into gridDim.y chunks so that graphs with few nodes but many
  // features (Cora: 2.7k x 1434) still fill the chip; with sum_features the chunks meet in `sum_partial`
  int feat_chunk;
  float* sum_partial;  // [chunks, n, C] or nullptr (single chunk)
  // training-mode Dropout (lane kernel): keep iff hash(seed, node, feature, layer, unit) >= drop_thresh, scale 1 / (1 - p)
  uint32_t drop_thresh;
  float drop_scale;
  uint64_t drop_seed;
};

constexpr int JB = 8;

// out[j'] = act(b[j'] + sum_j W[j', j] * in[j]) for one lane; in/out are LDS columns (stride 64).
template <bool RELU, typename Emit>
__device__ __forceinline__ void dense_layer(const float* __restrict__ W, const float* __restrict__ b, int n_out,
                                            int n_in, const float* in_col, Emit emit) {
  for (int j0 = 0; j0 < n_out; j0 += JB) {
    float acc[JB];
#pragma unroll
    for (int t = 0; t < JB; ++t) acc[t] = (b && j0 + t < n_out) ? b[j0 + t] : 0.f;
    for (int j = 0; j < n_in; ++j) {
      const float a = in_col[j * kWave];
#pragma unroll
      for (int t = 0; t < JB; ++t) {
        const int r = j0 + t < n_out ? j0 + t : n_out - 1;  // clamp: uniform, keeps loads in range
        acc[t] = fmaf(W[static_cast<int64_t>(r) * n_in + j], a, acc[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < JB; ++t) {
      if (j0 + t < n_out) emit(j0 + t, RELU ? fmaxf(acc[t], 0.f) : acc[t]);
    }
  }
}

__global__ __launch_bounds__(256) void fmlp_lane_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int waves = blockDim.x / kWave;
  const int H = p.H, C = p.C;
  const int per_wave = (2 * H + (p.sum_features ? C : 0)) * kWave;
  float* buf_a = smem + wave * per_wave + lane;
  float* buf_b = buf_a + H * kWave;
  float* sacc = buf_b + H * kWave;
  const int64_t node = (static_cast<int64_t>(blockIdx.x) * waves + wave) * kWave + lane;
  const bool valid = node < p.n;
  if (p.sum_features)
    for (int c = 0; c < C; ++c) sacc[c * kWave] = 0.f;

  for (int k = 0; k < p.F; ++k) {
    const float xv = valid ? p.x[node * p.x_stride + k] : 0.f;
    auto emit_out = [&](int c, float v) {
      if (p.sum_features) {
        sacc[c * kWave] += v;
      } else if (valid) {
        p.out[node * p.out_stride + static_cast<int64_t>(k) * C + c] = v;
      }
    };
    if (p.L == 1) {
      for (int c = 0; c < C; ++c) {
        const float b = p.b_last ? p.b_last[k * C + c] : 0.f;
        emit_out(c, fmaf(xv, p.w_last[k * C + c], b));
      }
      continue;
    }
    const bool drop = p.drop_thresh != 0u;
    const uint32_t dbase = drop ? gnan::drop_base(p.drop_seed, node, k) : 0u;
    for (int j = 0; j < H; ++j) {
      const float b = p.b_first ? p.b_first[k * H + j] : 0.f;
      float h = fmaxf(fmaf(xv, p.w_first[k * H + j], b), 0.f);
      if (drop) h = gnan::drop_keep(dbase, 0, j, p.drop_thresh) ? h * p.drop_scale : 0.f;
      buf_a[j * kWave] = h;
    }
    float* cur = buf_a;
    float* nxt = buf_b;
    for (int l = 0; l < p.L - 2; ++l) {
      const int64_t f = static_cast<int64_t>(l) * p.F + k;
      dense_layer<true>(p.w_mid + f * H * H, p.b_mid ? p.b_mid + f * H : nullptr, H, H, cur,
                        [&](int j, float v) {
                          if (drop) v = gnan::drop_keep(dbase, l + 1, j, p.drop_thresh) ? v * p.drop_scale : 0.f;
                          nxt[j * kWave] = v;
                        });
      float* t = cur; cur = nxt; nxt = t;
    }
    dense_layer<false>(p.w_last + static_cast<int64_t>(k) * C * H, p.b_last ? p.b_last + k * C : nullptr, C, H, cur,
                       emit_out);
  }
  if (p.sum_features && valid)
    for (int c = 0; c < C; ++c) p.out[node * p.out_stride + c] = sacc[c * kWave];
}

// fmlp_point_kernel (2 <= L <= 3 with H <= 64; a handful of evaluations): ONE WAVE per (node, feature), lane = hidden unit.
// The rho table of a truncated-hop graph is rho at three points, asked for on every training step: the matrix-core route
// packs the weights first (two launches, 9 + 7 us) and the lane kernel walks the H x H layer in ONE lane (70 us).  Here the
// wave holds a layer in its lanes: layer 1 is one fused multiply-add per lane, layer 2 a lane's own row of W2 (sixteen
// 16-byte loads) against the layer-1 values handed round by v_readlane, the C outputs a fixed butterfly each.  Sums in
// float32 like the other two kernels (another order: the three agree to a few ulp, tests/test_gpu_kernels.py).
__global__ __launch_bounds__(64) void fmlp_point_kernel(const Params p) {
  const int lane = threadIdx.x;
  const int64_t node = blockIdx.x / p.F;
  const int k = static_cast<int>(blockIdx.x - node * p.F);
  const int H = p.H, C = p.C;
  const float xv = p.x[node * p.x_stride + k];
  const bool on = lane < H;
  float h = 0.f;
  if (on) h = fmaxf(fmaf(xv, p.w_first[k * H + lane], p.b_first ? p.b_first[k * H + lane] : 0.f), 0.f);
  if (p.L == 3) {
    const float* row = p.w_mid + (static_cast<int64_t>(k) * H + (on ? lane : 0)) * H;     // W2[k][lane][:]
    float acc = (on && p.b_mid) ? p.b_mid[k * H + lane] : 0.f;
    if ((H & 3) == 0 && (reinterpret_cast<uintptr_t>(p.w_mid) & 15) == 0) {
      for (int i = 0; i < H; i += 4) {
        const float4 w = *reinterpret_cast<const float4*>(row + i);
        acc = fmaf(w.x, __shfl(h, i), acc);
        acc = fmaf(w.y, __shfl(h, i + 1), acc);
        acc = fmaf(w.z, __shfl(h, i + 2), acc);
        acc = fmaf(w.w, __shfl(h, i + 3), acc);
      }
    } else {
      for (int i = 0; i < H; ++i) acc = fmaf(row[i], __shfl(h, i), acc);
    }
    h = on ? fmaxf(acc, 0.f) : 0.f;
  }
  float* out = p.out + node * p.out_stride + (p.sum_features ? 0 : static_cast<int64_t>(k) * C);
  for (int c = 0; c < C; ++c) {
    float v = on ? p.w_last[(static_cast<int64_t>(k) * C + c) * H + lane] * h : 0.f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) out[c] = v + (p.b_last ? p.b_last[k * C + c] : 0.f);
  }
}

// =============================================================================================
// MFMA path
// =============================================================================================
using f32x16 = __attribute__((ext_vector_type(16))) float;

// Accumulator register r of lane-half h holds row (r&3) + 8*(r>>2) + 4*h of a 32-row MFMA tile.
__host__ __device__ constexpr int drow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// K step s (two hidden units per step, one per lane half) -> hidden unit, in accumulator order.
__host__ __device__ constexpr int kperm(int s, int h) { return 32 * (s / 16) + drow(s % 16, h); }

__host__ __device__ constexpr int round4(int v) { return (v + 3) & ~3; }
__host__ __device__ constexpr int vec_floats(int HT, int NMID, int CT) {
  return round4(64 * HT + NMID * 32 * HT + CT * 32 * HT + CT);
}
// floats per feature in the packed stream, padded so that each of the 4 waves copies the same number of
// 1-KiB pieces (64 lanes x 16 B)
__host__ __device__ constexpr int feature_floats(int HT, int NMID, int CT) {
  return (NMID * 32 * HT * 32 * HT + vec_floats(HT, NMID, CT) + 1023) & ~1023;
}

// Asynchronous global -> LDS copy of one feature's packed weights (global_load_lds_dwordx4: no VGPR
// round trip; the LDS image is lane-linear, which is exactly how the stream was packed).
template <int NCH>
__device__ __forceinline__ void stage_feature(const float* __restrict__ src, float* lds_dst, int lane, int wave) {
  static_assert(NCH % 4 == 0, "pieces are dealt evenly to the 4 waves");
#pragma unroll
  for (int c = 0; c < NCH / 4; ++c) {
    const int ch = c * 4 + wave;  // wave-uniform
    __builtin_amdgcn_global_load_lds(src + (ch * 64 + lane) * 4,
                                     (__attribute__((address_space(3))) void*)(lds_dst + ch * 256), 16, 0, 0);
  }
}

// Re-lays the reference-order weights of every feature into the order the MFMA kernel consumes:
//   mid[m][t_out][s/4][lane][s%4] = W_m[32*t_out + (lane&31)][kperm(s, lane>>5)]        (A fragments, b128 reads)
//   vec = { {w1, b1}[kperm(s, h)] for s, h } | b_mid[m][:] | w_last[c][:] | b_last[c]      (zero padded to HP)
__global__ __launch_bounds__(256) void fmlp_pack_kernel(const Params p, float* __restrict__ packed, int HT, int NMID,
                                                        int CT) {
  const int k = blockIdx.x;
  const int H = p.H, C = p.C, F = p.F;
  const int HP = 32 * HT, KS4 = 4 * HT;
  const int midf = NMID * HP * HP;
  const int fs = feature_floats(HT, NMID, CT);
  float* out = packed + static_cast<int64_t>(k) * fs;
  for (int idx = threadIdx.x; idx < fs; idx += blockDim.x) {
    float v = 0.f;
    if (idx < midf) {
      const int m = idx / (HP * HP);
      int rem = idx % (HP * HP);
      const int t_out = rem / (KS4 * 256);
      rem %= KS4 * 256;
      const int s4 = rem / 256, lane = (rem % 256) / 4, e = rem % 4;
      const int row = 32 * t_out + (lane & 31);
      const int col = kperm(4 * s4 + e, lane >> 5);
      if (row < H && col < H) v = p.w_mid[((static_cast<int64_t>(m) * F + k) * H + row) * H + col];
    } else {
      int j = idx - midf;
      if (j < 64 * HT) {
        const int hid = kperm(j / 4, (j / 2) % 2);
        if (hid < H) v = (j % 2) ? (p.b_first ? p.b_first[k * H + hid] : 0.f) : p.w_first[k * H + hid];
      } else if ((j -= 64 * HT) < NMID * HP) {
        const int m = j / HP, hid = j % HP;
        if (hid < H && p.b_mid) v = p.b_mid[(static_cast<int64_t>(m) * F + k) * H + hid];
      } else if ((j -= NMID * HP) < CT * HP) {
        const int c = j / HP, hid = j % HP;
        if (c < C && hid < H) v = p.w_last[(static_cast<int64_t>(k) * C + c) * H + hid];
      } else if ((j -= CT * HP) < CT) {
        if (j < C && p.b_last) v = p.b_last[k * C + j];
      }
    }
    out[idx] = v;
  }
}

template <int HT, int NMID, int CT, bool SUM, int NT>
__global__ __launch_bounds__(256, 2) void fmlp_mfma_kernel(const Params p, const float* __restrict__ packed) {
  constexpr int HP = 32 * HT, KS = 16 * HT, KS4 = 4 * HT;
  constexpr int MIDF = NMID * HP * HP;
  constexpr int FS = feature_floats(HT, NMID, CT);
  constexpr int NCH = FS / 256;  // wave-wide 1-KiB pieces per feature
  extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 * FS floats
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int half = lane >> 5, nl = lane & 31;
  const int64_t tile0 = (static_cast<int64_t>(blockIdx.x) * 4 + wave) * NT;
  int64_t node[NT];
  bool valid[NT];
  float xcur[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    node[nt] = (tile0 + nt) * 32 + nl;
    valid[nt] = node[nt] < p.n;
    xcur[nt] = valid[nt] ? p.x[node[nt] * p.x_stride + blockIdx.y * p.feat_chunk] : 0.f;
  }
  const int k_lo = blockIdx.y * p.feat_chunk;
  const int k_hi = k_lo + p.feat_chunk < p.F ? k_lo + p.feat_chunk : p.F;
  stage_feature<NCH>(packed + static_cast<int64_t>(k_lo) * FS, smem, lane, wave);
  __syncthreads();

  float sum[NT][CT];
  float blsum[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    blsum[c] = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) sum[nt][c] = 0.f;
  }

  // One feature: `buf` holds its packed weights, `nbuf` receives the next feature's.  Called with the
  // two LDS buffers as distinct compile-time offsets so the compiler can see that the in-flight LDS-DMA
  // does not alias the ds_reads of the current feature (otherwise it drains vmcnt before the first read).
  auto feature_step = [&](const int k, const float* __restrict__ buf, float* __restrict__ nbuf) {
    const float* vec = buf + MIDF;
    // prefetch the next feature's packed weights and x column into registers (the last iteration
    // harmlessly re-fetches its own feature: keeps the staging registers unconditional)
    const int kn = k + 1 < k_hi ? k + 1 : k;
    float xnext[NT];
    stage_feature<NCH>(packed + static_cast<int64_t>(kn) * FS, nbuf, lane, wave);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) xnext[nt] = valid[nt] ? p.x[node[nt] * p.x_stride + kn] : 0.f;

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      // layer 1 straight into the B operand (K order = accumulator order)
      float B[KS];
      const float xv = xcur[nt];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float2 wb = *reinterpret_cast<const float2*>(vec + (s * 2 + half) * 2);
        B[s] = fmaxf(fmaf(xv, wb.x, wb.y), 0.f);
      }
      // hidden layers on the matrix cores; accumulators start at the bias
#pragma unroll
      for (int m = 0; m < NMID; ++m) {
        f32x16 acc[HT];
        const float* bm = vec + 64 * HT + m * HP;
#pragma unroll
        for (int t = 0; t < HT; ++t) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 b4 = *reinterpret_cast<const float4*>(bm + 32 * t + 8 * q + 4 * half);
            acc[t][4 * q + 0] = b4.x; acc[t][4 * q + 1] = b4.y; acc[t][4 * q + 2] = b4.z; acc[t][4 * q + 3] = b4.w;
          }
        }
        const float* wm = buf + m * HP * HP;
#pragma unroll
        for (int s4 = 0; s4 < KS4; ++s4) {
#pragma unroll
          for (int t = 0; t < HT; ++t) {
            const float4 a4 = *reinterpret_cast<const float4*>(wm + ((t * KS4 + s4) * 64 + lane) * 4);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, B[4 * s4 + 0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, B[4 * s4 + 1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, B[4 * s4 + 2], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, B[4 * s4 + 3], acc[t], 0, 0, 0);
          }
        }
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) B[16 * t + r] = fmaxf(acc[t][r], 0.f);
      }
      // last layer: C dot products over this lane's hidden units (the other half's units are added below)
      const float* wl = vec + 64 * HT + NMID * HP;
      float part[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < HT; ++t) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 w4 = *reinterpret_cast<const float4*>(wl + c * HP + 32 * t + 8 * q + 4 * half);
            a = fmaf(w4.x, B[16 * t + 4 * q + 0], a);
            a = fmaf(w4.y, B[16 * t + 4 * q + 1], a);
            a = fmaf(w4.z, B[16 * t + 4 * q + 2], a);
            a = fmaf(w4.w, B[16 * t + 4 * q + 3], a);
          }
        }
        part[c] = a;
      }
      if constexpr (SUM) {
#pragma unroll
        for (int c = 0; c < CT; ++c) sum[nt][c] += part[c];
      } else {
        const float* bl = wl + CT * HP;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const float v = part[c] + __shfl_xor(part[c], 32) + bl[c];
          if (half == 0 && valid[nt] && c < p.C) p.out[node[nt] * p.out_stride + static_cast<int64_t>(k) * p.C + c] = v;
        }
      }
    }
    if constexpr (SUM) {
      const float* bl = vec + 64 * HT + NMID * HP + CT * HP;
#pragma unroll
      for (int c = 0; c < CT; ++c) blsum[c] += bl[c];
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) xcur[nt] = xnext[nt];
    __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and publishes the next feature's buffer
  };
  for (int k = k_lo; k < k_hi; k += 2) {
    feature_step(k, smem, smem + FS);
    if (k + 1 < k_hi) feature_step(k + 1, smem + FS, smem);
  }
  if constexpr (SUM) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const float v = sum[nt][c] + __shfl_xor(sum[nt][c], 32) + blsum[c];
        if (half == 0 && valid[nt] && c < p.C) {
          if (p.sum_partial) {
            p.sum_partial[(static_cast<int64_t>(blockIdx.y) * p.n + node[nt]) * p.C + c] = v;
          } else {
            p.out[node[nt] * p.out_stride + c] = v;
          }
        }
      }
    }
  }
}

// out[n, c] = sum over feature chunks of sum_partial[chunk, n, c], in chunk order (deterministic).
__global__ __launch_bounds__(256) void fmlp_chunk_sum_kernel(const Params p, int chunks) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= p.n * p.C) return;
  float a = 0.f;
  for (int ch = 0; ch < chunks; ++ch) a += p.sum_partial[static_cast<int64_t>(ch) * p.n * p.C + i];
  p.out[(i / p.C) * p.out_stride + i % p.C] = a;
}

constexpr int kNT = 2;  // 32-node tiles per wave

struct MfmaShape {
  int HT, NMID, CT;
};

bool mfma_shape(const Params& p, MfmaShape* s) {
  if (p.L < 3 || p.L > 4 || p.H < 1 || p.H > 64 || p.C > 8) return false;
  s->HT = p.H <= 32 ? 1 : 2;
  s->NMID = p.L - 2;
  s->CT = p.C <= 1 ? 1 : (p.C <= 2 ? 2 : (p.C <= 4 ? 4 : 8));
  return true;
}

// Feature chunks (gridDim.y): enough workgroups for ~4 per CU when the node axis alone is too short.
int feature_chunks(int64_t n, int F) {
  const int64_t node_blocks = (n + 4 * kNT * 32 - 1) / (4 * kNT * 32);
  int64_t chunks = (1024 + node_blocks - 1) / node_blocks;
  const int64_t max_chunks = (F + 1) / 2;          // at least two features per chunk (double-buffered pairs)
  chunks = chunks < 1 ? 1 : (chunks > max_chunks ? max_chunks : chunks);
  return static_cast<int>(chunks);
}

template <int HT, int NMID, int CT>
int launch_mfma(const Params& p, const float* packed, hipStream_t st) {
  constexpr size_t lds = 2 * static_cast<size_t>(feature_floats(HT, NMID, CT)) * sizeof(float);
  const int64_t nodes_per_block = 4 * kNT * 32;
  const int64_t blocks = (p.n + nodes_per_block - 1) / nodes_per_block;
  if (blocks > 0x7fffffffLL) return gnan::fail(GNAN_ERR_UNSUPPORTED, "fmlp: too many nodes for one launch");
  const int chunks = (p.F + p.feat_chunk - 1) / p.feat_chunk;
  auto go = [&](auto kernel) {
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "fmlp: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(blocks), static_cast<unsigned>(chunks)), dim3(256), lds, st,
                       p, packed);
    return gnan::check_launch("fmlp_mfma_kernel");
  };
  if (!p.sum_features) return go(fmlp_mfma_kernel<HT, NMID, CT, false, kNT>);
  if (int rc = go(fmlp_mfma_kernel<HT, NMID, CT, true, kNT>)) return rc;
  if (p.sum_partial) {
    const int64_t total = p.n * p.C;
    hipLaunchKernelGGL(fmlp_chunk_sum_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st, p,
                       chunks);
    return gnan::check_launch("fmlp_chunk_sum_kernel");
  }
  return GNAN_OK;
}

template <int HT, int NMID>
int launch_mfma_ct(const Params& p, int CT, const float* packed, hipStream_t st) {
  switch (CT) {
    case 1: return launch_mfma<HT, NMID, 1>(p, packed, st);
    case 2: return launch_mfma<HT, NMID, 2>(p, packed, st);
    case 4: return launch_mfma<HT, NMID, 4>(p, packed, st);
    default: return launch_mfma<HT, NMID, 8>(p, packed, st);
  }
}

size_t packed_bytes(const Params& p, const MfmaShape& s) {
  return static_cast<size_t>(p.F) * feature_floats(s.HT, s.NMID, s.CT) * sizeof(float);
}

int run_mfma(Params p, const MfmaShape& s, float* packed, hipStream_t st) {
  const int chunks = feature_chunks(p.n, p.F);
  p.feat_chunk = (p.F + chunks - 1) / chunks;
  p.feat_chunk += p.feat_chunk & 1;                  // even: chunks start on a double-buffer pair boundary
  const int real_chunks = (p.F + p.feat_chunk - 1) / p.feat_chunk;
  p.sum_partial = (p.sum_features && real_chunks > 1)
                      ? reinterpret_cast<float*>(reinterpret_cast<char*>(packed) + packed_bytes(p, s))
                      : nullptr;
  hipLaunchKernelGGL(fmlp_pack_kernel, dim3(p.F), dim3(256), 0, st, p, packed, s.HT, s.NMID, s.CT);
  if (int rc = gnan::check_launch("fmlp_pack_kernel")) return rc;
  if (s.HT == 1) return s.NMID == 1 ? launch_mfma_ct<1, 1>(p, s.CT, packed, st) : launch_mfma_ct<1, 2>(p, s.CT, packed, st);
  return s.NMID == 1 ? launch_mfma_ct<2, 1>(p, s.CT, packed, st) : launch_mfma_ct<2, 2>(p, s.CT, packed, st);
}

Params make_params(const gnan_fmlp_args* a) {
  Params p;
  p.x = a->x; p.n = a->n; p.x_stride = a->x_stride;
  p.F = a->F; p.L = a->L; p.H = a->L >= 2 ? a->H : 0; p.C = a->C;
  p.w_first = a->w_first; p.b_first = a->b_first; p.w_mid = a->w_mid; p.b_mid = a->b_mid;
  p.w_last = a->w_last; p.b_last = a->b_last;
  p.sum_features = a->sum_features; p.out = a->out; p.out_stride = a->out_stride;
  p.feat_chunk = a->F; p.sum_partial = nullptr;
  const bool drop = a->dropout_p > 0.f && a->L >= 2;
  p.drop_thresh = drop ? gnan::drop_threshold(a->dropout_p) : 0u;
  p.drop_scale = drop ? 1.f / (1.f - a->dropout_p) : 1.f;
  p.drop_seed = a->dropout_seed;
  return p;
}

}  // namespace

// a handful of evaluations under AUTO (no Dropout; with the feature sum only for a single feature): fmlp_point_kernel
static bool point_route(const gnan_fmlp_args* a) {
  return a->algo == GNAN_FMLP_AUTO && a->n * static_cast<int64_t>(a->F) <= 64 && (a->L == 2 || a->L == 3) && a->H >= 1 && a->H <= 64 &&
         !(a->dropout_p > 0.f) && (!a->sum_features || a->F == 1);
}

extern "C" size_t gnan_fmlp_fwd_workspace_bytes(const gnan_fmlp_args* a) {
  if (!a || a->algo == GNAN_FMLP_LANE || point_route(a)) return 0;
  const Params p = make_params(a);
  MfmaShape s;
  if (!mfma_shape(p, &s) || p.drop_thresh != 0u) return 0;
  size_t bytes = packed_bytes(p, s);
  if (p.sum_features) {                               // room for the per-chunk partial sums (upper bound)
    const int chunks = feature_chunks(p.n, p.F);
    if (chunks > 1) bytes += static_cast<size_t>(chunks) * p.n * p.C * sizeof(float);
  }
  return bytes;
}

extern "C" int gnan_fmlp_fwd(const gnan_fmlp_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "fmlp: null args");
  GNAN_REQUIRE(a->n >= 0 && a->F >= 1 && a->L >= 1 && a->C >= 1, "fmlp: bad sizes n=%lld F=%d L=%d C=%d",
               static_cast<long long>(a->n), a->F, a->L, a->C);
  if (a->n == 0) return GNAN_OK;
  GNAN_REQUIRE(a->x && a->out && a->w_last, "fmlp: null x / out / w_last");
  GNAN_REQUIRE(a->x_stride >= a->F, "fmlp: x row stride smaller than F");
  if (a->L >= 2) GNAN_REQUIRE(a->H >= 1 && a->w_first, "fmlp: L >= 2 needs H >= 1 and w_first");
  if (a->L >= 3) GNAN_REQUIRE(a->w_mid != nullptr, "fmlp: L >= 3 needs w_mid");
  const int64_t ow = a->sum_features ? a->C : static_cast<int64_t>(a->F) * a->C;
  GNAN_REQUIRE(a->out_stride >= ow, "fmlp: out row stride smaller than the output width");

  GNAN_REQUIRE(a->dropout_p >= 0.f && a->dropout_p < 1.f, "fmlp: dropout_p must be in [0, 1)");
  const Params p = make_params(a);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (point_route(a)) {
    hipLaunchKernelGGL(fmlp_point_kernel, dim3(static_cast<unsigned>(a->n * a->F)), dim3(64), 0, st, p);
    return gnan::check_launch("fmlp_point_kernel");
  }
  MfmaShape shape;
  const bool can_mfma = mfma_shape(p, &shape) && p.drop_thresh == 0u;        // Dropout: lane kernel
  if (a->algo == GNAN_FMLP_MFMA && p.drop_thresh != 0u)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "fmlp: training-mode Dropout runs on the lane kernel (algo AUTO or LANE)");
  if (a->algo == GNAN_FMLP_MFMA && !can_mfma)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "fmlp: MFMA path covers 3 <= L <= 4, H <= 64, C <= 8 (got L=%d H=%d C=%d)",
                      p.L, p.H, p.C);
  if (can_mfma && a->algo != GNAN_FMLP_LANE) {
    const size_t need = gnan_fmlp_fwd_workspace_bytes(a);
    if (a->workspace == nullptr || a->workspace_bytes < need)
      return gnan::fail(GNAN_ERR_WORKSPACE, "fmlp: workspace %zu B < required %zu B", a->workspace_bytes, need);
    GNAN_REQUIRE(reinterpret_cast<uintptr_t>(a->workspace) % 16 == 0, "fmlp: workspace must be 16-byte aligned");
    return run_mfma(p, shape, static_cast<float*>(a->workspace), st);
  }

  const size_t per_wave = static_cast<size_t>(2 * p.H + (p.sum_features ? p.C : 0)) * gnan::kWave * sizeof(float);
  const size_t budget = 64 * 1024;
  if (per_wave > 160 * 1024)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "fmlp: hidden width %d needs %zu B of LDS per wave (> 160 KiB)", p.H, per_wave);
  int waves = per_wave == 0 ? 4 : static_cast<int>(budget / per_wave);
  waves = waves < 1 ? 1 : (waves > 4 ? 4 : waves);
  const size_t lds = per_wave * waves;
  const int64_t nodes_per_block = static_cast<int64_t>(waves) * gnan::kWave;
  const int64_t blocks = (p.n + nodes_per_block - 1) / nodes_per_block;
  if (blocks > 0x7fffffffLL) return gnan::fail(GNAN_ERR_UNSUPPORTED, "fmlp: too many nodes for one launch");
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fmlp_lane_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "fmlp: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(fmlp_lane_kernel, dim3(static_cast<unsigned>(blocks)), dim3(waves * gnan::kWave), lds, st, p);
  return gnan::check_launch("fmlp_lane_kernel");
}


namespace {
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint64_t seed, uint32_t thresh, int64_t n, int F, int LH, int H,
                                                           uint8_t* __restrict__ mask) {
  const int64_t total = n * F * LH * H;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < total; e += static_cast<int64_t>(gridDim.x) * 256) {
    const int j = static_cast<int>(e % H);
    const int l = static_cast<int>((e / H) % LH);
    const int k = static_cast<int>((e / (static_cast<int64_t>(H) * LH)) % F);
    const int64_t node = e / (static_cast<int64_t>(H) * LH * F);
    mask[e] = gnan::drop_keep(gnan::drop_base(seed, node, k), l, j, thresh) ? 1 : 0;
  }
}
}  // namespace

extern "C" int gnan_dropout_mask(uint64_t seed, float p, int64_t n_nodes, int32_t F, int32_t n_hidden_layers, int32_t H,
                                 uint8_t* mask, gnan_stream_t stream) {
  GNAN_REQUIRE(p >= 0.f && p < 1.f && n_nodes >= 0 && F >= 1 && n_hidden_layers >= 1 && H >= 1, "dropout_mask: bad arguments");
  if (n_nodes == 0) return GNAN_OK;
  GNAN_REQUIRE(mask != nullptr, "dropout_mask: null output");
  const int64_t total = n_nodes * F * n_hidden_layers * H;
  int64_t blocks = (total + 255) / 256;
  blocks = blocks > 65536 ? 65536 : blocks;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), seed,
                     gnan::drop_threshold(p), n_nodes, F, n_hidden_layers, H, mask);
  return gnan::check_launch("dropout_mask_kernel");
}
