// A small graph-level task whose read-out is a NAM over the per-feature aggregates (models.py:358-384 with is_graph_task and
// readout_n_layers > 0):  out[c] = sum_k nam_k( hidden_k )[c],  hidden_k = sum_i sum_j m_ij f_k(x_jk) = sum_j colw_j f_k(x_jk),
// colw_j = sum_i rho(u_code(i,j)) / max(cnt(i, code(i,j)), 1).  f and rho are one-wide here (models.py:320-321).
//
// Forward, ONE launch, F workgroups that never wait for each other: workgroup k evaluates rho on the D distinct distances
// itself (a block of 64 inputs, what one block of nodes costs), forms the rows' weights and the column sums colw over
// LDS-resident hop codes, evaluates f_k on the nodes, reduces hidden_k with a fixed butterfly, and runs NAM_k on it (the MLP
// block with every lane holding the same input); the last workgroup to arrive adds the F contributions in feature order.
// Left behind for the backward pass: fx [F, n], the rho table [D], hidden [F].
//
// Backward, ONE launch, F + 1 workgroups.  Workgroup k < F: NAM_k's parameter gradients and d hidden_k (gnan_fmlp_bwd's body
// on the single input, with its input gradient), published to the workspace; colw again; f_k's parameter gradients from
// dfx[j] = colw_j * d hidden_k.  Workgroup F (the highest index: dispatched after the others, which depend on nothing, so its
// wait cannot starve them): stages its tables, waits for the F published values, T_j = sum_k d hidden_k fx[k, j], then the
// table gradient dlut[d] = sum_i 1 / max(cnt(i, d), 1) sum_{j : code(i, j) == d} T_j by small_graph_bwd_kernel's binning and
// rho's parameter gradients.  Fixed orders throughout: bit-reproducible.
#include "small_graph_body.hpp"

namespace {

using namespace gnan_small;

struct NamParams {
  const float* x;
  int64_t x_stride;
  int n, F;
  Mlp f, r, nam;             // nam.L may be 1 (Linear(1, C): w_last [F, C])
  const uint8_t* code;
  int D;
  const int32_t* cnt;
  int64_t cnt_stride;
  float *fx, *lut, *hidden, *out, *part;
  unsigned* counter;
};

constexpr int kNodes = kMaxNodes;                     // both kernels are built for two blocks of 64 nodes
constexpr int kWordsPer = kNodes * kNodes / 4 / (kWaves * kWave);
constexpr int kCntPer = kNodes * kWave / (kWaves * kWave);
constexpr int nam_cols_floats() {
  constexpr int tables = kNodes * kWave + kNodes * kNodes / 4;            // rows' weights [n][64] | hop codes [n][n]
  return tables > 2 * kMaxH * kWave ? tables : 2 * kMaxH * kWave;
}

// column sums of the normalised weights from LDS-resident rows' weights and codes
__device__ __forceinline__ void column_weights(const float* s_w, const uint8_t* s_code, int n, int D, float* s_colw) {
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    float acc = 0.f;
#pragma unroll 8
    for (int i = 0; i < n; ++i) {
      int d = s_code[i * n + j];
      d = d < D - 1 ? d : D - 1;
      acc += s_w[i * kWave + d];
    }
    s_colw[j] = acc;
  }
}

__global__ __launch_bounds__(256) void small_graph_nam_kernel(const NamParams p) {
  extern __shared__ __attribute__((aligned(16))) float cols[];               // nam_cols_floats()
  __shared__ __attribute__((aligned(16))) float weights[kWeightFloats];
  __shared__ float s_lut[kWave], s_colw[kNodes], s_fx[kNodes];
  __shared__ unsigned s_last;
  float* col_a = cols;
  float* col_b = cols + kMaxH * kWave;
  float* s_w = cols;
  uint8_t* s_code = reinterpret_cast<uint8_t*>(cols + kNodes * kWave);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int k = blockIdx.x, n = p.n, D = p.D, C = p.nam.C;
  const MlpLds wl_ = carve(weights);
  float out[kMaxC];
  // everything this workgroup reads that is not a weight, requested up front
  const int code_words = n * n / 4;
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
      pre_cnt[t] = (p.cnt && e < n * D) ? p.cnt[(e / D) * p.cnt_stride + e % D] : 1;
    }
  }
  float xv[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int j = b * kWave + lane;
    xv[b] = j < n ? p.x[static_cast<int64_t>(j) * p.x_stride + k] : 0.f;
  }
  // ---- rho on the distinct distances -------------------------------------------------------------------------------------
  stage_weights(p.r, 0, wl_);
  {
    const float u = lane < D - 1 ? 1.0f / (static_cast<float>(lane) + 1.0f) : 0.f;      // graph.hop_inputs
    mlp_block(p.r, wl_, u, col_a, col_b, lane, wave, out);
    if (wave == 0 && lane < D) {
      s_lut[lane] = out[0];
      if (k == 0) p.lut[lane] = out[0];
    }
  }
  __syncthreads();
  // ---- the rows' weights, the codes, the column sums -------------------------------------------------------------------------
#pragma unroll
  for (int t = 0; t < kCntPer; ++t) {
    const int e = threadIdx.x + t * 256;
    if (e < n * D) {
      const float l = s_lut[e % D];
      s_w[(e / D) * kWave + e % D] = p.cnt ? l / static_cast<float>(pre_cnt[t] > 1 ? pre_cnt[t] : 1) : l;
    }
  }
  {
    uint32_t* dw = reinterpret_cast<uint32_t*>(s_code);
#pragma unroll
    for (int t = 0; t < kWordsPer; ++t) {
      const int i = threadIdx.x + t * 256;
      if (i < code_words) dw[i] = pre_code[t];
    }
    for (int i = code_words * 4 + threadIdx.x; i < n * n; i += 256) s_code[i] = p.code[i];
  }
  __syncthreads();
  column_weights(s_w, s_code, n, D, s_colw);
  __syncthreads();                                   // (the tables are dead: the columns take their place)
  // ---- f_k on the nodes, hidden_k ------------------------------------------------------------------------------------------------
  stage_weights(p.f, k, wl_);
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int j = b * kWave + lane;
    if (b * kWave < n) {                             // (uniform)
      mlp_block(p.f, wl_, xv[b], col_a, col_b, lane, wave, out);
      if (wave == 0 && j < n) {
        s_fx[j] = out[0];
        p.fx[static_cast<int64_t>(k) * n + j] = out[0];
      }
    }
  }
  __syncthreads();
  float h = 0.f;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int j = b * kWave + lane;
    if (j < n) h = fmaf(s_colw[j], s_fx[j], h);
  }
#pragma unroll
  for (int off = kWave / 2; off >= 1; off >>= 1) h += __shfl_xor(h, off, kWave);      // the same sum in every lane of every wave
  if (threadIdx.x == 0) p.hidden[k] = h;
  // ---- NAM_k -------------------------------------------------------------------------------------------------------------------
  if (p.nam.L == 1) {
    if (static_cast<int>(threadIdx.x) < C)
      p.part[k * C + threadIdx.x] = fmaf(p.nam.w_last[k * C + threadIdx.x], h, p.nam.b_last ? p.nam.b_last[k * C + threadIdx.x] : 0.f);
  } else {
    stage_weights(p.nam, k, wl_);
    mlp_block(p.nam, wl_, h, col_a, col_b, lane, wave, out);
    if (threadIdx.x == 0)
      for (int c = 0; c < C; ++c) p.part[k * C + c] = out[c];
  }
  // ---- join: the last workgroup adds the features in order ------------------------------------------------------------------------
  if (!last_to_arrive(p.counter, gridDim.x, &s_last)) return;
  // (every term fetched by a thread of its own — all loads in flight together — then added in feature order from LDS)
  const int chunk = kWeightFloats / C * C;           // whole features per pass through the (now free) weight image
  for (int e0 = 0; e0 < p.F * C; e0 += chunk) {
    const int en = p.F * C - e0 < chunk ? p.F * C - e0 : chunk;
    __syncthreads();
    for (int e = threadIdx.x; e < en; e += 256) weights[e] = p.part[e0 + e];
    __syncthreads();
    if (static_cast<int>(threadIdx.x) < C) {
      float s = e0 == 0 ? 0.f : s_colw[threadIdx.x];
      for (int e = threadIdx.x; e < en; e += C) s += weights[e];
      s_colw[threadIdx.x] = s;
    }
  }
  __syncthreads();
  if (static_cast<int>(threadIdx.x) < C) p.out[threadIdx.x] = s_colw[threadIdx.x];
  if (threadIdx.x == 0) *p.counter = 0u;
}

// ---------------------------------------------------------------------------------------------------------------------------------
struct NamBwdParams {
  const float* x;
  int64_t x_stride;
  int n, F;
  gnan_bwd::Weights f, r, nam;
  int f_mid, r_mid, nam_L;
  const uint8_t* code;
  int D;
  const int32_t* cnt;
  int64_t cnt_stride;
  const float *fx, *lut, *hidden, *d_out;
  float* dh;                 // [F] workspace: d hidden_k, published by workgroup k
  unsigned* counter;         // zero before the launch; workgroup F zeroes it again
};

constexpr int kUFloats = kWave * (kNodes | 1);            // rows' weights [n][64]; one wave's bins [D <= 64][n | 1] (rho)
constexpr size_t nam_bwd_dyn_bytes() { return static_cast<size_t>(kUFloats) * sizeof(float) + static_cast<size_t>(kNodes) * kNodes; }

template <int CN>
__global__ __launch_bounds__(256) void small_graph_nam_bwd_kernel(const NamBwdParams p) {
  __shared__ gnan_bwd::RedBuffer red;
  extern __shared__ __attribute__((aligned(16))) float dyn[];                   // s_u | s_code
  float* s_u = dyn;                                   // rows' weights [n][64] (features) | bins [waves][D][n | 1] (rho)
  uint8_t* s_code = reinterpret_cast<uint8_t*>(dyn + kUFloats);
  __shared__ float s_g[kNodes];                       // colw (features) | T (rho)
  __shared__ float s_l[kWave];                        // rho table, then its gradient
  __shared__ float s_dh[kMaxH];                       // d hidden (features: entry 0; rho: all F... see below)
  __shared__ double s_part[kWaves][kWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int k = blockIdx.x, n = p.n, D = p.D;
  const bool is_rho = k == p.F;
  {
    const int words = n * n / 4;
    const uint32_t* cw = reinterpret_cast<const uint32_t*>(p.code);
    uint32_t v[kWordsPer];
#pragma unroll
    for (int t = 0; t < kWordsPer; ++t) {
      const int i = threadIdx.x + t * 256;
      v[t] = i < words ? cw[i] : 0u;
    }
    int q[kCntPer];
#pragma unroll
    for (int t = 0; t < kCntPer; ++t) {
      const int e = threadIdx.x + t * 256;
      q[t] = (!is_rho && p.cnt && e < n * D) ? p.cnt[(e / D) * p.cnt_stride + e % D] : 1;
    }
    const float l = lane < D ? p.lut[lane] : 0.f;
    uint32_t* dw = reinterpret_cast<uint32_t*>(s_code);
#pragma unroll
    for (int t = 0; t < kWordsPer; ++t) {
      const int i = threadIdx.x + t * 256;
      if (i < words) dw[i] = v[t];
    }
    for (int i = words * 4 + threadIdx.x; i < n * n; i += 256) s_code[i] = p.code[i];
    if (wave == 0 && lane < D) s_l[lane] = l;
    __syncthreads();
    if (!is_rho) {
#pragma unroll
      for (int t = 0; t < kCntPer; ++t) {
        const int e = threadIdx.x + t * 256;
        if (e < n * D) {
          const float lv = s_l[e % D];
          s_u[(e / D) * kWave + e % D] = p.cnt ? lv / static_cast<float>(q[t] > 1 ? q[t] : 1) : lv;
        }
      }
    }
    __syncthreads();
  }
  const gnan_bwd::Drop nodrop{0u, 1.f, 0ull};
  if (!is_rho) {
    // ---- NAM_k: parameter gradients and d hidden_k ------------------------------------------------------------------------------
    const float h = p.hidden[k];
    if (p.nam_L == 1) {
      if (static_cast<int>(threadIdx.x) < CN) {
        const float g = p.d_out[threadIdx.x];
        p.nam.d_w_last[k * CN + threadIdx.x] = g * h;
        if (p.nam.d_b_last) p.nam.d_b_last[k * CN + threadIdx.x] = g;
      }
      if (threadIdx.x == 0) {
        float s = 0.f;
        for (int c = 0; c < CN; ++c) s = fmaf(p.d_out[c], p.nam.w_last[k * CN + c], s);
        s_dh[0] = s;
      }
    } else {
      auto h_of = [&](int64_t) { return h; };
      auto go_of = [&](int64_t, int c) { return p.d_out[c]; };
      if (p.nam_L == 3) gnan_bwd::feature_grads<CN, true, true>(p.nam, k, 0, 1, 0, nodrop, h_of, go_of, red, s_dh);
      else gnan_bwd::feature_grads<CN, false, true>(p.nam, k, 0, 1, 0, nodrop, h_of, go_of, red, s_dh);
    }
    __syncthreads();
    const float dh = s_dh[0];
    if (threadIdx.x == 0) {
      __hip_atomic_store(p.dh + k, dh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence();
      atomicAdd(p.counter, 1u);
    }
    // ---- f_k's parameter gradients from dfx[j] = colw_j * d hidden_k ---------------------------------------------------------------
    column_weights(s_u, s_code, n, D, s_g);
    __syncthreads();
    auto x_of = [&](int64_t node) { return p.x[node * p.x_stride + k]; };
    auto g_of = [&](int64_t node, int) { return s_g[node] * dh; };
    if (p.f_mid) gnan_bwd::feature_grads<1, true>(p.f, k, 0, n, 0, nodrop, x_of, g_of, red);
    else gnan_bwd::feature_grads<1, false>(p.f, k, 0, n, 0, nodrop, x_of, g_of, red);
    return;
  }
  // ---- rho: wait for the F published d hidden_k (their producers were dispatched before this workgroup and wait for nobody) ------------
  if (threadIdx.x == 0) {
    while (__hip_atomic_load(p.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < static_cast<unsigned>(p.F))
      __builtin_amdgcn_s_sleep(8);
  }
  if (threadIdx.x == 0) __threadfence();               // (one acquire for the workgroup: see last_to_arrive)
  __syncthreads();
  // T_j = sum_k d hidden_k fx[k, j]: chunks of 64 features through LDS
  for (int j = threadIdx.x; j < n; j += 256) s_g[j] = 0.f;
  for (int k0 = 0; k0 < p.F; k0 += kMaxH) {
    __syncthreads();
    if (static_cast<int>(threadIdx.x) < kMaxH && k0 + static_cast<int>(threadIdx.x) < p.F)
      s_dh[threadIdx.x] = __hip_atomic_load(p.dh + k0 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int kn = p.F - k0 < kMaxH ? p.F - k0 : kMaxH;
    for (int j = threadIdx.x; j < n; j += 256) {
      float t = s_g[j];
      for (int kk = 0; kk < kn; kk += 8) {            // eight features' terms requested together (a load -> fma loop waits for each)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = kk + u < kn ? p.fx[static_cast<int64_t>(k0 + kk + u) * n + j] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) t = fmaf(kk + u < kn ? s_dh[kk + u] : 0.f, v[u], t);
      }
      s_g[j] = t;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) *p.counter = 0u;               // (every producer has added its one: nobody touches it again)
  // table gradient (small_graph_bwd_kernel's binning with dY = 1 and S = T)
  const int stride = n | 1;
  int nw = 1;
  while (nw < 4 && 2 * nw * D * stride <= kUFloats) nw *= 2;
  float* bins = s_u + wave * D * stride;
  double acc = 0.0;
  for (int r = 0; nw * r < n; ++r) {
    const int i = nw * r + wave;
    const bool live = wave < nw && i < n;
    if (live) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int j = b * kWave + lane;
        if (j < n) {
          for (int d = 0; d < D; ++d) bins[d * stride + j] = 0.f;
          int d = s_code[i * n + j];
          d = d < D - 1 ? d : D - 1;
          bins[d * stride + j] = s_g[j];
        }
      }
    }
    __syncthreads();
    if (live && lane < D) {
      float sum = 0.f;
#pragma unroll 8
      for (int l = 0; l < n; ++l) sum += bins[lane * stride + l];        // (unrolled: eight reads in flight, not one)
      if (p.cnt) {
        const int c = p.cnt[i * p.cnt_stride + lane];
        sum *= 1.f / static_cast<float>(c > 1 ? c : 1);
      }
      acc += static_cast<double>(sum);
    }
    __syncthreads();
  }
  s_part[wave][lane] = wave < nw ? acc : 0.0;
  __syncthreads();
  if (wave == 0 && lane < D) s_l[lane] = static_cast<float>(((s_part[0][lane] + s_part[1][lane]) + s_part[2][lane]) + s_part[3][lane]);
  __syncthreads();
  auto u_of = [&](int64_t node) { return node < D - 1 ? 1.0f / (static_cast<float>(node) + 1.0f) : 0.f; };
  auto gl_of = [&](int64_t node, int) { return s_l[node]; };
  if (p.r_mid) gnan_bwd::feature_grads<1, true>(p.r, 0, 0, D, 0, nodrop, u_of, gl_of, red);
  else gnan_bwd::feature_grads<1, false>(p.r, 0, 0, D, 0, nodrop, u_of, gl_of, red);
}

constexpr int kMaxBwdWorkgroups = 128;

template <int CN>
int launch_nam_bwd(const NamBwdParams& p, hipStream_t st) {
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&small_graph_nam_bwd_kernel<CN>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(nam_bwd_dyn_bytes()));
  if (attr != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "small_graph_nam_bwd: hipFuncSetAttribute: %s", hipGetErrorString(attr));
  hipLaunchKernelGGL((small_graph_nam_bwd_kernel<CN>), dim3(static_cast<unsigned>(p.F) + 1), dim3(256), nam_bwd_dyn_bytes(), st, p);
  return gnan::check_launch("small_graph_nam_bwd_kernel");
}

bool nam_ok(const gnan_small_mlp* m) {
  if (m->L == 1) return m->C >= 1 && m->C <= kMaxC && m->w_last != nullptr;
  return mlp_ok(m, kMaxC);
}

int check_common(const char* who, int n, int F, int D, const gnan_small_mlp* f, const gnan_small_mlp* rho, const gnan_small_mlp* nam) {
  GNAN_REQUIRE(n >= 1 && F >= 1 && D >= 1, "%s: bad sizes n=%d F=%d D=%d", who, n, F, D);
  if (n > kMaxNodes || D > kWave || !mlp_ok(f, 1) || !mlp_ok(rho, 1) || !nam_ok(nam))
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "%s: covers n <= %d nodes, D <= %d shells, one-wide f and rho with L in {2, 3}, H <= %d, a read-out "
                      "with L in {1, 2, 3} and C <= %d (got n=%d D=%d L=%d/%d/%d H=%d/%d/%d C=%d/%d/%d)", who, kMaxNodes, kWave, kMaxH,
                      kMaxC, n, D, f->L, rho->L, nam->L, f->H, rho->H, nam->H, f->C, rho->C, nam->C);
  return GNAN_OK;
}

}  // namespace

extern "C" size_t gnan_small_graph_nam_workspace_bytes(int32_t F, int32_t C) {
  const size_t fwd = 16 + static_cast<size_t>(F) * C * sizeof(float);          // counter (own 16 bytes) | part [F, C]
  const size_t bwd = 16 + static_cast<size_t>(F) * sizeof(float);              // counter | d hidden [F]
  return fwd > bwd ? fwd : bwd;
}

extern "C" int gnan_small_graph_nam_fwd(const gnan_small_graph_nam_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "small_graph_nam: null args");
  if (int rc = check_common("small_graph_nam", a->n, a->F, a->D, &a->f, &a->rho, &a->nam)) return rc;
  GNAN_REQUIRE(a->x && a->code && a->fx && a->lut && a->hidden && a->out, "small_graph_nam: null x / code / fx / lut / hidden / out");
  GNAN_REQUIRE(a->x_stride >= a->F && (a->cnt == nullptr || a->cnt_stride >= a->D), "small_graph_nam: row stride smaller than the width");
  const size_t need = gnan_small_graph_nam_workspace_bytes(a->F, a->nam.C);
  if (a->workspace == nullptr || a->workspace_bytes < need)
    return gnan::fail(GNAN_ERR_WORKSPACE, "small_graph_nam: workspace %zu B < required %zu B", a->workspace_bytes, need);
  NamParams p;
  p.x = a->x; p.x_stride = a->x_stride; p.n = a->n; p.F = a->F;
  p.f = to_mlp(&a->f); p.r = to_mlp(&a->rho); p.nam = to_mlp(&a->nam);
  p.code = a->code; p.D = a->D; p.cnt = a->cnt; p.cnt_stride = a->cnt_stride;
  p.fx = a->fx; p.lut = a->lut; p.hidden = a->hidden; p.out = a->out;
  p.counter = static_cast<unsigned*>(a->workspace);
  p.part = reinterpret_cast<float*>(static_cast<char*>(a->workspace) + 16);
  constexpr size_t lds = nam_cols_floats() * sizeof(float);
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&small_graph_nam_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  if (attr != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "small_graph_nam: hipFuncSetAttribute: %s", hipGetErrorString(attr));
  hipLaunchKernelGGL(small_graph_nam_kernel, dim3(static_cast<unsigned>(a->F)), dim3(kWaves * kWave), lds,
                     static_cast<hipStream_t>(stream), p);
  return gnan::check_launch("small_graph_nam_kernel");
}

extern "C" int gnan_small_graph_nam_bwd(const gnan_small_graph_nam_bwd_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "small_graph_nam_bwd: null args");
  if (int rc = check_common("small_graph_nam_bwd", a->n, a->F, a->D, &a->f, &a->rho, &a->nam)) return rc;
  // Workgroup F waits for the F producers of the same launch.  They wait for nobody, so the wait ends as soon as every one
  // of them has been dispatched — whatever order the dispatcher picks — PROVIDED a producer can always find a free slot next
  // to the waiting workgroup: the launch is kept small enough (at most one workgroup per compute unit of the smallest
  // gfx950 part) that producers and the waiter are resident together even with other work on the device.
  if (a->F + 1 > kMaxBwdWorkgroups)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "small_graph_nam_bwd: covers F <= %d features (one co-resident workgroup each; got F=%d)",
                      kMaxBwdWorkgroups - 1, a->F);
  GNAN_REQUIRE(a->x && a->code && a->fx && a->lut && a->hidden && a->d_out, "small_graph_nam_bwd: null x / code / fx / lut / hidden / d_out");
  GNAN_REQUIRE(a->x_stride >= a->F && (a->cnt == nullptr || a->cnt_stride >= a->D), "small_graph_nam_bwd: row stride smaller than the width");
  GNAN_REQUIRE(grads_ok(&a->f, &a->df) && grads_ok(&a->rho, &a->drho),
               "small_graph_nam_bwd: a gradient pointer for every weight, and for a bias exactly where there is one");
  if (a->nam.L == 1)
    GNAN_REQUIRE(a->dnam.w_last && ((a->nam.b_last == nullptr) == (a->dnam.b_last == nullptr)),
                 "small_graph_nam_bwd: gradient pointers of the one-layer read-out");
  else
    GNAN_REQUIRE(grads_ok(&a->nam, &a->dnam), "small_graph_nam_bwd: a gradient pointer for every weight of the read-out");
  const size_t need = gnan_small_graph_nam_workspace_bytes(a->F, a->nam.C);
  if (a->workspace == nullptr || a->workspace_bytes < need)
    return gnan::fail(GNAN_ERR_WORKSPACE, "small_graph_nam_bwd: workspace %zu B < required %zu B", a->workspace_bytes, need);
  NamBwdParams p;
  p.x = a->x; p.x_stride = a->x_stride; p.n = a->n; p.F = a->F;
  p.f = to_weights(&a->f, &a->df); p.r = to_weights(&a->rho, &a->drho); p.nam = to_weights(&a->nam, &a->dnam);
  p.f_mid = a->f.L == 3; p.r_mid = a->rho.L == 3; p.nam_L = a->nam.L;
  p.code = a->code; p.D = a->D; p.cnt = a->cnt; p.cnt_stride = a->cnt_stride;
  p.fx = a->fx; p.lut = a->lut; p.hidden = a->hidden; p.d_out = a->d_out;
  p.counter = static_cast<unsigned*>(a->workspace);
  p.dh = reinterpret_cast<float*>(static_cast<char*>(a->workspace) + 16);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (a->nam.C) {
    case 1: return launch_nam_bwd<1>(p, st);
    case 2: return launch_nam_bwd<2>(p, st);
    case 3: return launch_nam_bwd<3>(p, st);
    case 4: return launch_nam_bwd<4>(p, st);
    case 5: return launch_nam_bwd<5>(p, st);
    case 6: return launch_nam_bwd<6>(p, st);
    case 7: return launch_nam_bwd<7>(p, st);
    default: return launch_nam_bwd<8>(p, st);
  }
}
