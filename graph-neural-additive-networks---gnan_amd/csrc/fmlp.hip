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
#include "common.hpp"

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
    for (int j = 0; j < H; ++j) {
      const float b = p.b_first ? p.b_first[k * H + j] : 0.f;
      buf_a[j * kWave] = fmaxf(fmaf(xv, p.w_first[k * H + j], b), 0.f);
    }
    float* cur = buf_a;
    float* nxt = buf_b;
    for (int l = 0; l < p.L - 2; ++l) {
      const int64_t f = static_cast<int64_t>(l) * p.F + k;
      dense_layer<true>(p.w_mid + f * H * H, p.b_mid ? p.b_mid + f * H : nullptr, H, H, cur,
                        [&](int j, float v) { nxt[j * kWave] = v; });
      float* t = cur; cur = nxt; nxt = t;
    }
    dense_layer<false>(p.w_last + static_cast<int64_t>(k) * C * H, p.b_last ? p.b_last + k * C : nullptr, C, H, cur,
                       emit_out);
  }
  if (p.sum_features && valid)
    for (int c = 0; c < C; ++c) p.out[node * p.out_stride + c] = sacc[c * kWave];
}

}  // namespace

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

  Params p;
  p.x = a->x; p.n = a->n; p.x_stride = a->x_stride;
  p.F = a->F; p.L = a->L; p.H = a->L >= 2 ? a->H : 0; p.C = a->C;
  p.w_first = a->w_first; p.b_first = a->b_first; p.w_mid = a->w_mid; p.b_mid = a->b_mid;
  p.w_last = a->w_last; p.b_last = a->b_last;
  p.sum_features = a->sum_features; p.out = a->out; p.out_stride = a->out_stride;

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
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fmlp_lane_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "fmlp: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(fmlp_lane_kernel, dim3(static_cast<unsigned>(blocks)), dim3(waves * gnan::kWave), lds, st, p);
  return gnan::check_launch("fmlp_lane_kernel");
}
