// Parameter gradients of the shape functions for SMALL batches (gfx950): what autograd computes behind
// GNAN.py:57-62 when trainer.py:66 calls backward() on graphs of a few thousand nodes (Cora, Mutagenicity), where the
// forward evaluates the MLPs directly (gnan_fmlp_fwd) and the table route's moments do not pay.
//
//   f_k(x) = W3 relu(W2 relu(w1 x + b1) + b2) + b3          (L == 3; L == 2 without the middle layer; H <= 64, C <= 8)
//   given g[n, c] = dLoss / d f_k(x[n, k])[c]:   d{w1, b1, W2, b2, W3, b3} of every feature k, summed over the nodes.
//
// One 256-thread workgroup per feature, nothing crosses workgroups (no atomics: the result does not depend on the
// schedule).  A wave owns a node at a time; lane j is hidden unit j of both hidden layers.  The lane keeps row j and
// column j of W2 and row j of dW2 in registers (192 VGPRs: one wave per SIMD), and the activations and deltas of the
// other units reach it through v_readlane with constant lane numbers — scalar operands of the fmas, no LDS in the node
// loop.  Per node and wave: 3 H^2 fmas (z2, dW2 += dz2 x h1, dh1 = W2^T dz2).  The four waves' partial sums meet in LDS
// in wave order at the end.
#include "fmlp_bwd_body.hpp"

namespace {

using gnan::kWave;
using gnan_bwd::kCmax;
using gnan_bwd::kH;
using gnan_bwd::kWaves;

struct BwdParams {
  const float* x;
  int64_t n, x_stride;
  int F, H, C;
  const float *w_first, *b_first, *w_mid, *b_mid, *w_last, *b_last;
  int sum_features;
  const float* grad;
  int64_t grad_stride;
  float *d_w_first, *d_b_first, *d_w_mid, *d_b_mid, *d_w_last, *d_b_last;
  int64_t nodes_per_split;   // blockIdx.y walks nodes [y * nodes_per_split, (y + 1) * nodes_per_split)
  int64_t split_stride;      // floats between the gradient blocks of consecutive splits (0: one split, final outputs)
  uint32_t drop_thresh;      // training-mode Dropout of the forward pass (csrc/dropout.hpp); 0 = none
  float drop_scale;
  uint64_t drop_seed;
};

// One 256-thread workgroup per feature (and node range), nothing crosses workgroups (no atomics: the result does not depend on
// the schedule); the work is gnan_bwd::feature_grads (csrc/fmlp_bwd_body.hpp).  MID: L == 3; L == 2 has no hidden-to-hidden
// matrix — the same mapping without the readlane loops.
template <int C, bool MID>
__global__ __launch_bounds__(kWaves * kWave) void fmlp_bwd_kernel(const BwdParams p) {
  __shared__ gnan_bwd::RedBuffer red;
  const int k = blockIdx.x;                       // feature
  gnan_bwd::Weights w;
  w.H = p.H;
  w.w_first = p.w_first; w.b_first = p.b_first; w.w_mid = p.w_mid; w.b_mid = p.b_mid; w.w_last = p.w_last; w.b_last = p.b_last;
  w.d_w_first = p.d_w_first; w.d_b_first = p.d_b_first; w.d_w_mid = p.d_w_mid; w.d_b_mid = p.d_b_mid;
  w.d_w_last = p.d_w_last; w.d_b_last = p.d_b_last;
  const gnan_bwd::Drop drop{p.drop_thresh, p.drop_scale, p.drop_seed};
  const int64_t n_lo = static_cast<int64_t>(blockIdx.y) * p.nodes_per_split;
  const int64_t n_hi = n_lo + p.nodes_per_split < p.n ? n_lo + p.nodes_per_split : p.n;
  const int64_t so = static_cast<int64_t>(blockIdx.y) * p.split_stride;     // this split's block of partial gradients
  const int64_t goff = p.sum_features ? 0 : static_cast<int64_t>(k) * C;
  gnan_bwd::feature_grads<C, MID>(
      w, k, n_lo, n_hi, so, drop, [&](int64_t node) { return p.x[node * p.x_stride + k]; },
      [&](int64_t node, int c) { return p.grad[node * p.grad_stride + goff + c]; }, red);
}

// out[i] = sum over the splits of partial[s * stride + i], in split order — all six gradient tensors in ONE launch (a launch per
// tensor was six launches behind a 15-us kernel on a few hundred inputs)
struct ReduceOuts {
  float* out[6];
  int64_t off[6];       // start of the tensor inside a split's block; off[t + 1] - off[t] elements (off[6] = block size)
  int64_t end;
};

__global__ __launch_bounds__(256) void fmlp_bwd_reduce_kernel(const float* __restrict__ partial, int64_t stride, int splits,
                                                              const ReduceOuts r) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= r.end) return;
  float a = partial[i];
  for (int s = 1; s < splits; ++s) a += partial[static_cast<int64_t>(s) * stride + i];
  int t = 0;
#pragma unroll
  for (int u = 1; u < 6; ++u) t = i >= r.off[u] ? u : t;
  if (r.out[t]) r.out[t][i - r.off[t]] = a;
}

// How many node ranges a feature's nodes are cut into: enough workgroups to fill the chip twice over, at least 256 nodes each.
int node_splits(int64_t n, int F) {
  int64_t want = (512 + F - 1) / F;
  // (at least 32 inputs each — 8 per wave: a single scalar MLP over a few hundred inputs, rho under the pre-rho normalisation
  // of a small graph, used to be ONE workgroup walking 80 inputs per wave: 50 us)
  const int64_t most = (n + 31) / 32;
  if (want > most) want = most;
  if (want > 1024) want = 1024;
  return want < 1 ? 1 : static_cast<int>(want);
}

int64_t block_floats(const gnan_fmlp_bwd_args* a) {       // one split's gradients: [w_first | b_first | w_mid | b_mid | w_last | b_last]
  const int64_t F = a->F, H = a->H, C = a->C;
  return 2 * F * H + (a->L == 3 ? F * H * H + F * H : 0) + F * C * H + F * C;
}

template <int C>
int launch_bwd(const BwdParams& p, unsigned splits, hipStream_t st) {
  if (p.w_mid == nullptr) {
    hipLaunchKernelGGL((fmlp_bwd_kernel<C, false>), dim3(static_cast<unsigned>(p.F), splits), dim3(kWaves * kWave), 0, st, p);
    return gnan::check_launch("fmlp_bwd_kernel<L=2>");
  }
  hipLaunchKernelGGL((fmlp_bwd_kernel<C, true>), dim3(static_cast<unsigned>(p.F), splits), dim3(kWaves * kWave), 0, st, p);
  return gnan::check_launch("fmlp_bwd_kernel");
}

}  // namespace

extern "C" int gnan_fmlp_bwd(const gnan_fmlp_bwd_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "fmlp_bwd: null args");
  GNAN_REQUIRE(a->n >= 0 && a->F >= 1, "fmlp_bwd: bad sizes");
  if ((a->L != 2 && a->L != 3) || a->H < 1 || a->H > kH || a->C < 1 || a->C > kCmax)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "fmlp_bwd: covers L in {2, 3}, H <= %d, C <= %d (got L=%d H=%d C=%d)", kH, kCmax,
                      a->L, a->H, a->C);
  GNAN_REQUIRE(a->w_first && a->w_last && a->d_w_first && a->d_w_last, "fmlp_bwd: null weight / gradient pointer");
  GNAN_REQUIRE(a->L == 2 || (a->w_mid && a->d_w_mid), "fmlp_bwd: L == 3 needs w_mid and d_w_mid");
  GNAN_REQUIRE((a->b_first == nullptr) == (a->d_b_first == nullptr) &&
                   (a->L == 2 || (a->b_mid == nullptr) == (a->d_b_mid == nullptr)) &&
                   (a->b_last == nullptr) == (a->d_b_last == nullptr),
               "fmlp_bwd: a bias gradient is wanted exactly where there is a bias");
  const int64_t gw = a->sum_features ? a->C : static_cast<int64_t>(a->F) * a->C;
  GNAN_REQUIRE(a->n == 0 || (a->x && a->grad && a->x_stride >= a->F && a->grad_stride >= gw), "fmlp_bwd: bad x / grad");
  BwdParams p;
  p.x = a->x; p.n = a->n; p.x_stride = a->x_stride; p.F = a->F; p.H = a->H; p.C = a->C;
  p.w_first = a->w_first; p.b_first = a->b_first;
  p.w_mid = a->L == 3 ? a->w_mid : nullptr; p.b_mid = a->L == 3 ? a->b_mid : nullptr;
  p.w_last = a->w_last; p.b_last = a->b_last;
  p.sum_features = a->sum_features; p.grad = a->grad; p.grad_stride = a->grad_stride;
  p.d_w_first = a->d_w_first; p.d_b_first = a->d_b_first; p.d_w_mid = a->d_w_mid; p.d_b_mid = a->d_b_mid;
  p.d_w_last = a->d_w_last; p.d_b_last = a->d_b_last;
  GNAN_REQUIRE(a->dropout_p >= 0.f && a->dropout_p < 1.f, "fmlp_bwd: dropout_p must be in [0, 1)");
  p.drop_thresh = a->dropout_p > 0.f ? gnan::drop_threshold(a->dropout_p) : 0u;
  p.drop_scale = a->dropout_p > 0.f ? 1.f / (1.f - a->dropout_p) : 1.f;
  p.drop_seed = a->dropout_seed;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // few features and many nodes: a feature's nodes are cut into ranges, one workgroup each; the ranges' gradients land in
  // the workspace and are added in range order (still no atomics)
  const int splits = a->n > 0 ? node_splits(a->n, a->F) : 1;
  const int64_t blk = block_floats(a);
  p.nodes_per_split = a->n;
  p.split_stride = 0;
  const int64_t F = a->F, H = a->H, C = a->C;
  const int64_t off_b1 = F * H, off_w2 = 2 * F * H, off_b2 = off_w2 + (a->L == 3 ? F * H * H : 0),
                off_w3 = off_b2 + (a->L == 3 ? F * H : 0), off_b3 = off_w3 + F * C * H;
  if (splits > 1) {
    const size_t need = static_cast<size_t>(splits) * blk * sizeof(float);
    if (a->workspace == nullptr || a->workspace_bytes < need)
      return gnan::fail(GNAN_ERR_WORKSPACE, "fmlp_bwd: workspace %zu B < required %zu B", a->workspace_bytes, need);
    float* ws = static_cast<float*>(a->workspace);
    p.nodes_per_split = ((a->n + splits - 1) / splits + kWaves - 1) / kWaves * kWaves;
    p.split_stride = blk;
    p.d_w_first = ws;
    p.d_b_first = a->d_b_first ? ws + off_b1 : nullptr;
    p.d_w_mid = ws + off_w2;
    p.d_b_mid = a->d_b_mid ? ws + off_b2 : nullptr;
    p.d_w_last = ws + off_w3;
    p.d_b_last = a->d_b_last ? ws + off_b3 : nullptr;
  }
  int rc;
  const unsigned sp = static_cast<unsigned>(splits);
  switch (a->C) {
    case 1: rc = launch_bwd<1>(p, sp, st); break;
    case 2: rc = launch_bwd<2>(p, sp, st); break;
    case 3: rc = launch_bwd<3>(p, sp, st); break;
    case 4: rc = launch_bwd<4>(p, sp, st); break;
    case 5: rc = launch_bwd<5>(p, sp, st); break;
    case 6: rc = launch_bwd<6>(p, sp, st); break;
    case 7: rc = launch_bwd<7>(p, sp, st); break;
    default: rc = launch_bwd<8>(p, sp, st); break;
  }
  if (rc || splits == 1) return rc;
  const float* ws = static_cast<const float*>(a->workspace);
  ReduceOuts r;
  r.out[0] = a->d_w_first; r.out[1] = a->d_b_first; r.out[2] = a->L == 3 ? a->d_w_mid : nullptr;
  r.out[3] = a->L == 3 ? a->d_b_mid : nullptr; r.out[4] = a->d_w_last; r.out[5] = a->d_b_last;
  r.off[0] = 0; r.off[1] = off_b1; r.off[2] = off_w2; r.off[3] = off_b2; r.off[4] = off_w3; r.off[5] = off_b3;
  r.end = blk;
  hipLaunchKernelGGL(fmlp_bwd_reduce_kernel, dim3(static_cast<unsigned>((blk + 255) / 256)), dim3(256), 0, st, ws, blk, splits, r);
  return gnan::check_launch("fmlp_bwd_reduce_kernel");
}

extern "C" size_t gnan_fmlp_bwd_workspace_bytes(const gnan_fmlp_bwd_args* a) {
  if (!a || a->n <= 0 || a->F < 1 || a->H < 1 || a->C < 1) return 0;
  const int splits = node_splits(a->n, a->F);
  return splits > 1 ? static_cast<size_t>(splits) * block_floats(a) * sizeof(float) : 0;
}
