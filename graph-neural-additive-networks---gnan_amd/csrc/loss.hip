// The epoch loops' loss step in one launch (two for more than a few thousand rows): row selection by the task mask, the
// loss of trainer.py:61-64 (BCEWithLogitsLoss on one logit per row, CrossEntropyLoss on C > 1), its gradient w.r.t. the
// logits, the hit count of trainer.py:5-20 and the epoch's running totals (trainer.py:67-71).  Stock torch issues 20-25
// element-wise and reduction kernels for the same numbers — more than the model's own kernels on a 30-node graph.
//   BCE:  l_i = (1 - t_i) x_i - logsigmoid(x_i)        dl/dx_i = (sigmoid(x_i) - t_i) / n      hit: (sigmoid(x_i) > 0.5) == t_i
//   CE:   l_i = logsumexp(x_i) - x_i[t_i]              dl/dx_i = (softmax(x_i) - e_{t_i}) / n  hit: argmax x_i == t_i
//   loss = mean_i l_i   (float32 terms as torch computes them, float64 across rows, fixed order: bit-reproducible)
// The mean is over ALL n rows handed over (torch's default options): class labels outside [0, C) — ignore_index rows, which
// torch leaves out of the mean — are not this kernel's business; gnan_amd/harness.py keeps the eager loss for such labels, and
// callers that cannot see the labels on the host (replayed steps) hand over `label_flag`, which such a row sets.
#include "common.hpp"

namespace {

struct LossParams {
  const float* x;
  int64_t x_stride;
  int C;
  const int64_t* index;
  int64_t n;
  const float* t_f;
  const int64_t* t_i;
  float* loss;
  int64_t* hits;
  float* grad;
  int64_t grad_stride;
  float* loss_sum;
  float* hits_sum;
  const float* skip_sums;   // optional flag: non-zero = leave the running totals alone (a replayed step whose guard tripped)
  float* label_flag;        // optional flag: set to 1 by a cross-entropy row whose label lies outside [0, C)
  double* partial;   // [blocks, 2]
  int blocks;
};

__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }

__device__ __forceinline__ void finish(const LossParams& p, double loss_total, double hit_total) {
  const float loss = static_cast<float>(loss_total / static_cast<double>(p.n));
  *p.loss = loss;
  if (p.hits) *p.hits = static_cast<int64_t>(hit_total);
  if (p.skip_sums && *p.skip_sums != 0.f) return;     // the caller re-runs this step eagerly and counts it then
  if (p.loss_sum) *p.loss_sum += loss;
  if (p.hits_sum) *p.hits_sum += static_cast<float>(hit_total);
}

// rows blockIdx.x * rows_per_block ...; CE: one thread per row walks its C logits (C is small: classes)
template <bool CE>
__global__ __launch_bounds__(256) void loss_kernel(const LossParams p) {
  __shared__ double red[2][256];
  const int64_t per = (p.n + p.blocks - 1) / p.blocks;
  const int64_t lo = static_cast<int64_t>(blockIdx.x) * per;
  const int64_t hi = lo + per < p.n ? lo + per : p.n;
  const float inv_n = 1.f / static_cast<float>(p.n);
  double l_acc = 0.0, h_acc = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const int64_t r = p.index ? p.index[i] : i;
    const float* x = p.x + r * p.x_stride;
    if constexpr (!CE) {
      const float xv = x[0], t = p.t_f[i];
      const float l = (1.f - t) * xv - log_sigmoid(xv);
      const float s = 1.f / (1.f + expf(-xv));
      l_acc += static_cast<double>(l);
      h_acc += ((s > 0.5f ? 1.f : 0.f) == t) ? 1.0 : 0.0;
      if (p.grad) p.grad[r * p.grad_stride] = (s - t) * inv_n;
    } else {
      const int64_t t = p.t_i[i];
      float m = x[0];
      int arg = 0;
      for (int c = 1; c < p.C; ++c)
        if (x[c] > m) { m = x[c]; arg = c; }
      float z = 0.f;
      for (int c = 0; c < p.C; ++c) z += expf(x[c] - m);
      const float lz = logf(z);
      const float xt = (t >= 0 && t < p.C) ? x[t] : 0.f;
      if ((t < 0 || t >= p.C) && p.label_flag) *p.label_flag = 1.f;
      l_acc += static_cast<double>(-(xt - m - lz));
      h_acc += arg == t ? 1.0 : 0.0;
      if (p.grad) {
        float* g = p.grad + r * p.grad_stride;
        for (int c = 0; c < p.C; ++c) g[c] = (expf(x[c] - m - lz) - (c == t ? 1.f : 0.f)) * inv_n;
      }
    }
  }
  red[0][threadIdx.x] = l_acc;
  red[1][threadIdx.x] = h_acc;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (static_cast<int>(threadIdx.x) < st) {
      red[0][threadIdx.x] += red[0][threadIdx.x + st];
      red[1][threadIdx.x] += red[1][threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (p.blocks == 1) {
      finish(p, red[0][0], red[1][0]);
    } else {
      p.partial[2 * blockIdx.x] = red[0][0];
      p.partial[2 * blockIdx.x + 1] = red[1][0];
    }
  }
}

// Cross entropy over MANY classes (C > 8: ogbn-arxiv has 40): sixteen lanes per row, lane l holds classes l, l + 16, ... in
// registers (C <= 128), so a row's logits are one coalesced run per load instead of one strided load per class and thread —
// 100k rows of 40 logits: 200 us with a thread per row (every load instruction touched 64 different lines), 30 us this way.
// Max / first arg-max / sum of exponentials meet through shuffles inside the 16-lane group; same formulas, same float32
// terms; the block's row terms are added in row order as before.
constexpr int kWideLanes = 16, kWideRegs = 8;   // C <= 128

__global__ __launch_bounds__(256) void loss_wide_ce_kernel(const LossParams p) {
  __shared__ double red[2][256];
  const int64_t per = (p.n + p.blocks - 1) / p.blocks;
  const int64_t lo = static_cast<int64_t>(blockIdx.x) * per;
  const int64_t hi = lo + per < p.n ? lo + per : p.n;
  const float inv_n = 1.f / static_cast<float>(p.n);
  const int sub = threadIdx.x % kWideLanes, slot = threadIdx.x / kWideLanes;      // 16 rows per pass of the workgroup
  const int nreg = (p.C + kWideLanes - 1) / kWideLanes;                           // registers the class count fills (uniform)
  double l_acc = 0.0, h_acc = 0.0;
  for (int64_t i0 = lo; i0 < hi; i0 += 256 / kWideLanes) {
    const int64_t i = i0 + slot;
    const bool live = i < hi;
    const int64_t r = live ? (p.index ? p.index[i] : i) : 0;
    const float* x = p.x + r * p.x_stride;
    float v[kWideRegs];
    float m = -INFINITY;
    int arg = 0x7fffffff;
#pragma unroll
    for (int t = 0; t < kWideRegs; ++t) {
      const int c = sub + t * kWideLanes;
      v[t] = (t < nreg && live && c < p.C) ? x[c] : -INFINITY;
      if (v[t] > m) { m = v[t]; arg = c; }              // (classes ascending within the lane: the first maximum)
    }
#pragma unroll
    for (int off = 1; off < kWideLanes; off <<= 1) {
      const float om = __shfl_xor(m, off);
      const int oa = __shfl_xor(arg, off);
      if (om > m || (om == m && oa < arg)) { m = om; arg = oa; }
    }
    float z = 0.f;
#pragma unroll
    for (int t = 0; t < kWideRegs; ++t)
      if (t < nreg) z += (sub + t * kWideLanes < p.C) ? expf(v[t] - m) : 0.f;
#pragma unroll
    for (int off = 1; off < kWideLanes; off <<= 1) z += __shfl_xor(z, off);
    const float lz = logf(z);
    const int64_t t_cls = live ? p.t_i[i] : 0;
    if (p.grad && live) {
      float* g = p.grad + r * p.grad_stride;
#pragma unroll
      for (int t = 0; t < kWideRegs; ++t) {
        const int c = sub + t * kWideLanes;
        if (t < nreg && c < p.C) g[c] = (expf(v[t] - m - lz) - (c == t_cls ? 1.f : 0.f)) * inv_n;
      }
    }
    // the target's logit sits in lane t_cls % 16, register t_cls / 16
    float xt = 0.f;
#pragma unroll
    for (int t = 0; t < kWideRegs; ++t) xt = (sub + t * kWideLanes == t_cls) ? v[t] : xt;
#pragma unroll
    for (int off = 1; off < kWideLanes; off <<= 1) xt += __shfl_xor(xt, off);
    if (live && sub == 0) {
      const float target = (t_cls >= 0 && t_cls < p.C) ? xt : 0.f;
      if ((t_cls < 0 || t_cls >= p.C) && p.label_flag) *p.label_flag = 1.f;
      l_acc += static_cast<double>(-(target - m - lz));
      h_acc += arg == t_cls ? 1.0 : 0.0;
    }
  }
  red[0][threadIdx.x] = l_acc;
  red[1][threadIdx.x] = h_acc;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (static_cast<int>(threadIdx.x) < st) {
      red[0][threadIdx.x] += red[0][threadIdx.x + st];
      red[1][threadIdx.x] += red[1][threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (p.blocks == 1) {
      finish(p, red[0][0], red[1][0]);
    } else {
      p.partial[2 * blockIdx.x] = red[0][0];
      p.partial[2 * blockIdx.x + 1] = red[1][0];
    }
  }
}

__global__ __launch_bounds__(256) void loss_final_kernel(const LossParams p) {
  __shared__ double red[2][256];
  double l = 0.0, h = 0.0;
  for (int b = threadIdx.x; b < p.blocks; b += 256) {
    l += p.partial[2 * b];
    h += p.partial[2 * b + 1];
  }
  red[0][threadIdx.x] = l;
  red[1][threadIdx.x] = h;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (static_cast<int>(threadIdx.x) < st) {
      red[0][threadIdx.x] += red[0][threadIdx.x + st];
      red[1][threadIdx.x] += red[1][threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) finish(p, red[0][0], red[1][0]);
}

// rows the mask leaves out have no part in the loss: their gradient rows are zero
__global__ __launch_bounds__(256) void loss_zero_grad_kernel(float* g, int64_t rows, int C, int64_t stride) {
  const int64_t total = rows * C;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < total; e += static_cast<int64_t>(gridDim.x) * 256)
    g[(e / C) * stride + e % C] = 0.f;
}

constexpr int64_t kOneBlockRows = 512;    // up to this many rows: one workgroup, one launch (1600 rows x 7 classes in one workgroup: 28 us)
constexpr int kMaxBlocks = 1024;

int blocks_for(int64_t n) {               // beyond that: 256 rows per workgroup until the chip is full four times over
  if (n <= kOneBlockRows) return 1;
  const int64_t b = (n + 255) / 256;
  return static_cast<int>(b > kMaxBlocks ? kMaxBlocks : b);
}

}  // namespace

extern "C" size_t gnan_loss_workspace_bytes(int64_t n) {
  const int b = blocks_for(n);
  return b > 1 ? static_cast<size_t>(b) * 2 * sizeof(double) : 0;
}

extern "C" int gnan_loss_step(const gnan_loss_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "loss: null args");
  GNAN_REQUIRE(a->kind == GNAN_LOSS_BCE_LOGITS || a->kind == GNAN_LOSS_CROSS_ENTROPY, "loss: unknown kind %d", a->kind);
  GNAN_REQUIRE(a->n >= 1 && a->n_rows >= 0 && a->C >= 1, "loss: bad sizes n=%lld rows=%lld C=%d (the mean of no rows is undefined)",
               static_cast<long long>(a->n), static_cast<long long>(a->n_rows), a->C);
  GNAN_REQUIRE(a->logits && a->labels && a->loss, "loss: null logits / labels / loss");
  GNAN_REQUIRE(a->stride >= a->C && (a->grad == nullptr || a->grad_stride >= a->C), "loss: row stride smaller than C");
  GNAN_REQUIRE(a->index != nullptr || a->n <= a->n_rows, "loss: more rows asked for than there are");
  if (a->kind == GNAN_LOSS_BCE_LOGITS && a->C != 1)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "loss: the binary loss takes one logit per row (got C=%d)", a->C);
  if (a->kind == GNAN_LOSS_CROSS_ENTROPY && a->C < 2)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "loss: cross entropy needs at least two classes");
  LossParams p;
  p.x = a->logits; p.x_stride = a->stride; p.C = a->C; p.index = a->index; p.n = a->n;
  p.t_f = a->kind == GNAN_LOSS_BCE_LOGITS ? static_cast<const float*>(a->labels) : nullptr;
  p.t_i = a->kind == GNAN_LOSS_CROSS_ENTROPY ? static_cast<const int64_t*>(a->labels) : nullptr;
  p.loss = a->loss; p.hits = a->hits; p.grad = a->grad; p.grad_stride = a->grad_stride;
  p.loss_sum = a->loss_sum; p.hits_sum = a->hits_sum; p.skip_sums = a->skip_sums; p.label_flag = a->label_flag;
  p.blocks = blocks_for(a->n);
  p.partial = static_cast<double*>(a->workspace);
  const size_t need = gnan_loss_workspace_bytes(a->n);
  if (need > 0 && (a->workspace == nullptr || a->workspace_bytes < need))
    return gnan::fail(GNAN_ERR_WORKSPACE, "loss: workspace %zu B < required %zu B", a->workspace_bytes, need);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a->grad && a->index && a->n_rows > 0) {
    int64_t zb = (a->n_rows * a->C + 255) / 256;
    zb = zb > 4096 ? 4096 : zb;
    hipLaunchKernelGGL(loss_zero_grad_kernel, dim3(static_cast<unsigned>(zb)), dim3(256), 0, st, a->grad, a->n_rows, a->C, a->grad_stride);
    if (int rc = gnan::check_launch("loss_zero_grad_kernel")) return rc;
  }
  if (a->kind == GNAN_LOSS_BCE_LOGITS)
    hipLaunchKernelGGL(loss_kernel<false>, dim3(static_cast<unsigned>(p.blocks)), dim3(256), 0, st, p);
  else if (a->C > 8 && a->C <= kWideLanes * kWideRegs)
    hipLaunchKernelGGL(loss_wide_ce_kernel, dim3(static_cast<unsigned>(p.blocks)), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(loss_kernel<true>, dim3(static_cast<unsigned>(p.blocks)), dim3(256), 0, st, p);
  if (int rc = gnan::check_launch("loss_kernel")) return rc;
  if (p.blocks > 1) {
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, p);
    return gnan::check_launch("loss_final_kernel");
  }
  return GNAN_OK;
}
