// Pre-rho normalisation (GNAN.py:65-67): the weight of a pair is rho(node_distances / normalization_matrix), i.e. rho is
// evaluated at u_d / |shell_i(d)| — a different argument for every (row, hop code).  The reference does that with an MLP
// pass over all N^2 pairs; the shell identity leaves N*D evaluations, and rho : R -> R^C is a ReLU MLP of a
// scalar, i.e. exactly piecewise linear like the shape functions (pwl_build.hip tabulates it as a one-feature table).  This
// kernel turns the shell counts straight into the per-row weight table of the aggregation kernel:
//     lut[i, d, :] = val[p] + slope[p] * (x - anchor[p]),   x = u[d] / max(cnt[i, d], 1),   p = #{ anchors 1.. <= x }
// (the same arithmetic per look-up as fpwl.hip) — D look-ups per row instead of a [N*D, H] activation tensor per layer.
// HBM-bound: 4 B of cnt in, 4*C B of lut out per (row, code); the anchors sit in LDS, (val, slope) rows come from L1/L2.
#include "common.hpp"

namespace {

struct RhoLutParams {
  const int32_t* cnt;
  int64_t cnt_stride, n_rows;
  int D, C, T_cap;
  const int32_t* n_pieces;   // device: real piece count (tables may sit in a buffer of full capacity), or null
  const float* u;
  const float* anchor;
  const float* val;
  const float* slope;
  float* lut;
  float* arg;
};

__global__ __launch_bounds__(256) void rho_row_lut_kernel(const RhoLutParams p) {
  extern __shared__ float an[];                         // anchors of pieces 1 .. T-1 (piece 0 shares piece 1's)
  int T = p.T_cap;
  if (p.n_pieces) { const int t = *p.n_pieces; T = t < T ? t : T; }
  for (int j = threadIdx.x; j < T; j += 256) an[j] = p.anchor[j];
  __syncthreads();
  const int64_t total = p.n_rows * p.D;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < total; e += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t i = e / p.D;
    const int d = static_cast<int>(e - i * p.D);
    const int c = p.cnt[i * p.cnt_stride + d];
    const float x = p.u[d] / static_cast<float>(c > 1 ? c : 1);      // IEEE division, as torch.div (GNAN.py:66)
    int lo = 0, hi = T - 1;                             // piece = #{ j in 1..T-1 : an[j] <= x }
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (an[mid] <= x) lo = mid; else hi = mid - 1;
    }
    const float dx = x - an[lo];
    const float* v = p.val + static_cast<int64_t>(lo) * p.C;
    const float* s = p.slope + static_cast<int64_t>(lo) * p.C;
    float* o = p.lut + e * p.C;
    for (int ch = 0; ch < p.C; ++ch) o[ch] = fmaf(s[ch], dx, v[ch]);
    if (p.arg) p.arg[e] = x;
  }
}

}  // namespace

extern "C" int gnan_rho_row_lut(const gnan_rho_lut_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "rho_row_lut: null args");
  GNAN_REQUIRE(a->n_rows >= 0 && a->D >= 1 && a->D <= GNAN_MAX_CODES && a->C >= 1, "rho_row_lut: bad sizes");
  GNAN_REQUIRE(a->max_pieces >= 1 && a->max_pieces <= 16384, "rho_row_lut: max_pieces must be in [1, 16384] (got %d)", a->max_pieces);
  GNAN_REQUIRE(a->cnt_stride >= a->D, "rho_row_lut: cnt row stride smaller than D");
  if (a->n_rows == 0) return GNAN_OK;
  GNAN_REQUIRE(a->cnt && a->u && a->anchor && a->val && a->slope && a->lut, "rho_row_lut: null pointer");
  RhoLutParams p;
  p.cnt = a->cnt; p.cnt_stride = a->cnt_stride; p.n_rows = a->n_rows; p.D = a->D; p.C = a->C; p.T_cap = a->max_pieces;
  p.n_pieces = a->n_pieces; p.u = a->u; p.anchor = a->anchor; p.val = a->val; p.slope = a->slope;
  p.lut = a->lut; p.arg = a->arg;
  const int64_t total = a->n_rows * a->D;
  int64_t blocks = (total + 256 * 4 - 1) / (256 * 4);
  blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
  hipLaunchKernelGGL(rho_row_lut_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), static_cast<size_t>(a->max_pieces) * sizeof(float),
                     static_cast<hipStream_t>(stream), p);
  return gnan::check_launch("rho_row_lut_kernel");
}
