// The row pass of the one-column backward on the row-parallel route (gnan_spmm_pack_z, include/gnan_hip.h): with the forward's
// per-code shell sums T[i, d] kept (gnan_spmm_args.shell_out) the gradient of the weight table is a sum over ROWS,
//   dlut[d] = sum_i dY_i / cnt(i, d) * T[i, d],
// and the operand gradient a gather of ONE pre-weighted number per (row, hop code) over the transposed adjacency
// (Z[i, d] = (lut[d] / cnt(i, d) - lut[rest] / cnt(i, rest)) dY_i, gnan_spmm_fwd with s_by_code) — autograd through GNAN.py:64-70
// w.r.t. f's output and rho's table.  It replaces gnan_spmm_pack_bwd_rows + gnan_spmm_bwd_narrow, which walk the transposed pairs
// with two numbers per row to get both gradients out of one pass (arxiv shape: 52 us -> 28).
#include "common.hpp"

namespace {

struct PackZParams {
  int64_t n;
  const float* dY;
  int64_t dy_stride;
  const int32_t* cnt;
  int64_t cnt_stride;
  int D, with_rest;
  const float* lut;
  const float* shell;
  float* Z;
  double* partial;      // [blocks, 4]: q | g_0 | g_1 | g_2 | (the rest term's sum rides in slot D - 1 <= 3)
};

__global__ __launch_bounds__(256) void pack_z_kernel(const PackZParams p) {
  const int D = p.D, rest = D - 1;
  float l[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) l[d] = d < D ? p.lut[d] : 0.f;
  const float lr = p.with_rest ? l[rest] : 0.f;
  const bool has_cnt = p.cnt != nullptr;
  const int32_t* cnt_base = has_cnt ? p.cnt : reinterpret_cast<const int32_t*>(p.lut);       // (no load behind a condition)
  const int64_t cnt_step = has_cnt ? p.cnt_stride : 0;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  double q = 0.0, g[3] = {0.0, 0.0, 0.0}, gr = 0.0;
  if (i < p.n) {
    const float dy = p.dY[i * p.dy_stride];
    int c[4];
    float t[3];
#pragma unroll
    for (int d = 0; d < 4; ++d) c[d] = cnt_base[i * cnt_step + (has_cnt && d < D ? d : 0)];
#pragma unroll
    for (int d = 0; d < 3; ++d) t[d] = p.shell[i * rest + (d < rest ? d : 0)];
    float a[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) a[d] = dy / (has_cnt ? static_cast<float>(c[d] > 1 ? c[d] : 1) : 1.f);      // IEEE division, as torch.div
    const float ar = p.with_rest ? a[rest] : 0.f;
    double tsum = 0.0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (d < rest) {
        p.Z[i * D + d] = fmaf(l[d], a[d], -(lr * ar));
        g[d] = static_cast<double>(a[d]) * static_cast<double>(t[d]);
        tsum += static_cast<double>(t[d]);
      }
    }
    p.Z[i * D + rest] = 0.f;
    q = static_cast<double>(ar);
    gr = static_cast<double>(ar) * tsum;
  }
  __shared__ double red[256][5];
  red[threadIdx.x][0] = q; red[threadIdx.x][1] = g[0]; red[threadIdx.x][2] = g[1]; red[threadIdx.x][3] = g[2]; red[threadIdx.x][4] = gr;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (static_cast<int>(threadIdx.x) < off) {
#pragma unroll
      for (int k = 0; k < 5; ++k) red[threadIdx.x][k] += red[threadIdx.x + off][k];
    }
    __syncthreads();
  }
  if (threadIdx.x < 5) p.partial[static_cast<int64_t>(blockIdx.x) * 5 + threadIdx.x] = red[0][threadIdx.x];
}

__global__ __launch_bounds__(1024) void pack_z_final_kernel(const double* __restrict__ partial, int64_t n_blocks, int D, int with_rest,
                                                            const float* s_total, float* __restrict__ q_out, float* __restrict__ dlut) {
  __shared__ double red[1024][5];
  double g[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int64_t b = threadIdx.x; b < n_blocks; b += 1024) {
#pragma unroll
    for (int k = 0; k < 5; ++k) g[k] += partial[b * 5 + k];
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) red[threadIdx.x][k] = g[k];
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (static_cast<int>(threadIdx.x) < off) {
#pragma unroll
      for (int k = 0; k < 5; ++k) red[threadIdx.x][k] += red[threadIdx.x + off][k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    q_out[0] = static_cast<float>(red[0][0]);
    for (int d = 0; d < D - 1; ++d) dlut[d] = static_cast<float>(red[0][1 + d]);
    dlut[D - 1] = with_rest ? static_cast<float>(-red[0][4] + (s_total ? static_cast<double>(s_total[0]) * static_cast<double>(q_out[0]) : 0.0))
                            : 0.f;
  }
}

}  // namespace

extern "C" size_t gnan_spmm_pack_z_workspace_bytes(int64_t n) {
  if (n <= 0) return 64;
  return static_cast<size_t>((n + 255) / 256) * 5 * sizeof(double);
}

extern "C" int gnan_spmm_pack_z(const gnan_pack_z_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr && a->n >= 0 && a->D >= 2 && a->D <= 4, "gnan_spmm_pack_z: bad sizes");
  GNAN_REQUIRE(a->q && a->dlut && a->lut && (a->n == 0 || (a->dY && a->shell && a->Z)), "gnan_spmm_pack_z: null pointer");
  GNAN_REQUIRE(a->dy_stride >= 1 && (a->cnt == nullptr || a->cnt_stride >= a->D), "gnan_spmm_pack_z: row stride smaller than the width");
  GNAN_REQUIRE(a->workspace && a->workspace_bytes >= gnan_spmm_pack_z_workspace_bytes(a->n) &&
                   reinterpret_cast<uintptr_t>(a->workspace) % 16 == 0,
               "gnan_spmm_pack_z: workspace of %zu bytes, 16-byte aligned", gnan_spmm_pack_z_workspace_bytes(a->n));
  const int64_t blocks = a->n > 0 ? (a->n + 255) / 256 : 0;
  GNAN_REQUIRE(blocks < (int64_t{1} << 31), "gnan_spmm_pack_z: too many rows for one launch");
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* partial = static_cast<double*>(a->workspace);
  if (blocks > 0) {
    PackZParams p;
    p.n = a->n; p.dY = a->dY; p.dy_stride = a->dy_stride; p.cnt = a->cnt; p.cnt_stride = a->cnt_stride; p.D = a->D;
    p.with_rest = a->with_rest; p.lut = a->lut; p.shell = a->shell; p.Z = a->Z; p.partial = partial;
    hipLaunchKernelGGL(pack_z_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, p);
    if (int rc = gnan::check_launch("pack_z_kernel")) return rc;
  }
  hipLaunchKernelGGL(pack_z_final_kernel, dim3(1), dim3(1024), 0, st, partial, blocks, a->D, a->with_rest, a->s_total, a->q, a->dlut);
  return gnan::check_launch("pack_z_final_kernel");
}
