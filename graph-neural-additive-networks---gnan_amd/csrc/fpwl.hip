// Shape functions by exact piecewise-linear table look-up (gfx950).
//
// f_k is a ReLU MLP of a scalar, i.e. exactly piecewise linear; gnan_amd/pwl.py tabulates it per forward
// (anchor / value / slope per piece, ~130 pieces for H = 64, L = 3).  This kernel evaluates
//     f_k(x) = val[i] + slope[i] * (x - anchor[i]),   i = #{ breakpoints of f_k <= x }
// for every (node, feature): it replaces the F x L addmm/relu launches of GNAN.py:57-62 by N*F binary
// searches in LDS.  HBM-bound (x in, fx out), no matrix work at all.
//
// Mapping: a workgroup owns a contiguous block of nodes and a group of <= FG consecutive features whose
// tables it copies to LDS once (amortised over the node block).  Thread = node: it loads the node's FG x
// values (one 64-B sector for FG = 16), runs the FG searches in lock-step (FG independent LDS reads per
// step hide the LDS latency; all lanes of a wave search the same table), and stores FG*C results.
#include "common.hpp"

namespace {

struct Params {
  const float* x;
  int64_t n, x_stride;
  int F, C;
  const int32_t* off;
  const float* anchor;
  const float* val;
  const float* slope;
  int step0;        // largest power of two <= max breakpoints per feature (0 if none)
  int n_groups;
  int nodes_per_block;
  int sum_features;
  int vec_x, vec_out;
  float* out;
  int64_t out_stride;
};

template <int FG>
__global__ __launch_bounds__(256) void fpwl_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_off[FG + 1];
  const int tid = threadIdx.x;
  const int C = p.C;
  const int64_t n_lo = static_cast<int64_t>(blockIdx.x) * p.nodes_per_block;
  const int64_t n_hi = n_lo + p.nodes_per_block < p.n ? n_lo + p.nodes_per_block : p.n;
  const int g_lo = p.sum_features ? 0 : blockIdx.y;
  const int g_hi = p.sum_features ? p.n_groups : g_lo + 1;

  for (int g = g_lo; g < g_hi; ++g) {
    const int k0 = g * FG;
    const int nf = p.F - k0 < FG ? p.F - k0 : FG;
    const int base = p.off[k0];
    const int tot = p.off[k0 + nf] - base;
    float* anchor_l = smem;
    float* val_l = smem + tot;
    float* slope_l = val_l + static_cast<int64_t>(tot) * C;
    __syncthreads();  // previous group's searches are done with the LDS tables
    for (int i = tid; i < tot; i += 256) anchor_l[i] = p.anchor[base + i];
    for (int i = tid; i < tot * C; i += 256) {
      val_l[i] = p.val[static_cast<int64_t>(base) * C + i];
      slope_l[i] = p.slope[static_cast<int64_t>(base) * C + i];
    }
    if (tid <= nf) s_off[tid] = p.off[k0 + tid] - base;
    __syncthreads();
    int po[FG], pn[FG];  // piece offset / number of breakpoints of each feature of the group (uniform)
#pragma unroll
    for (int f = 0; f < FG; ++f) {
      po[f] = f < nf ? s_off[f] : 0;
      pn[f] = f < nf ? s_off[f + 1] - s_off[f] - 1 : 0;
    }

    for (int64_t n = n_lo + tid; n < n_hi; n += 256) {
      float xv[FG];
      const float* xr = p.x + n * p.x_stride + k0;
      if (p.vec_x && nf == FG) {
#pragma unroll
        for (int q = 0; q < FG / 4; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(xr + 4 * q);
          xv[4 * q + 0] = t.x; xv[4 * q + 1] = t.y; xv[4 * q + 2] = t.z; xv[4 * q + 3] = t.w;
        }
      } else {
#pragma unroll
        for (int f = 0; f < FG; ++f) xv[f] = f < nf ? xr[f] : 0.f;
      }
      int idx[FG];
#pragma unroll
      for (int f = 0; f < FG; ++f) idx[f] = 0;
      for (int step = p.step0; step > 0; step >>= 1) {
#pragma unroll
        for (int f = 0; f < FG; ++f) {
          const int j = idx[f] + step;
          const int jj = j <= pn[f] ? j : 0;                     // out of range -> harmless in-range read
          const float a = anchor_l[po[f] + jj];
          idx[f] = (j <= pn[f] && a <= xv[f]) ? j : idx[f];
        }
      }
      float d[FG];
#pragma unroll
      for (int f = 0; f < FG; ++f) {
        idx[f] += po[f];
        d[f] = xv[f] - anchor_l[idx[f]];
      }
      if (p.sum_features) {
        float* o = p.out + n * p.out_stride;
        for (int c = 0; c < C; ++c) {
          float a = 0.f;
#pragma unroll
          for (int f = 0; f < FG; ++f)
            if (f < nf) a += fmaf(slope_l[idx[f] * C + c], d[f], val_l[idx[f] * C + c]);
          o[c] = g == 0 ? a : o[c] + a;   // groups run one after the other inside the workgroup: no race
        }
      } else {
        float* o = p.out + n * p.out_stride + static_cast<int64_t>(k0) * C;
        if (C == 1 && p.vec_out && nf == FG) {
#pragma unroll
          for (int q = 0; q < FG / 4; ++q) {
            float4 t;
            t.x = fmaf(slope_l[idx[4 * q + 0]], d[4 * q + 0], val_l[idx[4 * q + 0]]);
            t.y = fmaf(slope_l[idx[4 * q + 1]], d[4 * q + 1], val_l[idx[4 * q + 1]]);
            t.z = fmaf(slope_l[idx[4 * q + 2]], d[4 * q + 2], val_l[idx[4 * q + 2]]);
            t.w = fmaf(slope_l[idx[4 * q + 3]], d[4 * q + 3], val_l[idx[4 * q + 3]]);
            *reinterpret_cast<float4*>(o + 4 * q) = t;
          }
        } else {
#pragma unroll
          for (int f = 0; f < FG; ++f)
            if (f < nf)
              for (int c = 0; c < C; ++c) o[f * C + c] = fmaf(slope_l[idx[f] * C + c], d[f], val_l[idx[f] * C + c]);
        }
      }
    }
  }
}

template <int FG>
int launch(const Params& p, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fpwl_kernel<FG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "fpwl: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const int64_t bx = (p.n + p.nodes_per_block - 1) / p.nodes_per_block;
  if (bx > 0x7fffffffLL) return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl: too many nodes for one launch");
  const dim3 grid(static_cast<unsigned>(bx), p.sum_features ? 1u : static_cast<unsigned>(p.n_groups));
  hipLaunchKernelGGL(fpwl_kernel<FG>, grid, dim3(256), lds, st, p);
  return gnan::check_launch("fpwl_kernel");
}

}  // namespace

extern "C" int gnan_fpwl_fwd(const gnan_fpwl_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "fpwl: null args");
  GNAN_REQUIRE(a->n >= 0 && a->F >= 1 && a->C >= 1, "fpwl: bad sizes");
  if (a->n == 0) return GNAN_OK;
  GNAN_REQUIRE(a->x && a->out && a->off && a->anchor && a->val && a->slope, "fpwl: null pointer");
  GNAN_REQUIRE(a->x_stride >= a->F, "fpwl: x row stride smaller than F");
  GNAN_REQUIRE(a->max_pieces >= 1 && a->max_group_pieces >= 1, "fpwl: max_pieces / max_group_pieces must be >= 1");
  const int fg = a->features_per_group;
  GNAN_REQUIRE(fg == 1 || fg == 2 || fg == 4 || fg == 8 || fg == 16, "fpwl: features_per_group must be 1, 2, 4, 8 or 16");
  const int64_t ow = a->sum_features ? a->C : static_cast<int64_t>(a->F) * a->C;
  GNAN_REQUIRE(a->out_stride >= ow, "fpwl: out row stride smaller than the output width");
  const size_t lds = static_cast<size_t>(a->max_group_pieces) * (1 + 2 * static_cast<size_t>(a->C)) * sizeof(float);
  if (lds > 150 * 1024)
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl: %zu B of tables per feature group exceed LDS; use fewer features per group",
                      lds);
  Params p;
  p.x = a->x; p.n = a->n; p.x_stride = a->x_stride; p.F = a->F; p.C = a->C;
  p.off = a->off; p.anchor = a->anchor; p.val = a->val; p.slope = a->slope;
  int step0 = 0;
  while ((step0 ? step0 * 2 : 1) <= a->max_pieces - 1) step0 = step0 ? step0 * 2 : 1;
  p.step0 = step0;
  p.n_groups = (a->F + fg - 1) / fg;
  int64_t npb = (a->n / 1024 + 255) / 256 * 256;           // aim at ~1024 workgroups along the node axis
  p.nodes_per_block = static_cast<int>(npb < 256 ? 256 : (npb > 4096 ? 4096 : npb));
  p.sum_features = a->sum_features;
  auto aligned = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) % 16) == 0; };
  p.vec_x = fg % 4 == 0 && a->F % 4 == 0 && a->x_stride % 4 == 0 && aligned(a->x);
  p.vec_out = fg % 4 == 0 && a->F % 4 == 0 && a->out_stride % 4 == 0 && aligned(a->out);
  p.out = a->out; p.out_stride = a->out_stride;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (fg) {
    case 1: return launch<1>(p, lds, st);
    case 2: return launch<2>(p, lds, st);
    case 4: return launch<4>(p, lds, st);
    case 8: return launch<8>(p, lds, st);
    default: return launch<16>(p, lds, st);
  }
}
