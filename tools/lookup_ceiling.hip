// What does the ACCESS PATTERN of the shape-function look-up cost without any look-up?  (tools/: evidence for DESIGN.md
// section 4.2, not product code.)  x [N, 64] fp32 in, y [N, 64] fp32 out, y = 2 x, with the kernels' mappings:
//   flat      contiguous 16-byte elements, grid-stride                                   (what torch.mul does)
//   quad      workgroup = (node block, 16-feature group), thread = (node, 4 features), XCD-aware block map: the
//             look-up kernels' mapping — every wave touches 16 half lines (64 of 128 bytes) of x and of y
//   quad+img  ... plus an LDS image of IMG bytes loaded per workgroup before its node loop (the tables)
//   quad+pf   ... U = 2 nodes per round with the next round's rows requested first
//   row       thread = 16 bytes of a WHOLE row: a wave covers 4 complete 256-byte rows (all 64 features in one workgroup)
//   read      quad mapping, loads only (the feature-sum mode's traffic): one 4-byte store per node
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/lookup_ceiling tools/lookup_ceiling.hip && tools/_bin/lookup_ceiling
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int F = 64;

__global__ __launch_bounds__(512) void flat(const float4* __restrict__ x, float4* __restrict__ y, int64_t n4) {
  for (int64_t i = blockIdx.x * 512ll + threadIdx.x; i < n4; i += gridDim.x * 512ll) {
    float4 t = x[i];
    y[i] = make_float4(2 * t.x, 2 * t.y, 2 * t.z, 2 * t.w);
  }
}

template <int BS, int U, bool READ_ONLY, int TPN = 4, bool NT = false>
__global__ __launch_bounds__(BS) void quad(const float* __restrict__ x, float* __restrict__ y, int64_t n, int npb, int n_groups,
                                           const float* __restrict__ img, int img_floats) {
  extern __shared__ float smem[];
  constexpr int NODES = BS / TPN;
  const int tid = threadIdx.x, q = tid % TPN, nl = tid / TPN;
  const int64_t id = blockIdx.x;
  const int g = static_cast<int>((id >> 3) % n_groups);
  const int64_t nb = ((id >> 3) / n_groups) * 8 + (id & 7);
  const int64_t n_lo = nb * npb;
  if (n_lo >= n) return;
  const int64_t n_hi = n_lo + npb < n ? n_lo + npb : n;
  for (int i = tid; i < img_floats; i += BS) smem[i] = img[i];
  __syncthreads();
  const float bias = img_floats ? smem[(tid * 7) % img_floats] : 0.f;
  const float* xq = x + g * (4 * TPN) + q * 4;
  float* yq = y + g * (4 * TPN) + q * 4;
  auto put = [&](float* dst, float4 v) {
    if constexpr (NT) {
      __builtin_nontemporal_store(v.x, dst); __builtin_nontemporal_store(v.y, dst + 1);
      __builtin_nontemporal_store(v.z, dst + 2); __builtin_nontemporal_store(v.w, dst + 3);
    } else {
      *reinterpret_cast<float4*>(dst) = v;
    }
  };
  const float* xlast = xq + (n_hi - 1) * F;
  float acc = 0.f;
  if constexpr (U == 1) {
    for (int64_t m = n_lo + nl; m < n_hi; m += NODES) {
      const float4 t = *reinterpret_cast<const float4*>(xq + m * F);
      if constexpr (READ_ONLY) acc += t.x + t.y + t.z + t.w;
      else put(yq + m * F, make_float4(2 * t.x + bias, 2 * t.y, 2 * t.z, 2 * t.w));
    }
  } else {
    const int64_t step = static_cast<int64_t>(NODES) * F;
    const float* xp = xq + (n_lo + nl) * F;
    auto row = [&](const float* p) { return *reinterpret_cast<const float4*>(p <= xlast ? p : xlast); };
    float4 cur[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = row(xp + u * step);
    for (int64_t m = n_lo + nl; m < n_hi; m += U * NODES) {
#pragma unroll
      for (int u = 0; u < U; ++u) nxt[u] = row(xp + (U + u) * step);
      xp += U * step;
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (m + u * NODES < n_hi) {
          const float4 t = cur[u];
          if constexpr (READ_ONLY) acc += t.x + t.y + t.z + t.w;
          else put(yq + (m + u * NODES) * F, make_float4(2 * t.x + bias, 2 * t.y, 2 * t.z, 2 * t.w));
        }
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
  }
  if constexpr (READ_ONLY) if (acc == 12345.678f) y[blockIdx.x] = acc;
}

// what an element-wise framework kernel does: a workgroup owns one contiguous 16-KB chunk, 4 loads per thread in flight
__global__ __launch_bounds__(256) void flat4(const float4* __restrict__ x, float4* __restrict__ y, int64_t n4) {
  const int64_t base = blockIdx.x * 1024ll + threadIdx.x;
  float4 t[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = base + i * 256 < n4 ? x[base + i * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (base + i * 256 < n4) y[base + i * 256] = make_float4(2 * t[i].x, 2 * t[i].y, 2 * t[i].z, 2 * t[i].w);
}

template <int BS>
__global__ __launch_bounds__(BS) void whole_row(const float* __restrict__ x, float* __restrict__ y, int64_t n, int npb) {
  constexpr int NODES = BS / 16;
  const int tid = threadIdx.x, q = tid % 16, nl = tid / 16;
  const int64_t n_lo = static_cast<int64_t>(blockIdx.x) * npb;
  const int64_t n_hi = n_lo + npb < n ? n_lo + npb : n;
  for (int64_t m = n_lo + nl; m < n_hi; m += NODES) {
    const float4 t = *reinterpret_cast<const float4*>(x + m * F + q * 4);
    *reinterpret_cast<float4*>(y + m * F + q * 4) = make_float4(2 * t.x, 2 * t.y, 2 * t.z, 2 * t.w);
  }
}

template <typename Fn>
int timeit(const char* name, double bytes, Fn launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); launch();
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0;
  for (int rep = 0; rep < 10; ++rep) {
    CK(hipEventRecord(a));
    launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best; sum += ms;
  }
  CK(hipGetLastError());
  printf("{\"variant\": \"%s\", \"ms_min\": %.4f, \"ms_mean\": %.4f, \"TB_per_s\": %.3f}\n", name, best, sum / 10, bytes / best / 1e9);
  fflush(stdout);
  return 0;
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
  float *x, *y, *img;
  CK(hipMalloc(&x, n * F * 4)); CK(hipMalloc(&y, n * F * 4)); CK(hipMalloc(&img, 1 << 20));
  CK(hipMemset(x, 0, n * F * 4)); CK(hipMemset(img, 0, 1 << 20));
  const double rw = 2.0 * n * F * 4, ro = 1.0 * n * F * 4;
  if (timeit("flat", rw, [&] { hipLaunchKernelGGL(flat, dim3(256 * 8), dim3(512), 0, 0, (const float4*)x, (float4*)y, n * F / 4); })) return 1;
  if (timeit("flat4", rw, [&] { hipLaunchKernelGGL(flat4, dim3((n * F / 4 + 1023) / 1024), dim3(256), 0, 0, (const float4*)x, (float4*)y, n * F / 4); })) return 1;
  char name[128];
  auto sweep = [&](auto kernel, const char* tag, int tpn, int bs, double bytes) -> int {
    CK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int ng = 16 / tpn;
    for (int npb : {2048, 4096}) {
      const int64_t bx = ((n + npb - 1) / npb + 7) / 8 * 8;
      for (int img_kb : {0, 57, 82, 132}) {
        const int fl = img_kb * 256;
        snprintf(name, sizeof name, "%s_bs%d_npb%d_img%dK", tag, bs, npb, img_kb);
        if (timeit(name, bytes, [&] { hipLaunchKernelGGL(kernel, dim3(bx * ng), dim3(bs), fl * 4, 0, x, y, n, npb, ng, img, fl); })) return 1;
      }
    }
    return 0;
  };
  if (sweep(quad<512, 2, false, 4>, "q16_pf2", 4, 512, rw)) return 1;
  if (sweep(quad<512, 4, false, 4>, "q16_pf4", 4, 512, rw)) return 1;
  if (sweep(quad<512, 2, false, 4, true>, "q16_pf2_nt", 4, 512, rw)) return 1;
  if (sweep(quad<512, 2, false, 8>, "q32_pf2", 8, 512, rw)) return 1;
  if (sweep(quad<1024, 2, false, 8>, "q32_pf2", 8, 1024, rw)) return 1;
  if (sweep(quad<1024, 4, false, 8>, "q32_pf4", 8, 1024, rw)) return 1;
  if (sweep(quad<512, 2, false, 16>, "row_pf2", 16, 512, rw)) return 1;
  if (sweep(quad<1024, 2, false, 16>, "row_pf2", 16, 1024, rw)) return 1;
  if (sweep(quad<1024, 4, false, 16>, "row_pf4", 16, 1024, rw)) return 1;
  if (sweep(quad<1024, 2, false, 16, true>, "row_pf2_nt", 16, 1024, rw)) return 1;
  if (sweep(quad<512, 2, true, 4>, "read_q16_pf2", 4, 512, ro)) return 1;
  if (sweep(quad<1024, 2, true, 8>, "read_q32_pf2", 8, 1024, ro)) return 1;
  if (sweep(quad<1024, 2, true, 16>, "read_row_pf2", 16, 1024, ro)) return 1;
  return 0;
}
