// How fast are LDS atomics on gfx950?  (tools/: evidence for the per-piece moment kernel, not product code)
// Every lane issues `iters` atomic adds to pseudo-random bins of a 2304-bin LDS table (the footprint of 16 features x
// 144 pieces), as float, uint32 and uint64.   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate tools/lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <typename T, int SPREAD>
__global__ __launch_bounds__(512) void hammer(int iters, T* out) {
  __shared__ T bins[2304];
  for (int i = threadIdx.x; i < 2304; i += 512) bins[i] = T(0);
  __syncthreads();
  uint32_t h = blockIdx.x * 512 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    h = mix(h + it);
    // SPREAD bins per 144-bin feature are actually hit (x ~ U[0,1) lands in a few pieces only)
    const int bin = (h % 16) * 144 + ((h >> 8) % SPREAD);
    atomicAdd(&bins[bin], T(1));
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = bins[0];
}

template <typename T, int SPREAD>
int run(const char* name) {
  T* out;
  CK(hipMalloc(&out, 4096 * sizeof(T)));
  const int blocks = 1024, iters = 2048;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((hammer<T, SPREAD>), dim3(blocks), dim3(512), 0, 0, iters, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((hammer<T, SPREAD>), dim3(blocks), dim3(512), 0, 0, iters, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("{\"type\": \"%s\", \"bins_hit_per_feature\": %d, \"ms\": %.3f, \"G_atomics_per_s\": %.1f}\n", name, SPREAD, ms,
         double(blocks) * 512 * iters / ms / 1e6);
  return 0;
}

int main() {
  if (run<float, 144>("f32")) return 1;
  if (run<float, 16>("f32")) return 1;
  if (run<float, 4>("f32")) return 1;
  if (run<unsigned, 144>("u32")) return 1;
  if (run<unsigned, 16>("u32")) return 1;
  if (run<unsigned, 4>("u32")) return 1;
  if (run<unsigned long long, 144>("u64")) return 1;
  if (run<unsigned long long, 16>("u64")) return 1;
  if (run<unsigned long long, 4>("u64")) return 1;
  return 0;
}
