// What bounds LDS atomics on gfx950: bank conflicts or the atomic unit?  (tools/: evidence for the moment kernel, not product code)
// Patterns: 0 = random bins (as lds_atomic_rate.hip), 1 = conflict-free (bin = lane), 2 = one bin for the whole wave,
// 3 = random but each lane inside its own bank (bin = lane + 64 * r), 4 = two 64-bit atomics to [piece] and [tot + piece]
// (the moment kernel's pair), 5 = the pair as one 32-bit atomic each, 6 = pair interleaved (M0, M1 adjacent: 16 bytes).
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate2 tools/lds_atomic_rate2.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

constexpr int kBins = 4608;

template <typename T, int PATTERN, int BS>
__global__ __launch_bounds__(BS) void hammer(int iters, T* out) {
  __shared__ T bins[2 * kBins];
  for (int i = threadIdx.x; i < 2 * kBins; i += BS) bins[i] = T(0);
  __syncthreads();
  uint32_t h = blockIdx.x * BS + threadIdx.x;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    h = mix(h + it);
    int bin;
    if (PATTERN == 0 || PATTERN >= 4) bin = (h % 32) * 144 + ((h >> 8) % 16);
    else if (PATTERN == 1) bin = lane + 64 * (it & 31);
    else if (PATTERN == 2) bin = (it & 1023);
    else bin = lane + 64 * ((h >> 8) % 64);
    if (PATTERN < 4) {
      atomicAdd(&bins[bin], T(1));
    } else if (PATTERN == 4 || PATTERN == 5) {
      atomicAdd(&bins[bin], T(1));
      atomicAdd(&bins[kBins + bin], T(h));
    } else {
      atomicAdd(&bins[2 * bin], T(1));
      atomicAdd(&bins[2 * bin + 1], T(h));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = bins[0] + bins[kBins];
}

// the pair as split 64-bit sums: the low words by returning 32-bit atomics, the high words only when they change
// (sign extension + carry != 0): terms of SHIFT significant bits
template <int SHIFT, int BS>
__global__ __launch_bounds__(BS) void hammer_split(int iters, unsigned* out) {
  __shared__ unsigned lo[2 * kBins];
  __shared__ int hi[2 * kBins];
  for (int i = threadIdx.x; i < 2 * kBins; i += BS) { lo[i] = 0u; hi[i] = 0; }
  __syncthreads();
  uint32_t h = blockIdx.x * BS + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    h = mix(h + it);
    const int bin = (h % 32) * 144 + ((h >> 8) % 16);
    const long long t0 = static_cast<long long>(static_cast<int>(h)) >> (32 - SHIFT);
    const long long t1 = static_cast<long long>(static_cast<int>(h * 2654435761u)) >> (32 - SHIFT);
    const unsigned l0 = static_cast<unsigned>(t0), l1 = static_cast<unsigned>(t1);
    const unsigned o0 = atomicAdd(&lo[bin], l0);
    const unsigned o1 = atomicAdd(&lo[kBins + bin], l1);
    const int d0 = static_cast<int>(t0 >> 32) + (o0 + l0 < l0 ? 1 : 0);
    const int d1 = static_cast<int>(t1 >> 32) + (o1 + l1 < l1 ? 1 : 0);
    if (d0 != 0) atomicAdd(&hi[bin], d0);
    if (d1 != 0) atomicAdd(&hi[kBins + bin], d1);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = lo[0] + lo[kBins] + hi[0] + hi[kBins];
}

template <int SHIFT, int BS>
int run_split() {
  unsigned* out;
  CK(hipMalloc(&out, 4096 * sizeof(unsigned)));
  const int blocks = 1024 * 512 / BS, iters = 2048;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((hammer_split<SHIFT, BS>), dim3(blocks), dim3(BS), 0, 0, iters, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((hammer_split<SHIFT, BS>), dim3(blocks), dim3(BS), 0, 0, iters, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("{\"type\": \"split\", \"term_bits\": %d, \"block\": %d, \"ms\": %.3f, \"G_pairs_x2_per_s\": %.1f}\n", SHIFT, BS, ms,
         2.0 * double(blocks) * BS * iters / ms / 1e6);
  CK(hipFree(out));
  return 0;
}

template <typename T, int PATTERN, int BS>
int run(const char* name) {
  T* out;
  CK(hipMalloc(&out, 4096 * sizeof(T)));
  const int blocks = 1024 * 512 / BS, iters = 2048;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((hammer<T, PATTERN, BS>), dim3(blocks), dim3(BS), 0, 0, iters, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((hammer<T, PATTERN, BS>), dim3(blocks), dim3(BS), 0, 0, iters, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double per = PATTERN >= 4 ? 2.0 : 1.0;
  printf("{\"type\": \"%s\", \"pattern\": %d, \"block\": %d, \"ms\": %.3f, \"G_atomics_per_s\": %.1f}\n", name, PATTERN, BS, ms,
         per * double(blocks) * BS * iters / ms / 1e6);
  CK(hipFree(out));
  return 0;
}

int main() {
  typedef unsigned long long u64;
  if (run<u64, 0, 512>("u64")) return 1;
  if (run<u64, 1, 512>("u64")) return 1;
  if (run<u64, 2, 512>("u64")) return 1;
  if (run<u64, 3, 512>("u64")) return 1;
  if (run<u64, 4, 512>("u64")) return 1;
  if (run<u64, 6, 512>("u64")) return 1;
  if (run<unsigned, 0, 512>("u32")) return 1;
  if (run<unsigned, 1, 512>("u32")) return 1;
  if (run<unsigned, 2, 512>("u32")) return 1;
  if (run<unsigned, 3, 512>("u32")) return 1;
  if (run<unsigned, 5, 512>("u32")) return 1;
  if (run<u64, 0, 1024>("u64")) return 1;
  if (run<u64, 4, 1024>("u64")) return 1;
  if (run<u64, 0, 256>("u64")) return 1;
  if (run_split<20, 512>()) return 1;
  if (run_split<24, 512>()) return 1;
  if (run_split<28, 512>()) return 1;
  if (run_split<31, 512>()) return 1;
  if (run_split<24, 1024>()) return 1;
  return 0;
}
