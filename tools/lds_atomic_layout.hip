// Does the LAYOUT of the moment kernel's bins matter?  (tools/: evidence, not product code)
// A wave of the moment kernel is 16 nodes x 4 feature quads; instruction f of a thread adds to feature 4q+f's bins (M0 and
// M1: two 64-bit LDS atomics).  x ~ U[0,1) lands in HOT of a feature's ~144 pieces.
//   layout 0: compact (what the kernel does): bin = off[feature] + piece, M1 at + tot
//   layout 1: bank groups: the four features one instruction touches own 8 bank pairs each:
//             slot = (piece / 8) * 32 + q * 8 + piece % 8, the four f of a thread in four planes
//   layout 2: as 1, M0 and M1 interleaved in the plane (M1 = slot + 16 bank pairs away: q groups of 4)
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_layout tools/lds_atomic_layout.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

constexpr int kPieces = 144, kPad = 256;

template <int LAYOUT, int HOT>
__global__ __launch_bounds__(512) void hammer(int iters, unsigned long long* out) {
  extern __shared__ unsigned long long bins[];
  constexpr int kCompact = 2 * 16 * kPieces, kPlanes = 2 * 4 * 4 * kPad;
  constexpr int kTotal = LAYOUT == 0 ? kCompact : kPlanes;
  for (int i = threadIdx.x; i < kTotal; i += 512) bins[i] = 0ull;
  __syncthreads();
  const int q = threadIdx.x & 3;
  uint32_t h = blockIdx.x * 512 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      h = mix(h + it);
      const int piece = 40 + static_cast<int>(h % HOT);
      int i0, i1;
      if (LAYOUT == 0) {
        i0 = (4 * q + f) * kPieces + piece;
        i1 = i0 + 16 * kPieces;
      } else if (LAYOUT == 1) {
        i0 = f * (4 * kPad) + (piece >> 3) * 32 + q * 8 + (piece & 7);
        i1 = i0 + 4 * 4 * kPad;
      } else {
        i0 = f * (8 * kPad) + (piece >> 2) * 32 + q * 4 + (piece & 3);
        i1 = i0 + 16;
      }
      atomicAdd(&bins[i0], 1ull);
      atomicAdd(&bins[i1], static_cast<unsigned long long>(h));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = bins[0] + bins[kTotal - 1];
}

template <int LAYOUT, int HOT>
int run() {
  unsigned long long* out;
  CK(hipMalloc(&out, 4096 * sizeof(unsigned long long)));
  const int blocks = 1024, iters = 512;
  const size_t lds = (LAYOUT == 0 ? 2 * 16 * kPieces : 2 * 4 * 4 * kPad) * sizeof(unsigned long long);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&hammer<LAYOUT, HOT>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((hammer<LAYOUT, HOT>), dim3(blocks), dim3(512), lds, 0, iters, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((hammer<LAYOUT, HOT>), dim3(blocks), dim3(512), lds, 0, iters, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("{\"layout\": %d, \"hot_pieces\": %d, \"lds_bytes\": %zu, \"ms\": %.3f, \"G_atomics_per_s\": %.1f}\n", LAYOUT, HOT, lds, ms,
         2.0 * 4 * double(blocks) * 512 * iters / ms / 1e6);
  CK(hipFree(out));
  return 0;
}

int main() {
  if (run<0, 4>()) return 1;
  if (run<1, 4>()) return 1;
  if (run<2, 4>()) return 1;
  if (run<0, 12>()) return 1;
  if (run<1, 12>()) return 1;
  if (run<2, 12>()) return 1;
  if (run<0, 64>()) return 1;
  if (run<1, 64>()) return 1;
  if (run<2, 64>()) return 1;
  return 0;
}
