// What can an MI355X sustain for RANDOM row gathers?  (tools/: evidence for DESIGN.md section 4.1, not product code)
// Every group of LPR lanes reads one pseudo-random row of ROWB = 16*LPR bytes per step, DEPTH independent loads in
// flight per lane, rows drawn uniformly from a table of `n_rows` rows.  No index array is read: the row ids come
// from a hash, so this is the gather traffic alone — an upper bound for the aggregation kernel's gather leg.
//   hipcc --offload-arch=gfx950 -O3 -o gather_ceiling tools/gather_ceiling.hip && ./gather_ceiling
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int LPR, int DEPTH>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ tab, uint32_t n_rows, int steps, float* out) {
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
  const uint32_t grp = gid / LPR, sub = gid % LPR;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < steps; s += DEPTH) {
    float4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const uint32_t r = mix(grp * 2654435761u + (s + d) * 40503u) % n_rows;
      v[d] = tab[static_cast<uint64_t>(r) * LPR + sub];
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { acc.x += v[d].x; acc.y += v[d].y; acc.z += v[d].z; acc.w += v[d].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[gid] = acc.x;   // never true: keeps the loads alive
}

template <int LPR, int DEPTH>
int run(const float4* tab, uint64_t table_bytes, float* out, int wgs) {
  const uint32_t n_rows = static_cast<uint32_t>(table_bytes / (16 * LPR));
  const int steps = 256;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((gather<LPR, DEPTH>), dim3(wgs), dim3(256), 0, 0, tab, n_rows, steps, out);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather<LPR, DEPTH>), dim3(wgs), dim3(256), 0, 0, tab, n_rows, steps, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best;
  }
  const double rows = static_cast<double>(wgs) * 256 / LPR * steps;
  printf("{\"row_bytes\": %d, \"depth\": %d, \"table_GB\": %.2f, \"ms\": %.3f, \"Grows_per_s\": %.2f, \"TB_per_s\": %.3f}\n",
         16 * LPR, DEPTH, table_bytes / 1e9, best, rows / best / 1e6, rows * 16 * LPR / best / 1e9);
  fflush(stdout);
  return 0;
}

int main(int argc, char** argv) {
  const uint64_t cap = 16ull << 30;
  float4* tab; float* out;
  CK(hipMalloc(&tab, cap));
  CK(hipMemset(tab, 0, cap));
  CK(hipMalloc(&out, 1 << 28));
  const int wgs = 256 * 64;
  // ./gather_ceiling [table sizes in MiB ...]: small tables show what L2- (4 MiB per XCD) and MALL-resident (256 MiB)
  // rows reach — does the request-rate wall belong to L2 misses or to L2 requests?
  std::vector<uint64_t> sizes;
  for (int i = 1; i < argc; ++i) sizes.push_back(static_cast<uint64_t>(atoll(argv[i])) << 20);
  if (sizes.empty()) sizes = {1280ull << 20, 2560ull << 20, 14ull << 30};
  for (uint64_t bytes : sizes) {
    if (run<4, 4>(tab, bytes, out, wgs)) return 1;      // 64-B rows
    if (run<8, 4>(tab, bytes, out, wgs)) return 1;      // 128-B rows (64 bf16)
    if (run<8, 8>(tab, bytes, out, wgs)) return 1;
    if (run<16, 4>(tab, bytes, out, wgs)) return 1;     // 256-B rows (64 fp32)
    if (run<16, 8>(tab, bytes, out, wgs)) return 1;
    if (run<32, 4>(tab, bytes, out, wgs)) return 1;     // 512-B rows
  }
  // streaming reference: consecutive rows
  return 0;
}
