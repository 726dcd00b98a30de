// The direct-index (bucket) tables of the look-up, shared by their builder (csrc/fpwl_index.hip: gnan_fpwl_index_build) and by
// the table build, which writes them in its compaction pass when the caller hands it the value ranges (csrc/pwl_build.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

namespace gnan_index {

// The bucket of x: ONE definition for the look-up and for the builder (which buckets the anchors with it).
__device__ __forceinline__ int bucket_of(float x, float ks, float ko, float top) {
  return static_cast<int>(__builtin_amdgcn_fmed3f(__fmaf_rn(x, ks, ko), 0.0f, top));   // NaN -> 0 (med3 = min3 on NaN)
}


// The bucket table of ONE feature f (256 threads, every thread calls it): anchor_f = the feature's anchors (entry 0: anchor of
// piece 0, breakpoints 1 .. pn), range = [F, 2] value ranges, table [F, B], key [F, 2], stats [F] or NULL.
template <int LOGB>
__device__ __forceinline__ void build_bucket_index(int f, const float* __restrict__ anchor_f, int pn, const float* __restrict__ range,
                                                   uint16_t* __restrict__ table, float* __restrict__ key, int32_t* __restrict__ stats) {
  constexpr int B = 1 << LOGB;
  __shared__ int cnt[B];
  __shared__ int part[256];
  const int tid = threadIdx.x;
  float lo = range[2 * f], hi = range[2 * f + 1];
  if (!(lo <= hi) || !(fabsf(lo) < 3.0e38f) || !(fabsf(hi) < 3.0e38f)) { lo = 0.f; hi = 0.f; }     // empty / non-finite hint
  // buckets 1 .. B-2 span [lo, hi]: key(lo) = 1.5, key(hi) = B - 1.5.  The span is at least 2^-10 of the larger magnitude:
  // |lo * ks| then stays below 2^20 * B / 1024, so ko's rounding moves a key by < 0.1 bucket — x = lo never slips into
  // the end bucket — and a constant column (the ones column of pre_process_datasets.py:127) still gets a bucket of its own
  const float span = fmaxf(hi - lo, fmaxf(fmaxf(fabsf(lo), fabsf(hi)) * 0.0009765625f, 1e-30f));
  const float ks = static_cast<float>(B - 3) / span;
  const float ko = __fmaf_rn(-lo, ks, 1.5f);
  const float top = static_cast<float>(B - 1);
  for (int i = tid; i < B; i += 256) cnt[i] = 0;
  __syncthreads();
  // breakpoints are entries 1 .. pn of the feature's anchors (entry 0: anchor of piece 0)
  for (int j = 1 + tid; j <= pn; j += 256) atomicAdd(&cnt[bucket_of(anchor_f[j], ks, ko, top)], 1);
  __syncthreads();
  // exclusive prefix over the B buckets: thread t owns buckets [t * B / 256, (t + 1) * B / 256)
  constexpr int PER = B / 256 > 0 ? B / 256 : 1;
  int s = 0;
  if (tid * PER < B)
    for (int i = 0; i < PER; ++i) s += cnt[tid * PER + i];
  part[tid] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int flagged = 0;
  if (tid * PER < B) {
    int run = part[tid] - s;
    for (int i = 0; i < PER; ++i) {
      const int k = tid * PER + i, c = cnt[k];
      const int code = c <= 1 ? 0 : (c <= 3 ? 1 : 3);
      table[static_cast<int64_t>(f) * B + k] = static_cast<uint16_t>(run * 4 | code << 14);
      if (code == 3 && k > 0 && k < B - 1) ++flagged;
      run += c;
    }
  }
  if (tid == 0) { key[2 * f] = ks; key[2 * f + 1] = ko; }
  if (stats) {                                  // searched buckets inside the hinted range: the fast path's health
    __syncthreads();
    part[tid] = flagged;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if (tid < st) part[tid] += part[tid + st];
      __syncthreads();
    }
    if (tid == 0) stats[f] = part[0];
  }
}

}  // namespace gnan_index
