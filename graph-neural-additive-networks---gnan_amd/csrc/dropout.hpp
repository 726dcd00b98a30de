// Training-mode Dropout of the shape functions (GNAN.py:28,32: nn.Dropout behind every hidden ReLU) inside the kernels.
//
// The reference draws its masks from torch's generator, one Bernoulli per (node, hidden unit) in the order its Python loop
// over the features happens to issue them — a stream no fused kernel can (or needs to) reproduce; what is specified is the
// distribution.  The kernels draw theirs from a counter-based hash instead: keep(node, feature, hidden layer, unit) is a
// pure function of a 64-bit seed (taken from torch's generator once per forward, so torch.manual_seed() fixes it) and of
// the element's coordinates.  Forward and backward kernels recompute the same mask from the same seed — no mask tensor
// exists ([N, F, H] per layer: 1 GB per layer on the Cora shape) — and gnan_dropout_mask writes it out for the tests,
// which feed it to the oracle.  Kept activations are scaled by 1 / (1 - p), as torch does.
#pragma once
#include <cstdint>

namespace gnan {

__host__ __device__ __forceinline__ uint32_t mix32(uint32_t h) {      // murmur3's finaliser
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}

// one value per (node, feature): the per-unit draws hang off it with a single further mix
__host__ __device__ __forceinline__ uint32_t drop_base(uint64_t seed, int64_t node, int feature) {
  uint32_t a = mix32(static_cast<uint32_t>(seed) ^ static_cast<uint32_t>(node));
  return mix32(a ^ static_cast<uint32_t>(seed >> 32) ^ (static_cast<uint32_t>(static_cast<uint64_t>(node) >> 32) * 0x9e3779b1u) ^
               (static_cast<uint32_t>(feature) * 0x85ebca77u));
}

__host__ __device__ __forceinline__ bool drop_keep(uint32_t base, int layer, int unit, uint32_t thresh) {
  return mix32(base ^ (static_cast<uint32_t>(layer * 4096 + unit + 1) * 0xc2b2ae3du)) >= thresh;
}

// thresh = p * 2^32 (p in [0, 1)): P(keep) = 1 - p to 2^-32
__host__ __forceinline__ uint32_t drop_threshold(float p) {
  const double t = static_cast<double>(p) * 4294967296.0;
  return t <= 0.0 ? 0u : (t >= 4294967295.0 ? 4294967295u : static_cast<uint32_t>(t));
}

}  // namespace gnan
