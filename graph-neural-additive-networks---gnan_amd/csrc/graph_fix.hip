// hipGraph surgery: replace every memset node of a captured graph by a kernel node doing the same fill (gfx950, ROCm 7.2).
//
// Why: on this stack a hipMemsetAsync captured into a graph is only replayed correctly ONCE — from the second launch of
// the instantiated graph on, the node fills with garbage (tools/experiments/memset_in_graph.py: value 0 becomes
// 0x10101010 / pointer-like patterns).  The framework's own multi-block reductions zero their semaphores with such a
// memset, so a captured training step (gnan_amd/graphed.py) computed wrong sums from its second replay on.  Kernel
// nodes replay correctly; this pass keeps the graph's topology (same dependencies, same dependents) and swaps the node.
#include "common.hpp"

#include <vector>

namespace {

__global__ __launch_bounds__(256) void graph_fill_kernel(unsigned char* dst, unsigned value, unsigned esize, size_t width,
                                                        size_t height, size_t pitch) {
  const size_t total = width * height;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    unsigned char* at = dst + (i / width) * pitch + (i % width) * esize;
    if (esize == 4) *reinterpret_cast<unsigned*>(at) = value;
    else if (esize == 2) *reinterpret_cast<unsigned short*>(at) = static_cast<unsigned short>(value);
    else *at = static_cast<unsigned char>(value);
  }
}

#define HIP_TRY(call)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "hipGraph surgery: %s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

}  // namespace

extern "C" int gnan_graph_replace_memsets(void* graph_handle, int32_t* n_replaced) {
  GNAN_REQUIRE(graph_handle != nullptr, "graph_replace_memsets: null graph");
  hipGraph_t graph = static_cast<hipGraph_t>(graph_handle);
  size_t n = 0;
  HIP_TRY(hipGraphGetNodes(graph, nullptr, &n));
  std::vector<hipGraphNode_t> nodes(n);
  if (n) HIP_TRY(hipGraphGetNodes(graph, nodes.data(), &n));
  int replaced = 0;
  for (size_t k = 0; k < n; ++k) {
    hipGraphNodeType type;
    HIP_TRY(hipGraphNodeGetType(nodes[k], &type));
    if (type != hipGraphNodeTypeMemset) continue;
    hipMemsetParams mp;
    HIP_TRY(hipGraphMemsetNodeGetParams(nodes[k], &mp));
    if (mp.elementSize != 1 && mp.elementSize != 2 && mp.elementSize != 4)
      return gnan::fail(GNAN_ERR_UNSUPPORTED, "graph_replace_memsets: memset node with element size %u", mp.elementSize);
    size_t n_in = 0, n_out = 0;
    HIP_TRY(hipGraphNodeGetDependencies(nodes[k], nullptr, &n_in));
    std::vector<hipGraphNode_t> ins(n_in);
    if (n_in) HIP_TRY(hipGraphNodeGetDependencies(nodes[k], ins.data(), &n_in));
    HIP_TRY(hipGraphNodeGetDependentNodes(nodes[k], nullptr, &n_out));
    std::vector<hipGraphNode_t> outs(n_out);
    if (n_out) HIP_TRY(hipGraphNodeGetDependentNodes(nodes[k], outs.data(), &n_out));

    unsigned char* dst = static_cast<unsigned char*>(mp.dst);
    unsigned value = mp.value, esize = mp.elementSize;
    size_t width = mp.width, height = mp.height ? mp.height : 1, pitch = mp.pitch;
    void* args[] = {&dst, &value, &esize, &width, &height, &pitch};
    const size_t total = width * height;
    size_t blocks = (total + 256 * 4 - 1) / (256 * 4);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipKernelNodeParams kp = {};
    kp.func = reinterpret_cast<void*>(graph_fill_kernel);
    kp.gridDim = dim3(static_cast<unsigned>(blocks));
    kp.blockDim = dim3(256);
    kp.sharedMemBytes = 0;
    kp.kernelParams = args;
    kp.extra = nullptr;
    hipGraphNode_t fill;
    HIP_TRY(hipGraphAddKernelNode(&fill, graph, n_in ? ins.data() : nullptr, n_in, &kp));
    for (size_t o = 0; o < n_out; ++o) HIP_TRY(hipGraphAddDependencies(graph, &fill, &outs[o], 1));
    HIP_TRY(hipGraphDestroyNode(nodes[k]));
    ++replaced;
  }
  if (n_replaced) *n_replaced = replaced;
  return GNAN_OK;
}

extern "C" int gnan_graph_node_count(void* graph_handle, int32_t* n_kernels, int32_t* n_nodes) {
  GNAN_REQUIRE(graph_handle != nullptr, "graph_node_count: null graph");
  hipGraph_t graph = static_cast<hipGraph_t>(graph_handle);
  size_t n = 0;
  HIP_TRY(hipGraphGetNodes(graph, nullptr, &n));
  std::vector<hipGraphNode_t> nodes(n);
  if (n) HIP_TRY(hipGraphGetNodes(graph, nodes.data(), &n));
  int kernels = 0;
  for (size_t k = 0; k < n; ++k) {
    hipGraphNodeType type;
    HIP_TRY(hipGraphNodeGetType(nodes[k], &type));
    if (type == hipGraphNodeTypeKernel) ++kernels;
  }
  if (n_kernels) *n_kernels = kernels;
  if (n_nodes) *n_nodes = static_cast<int32_t>(n);
  return GNAN_OK;
}
