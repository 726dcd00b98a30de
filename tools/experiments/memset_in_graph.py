"""Does a hipMemsetAsync captured into a hipGraph run again on every replay?  (ROCm 7.2, torch 2.10, MI355X)"""
import ctypes
import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
dev = "cuda"


def trial(offset_bytes, nbytes, total=1024):
    buf = torch.full((total // 4,), 7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        rc = hip.hipMemsetAsync(buf.data_ptr() + offset_bytes, 0, nbytes, st)
        assert rc == 0, rc
        buf.add_(1)
    res = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        lo = offset_bytes // 4
        res.append((buf[lo:lo + max(1, nbytes // 4)].tolist()[:2], int(buf[-1])))
    return res


for off, nb in [(0, 1024), (0, 8), (16, 8), (16, 4), (64, 64), (256, 256), (0, 4096)]:
    print(f"memset offset {off:4d} bytes {nb:5d}:", trial(off, nb, total=max(1024, off + nb + 64)))

# what torch itself does
def torch_trial(fn, name):
    g = torch.cuda.CUDAGraph()
    out = None
    with torch.cuda.graph(g):
        out = fn()
    vals = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        vals.append(out.flatten()[:2].tolist())
    print(name, vals)

x = torch.rand(1 << 22, device=dev)
torch_trial(lambda: x.sum().reshape(1), "sum of 4M floats (multi-block reduce)   ")
torch_trial(lambda: (torch.zeros(3, device=dev) + 1), "zeros(3) + 1                            ")
torch_trial(lambda: torch.zeros(5, dtype=torch.int64, device=dev).add_(1), "zeros(5, int64).add_(1)                 ")
torch_trial(lambda: torch.nn.functional.binary_cross_entropy_with_logits(x, (x > 0.5).float()).reshape(1), "bce mean over 4M                        ")

# the same trials with the memset nodes swapped for kernel nodes (gnan_graph_replace_memsets)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gnan_amd
from gnan_amd import _lib


def fixed_trial(offset_bytes, nbytes, total=1024):
    buf = torch.full((total // 4,), 7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        assert hip.hipMemsetAsync(buf.data_ptr() + offset_bytes, 0, nbytes, st) == 0
        buf.add_(1)
    k = ctypes.c_int32(0)
    _lib.check(_lib.lib().gnan_graph_replace_memsets(g.raw_cuda_graph(), ctypes.byref(k)), "replace")
    g.instantiate()
    res = []
    for _ in range(4):
        g.replay()
        torch.cuda.synchronize()
        lo = offset_bytes // 4
        res.append((buf[lo:lo + max(1, nbytes // 4)].tolist()[:2], int(buf[-1])))
    return k.value, res


for off, nb in [(0, 1024), (16, 8), (16, 4), (64, 64), (3, 5), (0, 4096)]:
    print(f"FIXED memset offset {off:4d} bytes {nb:5d}:", fixed_trial(off, nb, total=max(1024, off + nb + 64)))
