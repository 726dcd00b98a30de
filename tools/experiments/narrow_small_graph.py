import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import microbench as mb
import torch
from gnan_amd import functional
from gnan_amd.functional import spmm_launch, column_sums
N, E = 169_343, 1_166_243
gen = torch.Generator(device="cuda").manual_seed(0)
src = torch.randint(0, N, (E,), generator=gen, device="cuda")
dst = (torch.rand(E, generator=gen, device="cuda") ** 3 * N).long().clamp_(0, N - 1)
g = mb.syn.hop1_csr(src, dst, N)
lut = torch.tensor([[0.7], [-0.3], [0.2]], device="cuda")
for W in (1, 2, 4):
    S = torch.rand(N, W, device="cuda"); tot = column_sums(S)
    for walk in (False, True):
        functional.NARROW_SORTED_WALK = walk
        for _ in range(5): y = spmm_launch(g, S, lut, True, True, s_total=tot)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): y = spmm_launch(g, S, lut, True, True, s_total=tot)
        b.record(); torch.cuda.synchronize()
        print(W, "sorted" if walk else "natural", round(a.elapsed_time(b) / 50 * 1000, 1), "us (incl. launch overhead)")
