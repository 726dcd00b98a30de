#!/usr/bin/env python3
"""One replayed training step of a 32-graph batch (batched.GraphedBatchStep) for a kernel trace."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gnan_amd  # noqa
from gnan_amd import batched
from batched_bench import graphs  # noqa

def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    data = graphs(bs * 8)
    torch.manual_seed(0)
    mod = batched.TensorGNAN(15, 8, 2, hidden_channels=16, device="cuda").to("cuda")
    with torch.no_grad():
        for _, p in mod.named_parameters():
            p.copy_(torch.randn(p.shape) * 0.5)
    opt = torch.optim.Adam(mod.parameters(), lr=1e-3)
    loss_fn = torch.nn.CrossEntropyLoss()
    batches = [batched.collate(data[i:i + bs]) for i in range(0, len(data), bs)]
    x0, b0, y0, _ = batches[0]
    gs = batched.GraphedBatchStep(mod, opt, lambda out, lab: loss_fn(out, lab), x0, b0, y0)
    for _ in range(3):
        for x, blocks, y, bv in batches:
            assert gs.run(x, blocks, y) is not None
    torch.cuda.synchronize()
    print("kernels per step", gs.kernel_nodes, "n_codes cap", gs.n_codes, "batch n_codes", [b[1].n_codes for b in batches])
main()
