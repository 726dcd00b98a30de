"""EXPERIMENT: consecutive forwards on two streams (throughput mode).  The look-up is bound by LDS / VALU / streaming, the
aggregation by the L2-miss request rate: do they overlap when forward k+1's look-up runs under forward k's aggregation?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gnan_amd
from gnan_amd import synthetic as syn
from gnan_amd.functional import feature_mlps, rho_aggregate, stack_mlps
from gnan_amd.graph import hop_inputs
from gnan_amd.models import TensorGNAN
dev = torch.device("cuda")
N, E, F = 10_000_000, 100_000_000, 64
src, dst = syn.rmat_edges(24, N, E, seed=0, device=dev)
g = syn.hop1_csr(src, dst, N); del src, dst
x = syn.block_features(N, F, 0, N, seed=1, device=dev)
torch.manual_seed(0)
m = TensorGNAN(F, 1, 3, hidden_channels=64, device="cuda")
with torch.no_grad():
    for p in m.parameters():
        torch.nn.init.xavier_normal_(p, gain=1.0) if p.dim() == 2 else p.normal_(0.0, 0.5)
m = m.to(dev).eval()
with torch.no_grad():
    st = stack_mlps(m.fs); lut = m.rho(hop_inputs(g.n_codes, dev).view(-1, 1))
def fwd():
    with torch.no_grad():
        S, total = feature_mlps(x, st, False, return_total=True)
        return rho_aggregate(g, S, lut, True, s_total=total, reduce_channels=1)
for _ in range(3): fwd()
torch.cuda.synchronize()
def run(n_streams, K=20):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    for s in streams: s.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = []
    for k in range(K):
        with torch.cuda.stream(streams[k % n_streams]):
            outs.append(float(0) if False else fwd())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3, float(outs[-1].double().sum())
for ns in (1, 2, 3, 1, 2):
    print(ns, "streams:", run(ns))
