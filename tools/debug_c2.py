import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnan_amd
from gnan_amd import HopGraph, functional, small_graph
from gnan_amd import synthetic as syn
from gnan_amd.models import TensorGNAN
from oracle import gnan_oracle as O
dev = "cuda"
torch.manual_seed(0)
model = TensorGNAN(15, 1, 3, hidden_channels=64, is_graph_task=True, readout_n_layers=0, device="cuda")
with torch.no_grad():
    for _, p in model.named_parameters():
        if p.dim() == 2:
            torch.nn.init.xavier_normal_(p, gain=1.0)
        else:
            p.normal_(0.0, 0.5)
model = model.to(dev).eval()
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
p64 = {k: v.double() for k, v in sd.items()}
print(list(sd.keys())[:8])
class Bag:
    def __init__(self, **kw): self.__dict__.update(kw)
for i, (ei, x, y) in enumerate(syn.mutagenicity_shaped_graphs(64, seed=0)):
    n = x.shape[0]
    hg = HopGraph.from_edge_index(torch.as_tensor(ei).to(dev), n)
    code = hg.code.long()
    nd = torch.where(code == 255, torch.zeros((), device=dev), 1.0 / (1.0 + code.float()))
    norm = torch.gather(hg.cnt.float(), 1, code.clamp_max(hg.n_codes - 1))
    nd_o, norm_o = O.pre_process_dense(ei, n)
    same_in = bool(torch.equal(nd.cpu(), nd_o)) and bool(torch.equal(norm.cpu(), norm_o))
    d = Bag(x=x.to(dev), edge_index=None, node_distances=nd, normalization_matrix=norm)
    with torch.no_grad():
        small_graph.SMALL_GRAPH_FORWARD = True
        a = float(model.forward(d))
        small_graph.SMALL_GRAPH_FORWARD = False
        b = float(model.forward(d))
        small_graph.SMALL_GRAPH_FORWARD = True
    t = float(O.tensor_gnan_forward_models(x.double(), nd_o.double(), norm_o.double(), p64, True, True, 0))
    t32 = float(O.tensor_gnan_forward_models(x, nd_o, norm_o, sd, True, True, 0))
    flag = "" if abs(a - t) <= 1e-5 * max(abs(t), 1e-3) else "  <<<<"
    print(i, "n", n, "D", hg.n_codes, "inputs_equal", same_in, "small", a, "general", b, "truth", t, "ref32", t32, flag)
