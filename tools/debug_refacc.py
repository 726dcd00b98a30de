import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gnan_amd
from gnan_amd import synthetic as syn, functional
from gnan_amd.functional import feature_mlps, column_sums
from gnan_amd.aggregate import rho_aggregate
from oracle import gnan_oracle as O
import test_gpu_multirank as T
dev = "cuda"
N = T.N
src, dst, x = T._problem(1)
m = T._model(1, dev, "pwl").eval()
g = syn.hop1_csr(src.to(dev), dst.to(dev), N)
p64 = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
fx64 = O.feature_mlps(x.double(), p64)[:, :, 0]
rp, cl, cd = g.rowptr.cpu().long().numpy(), g.col.cpu().numpy(), g.code.cpu().numpy(); cnt = g.cnt.cpu().long().numpy()
lut64 = O.rho_lut(p64, 3, torch.float64)
truth = O.spmm_csr_vectorised(rp, cl, cd, fx64, lut64, cnt).sum(1, keepdim=True)
scale = float(truth.abs().max())
def err(y): return float((y.double().cpu() - truth).abs().max()) / scale
with torch.no_grad():
    st = m._stacked("fs", m.fs)
    lut = m._lut_global(g)
    S, total = feature_mlps(x.to(dev), st, False, return_total=True)
    print("operand err", float((S.double().cpu() - fx64).abs().max()), "total rel err", float(((total.double().cpu() - fx64.sum(0)).abs() / fx64.sum(0).abs()).max()),
          "lut err", float((lut.double().cpu() - lut64).abs().max()))
    print("kernel, table operand, fused total     ", err(rho_aggregate(g, S, lut, True, s_total=total, reduce_channels=1)))
    Sx = fx64.float().to(dev)
    tot_x = fx64.sum(0).float().to(dev)
    print("kernel, exact operand, exact total     ", err(rho_aggregate(g, Sx, lut64.float().to(dev), True, s_total=tot_x, reduce_channels=1)))
    print("kernel, table operand, exact total     ", err(rho_aggregate(g, S, lut, True, s_total=S.double().sum(0).float(), reduce_channels=1)))
    print("kernel, exact operand, no fused readout", err(rho_aggregate(g, Sx, lut64.float().to(dev), True, s_total=tot_x).double().sum(1, keepdim=True)))
    y_cols = rho_aggregate(g, Sx, lut64.float().to(dev), True, s_total=tot_x)          # [N, F]
    cols64 = O.spmm_csr_vectorised(rp, cl, cd, fx64, lut64, cnt)
    print("per-column error / column scale", float((y_cols.double().cpu() - cols64).abs().max() / cols64.abs().max()), "column scale", float(cols64.abs().max()), "out scale", scale)
    # oracle fp32 in its own formulation
    sd32 = {k: v.detach().cpu().float() for k, v in m.state_dict().items()}
    wt32 = O.weight_table(O.rho_lut(sd32, 3), cnt).expand(N, -1, -1)
    ref32 = O.spmm_csr(rp, cl, cd, fx64.float(), wt32).sum(1, keepdim=True)
    print("oracle spmm_csr float32 on exact operand", err(ref32))
