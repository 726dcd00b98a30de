"""CPU restatement (oracle) of the GNAN distance-weighted additive aggregation path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this
module: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / reported CPU baseline.

Every function restates one piece of the reference's PyTorch path and cites the
reference file:line (paths relative to the upstream repository root) whose
*behaviour* it follows.  The restatement is pure ``torch`` on CPU tensors and is
dtype-generic: feed it float32 tensors for the reference's own arithmetic or
float64 tensors for the "truth" used by the tolerance rule.

Pinning: the reference repository ships no tests or golden vectors for this
path, so the oracle is pinned against outputs of the *imported*
reference classes, generated in the build container by
``tests/golden/make_golden.py`` and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` replays them.

Parameter convention: a ``dict`` with the reference ``state_dict`` key names
(``fs.{k}.{i}.weight`` …, ``rho.{i}.weight`` …, ``readout_nam.fs.{k}.{i}.…``)
mapping to CPU tensors.
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
Params = Dict[str, Tensor]
Layer = Tuple[Tensor, Optional[Tensor]]


# --------------------------------------------------------------------------
# parameter plumbing
# --------------------------------------------------------------------------
def mlp_layers(params: Params, prefix: str) -> List[Layer]:
    """Collect the (weight, bias) pairs of one ``nn.Sequential`` MLP.

    Key layout follows the reference modules: f MLPs step by 3
    (Linear, ReLU, Dropout — GNAN.py:24-34), the rho MLP by 2
    (Linear, ReLU — GNAN.py:38-47).  Only the numeric order matters here.
    """
    pat = re.compile(re.escape(prefix) + r"\.(\d+)\.weight$")
    idx = sorted(int(m.group(1)) for k in params for m in [pat.match(k)] if m)
    if not idx:
        raise KeyError(f"no layers under prefix {prefix!r}")
    return [(params[f"{prefix}.{i}.weight"], params.get(f"{prefix}.{i}.bias")) for i in idx]


def n_features(params: Params, prefix: str = "fs") -> int:
    pat = re.compile(re.escape(prefix) + r"\.(\d+)\.")
    return 1 + max(int(m.group(1)) for k in params for m in [pat.match(k)] if m)


def mlp_apply(layers: Sequence[Layer], v: Tensor, keep: Optional[Sequence[Tensor]] = None, drop_p: float = 0.0) -> Tensor:
    """``Linear -> ReLU -> ... -> Linear`` on ``v [M, in]`` (eval mode: Dropout is identity).

    Follows the ``nn.Sequential`` built at GNAN.py:24-34 / GNAN.py:38-47
    (``n_layers == 1`` degenerates to one Linear, GNAN.py:25-26).  Training mode: ``keep[i] [M, H]`` is the Bernoulli
    keep-mask of the ``nn.Dropout`` behind hidden layer ``i`` (GNAN.py:28,32) — whatever drew it — and kept units are
    scaled by ``1 / (1 - drop_p)``, which is what ``nn.Dropout`` computes.
    """
    h = v
    last = len(layers) - 1
    for i, (w, b) in enumerate(layers):
        h = torch.nn.functional.linear(h, w.to(h.dtype), None if b is None else b.to(h.dtype))
        if i != last:
            h = torch.relu(h)
            if keep is not None:
                h = h * keep[i].to(h.dtype) / (1.0 - drop_p)
    return h


# --------------------------------------------------------------------------
# a2: per-feature shape functions
# --------------------------------------------------------------------------
def feature_mlps(x: Tensor, params: Params, prefix: str = "fs", keep: Optional[Tensor] = None, drop_p: float = 0.0) -> Tensor:
    """``fx[n, k, :] = f_k(x[n, k])`` — GNAN.py:57-62, models.py:360-365, models.py:292-297.
    ``keep [N, F, n_hidden, H]``: training-mode Dropout keep-masks (see :func:`mlp_apply`)."""
    F = x.shape[1]
    cols = []
    for k in range(F):
        masks = None if keep is None else [keep[:, k, l] for l in range(keep.shape[2])]
        cols.append(mlp_apply(mlp_layers(params, f"{prefix}.{k}"), x[:, k].reshape(-1, 1), masks, drop_p))
    return torch.stack(cols, dim=1)  # [N, F, C]


def nam_forward(x: Tensor, params: Params, prefix: str = "fs") -> Tensor:
    """``NAM.forward`` — models.py:291-300: sum over features of f_k(x[:, k]) -> [N, C]."""
    return feature_mlps(x, params, prefix).sum(dim=1)


# --------------------------------------------------------------------------
# a3-a5: dense tensorised forward, both upstream variants
# --------------------------------------------------------------------------
def _rho_dense(nd: Tensor, params: Params) -> Tensor:
    n = nd.shape[0]
    m = mlp_apply(mlp_layers(params, "rho"), nd.reshape(-1, 1))
    return m.reshape(n, nd.shape[1], -1)  # [N, N, Crho]


def tensor_gnan_forward_standalone(x: Tensor, nd: Tensor, norm: Optional[Tensor], params: Params,
                                   normalize_rho: bool = True, is_graph_task: bool = False) -> Tensor:
    """``TensorGNAN.forward`` of the stand-alone model file — GNAN.py:55-79.

    Normalisation divides the *distance* before rho (GNAN.py:65-66); rho always
    has ``out_channels`` outputs (GNAN.py:39,46).  Node task returns ``[N, C]``
    (GNAN.py:72-73,79), graph task ``[C, 1]`` (GNAN.py:75-79).
    """
    fx = feature_mlps(x, params)                     # [N, F, C]       GNAN.py:57-62
    u = nd / norm if normalize_rho else nd           #                 GNAN.py:65-66
    m = _rho_dense(u, params)                        # [N, N, C]       GNAN.py:67
    mf = torch.matmul(m.permute(2, 0, 1), fx.permute(2, 0, 1))  # [C, N, F]  GNAN.py:64,68,70
    if not is_graph_task:
        return mf.sum(dim=2).T                       # [N, C]          GNAN.py:72-73,79
    hidden = mf.sum(dim=1)                           # [C, F]          GNAN.py:76
    return hidden.sum(dim=1).reshape(1, -1).T        # [C, 1]          GNAN.py:78-79


def tensor_gnan_forward_models(x: Tensor, nd: Tensor, norm: Optional[Tensor], params: Params,
                               normalize_rho: bool = True, is_graph_task: bool = False,
                               readout_n_layers: int = 0) -> Tensor:
    """``TensorGNAN.forward`` of the copy ``main.py`` imports — models.py:358-384.

    rho is applied to the raw distance and the result is divided by the
    normalisation matrix (models.py:368-370); rho / f widths follow
    models.py:320-321 and are read off the weight shapes here; graph read-out is
    a plain sum (models.py:383) or the NAM read-out (models.py:380-381).
    """
    fx = feature_mlps(x, params)                     # [N, F, Cf]      models.py:360-365
    m = _rho_dense(nd, params)                       # [N, N, Crho]    models.py:368
    if normalize_rho:
        m = m / norm.unsqueeze(-1)                   #                 models.py:369-370
    mf = torch.matmul(m.permute(2, 0, 1), fx.permute(2, 0, 1))  # broadcast over c if Crho==1
    if not is_graph_task:
        return mf.sum(dim=2).T                       # [N, C]          models.py:375-376,384
    hidden = mf.sum(dim=1)                           # [Cf, F]         models.py:379
    if readout_n_layers > 0:
        ro = {k[len("readout_nam."):]: v for k, v in params.items() if k.startswith("readout_nam.")}
        return nam_forward(hidden, ro).T             # [C, 1]          models.py:380-381,384
    return hidden.sum(dim=1).reshape(1, -1).T        # [C, 1]          models.py:383-384


# --------------------------------------------------------------------------
# a7: per-node loop forward
# --------------------------------------------------------------------------
def gnan_forward(x: Tensor, nd: Tensor, norm: Tensor, params: Params, normalize_rho: bool = True,
                 node_ids: Optional[Sequence[int]] = None) -> Tensor:
    """``GNAN.forward`` — GNAN.py:146-172 and models.py:451-477 (identical bodies).

    Sum-first: ``f_sums = sum_k fx`` (GNAN.py:157), then per requested node
    ``sum_j (rho(nd[node, j]) / norm[node, j]) * f_sums[j]`` (GNAN.py:159-170).
    """
    f_sums = feature_mlps(x, params).sum(dim=1)      # [N, C]          GNAN.py:150-157
    ids = range(x.shape[0]) if node_ids is None else [int(i) for i in node_ids]
    rho = mlp_layers(params, "rho")
    rows = []
    for node in ids:
        r = mlp_apply(rho, nd[node].reshape(-1, 1))  # [N, Crho]       GNAN.py:162
        if normalize_rho:
            r = r / norm[node].reshape(-1, 1)        #                 GNAN.py:163-168
        rows.append((r * f_sums).sum(dim=0))         #                 GNAN.py:169
    return torch.stack(rows, dim=0)                  # [len(ids), C]   GNAN.py:170-172


# --------------------------------------------------------------------------
# f-2: batched block-diagonal variant
# --------------------------------------------------------------------------
def batched_tensor_gnan_forward(x: Tensor, dist: Tensor, batch: Tensor, params: Params,
                                is_graph_task: bool = True) -> Tensor:
    """``TensorGNAN.forward`` of the batched script — batched_pyg_main.py:133-184.

    rho acts on raw hop counts; pairs marked ``-1`` (cross-graph or
    unreachable) are zeroed (batched_pyg_main.py:158-159); node outputs are
    scatter-added per graph (batched_pyg_main.py:173-181).
    """
    fx = feature_mlps(x, params)                                   # :140-145
    m = _rho_dense(dist, params)                                   # :151-152
    m = torch.where((dist >= 0).unsqueeze(-1), m, torch.zeros_like(m))  # :155-156
    mf = torch.matmul(m.permute(2, 0, 1), fx.permute(2, 0, 1)).sum(dim=2).T  # [N, C]  :160-171
    if not is_graph_task:
        return mf
    g = int(batch.max()) + 1
    out = torch.zeros(g, mf.shape[1], dtype=mf.dtype)
    out.index_add_(0, batch.long(), mf)                            # :176-181
    return out


# --------------------------------------------------------------------------
# a8 / f-1: producer of the two dense inputs
# --------------------------------------------------------------------------
def pre_process_dense(edge_index: np.ndarray, num_nodes: int) -> Tuple[Tensor, Tensor]:
    """Dense ``node_distances`` / ``normalization_matrix`` — pre_process_datasets.py:109-121,128-140.

    Directed unit-weight shortest paths (duplicate edges are summed by the
    COO->LIL conversion and therefore become weight-2 edges),
    ``1/(1+d)`` with unreachable -> 0, and per-row counts of equal entries.
    """
    import scipy.sparse
    from scipy.sparse.csgraph import dijkstra

    ei = np.asarray(edge_index).reshape(2, -1)
    adj = scipy.sparse.coo_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(num_nodes, num_nodes))
    d = dijkstra(scipy.sparse.lil_matrix(adj))                     # :109-110
    nd = torch.from_numpy(d).float()
    nd = 1.0 / (nd + 1.0)                                          # :112-114 (inf -> 0)
    norm = torch.empty_like(nd)
    for i in range(num_nodes):                                     # :117-121
        vals, inv, cnt = torch.unique(nd[i], return_inverse=True, return_counts=True)
        norm[i] = cnt[inv].to(nd.dtype)
    return nd, norm


# --------------------------------------------------------------------------
# hop-code / shell / CSR restatements (integer work: bit-exact targets)
# --------------------------------------------------------------------------
REST = -1  # marker for "unreachable / not listed" in int hop matrices


def hop_codes_from_dense(nd: Tensor) -> np.ndarray:
    """int32 hop matrix from ``node_distances``: ``round(1/nd) - 1``, ``REST`` where ``nd == 0``.

    Exact for every value the reference's preprocessing can emit
    (pre_process_datasets.py:112-114).
    """
    a = nd.detach().cpu().double().numpy()
    out = np.full(a.shape, REST, dtype=np.int32)
    nz = a > 0
    out[nz] = np.rint(1.0 / a[nz]).astype(np.int32) - 1
    return out


def shell_counts(hops: np.ndarray, n_codes: int) -> np.ndarray:
    """``cnt[i, d]`` = #{j : hop(i, j) == d} for d < n_codes-1; last column = everything else.

    This is the reference's counting rule (pre_process_datasets.py:117-121)
    expressed per shell, with hops beyond ``n_codes-2`` folded into the rest
    bucket (SURVEY A.4, K-hop truncation).
    """
    n = hops.shape[0]
    k = n_codes - 1
    cnt = np.zeros((n, n_codes), dtype=np.int64)
    for d in range(k):
        cnt[:, d] = (hops == d).sum(axis=1)
    cnt[:, k] = hops.shape[1] - cnt[:, :k].sum(axis=1)
    return cnt


def csr_from_hops(hops: np.ndarray, max_hop: int):
    """List pairs with ``0 <= hop <= max_hop`` as CSR ``(rowptr int64, col int32, code uint8)``."""
    keep = (hops >= 0) & (hops <= max_hop)
    rowptr = np.zeros(hops.shape[0] + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(keep.sum(axis=1))
    rows, cols = np.nonzero(keep)
    return rowptr, cols.astype(np.int32), hops[rows, cols].astype(np.uint8)


def truncate_dense(nd: Tensor, max_hop: int) -> Tuple[Tensor, Tensor]:
    """The dense input a K-truncated CSR is *defined* to be equivalent to:
    entries with hop > K are set to 0 and the normalisation matrix is recounted
    by the reference's own rule (pre_process_datasets.py:136-140)."""
    hops = hop_codes_from_dense(nd)
    nd2 = nd.clone()
    nd2[torch.from_numpy((hops > max_hop) | (hops < 0))] = 0
    norm = torch.empty_like(nd2)
    for i in range(nd2.shape[0]):
        _, inv, cnt = torch.unique(nd2[i], return_inverse=True, return_counts=True)
        norm[i] = cnt[inv].to(nd2.dtype)
    return nd2, norm


def hop_inputs(n_codes: int) -> Tensor:
    """The distinct values ``node_distances`` can take, as the float32 numbers the reference stores:
    ``float32(1/(1+d))`` for d < n_codes-1, then 0 (pre_process_datasets.py:112-114)."""
    u = torch.zeros(n_codes, dtype=torch.float32)
    u[: n_codes - 1] = 1.0 / (torch.arange(n_codes - 1, dtype=torch.float32) + 1.0)
    return u


def rho_lut(params: Params, n_codes: int, dtype=torch.float32) -> Tensor:
    """``lut[d] = rho(1/(1+d))`` for d < n_codes-1 and ``lut[n_codes-1] = rho(0)``."""
    u = hop_inputs(n_codes).to(dtype)
    return mlp_apply(mlp_layers(params, "rho"), u.reshape(-1, 1))  # [n_codes, Crho]


def spmm_csr(rowptr: np.ndarray, col: np.ndarray, code: np.ndarray, S: Tensor, wtab: Tensor,
             with_rest: bool = True) -> Tensor:
    """rho-weighted neighbourhood sum over a hop-coded CSR (shell form).

    ``wtab [N, D, Cw]`` is the per-row weight table (already normalised);
    column ``w`` of ``S [Ncols, W]`` uses weight channel ``w % Cw``.
    ``out[i] = sum_e wtab[i, code_e] * S[col_e] + wtab[i, D-1] * (sum_j S[j] - sum_e S[col_e])``.
    """
    n = len(rowptr) - 1
    W = S.shape[1]
    cw = wtab.shape[2]
    reps = (W + cw - 1) // cw
    out = torch.zeros(n, W, dtype=S.dtype)
    total = S.sum(dim=0)
    colt = torch.from_numpy(np.asarray(col, dtype=np.int64))
    codet = torch.from_numpy(np.asarray(code, dtype=np.int64))
    for i in range(n):
        lo, hi = int(rowptr[i]), int(rowptr[i + 1])
        rows = S[colt[lo:hi]]                                      # gather
        w = wtab[i][codet[lo:hi]].repeat(1, reps)[:, :W] if cw > 1 else wtab[i][codet[lo:hi]]
        acc = (w * rows).sum(dim=0)
        if with_rest:
            wr = wtab[i, -1].repeat(reps)[:W] if cw > 1 else wtab[i, -1]
            acc = acc + wr * (total - rows.sum(dim=0))
        out[i] = acc
    return out


def spmm_csr_vectorised(rowptr: np.ndarray, col: np.ndarray, code: np.ndarray, S: Tensor, lut: Tensor,
                        cnt: Optional[np.ndarray]) -> Tensor:
    """:func:`spmm_csr` with the post-rho table ``lut[d] / cnt[i, d]`` (models.py:368-370 per shell), written with
    ``index_add`` so that it runs on graphs of 10^5..10^6 rows and stays differentiable w.r.t. ``S`` and ``lut`` (the
    float64 truth of the full-shape config-3 test).  Same identity as :func:`spmm_csr_sparse`:
    ``out[i] = sum_e (w_e - w_rest[i]) * S[col_e] + w_rest[i] * sum_j S[j]``; weight channel ``w % Cw`` for column ``w``."""
    n = len(rowptr) - 1
    W = S.shape[1]
    rp = torch.from_numpy(np.asarray(rowptr, dtype=np.int64))
    row_of = torch.repeat_interleave(torch.arange(n), rp[1:] - rp[:-1])
    codet = torch.from_numpy(np.asarray(code, dtype=np.int64))
    colt = torch.from_numpy(np.asarray(col, dtype=np.int64))
    cw = lut.shape[1]
    l = lut.to(S.dtype)
    if cw not in (1, W):
        l = l.repeat(1, (W + cw - 1) // cw)[:, :W]
    if cnt is None:
        w_e, w_rest = l[codet], l[-1].unsqueeze(0).expand(n, -1)
    else:
        c = torch.from_numpy(np.maximum(cnt, 1)).to(S.dtype)
        w_e = l[codet] / c[row_of, codet].unsqueeze(1)
        w_rest = l[-1].unsqueeze(0) / c[:, -1:]
    out = torch.zeros(n, W, dtype=S.dtype).index_add(0, row_of, (w_e - w_rest[row_of]) * S[colt])
    return out + w_rest * S.sum(dim=0, keepdim=True)


def weight_table(lut: Tensor, cnt: Optional[np.ndarray]) -> Tensor:
    """Post-rho normalised table ``wtab[i, d, :] = lut[d, :] / cnt[i, d]`` (models.py:369-370 per shell)."""
    if cnt is None:
        return lut.unsqueeze(0)
    c = torch.from_numpy(np.maximum(cnt, 1)).to(lut.dtype)
    return lut.unsqueeze(0) / c.unsqueeze(-1)


def row_lut_pre_rho(params: Params, cnt: np.ndarray, dtype=torch.float32) -> Tensor:
    """Pre-rho normalised per-row table ``wtab[i, d] = rho(u_d / cnt[i, d])`` (GNAN.py:65-67 per shell)."""
    n, D = cnt.shape
    u = hop_inputs(D).to(dtype)
    arg = u.unsqueeze(0) / torch.from_numpy(np.maximum(cnt, 1)).to(dtype)
    return mlp_apply(mlp_layers(params, "rho"), arg.reshape(-1, 1)).reshape(n, D, -1)


def rel_err(y: Tensor, truth: Tensor) -> float:
    """``max|y - truth| / max|truth|`` — the quantity the tolerance rule bounds."""
    t = truth.double()
    den = float(t.abs().max())
    return float((y.double() - t).abs().max()) / (den if den > 0 else 1.0)


def spmm_csr_sparse(rowptr: np.ndarray, col: np.ndarray, code: np.ndarray, S: Tensor, lut: Tensor,
                    cnt: Optional[np.ndarray]) -> Tensor:
    """:func:`spmm_csr` for one weight channel, vectorised through ``torch.sparse_csr`` (MKL, all host cores).

    This is the CPU comparator for graphs the dense reference cannot hold: edge values
    ``lut[code] / cnt[row, code]`` from the same table as the kernels use, with the rest-bucket term
    folded in as ``(w_e - w_rest[row]) * S[col]  +  w_rest[row] * total``.
    """
    n = len(rowptr) - 1
    rp = torch.from_numpy(np.asarray(rowptr, dtype=np.int64))
    deg = rp[1:] - rp[:-1]
    row_of_edge = torch.repeat_interleave(torch.arange(n), deg)
    codet = torch.from_numpy(np.asarray(code, dtype=np.int64))
    l = lut.reshape(-1).to(S.dtype)
    if cnt is None:
        w_e, w_rest = l[codet], l[-1].expand(n)
    else:
        c = torch.from_numpy(np.maximum(cnt, 1)).to(S.dtype)
        w_e = l[codet] / c[row_of_edge, codet]
        w_rest = l[-1] / c[:, -1]
    A = torch.sparse_csr_tensor(rp, torch.from_numpy(np.asarray(col, dtype=np.int64)), w_e - w_rest[row_of_edge],
                                size=(n, S.shape[0]))
    return A @ S + w_rest.unsqueeze(1) * S.sum(dim=0, keepdim=True)
