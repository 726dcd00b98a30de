"""The path for CPU tensors — the reference's default device (``device='cpu'``, GNAN.py:10-11; main.py:49-52 falls back to it;
BASELINE's first configuration is "TensorGNAN on PyTorch CPU").

This is NOT a fallback of the HIP path: a module whose parameters and inputs live on the GPU runs the kernels of
``libgnan_hip.so`` or raises — it never comes here, whatever is missing.  A module that was never moved off the CPU, called
with CPU inputs, is evaluated by this file: the same shell / table identity the kernels use, written in plain
torch so that autograd differentiates it —

* ``rho`` is evaluated on the D distinct distances a graph holds (post-rho normalisation, models.py:368-370 / GNAN.py:159-170)
  or on the N x D distinct quotients ``u_d / |shell|`` (pre-rho, GNAN.py:65-67), never on N^2 pairs;
* the per-feature networks run as batched matrix products over chunks of features (GNAN.py:57-62 without the Python loop);
* the aggregation is one gather of the per-(row, shell) weights and one matrix product (dense inputs), or an ``index_add`` over the
  listed pairs plus the rest-bucket term (hop-coded CSR inputs).

Nothing here imports ``oracle/`` (test infrastructure) and nothing under ``tests -m gpu``, ``bench.py``'s timed region or
``__graft_entry__.smoke()`` reaches this file.  It is also an independent restatement the HIP kernels can be compared with
at no GPU cost (tests/test_cpu_route.py checks it against the golden vectors of the reference by the rule of SURVEY.md 8c).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .graph import HopGraph, hop_inputs

FEATURE_CHUNK = 64          # features per batched product: bounds the [chunk, N, H] activations (Cora: 44 MB per layer)


def applies(module, *tensors) -> bool:
    """True: CPU module and CPU inputs — this file evaluates the call.  False: everything on the GPU — the HIP path does.
    A mixture raises (the reference would fail inside the first ``Linear`` with torch's device error)."""
    devs = {t.device.type for t in tensors if torch.is_tensor(t)}
    p = next(iter(torch.nn.Module.parameters(module)), None)
    if p is not None:
        devs.add(p.device.type)
    if devs <= {"cpu"}:
        return True
    if "cpu" in devs:
        raise _lib.GnanHipError("gnan_amd: module and inputs must live on ONE device — all on the MI355X (the HIP path) or all on "
                                f"the CPU (gnan_amd.cpu_route); got {sorted(devs)}")
    return False


# =============================================================================
# inputs -> hop codes and shell sizes (pre_process_datasets.py:112-121, read backwards)
# =============================================================================
def graph_from_dense(node_distances: torch.Tensor, normalization_matrix: Optional[torch.Tensor]) -> HopGraph:
    """What ``HopGraph.from_dense`` derives on the GPU (``gnan_dense_to_code``), by torch on the CPU: ``code[i, j]`` = hop count of
    the pair (255: unreachable), ``cnt[i, d]`` = size of row i's shell d (last column: the unreachable ones).  Raises, like the
    GPU path, if the inputs are not of the reference's form."""
    nd = node_distances.detach().float()
    if nd.dim() != 2:
        raise ValueError(f"node_distances must be 2-D, got {tuple(nd.shape)}")
    n_rows, n_cols = nd.shape
    reach = nd > 0
    hop = torch.round(1.0 / nd[reach]) - 1.0
    if bool((nd < 0).any()) or hop.numel() and (float(hop.max()) > 254 or bool(((1.0 / (hop + 1.0)).float() != nd[reach]).any())):
        raise _lib.GnanHipError("node_distances holds values that are not float32(1/(1+hop)) with hop <= 254 (or 0 for "
                                "unreachable pairs); the shell form cannot represent it")
    max_hop = int(hop.max()) if hop.numel() else 0
    D = max_hop + 2
    code = torch.full((n_rows, n_cols), 255, dtype=torch.uint8)
    code[reach] = hop.to(torch.uint8)
    shell = code.long().clamp_(max=D - 1)
    cnt = torch.zeros((n_rows, D), dtype=torch.int32).scatter_add_(1, shell, torch.ones_like(shell, dtype=torch.int32))
    if normalization_matrix is not None:
        norm = normalization_matrix.detach().float()
        if norm.shape != nd.shape:
            raise ValueError("normalization_matrix and node_distances differ in shape")
        if bool((norm != cnt.gather(1, shell).float()).any()):
            raise _lib.GnanHipError("normalization_matrix is not the per-row count of equal node_distances entries "
                                    "(pre_process_datasets.py:117-121); the shell form cannot represent it")
    return HopGraph(n_rows=n_rows, n_cols=n_cols, n_codes=D, code=code, cnt=cnt)


def graph_of(module, inputs, want_norm: bool) -> HopGraph:
    """The hop-coded graph of ``inputs`` (dense matrices, or the CSR extension attributes), cached on the module like the GPU
    path's — by the identity and version of the input tensors."""
    g = getattr(inputs, "gnan_graph", None)
    if g is not None:
        return g
    cache = module._graph_cache
    if hasattr(inputs, "gnan_rowptr"):
        cnt = getattr(inputs, "gnan_cnt", None)
        src = (inputs.gnan_rowptr, inputs.gnan_col, inputs.gnan_code, cnt)
        extra = ("csr", int(inputs.x.shape[0]), int(inputs.gnan_n_codes))
        g = cache.get(src, extra)
        if g is None:
            g = cache.put(src, extra, HopGraph.from_csr(inputs.gnan_rowptr, inputs.gnan_col, inputs.gnan_code,
                                                        n_cols=inputs.x.shape[0], n_codes=int(inputs.gnan_n_codes), cnt=cnt))
        return g
    nd = inputs.node_distances
    norm = inputs.normalization_matrix if want_norm else getattr(inputs, "normalization_matrix", None)
    g = cache.get((nd, norm), "dense-cpu")
    if g is None:
        g = cache.put((nd, norm), "dense-cpu", graph_from_dense(nd, norm))
    return g


# =============================================================================
# the per-feature networks (GNAN.py:57-62), batched over chunks of features
# =============================================================================
def shape_functions(x: torch.Tensor, p, sum_features: bool, drop_p: float = 0.0) -> torch.Tensor:
    """``fx[n, k*C + c] = f_k(x[n, k])[c]`` or, with ``sum_features``, ``sum_k f_k(x[n, k])`` — ``p``: the stacked weights
    (``functional.StackedMLP``; views of the module's parameter store, so gradients reach the per-layer Parameters).
    ``drop_p > 0``: ``F.dropout`` behind every hidden ReLU (GNAN.py:28,32; training mode only)."""
    n, F = x.shape
    if F != p.F:
        raise ValueError(f"x has {F} feature columns, the model was built for {p.F}")
    x = x.float()
    drop = (lambda h: torch.nn.functional.dropout(h, drop_p, training=True)) if drop_p > 0 else (lambda h: h)
    out = x.new_zeros((n, p.C)) if sum_features else []
    for k0 in range(0, F, FEATURE_CHUNK):
        k1 = min(F, k0 + FEATURE_CHUNK)
        xt = x[:, k0:k1].t().unsqueeze(-1)                                    # [f, n, 1]
        if p.L == 1:
            h = xt * p.w_last[k0:k1].unsqueeze(1)                             # [f, n, C]
            if p.b_last is not None:
                h = h + p.b_last[k0:k1].unsqueeze(1)
        else:
            h = xt * p.w_first[k0:k1].unsqueeze(1)
            if p.b_first is not None:
                h = h + p.b_first[k0:k1].unsqueeze(1)
            h = drop(torch.relu(h))                                           # [f, n, H]
            for l in range(p.L - 2):
                h = torch.bmm(h, p.w_mid[l, k0:k1].transpose(1, 2))
                if p.b_mid is not None:
                    h = h + p.b_mid[l, k0:k1].unsqueeze(1)
                h = drop(torch.relu(h))
            h = torch.bmm(h, p.w_last[k0:k1].transpose(1, 2))                 # [f, n, C]
            if p.b_last is not None:
                h = h + p.b_last[k0:k1].unsqueeze(1)
        if sum_features:
            out = out + h.sum(0)
        else:
            out.append(h.permute(1, 0, 2).reshape(n, -1))
    return out if sum_features else torch.cat(out, dim=1)


# =============================================================================
# the aggregation (GNAN.py:64-70 / models.py:367-373 / GNAN.py:159-170) in shell form
# =============================================================================
def shell_weights(g: HopGraph, lut: torch.Tensor, use_cnt: bool) -> torch.Tensor:
    """Post-rho table ``wt[i, d, :] = lut[d, :] / |shell_i(d)|`` (models.py:368-370), ``[N, D, C_rho]``; without normalisation the
    global table broadcast over the rows."""
    if not use_cnt:
        return lut.unsqueeze(0).expand(g.n_rows, -1, -1)
    return lut.unsqueeze(0) / g.cnt.clamp_min(1).to(lut.dtype).unsqueeze(-1)


def aggregate(g: HopGraph, S: torch.Tensor, wt: torch.Tensor, rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``Y[q, w] = sum_j wt[i_q, shell(i_q, j), w % C_rho] * S[j, w]`` over ALL nodes j: the listed pairs by their hop code, every
    other pair through the rest bucket ``wt[i, D-1] * (sum_j S[j] - sum_listed S[j])``.  ``rows``: the output rows
    wanted (``GNAN.forward(inputs, node_ids)``); ``wt [N, D, C_rho]`` per adjacency row."""
    D, cw = g.n_codes, wt.shape[-1]
    W = S.shape[1]
    if W % cw:
        raise ValueError("operand width must be a multiple of the weight-channel count")
    if g.is_dense:
        shell = g.code.long().clamp(max=D - 1)                                # [N, N]; 255 (unreachable) -> the rest shell
        if rows is not None:
            shell, wt = shell[rows], wt[rows]
        per_pair = torch.gather(wt, 1, shell.unsqueeze(-1).expand(-1, -1, cw))        # [n_out, N, C_rho]
        if cw == 1:
            return per_pair[:, :, 0] @ S
        return torch.einsum("ijc,jrc->irc", per_pair, S.view(S.shape[0], W // cw, cw)).reshape(-1, W)
    rowptr = g.rowptr.long()
    deg = rowptr[1:] - rowptr[:-1]
    row_of = torch.repeat_interleave(torch.arange(g.n_rows), deg)
    col, code = g.col.long(), g.code.long()
    reps = W // cw
    w_e = wt[row_of, code].repeat(1, reps)                                    # [nnz, W]
    w_rest = wt[:, D - 1].repeat(1, reps)                                     # [N, W]
    gathered = S[col]
    Y = torch.zeros((g.n_rows, W), dtype=S.dtype).index_add(0, row_of, (w_e - w_rest[row_of]) * gathered)
    Y = Y + w_rest * S.sum(dim=0, keepdim=True)
    return Y if rows is None else Y[rows]


def _rho_lut(module, g: HopGraph) -> torch.Tensor:
    """``rho`` on the D distinct distances of the graph, ``[D, C_rho]`` (the same float32 inputs the reference feeds per pair)."""
    return _rho(module, hop_inputs(g.n_codes, "cpu"))


def _rho(module, arg: torch.Tensor) -> torch.Tensor:
    """``rho`` on a vector of arguments, ``[n, C_rho]`` — through the module's parameter store like the shape functions (so that
    the gradient reaches BOTH faces of the parameters: the per-layer tensors and the flat buffers ``flat_parameters()`` lists)."""
    return shape_functions(arg.reshape(-1, 1), module._stacked("rho", [module.rho]), False)


def _pre_rho_weights(module, g: HopGraph) -> torch.Tensor:
    """``wt[i, d, :] = rho(u_d / |shell_i(d)|)`` — GNAN.py:65-67 per shell: N x D evaluations of rho instead of N^2."""
    u = hop_inputs(g.n_codes, "cpu")
    arg = u.unsqueeze(0) / g.cnt.clamp_min(1).float()                         # torch.div(node_distances, normalization_matrix)
    return _rho(module, arg).view(g.n_rows, g.n_codes, -1)


def _drop(module) -> float:
    return float(module.dropout) if (module.training and module.dropout and module.dropout > 0) else 0.0


def _readout(module, Y: torch.Tensor) -> torch.Tensor:
    return Y.sum(dim=0).view(-1, 1) if getattr(module, "is_graph_task", False) else Y


# =============================================================================
# the four forwards
# =============================================================================
def forward_nam(module, x: torch.Tensor) -> torch.Tensor:
    """models.py:292-300."""
    return shape_functions(x, module._stacked("fs", module.fs), True, _drop(module))


def forward_standalone_tensor(module, inputs) -> torch.Tensor:
    """GNAN.py:55-79: pre-rho normalisation, rho ``out_channels`` wide."""
    g = graph_of(module, inputs, want_norm=bool(module.normalize_rho))
    S = shape_functions(inputs.x, module._stacked("fs", module.fs), True, _drop(module))
    wt = _pre_rho_weights(module, g) if module.normalize_rho else shell_weights(g, _rho_lut(module, g), False)
    return _readout(module, aggregate(g, S, wt))


def forward_gnan(module, inputs, node_ids=None) -> torch.Tensor:
    """GNAN.py:146-172 / models.py:451-477: post-rho normalisation, optionally on the requested nodes only."""
    g = graph_of(module, inputs, want_norm=True)
    S = shape_functions(inputs.x, module._stacked("fs", module.fs), True, _drop(module))
    wt = shell_weights(g, _rho_lut(module, g), bool(module.normalize_rho))
    rows = None
    if node_ids is not None:
        rows = torch.as_tensor(list(node_ids) if not torch.is_tensor(node_ids) else node_ids, dtype=torch.int64)
    return aggregate(g, S, wt, rows)


def forward_models_tensor(module, inputs) -> torch.Tensor:
    """models.py:358-384: post-rho normalisation; graph tasks with ``readout_n_layers > 0`` end in the NAM read-out over the
    per-feature aggregates.  (``aggregation_order`` does not change the function: the CPU route sums the features first.)"""
    g = graph_of(module, inputs, want_norm=bool(module.normalize_rho))
    wt = shell_weights(g, _rho_lut(module, g), bool(module.normalize_rho))
    if module.is_graph_task and module.readout_n_layers > 0:
        fx = shape_functions(inputs.x, module._stacked("fs", module.fs), False, _drop(module))     # [N, F] (f is one wide)
        hidden = aggregate(g, fx, wt).sum(dim=0).view(1, -1)                                        # models.py:379
        return forward_nam(module.readout_nam, hidden).T                                            # [C, 1]  models.py:380-384
    S = shape_functions(inputs.x, module._stacked("fs", module.fs), True, _drop(module))
    return _readout(module, aggregate(g, S, wt))
