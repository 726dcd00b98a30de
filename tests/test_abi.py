"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports what the header declares."""
import ctypes
import os
import re

import pytest
import torch

import gnan_amd
from gnan_amd import _lib
from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gnan_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gnan_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    handle = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 6
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/gnan_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == names, "ctypes binding and header disagree"


def test_library_exports_nothing_but_the_declared_symbols():
    """-fvisibility=hidden + the header's visibility pragma: the dynamic symbol table holds the C ABI and nothing else
    (no C++ helpers of the library; `__hip_*` / `__hipRegister*` objects are the toolchain's fat-binary registration data)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({line.split()[-1] for line in out.splitlines() if line.strip()})
    ours = [s for s in exported if not s.startswith(("__hip_", "__hipRegister", "_init", "_fini"))]
    assert ours == declared_symbols(), sorted(set(ours) ^ set(declared_symbols()))


def integration_stub():
    """The python block of INTEGRATION.md section 2 (what a maintainer of the reference would paste into GNAN.py)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. Binding the C ABI directly"):]
    block = re.search(r"```python\n(.*?)```", sec, flags=re.S).group(1)
    return block.replace('C.CDLL("libgnan_hip.so")', "C.CDLL(%r)" % _lib.LIB_PATH)


def test_integration_stub_runs_against_the_built_library():
    """The stub loads the library, passes its own ABI assertion, and its struct is the header's (so the document cannot
    drift from the library unnoticed)."""
    ns = {}
    exec(integration_stub(), ns)                     # asserts gnan_abi_version() itself
    assert [f[0] for f in ns["FmlpArgs"]._fields_] == [f[0] for f in _lib.FmlpArgs._fields_]
    assert [f[1] for f in ns["FmlpArgs"]._fields_] == [f[1] for f in _lib.FmlpArgs._fields_]
    assert callable(ns["f_sums"])


def test_abi_version_and_error_channel():
    lib = _lib.lib()
    assert lib.gnan_abi_version() == _lib.ABI_VERSION
    # argument validation happens on the host before any HIP call: usable without a GPU
    a = _lib.SpmmArgs(n_rows=4, n_cols=4, W=0, D=2, Cw=1)
    rc = lib.gnan_spmm_fwd(a, None)
    assert rc == -1 and b"W must be" in lib.gnan_last_error()
    f = _lib.FmlpArgs(n=-1, F=1, L=1, C=1)
    assert lib.gnan_fmlp_fwd(f, None) == -1


def test_struct_layout_matches_header():
    """Field order of the ctypes structs == field order of the C structs (names must line up)."""
    text = open(os.path.join(ROOT, "include", "gnan_hip.h")).read()
    for struct, cls in (("gnan_fmlp_args", _lib.FmlpArgs), ("gnan_spmm_args", _lib.SpmmArgs),
                        ("gnan_fpwl_args", _lib.FpwlArgs), ("gnan_pwl_build_args", _lib.PwlBuildArgs),
                        ("gnan_fpwl_grad_args", _lib.FpwlGradArgs), ("gnan_fmlp_bwd_args", _lib.FmlpBwdArgs),
                        ("gnan_rho_lut_args", _lib.RhoLutArgs), ("gnan_moment_scales_args", _lib.MomentScalesArgs),
                        ("gnan_spmm_lut_grad_args", _lib.SpmmLutGradArgs), ("gnan_pack_bwd_rows_args", _lib.PackBwdRowsArgs),
                        ("gnan_spmm_bwd_narrow_args", _lib.SpmmBwdNarrowArgs), ("gnan_bfs_dense_args", _lib.BfsDenseArgs),
                        ("gnan_spmm_pb_args", _lib.SpmmPbArgs), ("gnan_spmm_pb_bwd_args", _lib.SpmmPbBwdArgs),
                        ("gnan_pb_pack1_args", _lib.PbPack1Args), ("gnan_pack_z_args", _lib.PackZArgs),
                        ("gnan_sorted_csr_args", _lib.SortedCsrArgs), ("gnan_pb_keys_args", _lib.PbKeysArgs),
                        ("gnan_pb_fill_args", _lib.PbFillArgs), ("gnan_csr_transpose_args", _lib.CsrTransposeArgs),
                        ("gnan_bfs_khop_args", _lib.BfsKhopArgs), ("gnan_loss_args", _lib.LossArgs),
                        ("gnan_small_mlp", _lib.SmallMlp), ("gnan_small_graph_args", _lib.SmallGraphArgs),
                        ("gnan_small_mlp_grads", _lib.SmallMlpGrads), ("gnan_small_graph_bwd_args", _lib.SmallGraphBwdArgs),
                        ("gnan_small_batch_args", _lib.SmallBatchArgs), ("gnan_fpwl_index_args", _lib.FpwlIndexArgs),
                        ("gnan_small_graph_nam_args", _lib.SmallGraphNamArgs),
                        ("gnan_small_graph_nam_bwd_args", _lib.SmallGraphNamBwdArgs),
                        ("gnan_small_batch_bwd_args", _lib.SmallBatchBwdArgs)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), text, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(None, 1)[1] if not decl.startswith("const") else decl.split(None, 2)[2]
            for n in names.split(","):
                fields.append(n.strip().lstrip("*").strip())
        assert fields == [f[0] for f in cls._fields_], struct


def test_the_hip_path_has_no_cpu_fallback():
    """The kernels' launch wrappers reject CPU tensors loudly, and a module on one device called with inputs on another raises:
    nothing silently moves work to the CPU.  (A module that LIVES on the CPU, called with CPU inputs, is the CPU route's —
    tests/test_cpu_route.py.)"""
    from gnan_amd import HopGraph
    from gnan_amd.aggregate import rho_aggregate
    from gnan_amd.functional import feature_mlps, stack_mlps
    from gnan_amd.models import TensorGNAN
    m = TensorGNAN(3, 2, 3, hidden_channels=8).eval()
    with pytest.raises(_lib.GnanHipError, match="no CPU fallback"):
        feature_mlps(torch.rand(5, 3), stack_mlps(m.fs), True)
    g = HopGraph.from_csr(torch.tensor([0, 1, 2]), torch.tensor([0, 1], dtype=torch.int32), torch.zeros(2, dtype=torch.uint8),
                          n_cols=2, n_codes=2)
    with pytest.raises(_lib.GnanHipError, match="no CPU fallback"):
        rho_aggregate(g, torch.rand(2, 1), torch.rand(2, 1), True)

    class Bag:
        pass

    d = Bag()
    d.x = torch.rand(5, 3, device="meta")              # inputs on another device than the module
    with pytest.raises(_lib.GnanHipError, match="ONE device"):
        m.forward(d)


def test_flat_parameter_store_keeps_the_module_contract():
    """Parameters become views of six contiguous buffers: names, values, optimizer steps, load_state_dict and
    gradient accumulation behave exactly as with independent Parameters."""
    from gnan_amd.functional import _fmlp_eager, stack_mlps
    from gnan_amd.models import TensorGNAN
    torch.manual_seed(0)
    m = TensorGNAN(5, 2, 3, hidden_channels=8)
    keys = list(m.state_dict().keys())
    before = {k: v.clone() for k, v in m.state_dict().items()}
    st = m._stacked("fs", m.fs)
    assert list(m.state_dict().keys()) == keys
    assert all(torch.equal(m.state_dict()[k], before[k]) for k in keys)
    for a, b in zip(st[:6], stack_mlps(m.fs)[:6]):
        assert torch.equal(a.detach(), b.detach())
    x = torch.rand(7, 5)
    _fmlp_eager(x, st, True).pow(2).sum().backward()
    got = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    m.zero_grad(set_to_none=True)
    _fmlp_eager(x, stack_mlps(m.fs), True).pow(2).sum().backward()
    for n, p in m.named_parameters():
        if n.startswith("fs."):
            assert torch.allclose(p.grad, got[n], atol=1e-6), n
    m.zero_grad(set_to_none=False)                          # grads stay linked to the flat buffer: accumulate twice
    for _ in range(2):
        _fmlp_eager(x, m._stacked("fs", m.fs), True).pow(2).sum().backward()
    for n, p in m.named_parameters():
        if n.startswith("fs."):
            assert torch.allclose(p.grad, 2 * got[n], atol=1e-5), n
    torch.optim.SGD(m.parameters(), lr=0.1).step()           # in-place update is visible through the buffers
    assert torch.equal(m._stacked("fs", m.fs).w_last.detach(), stack_mlps(m.fs).w_last.detach())
    m.load_state_dict(before)
    assert torch.equal(m._stacked("fs", m.fs).w_mid.detach(), stack_mlps(m.fs).w_mid.detach())
    import copy
    m2 = copy.deepcopy(m)                                     # copies lose the sharing; the guard re-homes them
    with torch.no_grad():
        m2.fs[1][3].weight.add_(1.0)
    assert torch.equal(m2._stacked("fs", m2.fs).w_mid.detach(), stack_mlps(m2.fs).w_mid.detach())
    assert m.double()._stacked("fs", m.fs).w_last.dtype == torch.float64
