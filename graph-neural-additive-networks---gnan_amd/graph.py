"""Hop-coded adjacency resident in HBM — the data structure the aggregation kernels read.

The reference ships two dense fp32 ``N x N`` matrices per graph
(``node_distances`` and ``normalization_matrix``, pre_process_datasets.py:104-142).
Both are piecewise constant over hop shells, so one byte per listed pair (the hop
index) plus a per-row table of shell sizes carries the same information:

* dense layout — ``code [N, N] uint8``; every pair is listed, ``code == D-1`` marks
  unreachable pairs.  Produced on the GPU from the reference's dense inputs.
* CSR layout   — ``rowptr``/``col``/``code``; only pairs within ``K`` hops are listed,
  everything else falls into the rest bucket ``D-1``.  The only way
  the large configurations can exist at all.

``cnt [N, D] int32`` holds the shell sizes ``|{j : hop(i, j) == d}|`` (last column:
size of the rest bucket), i.e. the distinct values of ``normalization_matrix`` row i.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch

from . import _lib

LONG_ROW_THRESHOLD = 512   # rows with more listed pairs than this go through the slice kernel
LONG_ROW_THRESHOLD_NARROW = 64   # ... for operand rows of one or two lanes: a single LANE walks such a row, and the few rows
                                 # of a few hundred pairs set the kernel's duration (arxiv-shaped backward: 127 -> 25 us)
NARROW_PLAN_MAX_ROWS = 8192
SLICE_EDGES = 2048         # pairs per slice of a hub row
PACK_SHIFT = 29            # packed index entries: column id in the low 29 bits, hop code above (gnan_spmm_args.packed_index)
HOT_COLUMNS_MIN = 16384     # ... and no fewer than this (smaller graphs: their hot values survive in L2 as they are)
HOT_COLUMNS = 262144        # neighbours whose operand rows get a compact second copy (narrow operands, HopGraph.hot_columns)
HOT_COLUMNS_MIN_NNZ = 1 << 24    # below this the two extra launches that fill the copy cost more than the gathers save
HOT_COLUMNS_MIN_SHARE = 0.15     # ... and so does a graph whose K most listed neighbours receive less than this share of the pairs


LONG_PLAN_IN_HIP = True     # HopGraph.long_row_plan() by gnan_long_row_plan_count / _fill
TRANSPOSE_IN_HIP = True     # HopGraph.transposed() by gnan_csr_transpose
PB_PLAN_IN_HIP = True       # the pair-level work of HopGraph.pb_plan by gnan_pb_plan_* (False: framework ops; CPU tensors always)
SORTED_COPY_IN_HIP = True   # the degree-sorted copy by gnan_degree_sorted_csr (False: the framework ops below; CPU tensors always)
PB_LDS_BYTES = 65536        # propagation-blocked narrow aggregation (csrc/spmm_pb.hip): LDS of a column block / of a bin's accumulators
PB_SLOT_PAIRS = 512         # ... entries per accumulator slot: a row with more owns several (no LDS address is hit by a whole wavefront)
PB_CHUNK = 16               # ... tiles are padded to whole chunks of this many entries


@dataclass
class PbPlan:
    """The listed pairs of a CSR graph bucketed into (row bin, column block) tiles for ``gnan_spmm_pb_fwd`` (see
    ``include/gnan_hip.h``): static index work, built once per graph and operand width."""
    W: int
    n_entries: int
    src: torch.Tensor             # int16 [n_entries]
    dst: torch.Tensor             # int16 [n_entries]
    cb_width: int
    n_cblocks: int
    chunk_q: torch.Tensor         # int32 [n_chunks], column-block-major
    cb_chunk_ptr: torch.Tensor    # int32 [n_cblocks + 1]
    n_bins: int
    acc_per_bin: int
    bin_order: torch.Tensor       # int32 [n_bins], largest first
    bin_entry_ptr: torch.Tensor   # int32 [n_bins + 1]
    bin_row_ptr: torch.Tensor     # int32 [n_bins + 1]
    slot_ptr: torch.Tensor        # int32 [n_rows + 1]
    n_acc: int
    code_base: int
    self_col: Optional[torch.Tensor]   # int32 [n_rows] or None
    headroom_bits: int
    n_pairs: int                  # real entries (without pads and without the pairs self_col serves)
    self_is_row: bool = False     # every row's self_col entry is the row itself: the kernels need not read self_col


@dataclass
class LongRowPlan:
    """Hub rows of a CSR graph, cut into fixed-size slices (deterministic reduction order)."""
    rows: Optional[torch.Tensor]        # int32 [n_long] output-row indices
    slice_ptr: Optional[torch.Tensor]   # int32 [n_long + 1]
    n_long: int = 0
    n_slices: int = 0
    threshold: int = LONG_ROW_THRESHOLD
    slice_edges: int = SLICE_EDGES


@dataclass
class HopGraph:
    n_rows: int
    n_cols: int
    n_codes: int                         # D: hop shells 0..D-2 plus the rest bucket D-1
    code: torch.Tensor                   # uint8: [nnz] (CSR) or [n_rows, n_cols] (dense)
    cnt: torch.Tensor                    # int32 [n_rows, D]
    rowptr: Optional[torch.Tensor] = None   # int32/int64 [n_rows + 1]; None => dense layout
    col: Optional[torch.Tensor] = None      # int32 [nnz]
    colp: Optional[torch.Tensor] = None     # int32 [nnz]: col | code << 29 (degree-sorted copies of graphs below 2^29 neighbours)
    _plan: Optional[LongRowPlan] = field(default=None, repr=False)
    _transposed: Optional["HopGraph"] = field(default=None, repr=False)
    _degree_order: Optional[torch.Tensor] = field(default=None, repr=False)
    _degree_plan: Optional[LongRowPlan] = field(default=None, repr=False)
    _dense_plans: dict = field(default_factory=dict, repr=False)
    _plans: dict = field(default_factory=dict, repr=False)          # hub-row plans by threshold (other than the default)
    _sorted_copy: Optional["HopGraph"] = field(default=None, repr=False)
    _hot: Optional[tuple] = field(default=None, repr=False)           # hot_columns(): (ids or None,)
    _sorted_copy_hot: Optional["HopGraph"] = field(default=None, repr=False)
    _hot_head_share: Optional[dict] = field(default=None, repr=False)   # share of the pairs listing the first K hot neighbours
    _inv_rest: Optional[torch.Tensor] = field(default=None, repr=False)
    _pb_plans: dict = field(default_factory=dict, repr=False)       # pb_plan(): operand width -> PbPlan or None
    _cnt_by_col: bool = field(default=False, repr=False)     # transposed graphs: ``cnt`` rows belong to the neighbours

    @property
    def is_dense(self) -> bool:
        return self.rowptr is None

    @property
    def nnz(self) -> int:
        return self.n_rows * self.n_cols if self.is_dense else int(self.col.numel())

    @property
    def device(self) -> torch.device:
        return self.code.device

    # ------------------------------------------------------------------ builders
    @staticmethod
    def from_dense(node_distances: torch.Tensor, normalization_matrix: Optional[torch.Tensor] = None) -> "HopGraph":
        """Re-derive hop codes and shell counts from the reference's dense inputs, on the GPU.

        Raises if ``node_distances`` is not of the form ``float32(1/(1+hop))`` / 0
        (pre_process_datasets.py:112-114) or if ``normalization_matrix`` is not the
        per-row count of equal entries (pre_process_datasets.py:117-121): the shell
        kernels are only equivalent to the reference under those two properties.
        """
        nd = node_distances
        _lib.require_device(nd, normalization_matrix)
        if nd.dim() != 2:
            raise ValueError(f"node_distances must be 2-D, got {tuple(nd.shape)}")
        nd = nd.float()
        if nd.stride(1) != 1 or (nd.shape[0] > 1 and nd.stride(0) < nd.shape[1]):     # transposed or expanded views
            nd = nd.contiguous()
        norm = normalization_matrix
        if norm is not None:
            if norm.shape != nd.shape:
                raise ValueError("normalization_matrix and node_distances differ in shape")
            norm = norm.float()
            if norm.stride() != nd.stride():
                nd, norm = nd.contiguous(), norm.contiguous()
        n_rows, n_cols = nd.shape
        code = torch.empty((n_rows, n_cols), dtype=torch.uint8, device=nd.device)
        cnt256 = torch.empty((n_rows, _lib.MAX_CODES), dtype=torch.int32, device=nd.device)
        status = torch.zeros(2, dtype=torch.int32, device=nd.device)
        _lib.check(_lib.lib().gnan_dense_to_code(_lib.ptr(nd), _lib.ptr(norm), n_rows, n_cols, nd.stride(0),
                                                 _lib.ptr(code), _lib.ptr(cnt256), _lib.ptr(status),
                                                 _lib.stream_of(nd)), "gnan_dense_to_code")
        flags, max_hop = (int(v) for v in status.tolist())   # one small D2H read per graph
        if flags & 1:
            raise _lib.GnanHipError(
                "node_distances holds values that are not float32(1/(1+hop)) with hop <= 254 (or 0 for "
                "unreachable pairs); the shell kernels cannot represent it")
        if flags & 2:
            raise _lib.GnanHipError(
                "normalization_matrix is not the per-row count of equal node_distances entries "
                "(pre_process_datasets.py:117-121); the shell kernels cannot represent it")
        D = max_hop + 2
        cnt = torch.cat([cnt256[:, : D - 1], cnt256[:, _lib.MAX_CODES - 1:]], dim=1).contiguous()
        return HopGraph(n_rows=n_rows, n_cols=n_cols, n_codes=D, code=code, cnt=cnt)

    @staticmethod
    def from_csr(rowptr: torch.Tensor, col: torch.Tensor, code: torch.Tensor, n_cols: int, n_codes: int,
                 cnt: Optional[torch.Tensor] = None) -> "HopGraph":
        """Wrap a caller-built hop-coded CSR.  ``code`` values must be < n_codes-1.  Pure index work:
        runs wherever the tensors live (the kernels later insist on device memory)."""
        if rowptr.dtype not in (torch.int32, torch.int64):
            raise TypeError("rowptr must be int32 or int64")
        n_rows = rowptr.numel() - 1
        col = col.to(torch.int32).contiguous()
        code = code.to(torch.uint8).contiguous()
        if not 2 <= n_codes <= _lib.MAX_CODES:
            raise ValueError(f"n_codes must be in [2, {_lib.MAX_CODES}]")
        if cnt is None:
            cnt = shell_counts_csr(rowptr, code, n_cols, n_codes)
        cnt = cnt.to(torch.int32).contiguous()
        if cnt.shape != (n_rows, n_codes):
            raise ValueError(f"cnt must be [{n_rows}, {n_codes}], got {tuple(cnt.shape)}")
        return HopGraph(n_rows=n_rows, n_cols=n_cols, n_codes=n_codes, code=code, cnt=cnt,
                        rowptr=rowptr.contiguous(), col=col)

    @staticmethod
    def from_edge_index(edge_index: torch.Tensor, num_nodes: int, max_hops: Optional[int] = None,
                        layout: str = "auto", rows: Optional[Tuple[int, int]] = None) -> "HopGraph":
        """Preprocessing on the GPU: hop codes and shell counts straight from ``edge_index``.

        * ``max_hops is None``: all-pairs BFS (``gnan_bfs_dense``) -> dense layout, what ``pre_process``
          (pre_process_datasets.py:104-142) encodes in its two N x N matrices; needs N^2 bytes.
        * ``max_hops == 1``: the K = 1 hop-coded CSR (self pair + direct neighbours), any size.
        * ``max_hops = K >= 2``: dense while N^2 bytes fit (``layout='auto'``), else — or with ``layout='csr'`` —
          the K-hop truncated hop-coded CSR (``gnan_bfs_khop``: every node within K directed hops is listed with
          its hop count, everything farther joins the rest bucket); ``rows=(lo, hi)`` builds only that row block
          (vertex partition, global column ids).  Listed nodes are ordered by (hop, node id) inside a row.
        Edges are directed as given; duplicate edges count once (the reference's COO->LIL conversion would turn
        them into weight-2 edges, SURVEY.md A.7 — coalesce upstream if that quirk matters).
        """
        _lib.require_device(edge_index)
        if layout not in ("auto", "dense", "csr"):
            raise ValueError("layout must be 'auto', 'dense' or 'csr'")
        ei = edge_index.long()
        n = int(num_nodes)
        dev = ei.device
        keep = ei[0] != ei[1] if max_hops == 1 else torch.ones(ei.shape[1], dtype=torch.bool, device=dev)
        key = torch.unique(ei[0, keep] * n + ei[1, keep])             # coalesced, sorted by (src, dst)
        src, dst = key // n, key % n
        khop_csr = max_hops is not None and max_hops >= 2 and (layout == "csr" or (layout == "auto" and n * n > (1 << 33)))
        if khop_csr:
            return _khop_csr(src, dst, n, int(max_hops), rows)
        if rows is not None:
            raise ValueError("rows=(lo, hi) is supported by the K-hop CSR layout only (max_hops >= 2, layout='csr')")
        if max_hops == 1:
            rows = torch.cat([torch.arange(n, device=dev), src])
            cols = torch.cat([torch.arange(n, device=dev), dst])
            code = torch.cat([torch.zeros(n, dtype=torch.uint8, device=dev),
                              torch.ones(src.numel(), dtype=torch.uint8, device=dev)])
            order = torch.argsort(rows * n + cols)
            rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
            rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
            if int(rowptr[-1]) < 2 ** 31:
                rowptr = rowptr.to(torch.int32)
            return HopGraph.from_csr(rowptr, cols[order].to(torch.int32), code[order], n_cols=n, n_codes=3)
        if n * n > (1 << 33):
            raise _lib.GnanHipError(f"all-pairs hop codes of {n} nodes need {n * n / 2**30:.0f} GiB; use max_hops=1")
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(torch.bincount(src, minlength=n), 0)
        rowptr, col = rowptr.to(torch.int32), dst.to(torch.int32).contiguous()
        code = torch.empty((n, n), dtype=torch.uint8, device=dev)
        cnt256 = torch.empty((n, _lib.MAX_CODES), dtype=torch.int32, device=dev)
        status = torch.zeros(2, dtype=torch.int32, device=dev)
        need = _lib.lib().gnan_bfs_dense_workspace_bytes(n)
        ws = torch.empty(max(1, need // 4), dtype=torch.int32, device=dev)
        a = _lib.BfsDenseArgs(rowptr=_lib.ptr(rowptr), col=_lib.ptr(col), n=n, max_hops=254 if max_hops is None else max_hops,
                              code=_lib.ptr(code), cnt=_lib.ptr(cnt256), status=_lib.ptr(status), workspace=_lib.ptr(ws),
                              workspace_bytes=need)
        _lib.check(_lib.lib().gnan_bfs_dense(a, _lib.stream_of(code)), "gnan_bfs_dense")
        flags, max_hop = (int(v) for v in status.tolist())
        if flags & 1:
            raise _lib.GnanHipError("a shortest path longer than 254 hops cannot be coded in one byte")
        D = max_hop + 2
        cnt = torch.cat([cnt256[:, : D - 1], cnt256[:, _lib.MAX_CODES - 1:]], dim=1).contiguous()
        return HopGraph(n_rows=n, n_cols=n, n_codes=D, code=code, cnt=cnt)

    # ------------------------------------------------------------------ derived structures
    def long_row_plan(self, row_ids: Optional[torch.Tensor] = None, threshold: int = LONG_ROW_THRESHOLD) -> LongRowPlan:
        """Hub rows (more than ``threshold`` listed pairs) and their slices; cached for the identity row order."""
        if self.is_dense:
            return LongRowPlan(None, None)
        if row_ids is None:
            if threshold == LONG_ROW_THRESHOLD and self._plan is not None:
                return self._plan
            if threshold in self._plans:
                return self._plans[threshold]
        if row_ids is None and LONG_PLAN_IN_HIP and self.rowptr.is_cuda and 0 < self.n_rows < 2 ** 31:
            plan = self._long_row_plan_hip(threshold)
        else:
            plan = self._long_row_plan_torch(row_ids, threshold)
        if row_ids is None:
            if threshold == LONG_ROW_THRESHOLD:
                self._plan = plan
            else:
                self._plans[threshold] = plan
        return plan

    def _long_row_plan_hip(self, threshold: int) -> LongRowPlan:
        """gnan_long_row_plan_count / _fill (csrc/graph_build.hip): the same arrays as the framework route below."""
        L, dev = _lib.lib(), self.device
        need = L.gnan_long_row_plan_workspace_bytes(self.n_rows)
        ws = torch.empty((need + 255) // 256 * 64, dtype=torch.int32, device=dev)
        total = torch.empty(1, dtype=torch.int32, device=dev)
        is64 = int(self.rowptr.dtype == torch.int64)
        st = _lib.stream_of(self.rowptr)
        _lib.check(L.gnan_long_row_plan_count(_lib.ptr(self.rowptr), is64, self.n_rows, threshold, _lib.ptr(ws), ws.numel() * 4,
                                              _lib.ptr(total), st), "gnan_long_row_plan_count")
        n_long = int(total.item())
        if n_long == 0:
            return LongRowPlan(None, None, threshold=threshold)
        rows = torch.empty(n_long, dtype=torch.int32, device=dev)
        ptr = torch.empty(n_long + 1, dtype=torch.int32, device=dev)
        _lib.check(L.gnan_long_row_plan_fill(_lib.ptr(self.rowptr), is64, self.n_rows, threshold, SLICE_EDGES, n_long, _lib.ptr(ws),
                                             ws.numel() * 4, _lib.ptr(rows), _lib.ptr(ptr), st), "gnan_long_row_plan_fill")
        return LongRowPlan(rows, ptr, n_long, int(ptr[-1].item()), threshold=threshold)

    def _long_row_plan_torch(self, row_ids, threshold: int) -> LongRowPlan:
        deg = (self.rowptr[1:] - self.rowptr[:-1])
        if row_ids is not None:
            deg = deg[row_ids.long()]
        long_rows = torch.nonzero(deg > threshold).flatten()
        n_long = int(long_rows.numel())
        if n_long == 0:
            plan = LongRowPlan(None, None, threshold=threshold)
        else:
            n_sl = (deg[long_rows] + SLICE_EDGES - 1) // SLICE_EDGES
            ptr = torch.zeros(n_long + 1, dtype=torch.int64, device=self.device)
            ptr[1:] = torch.cumsum(n_sl, 0)
            plan = LongRowPlan(long_rows.to(torch.int32), ptr.to(torch.int32), n_long, int(ptr[-1]), threshold=threshold)
        return plan

    def narrow_row_plan(self) -> LongRowPlan:
        """Hub-row plan for operand rows of one or two lanes: the low threshold while only a FEW rows exceed it (the tail
        of a skewed graph, which a single lane per row would serialise) — with many such rows (R-MAT 10M/100M: hundreds of
        thousands) a workgroup per row costs more than the tail (1.88 -> 2.51 ms), so they stay with the default plan."""
        plan = self.long_row_plan(None, LONG_ROW_THRESHOLD_NARROW)
        return plan if plan.n_long <= NARROW_PLAN_MAX_ROWS else self.long_row_plan()

    def dense_slice_plan(self, n_out: int) -> LongRowPlan:
        """Dense layout with few output rows: cut EVERY row into column slices, one workgroup each (a dense row has
        n_cols pairs; one lane group per row would leave most of the 256 CUs idle on a Cora-sized graph)."""
        key = int(n_out)
        hit = self._dense_plans.get(key)
        if hit is None:
            if len(self._dense_plans) > 8:
                self._dense_plans.clear()
            spr = (self.n_cols + SLICE_EDGES - 1) // SLICE_EDGES
            rows = torch.arange(n_out, dtype=torch.int32, device=self.device)
            ptr = torch.arange(n_out + 1, dtype=torch.int32, device=self.device) * spr
            hit = self._dense_plans[key] = LongRowPlan(rows, ptr, n_out, n_out * spr, threshold=0)
        return hit

    def pb_plan(self, W: int = 1) -> Optional["PbPlan"]:
        """Bucketed copy of the adjacency for the propagation-blocked narrow aggregation (``gnan_spmm_pb_fwd``), or None where
        it does not apply (dense layout, no listed code besides the self pair's, 2^31 entries or more, a row too long for a bin).

        Index work only, bit-exact, on whatever device the graph lives (the CPU suite checks it against the oracle).  A row
        with more than ``PB_SLOT_PAIRS`` entries owns ``ceil(deg / PB_SLOT_PAIRS)`` accumulator slots and its entries are dealt
        over them round-robin; rows go to bins in order, a bin holding as many slots as fit its LDS; within a bin the entries
        are grouped by column block (as many operand rows as fit the expand kernel's LDS), every group padded to whole chunks.
        The pair with hop code 0 of every row (its self pair; at most one per row) is left out and served from ``self_col``."""
        if W in self._pb_plans:
            return self._pb_plans[W]
        if self.is_dense:
            plan = None
        elif PB_PLAN_IN_HIP and self.rowptr.is_cuda:
            plan = self._build_pb_plan_hip(int(W))
        else:
            plan = self._build_pb_plan(int(W))
        self._pb_plans[W] = plan
        return plan

    def _build_pb_plan_hip(self, W: int) -> Optional["PbPlan"]:
        """:meth:`_build_pb_plan` with everything that touches every listed pair done by the library (``gnan_pb_plan_rows / _keys /
        _fill``: one pass per row, one per pair, a stable radix sort of 4-byte keys, one scatter) and only the n_rows- and
        tile-sized scans left to the framework — the same arrays, bit for bit (``tests/test_gpu_kernels.py``)."""
        dev, n, D, nnz = self.device, self.n_rows, self.n_codes, self.nnz
        if W not in (1, 2, 4) or D < 2 or D > 4 or n == 0 or nnz == 0 or nnz + PB_CHUNK * 64 >= 2 ** 31:
            return None
        L = _lib.lib()
        st = _lib.stream_of(self.rowptr)
        R = PB_LDS_BYTES // (8 * W)
        cbw = PB_LDS_BYTES // (4 * W)
        i32 = dict(dtype=torch.int32, device=dev)
        i64 = dict(dtype=torch.int64, device=dev)
        is64 = int(self.rowptr.dtype == torch.int64)
        c0, self_col, self_pos = (torch.empty(n, **i32) for _ in range(3))
        deg = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)
        long_rows = torch.nonzero(deg > int(L.gnan_pb_plan_long_row_threshold())).flatten().to(torch.int32)     # walked by a workgroup each
        n_long = int(long_rows.numel())
        _lib.check(L.gnan_pb_plan_rows(_lib.ptr(self.rowptr), is64, _lib.ptr(self.col), _lib.ptr(self.code), n, _lib.ptr(long_rows), n_long,
                                       _lib.ptr(c0), _lib.ptr(self_col), _lib.ptr(self_pos), st), "gnan_pb_plan_rows")
        code_base, n_acc = 0, D - 1
        if D >= 3 and int(c0.max()) <= 1:
            deg2, code_base, n_acc = deg - c0, 1, D - 2
        else:
            deg2, self_col, self_pos = deg, None, None
        del c0
        m = int(deg2.sum())
        if m == 0:
            return None
        nslot = ((deg2 + PB_SLOT_PAIRS - 1) // PB_SLOT_PAIRS).clamp_(min=1)
        slot_ptr = torch.zeros(n + 1, **i64)
        slot_ptr[1:] = torch.cumsum(nslot, 0)
        slots_per_bin = (R - 1) // n_acc
        eff = slots_per_bin - int(nslot.max())
        if eff < slots_per_bin // 2:
            return None
        bin_of_row = slot_ptr[:-1] // eff
        n_bins = int(bin_of_row[-1]) + 1
        bin_row_ptr = torch.searchsorted(bin_of_row, torch.arange(n_bins + 1, **i64))
        bin_slot0 = slot_ptr[bin_row_ptr[:-1]]
        n_cb = -(-self.n_cols // cbw)
        n_tiles = n_bins * n_cb
        if n_tiles >= 2 ** 31 - 1:
            return None
        slot_ptr32, bin_of_row32, bin_slot032 = slot_ptr.to(torch.int32), bin_of_row.to(torch.int32), bin_slot0.to(torch.int32)
        key, val = torch.empty(nnz, **i32), torch.empty(nnz, **i32)
        tmp_src, tmp_dst = torch.empty(nnz, dtype=torch.int16, device=dev), torch.empty(nnz, dtype=torch.int16, device=dev)
        tile_cnt = torch.zeros(n_tiles + 1, **i32)
        ka = _lib.PbKeysArgs(rowptr=_lib.ptr(self.rowptr), rowptr_is64=is64, col=_lib.ptr(self.col), code=_lib.ptr(self.code), n_rows=n,
                             self_pos=_lib.ptr(self_pos), code_base=code_base, n_acc=n_acc, slot_ptr=_lib.ptr(slot_ptr32),
                             bin_of_row=_lib.ptr(bin_of_row32), bin_slot0=_lib.ptr(bin_slot032), n_cb=n_cb, cb_width=cbw, n_tiles=n_tiles,
                             key=_lib.ptr(key), val=_lib.ptr(val), tmp_src=_lib.ptr(tmp_src), tmp_dst=_lib.ptr(tmp_dst),
                             tile_cnt=_lib.ptr(tile_cnt), long_rows=_lib.ptr(long_rows), n_long=n_long)
        _lib.check(L.gnan_pb_plan_keys(ka, st), "gnan_pb_plan_keys")
        cnt = tile_cnt[:n_tiles].to(torch.int64)
        padded = (cnt + PB_CHUNK - 1) // PB_CHUNK * PB_CHUNK
        tile_ptr = torch.zeros(n_tiles + 1, **i64)
        tile_ptr[1:] = torch.cumsum(padded, 0)
        n_entries = int(tile_ptr[-1])
        if n_entries >= 2 ** 31:
            return None
        tile_start = torch.zeros(n_tiles + 1, **i64)
        tile_start[1:] = torch.cumsum(cnt, 0)
        chunks = (padded // PB_CHUNK).view(n_bins, n_cb).t().contiguous().view(-1)        # column-block-major
        first = torch.zeros(n_tiles + 1, **i64)
        first[1:] = torch.cumsum(chunks, 0)
        n_chunks = int(first[-1])
        src16 = torch.empty(n_entries, dtype=torch.int16, device=dev)
        dst16 = torch.empty(n_entries, dtype=torch.int16, device=dev)
        chunk_q = torch.empty(n_chunks, **i32)
        tile_ptr32, tile_start32, first32 = tile_ptr.to(torch.int32), tile_start.to(torch.int32), first.to(torch.int32)
        need = L.gnan_pb_plan_fill_workspace_bytes(nnz, n_tiles)
        ws = torch.empty((need + 255) // 256 * 64, **i32)
        fa = _lib.PbFillArgs(nnz=nnz, n_kept=m, n_tiles=n_tiles, n_bins=n_bins, n_cb=n_cb, dummy=R - 1, key=_lib.ptr(key), val=_lib.ptr(val),
                             tmp_src=_lib.ptr(tmp_src), tmp_dst=_lib.ptr(tmp_dst), tile_ptr=_lib.ptr(tile_ptr32),
                             tile_start=_lib.ptr(tile_start32), tile_cnt=_lib.ptr(tile_cnt), chunk_first=_lib.ptr(first32),
                             src16=_lib.ptr(src16), dst16=_lib.ptr(dst16), chunk_q=_lib.ptr(chunk_q), workspace=_lib.ptr(ws),
                             workspace_bytes=ws.numel() * 4)
        _lib.check(L.gnan_pb_plan_fill(fa, st), "gnan_pb_plan_fill")
        bin_entry_ptr = tile_ptr[torch.arange(n_bins + 1, **i64) * n_cb]
        cb_chunk_ptr = first[torch.arange(n_cb + 1, **i64) * n_bins]
        bin_order = torch.argsort(bin_entry_ptr[1:] - bin_entry_ptr[:-1], descending=True, stable=True)
        headroom = max(1, int(deg2.max()) - 1).bit_length()
        c32 = lambda t: t.to(torch.int32).contiguous()
        return PbPlan(W=W, n_entries=n_entries, src=src16, dst=dst16, cb_width=cbw, n_cblocks=n_cb, chunk_q=chunk_q,
                      cb_chunk_ptr=c32(cb_chunk_ptr), n_bins=n_bins, acc_per_bin=R, bin_order=c32(bin_order),
                      bin_entry_ptr=c32(bin_entry_ptr), bin_row_ptr=c32(bin_row_ptr), slot_ptr=slot_ptr32, n_acc=n_acc,
                      code_base=code_base, self_col=self_col, headroom_bits=headroom, n_pairs=m,
                      self_is_row=bool(self_col is not None and n <= self.n_cols
                                       and torch.equal(self_col, torch.arange(n, dtype=torch.int32, device=dev))))

    def _build_pb_plan(self, W: int) -> Optional["PbPlan"]:
        dev, n, D = self.device, self.n_rows, self.n_codes
        nnz = self.nnz
        if W not in (1, 2, 4) or D < 2 or D > 4 or n == 0 or nnz == 0 or nnz + PB_CHUNK * 64 >= 2 ** 31:
            return None
        R = PB_LDS_BYTES // (8 * W)                 # accumulators (W-vectors of int64) per bin; the last one is the pads' dummy
        cbw = PB_LDS_BYTES // (4 * W)               # operand rows per column block
        i64 = dict(dtype=torch.int64, device=dev)
        rowptr = self.rowptr.to(torch.int64)
        deg = rowptr[1:] - rowptr[:-1]
        row = torch.repeat_interleave(torch.arange(n, **i64), deg)
        col = self.col.to(torch.int64)
        code = self.code.to(torch.int64)
        self_col, code_base, n_acc = None, 0, D - 1
        if D >= 3:
            is0 = code == 0
            rows0 = row[is0]
            if rows0.numel() == 0 or int(torch.bincount(rows0, minlength=n).max()) <= 1:
                self_col = torch.full((n,), -1, dtype=torch.int32, device=dev)
                self_col[rows0] = col[is0].to(torch.int32)
                keep = ~is0
                row, col, code = row[keep], col[keep], code[keep]
                code_base, n_acc = 1, D - 2
                del keep
            del is0, rows0
        m = int(row.numel())
        if m == 0:
            return None
        a = code - code_base
        del code
        deg2 = torch.bincount(row, minlength=n)
        nslot = ((deg2 + PB_SLOT_PAIRS - 1) // PB_SLOT_PAIRS).clamp_(min=1)
        slot_ptr = torch.zeros(n + 1, **i64)
        slot_ptr[1:] = torch.cumsum(nslot, 0)
        slots_per_bin = (R - 1) // n_acc
        eff = slots_per_bin - int(nslot.max())     # a row's slots never straddle two bins: rows are assigned by their FIRST slot
        if eff < slots_per_bin // 2:
            return None
        bin_of_row = slot_ptr[:-1] // eff
        n_bins = int(bin_of_row[-1]) + 1
        bin_row_ptr = torch.searchsorted(bin_of_row, torch.arange(n_bins + 1, **i64))
        bin_slot0 = slot_ptr[bin_row_ptr[:-1]]
        rowptr2 = torch.zeros(n + 1, **i64)
        rowptr2[1:] = torch.cumsum(deg2, 0)
        k = torch.arange(m, **i64) - rowptr2[row]
        b = bin_of_row[row]
        dst = (slot_ptr[row] + k % nslot[row] - bin_slot0[b]) * n_acc + a
        del k, a, rowptr2
        n_cb = -(-self.n_cols // cbw)
        cb = torch.div(col, cbw, rounding_mode="floor")
        srcl = col - cb * cbw
        tile = b * n_cb + cb
        del b, cb, col, row
        n_tiles = n_bins * n_cb
        if n_tiles < 2 ** 31:                          # (a radix sort of 4-byte keys: half the passes of the 8-byte one)
            tile_s, order = torch.sort(tile.to(torch.int32), stable=True)
            tile_s = tile_s.to(torch.int64)
        else:
            tile_s, order = torch.sort(tile, stable=True)
        del tile
        tile_cnt = torch.bincount(tile_s, minlength=n_tiles)
        padded = (tile_cnt + PB_CHUNK - 1) // PB_CHUNK * PB_CHUNK
        tile_ptr = torch.zeros(n_tiles + 1, **i64)
        tile_ptr[1:] = torch.cumsum(padded, 0)
        n_entries = int(tile_ptr[-1])
        if n_entries >= 2 ** 31:
            return None
        start = torch.zeros(n_tiles + 1, **i64)
        start[1:] = torch.cumsum(tile_cnt, 0)
        pos = tile_ptr[tile_s] + (torch.arange(m, **i64) - start[tile_s])
        del tile_s, start
        src16 = torch.zeros(n_entries, dtype=torch.int16, device=dev)
        dst16 = torch.full((n_entries,), R - 1, dtype=torch.int16, device=dev)
        src16[pos] = srcl[order].to(torch.int16)
        dst16[pos] = dst[order].to(torch.int16)
        del pos, order, srcl, dst
        bin_entry_ptr = tile_ptr[torch.arange(n_bins + 1, **i64) * n_cb]
        chunks = (padded // PB_CHUNK).view(n_bins, n_cb).t().contiguous().view(-1)        # column-block-major
        starts = tile_ptr[:-1].view(n_bins, n_cb).t().contiguous().view(-1)
        first = torch.zeros(chunks.numel() + 1, **i64)
        first[1:] = torch.cumsum(chunks, 0)
        n_chunks = int(first[-1])
        chunk_q = torch.repeat_interleave(starts - first[:-1] * PB_CHUNK, chunks) + torch.arange(n_chunks, **i64) * PB_CHUNK
        cb_chunk_ptr = first[torch.arange(n_cb + 1, **i64) * n_bins]
        bin_order = torch.argsort(bin_entry_ptr[1:] - bin_entry_ptr[:-1], descending=True, stable=True)
        headroom = max(1, int(deg2.max()) - 1).bit_length()
        i32 = lambda t: t.to(torch.int32).contiguous()
        return PbPlan(W=W, n_entries=n_entries, src=src16, dst=dst16, cb_width=cbw, n_cblocks=n_cb, chunk_q=i32(chunk_q),
                      cb_chunk_ptr=i32(cb_chunk_ptr), n_bins=n_bins, acc_per_bin=R, bin_order=i32(bin_order),
                      bin_entry_ptr=i32(bin_entry_ptr), bin_row_ptr=i32(bin_row_ptr), slot_ptr=i32(slot_ptr), n_acc=n_acc,
                      code_base=code_base, self_col=self_col, headroom_bits=headroom, n_pairs=m,
                      self_is_row=bool(self_col is not None and n <= self.n_cols
                                       and torch.equal(self_col, torch.arange(n, dtype=torch.int32, device=dev))))

    def inv_rest_count(self) -> torch.Tensor:
        """``1 / max(cnt[:, D-1], 1)`` as float32 ``[n_rows, 1]`` (graph data, cached): the normalisation of the rest bucket."""
        if self._inv_rest is None:
            self._inv_rest = 1.0 / self.cnt[:, self.n_codes - 1:self.n_codes].clamp_min(1).float()
        return self._inv_rest

    def degree_sorted_copy(self):
        """``(copy, order, plan)``: the CSR stored in the processing order of :meth:`degree_schedule` — row ``q`` of the
        copy is row ``order[q]`` of this graph (its pairs in their original order), ``cnt`` permuted alike, ``plan`` the
        hub-row plan of the copy.  The aggregation kernel walks the copy front to back and stores row ``q`` at
        ``Y[order[q]]`` (``scatter_out = 2``): same arithmetic per row, hence bit-identical output, but ``rowptr``, ``cnt``
        and the index pairs of the lane groups sharing a wavefront are adjacent in memory instead of scattered — fewer L2
        requests, which is what bounds the kernel.  Costs a second copy of (col, code) in HBM; cached per graph."""
        if self._sorted_copy is None and SORTED_COPY_IN_HIP and self.rowptr.is_cuda and self.n_rows < 2 ** 31:
            self._sorted_copy = self._degree_sorted_copy_hip()
        if self._sorted_copy is None:
            order, _ = self.degree_schedule()
            o = order.long()
            deg = (self.rowptr[1:] - self.rowptr[:-1]).long()
            deg_s = deg[o]
            rowptr_s = torch.zeros(self.n_rows + 1, dtype=torch.int64, device=self.device)
            rowptr_s[1:] = torch.cumsum(deg_s, 0)
            nnz = int(self.col.numel())
            col_s = torch.empty_like(self.col)
            code_s = torch.empty_like(self.code)
            chunk = 1 << 27                                   # pairs per pass: bounds the int64 temporaries
            bounds = torch.searchsorted(rowptr_s, torch.arange(0, nnz + chunk, chunk, device=self.device).clamp_(max=nnz))
            bounds[-1] = self.n_rows
            src_start = self.rowptr.long()[:-1][o]
            for b in range(bounds.numel() - 1):
                r0, r1 = int(bounds[b]), int(bounds[b + 1])
                if r1 <= r0:
                    continue
                e0, e1 = int(rowptr_s[r0]), int(rowptr_s[r1])
                if e1 <= e0:
                    continue
                d = deg_s[r0:r1]
                src = torch.arange(e0, e1, device=self.device) + torch.repeat_interleave(src_start[r0:r1] - rowptr_s[r0:r1], d)
                col_s[e0:e1] = self.col[src]
                code_s[e0:e1] = self.code[src]
                del src
            rp = rowptr_s if self.rowptr.dtype == torch.int64 else rowptr_s.to(torch.int32)
            # a transposed graph carries the FORWARD graph's count table (one row per neighbour, `transposed`): it is
            # indexed by column there and must not be permuted with the rows (nor can it be when the graph is rectangular)
            cnt_s = self.cnt[o].contiguous() if self.cnt.shape[0] == self.n_rows and not self._cnt_by_col else self.cnt
            g = HopGraph(n_rows=self.n_rows, n_cols=self.n_cols, n_codes=self.n_codes, code=code_s,
                         cnt=cnt_s, rowptr=rp, col=col_s)
            g.colp = g._packed_index()
            g.long_row_plan()
            self._sorted_copy = g
        return self._sorted_copy, self._degree_order, self._sorted_copy._plan

    def _degree_sorted_copy_hip(self) -> "HopGraph":
        """:meth:`degree_sorted_copy` by ``gnan_degree_sorted_csr`` (csrc/graph_build.hip): a stable radix sort of the rows by
        length, a scan, one copy pass — the same arrays as the framework route below, bit for bit."""
        dev, n, nnz = self.device, self.n_rows, self.nnz
        order = torch.empty(n, dtype=torch.int32, device=dev)
        rowptr_s = torch.empty_like(self.rowptr)
        col_s, code_s = torch.empty_like(self.col), torch.empty_like(self.code)
        pack = not (self.n_cols > (1 << PACK_SHIFT) or self.n_codes > 4)
        colp = torch.empty_like(self.col) if pack else None
        permute_cnt = self.cnt.shape[0] == n and not self._cnt_by_col
        cnt = self.cnt.contiguous() if permute_cnt else None
        cnt_s = torch.empty_like(cnt) if permute_cnt else self.cnt
        need = _lib.lib().gnan_degree_sorted_csr_workspace_bytes(n)
        ws = torch.empty((need + 255) // 256 * 64, dtype=torch.int32, device=dev)            # (the caching allocator aligns to 512 B)
        a = _lib.SortedCsrArgs(n_rows=n, nnz=nnz, rowptr=_lib.ptr(self.rowptr), rowptr_is64=int(self.rowptr.dtype == torch.int64),
                               col=_lib.ptr(self.col), code=_lib.ptr(self.code), cnt=_lib.ptr(cnt),
                               D=int(self.cnt.shape[1]), pack_shift=PACK_SHIFT if pack else 0, order=_lib.ptr(order),
                               rowptr_s=_lib.ptr(rowptr_s), col_s=_lib.ptr(col_s), code_s=_lib.ptr(code_s), colp_s=_lib.ptr(colp),
                               cnt_s=_lib.ptr(cnt_s) if permute_cnt else None, workspace=_lib.ptr(ws), workspace_bytes=ws.numel() * 4)
        _lib.check(_lib.lib().gnan_degree_sorted_csr(a, _lib.stream_of(self.rowptr)), "gnan_degree_sorted_csr")
        if self._degree_order is None:
            self._degree_order = order
            self._degree_plan = None                                   # (the index schedule's plan: built on demand, degree_schedule)
        g = HopGraph(n_rows=n, n_cols=self.n_cols, n_codes=self.n_codes, code=code_s, cnt=cnt_s, rowptr=rowptr_s, col=col_s)
        g.colp = colp
        g.long_row_plan()
        return g

    def hot_columns(self) -> Optional[torch.Tensor]:
        """The ``HOT_COLUMNS`` most listed neighbours (int64 node ids, most listed first; ties by id), or ``None`` when
        the graph is too small or too flat for them to matter.  Narrow operand rows (4..16 bytes) are gathered one L2
        request each, and the few 10^4 neighbours that a power-law graph lists in half of its pairs are spread over as
        many different 128-byte lines, which do not survive in a 4-MiB L2 next to the cold stream.  The aggregation
        therefore appends a compact copy of their operand rows to the operand (rows ``[n_cols, n_cols + K)``, 32 hot
        neighbours per line at W = 1) and walks a copy of the column ids that points there (:meth:`degree_sorted_copy`
        with ``hot=True``): same values, same order of additions, bit-identical output (10M / 100M R-MAT, W = 1:
        1.79 -> 1.52 ms on top of the degree-sorted walk).  Pure index work; cached per graph."""
        if self._hot is None:
            ids = None
            K = HOT_COLUMNS                                           # at most a sixteenth of the neighbours, a power of two
            while K > 1 and 16 * K > self.n_cols:
                K //= 2
            if (not self.is_dense and self.nnz >= HOT_COLUMNS_MIN_NNZ and K >= min(HOT_COLUMNS, HOT_COLUMNS_MIN)
                    and self.n_cols + K < 2 ** 31):                               # the renumbered ids stay int32
                listed = torch.zeros(self.n_cols, dtype=torch.int64, device=self.device)
                for e0 in range(0, self.nnz, 1 << 27):                # pairs per pass: bounds the int64 temporaries
                    listed += torch.bincount(self.col[e0:e0 + (1 << 27)].long(), minlength=self.n_cols)
                top = torch.argsort(listed, descending=True, stable=True)[:K]
                if float(listed[top].sum()) >= HOT_COLUMNS_MIN_SHARE * self.nnz:
                    ids = top.contiguous()
                    # share of the pairs listing the first 4096 / 8192 / 16384 of them: what an LDS-resident head of the copy
                    # would serve (functional.HOT_LDS_MIN_SHARE decides whether spmm_hot_kernel is worth its lower occupancy)
                    cum = torch.cumsum(listed[top].double(), 0) / max(self.nnz, 1)
                    self._hot_head_share = {min(k, K): float(cum[min(k, K) - 1]) for k in (4096, 8192, 16384)}
            self._hot = (ids,)
        return self._hot[0]

    def _packed_index(self) -> Optional[torch.Tensor]:
        """``col | code << 29`` as int32, one 4-byte index stream for the aggregation kernel instead of two (its index
        loads are L2 requests like its gathers); ``None`` when ids or codes do not fit."""
        if self.is_dense or self.n_cols > (1 << PACK_SHIFT) or self.n_codes > 4:
            return None
        out = torch.empty_like(self.col)
        chunk = 1 << 27                                       # pairs per pass: bounds the int64 temporaries
        for e0 in range(0, int(self.col.numel()), chunk):
            v = self.col[e0:e0 + chunk].long() | (self.code[e0:e0 + chunk].long() << PACK_SHIFT)
            out[e0:e0 + chunk] = torch.where(v >= (1 << 31), v - (1 << 32), v).to(torch.int32)
        return out

    def _with_hot_columns(self, hot: torch.Tensor) -> "HopGraph":
        """This graph with the column id of every pair that lists ``hot[k]`` replaced by ``n_cols + k``."""
        rank = torch.full((self.n_cols,), -1, dtype=torch.int32, device=self.device)
        rank[hot] = torch.arange(hot.numel(), dtype=torch.int32, device=self.device)
        col_h = torch.empty_like(self.col)
        chunk = 1 << 27                                       # pairs per pass: bounds the temporaries
        for e0 in range(0, int(self.col.numel()), chunk):
            c = self.col[e0:e0 + chunk]
            r = rank[c.long()]
            col_h[e0:e0 + chunk] = torch.where(r >= 0, r + self.n_cols, c)
        g = HopGraph(n_rows=self.n_rows, n_cols=self.n_cols + int(hot.numel()), n_codes=self.n_codes, code=self.code,
                     cnt=self.cnt, rowptr=self.rowptr, col=col_h)
        g._plan, g._plans, g._cnt_by_col = self._plan, self._plans, self._cnt_by_col     # same rows, same hub-row plans

        g.colp = g._packed_index()
        return g

    def degree_sorted_copy_hot(self):
        """``(copy, order, hot)``: :meth:`degree_sorted_copy` whose column ids point at the appended hot rows
        (:meth:`hot_columns`; the caller appends ``S[hot]`` to the operand), or ``hot = None`` and the plain copy."""
        copy, order, _ = self.degree_sorted_copy()
        hot = self.hot_columns()
        if hot is None:
            return copy, order, None
        if self._sorted_copy_hot is None:
            self._sorted_copy_hot = copy._with_hot_columns(hot)
            self._sorted_copy_hot._hot_head_share = dict(self._hot_head_share or {})
        return self._sorted_copy_hot, order, hot

    def degree_schedule(self):
        """Rows sorted by number of listed pairs (stable, shortest first) and the hub-row plan in that order — the processing
        schedule of the aggregation kernel: the 4..32 rows that share a wavefront then have (almost) equal
        lengths, so no lane group idles while a neighbour finishes a longer row.  (Longest first — so that the workgroups
        dispatched last hold the shortest rows — was measured in round 6 and LOSES: whole graph W = 64 4.45 -> 4.55 ms,
        W = 1 0.89 -> 1.07 ms, a 1/8 share 0.911 -> 0.916 ms: the hub-row slices of the same launch already start first, and
        the short rows' scattered stores then land in a burst at the end.)  Cached per graph."""
        if self._degree_order is None:
            deg = (self.rowptr[1:] - self.rowptr[:-1])
            self._degree_order = torch.argsort(deg, stable=True).to(torch.int32)
        if self._degree_plan is None:
            self._degree_plan = self.long_row_plan(self._degree_order)
        return self._degree_order, self._degree_plan

    def transposed(self) -> "HopGraph":
        """Adjacency with the roles of row and neighbour swapped (used for the gradient w.r.t. S).

        ``cnt`` of the transposed graph is the *forward* graph's table: the kernels index it by
        the neighbour (``weight_by_col``), which is the forward pass's output row.
        """
        if self._transposed is not None:
            return self._transposed
        if self.is_dense:
            t = HopGraph(n_rows=self.n_cols, n_cols=self.n_rows, n_codes=self.n_codes,
                         code=self.code.t().contiguous(), cnt=self.cnt)
        elif (TRANSPOSE_IN_HIP and self.rowptr.is_cuda and 0 < self.nnz < 2 ** 32 and self.n_rows < 2 ** 31 and self.n_cols < 2 ** 31):
            # gnan_csr_transpose (csrc/graph_build.hip): a stable radix sort of the pairs by column id + one gather — the same arrays
            # as the framework route below, bit for bit
            L, dev = _lib.lib(), self.device
            deg = self.rowptr[1:] - self.rowptr[:-1]
            long_rows = torch.nonzero(deg > int(L.gnan_pb_plan_long_row_threshold())).flatten().to(torch.int32)
            rowptr_t = torch.empty(self.n_cols + 1, dtype=self.rowptr.dtype, device=dev)
            col_t, code_t = torch.empty_like(self.col), torch.empty_like(self.code)
            need = L.gnan_csr_transpose_workspace_bytes(self.nnz, self.n_cols)
            ws = torch.empty((need + 255) // 256 * 64, dtype=torch.int32, device=dev)
            a = _lib.CsrTransposeArgs(n_rows=self.n_rows, n_cols=self.n_cols, nnz=self.nnz, rowptr=_lib.ptr(self.rowptr),
                                      rowptr_is64=int(self.rowptr.dtype == torch.int64), col=_lib.ptr(self.col), code=_lib.ptr(self.code),
                                      long_rows=_lib.ptr(long_rows), n_long=int(long_rows.numel()), rowptr_t=_lib.ptr(rowptr_t),
                                      col_t=_lib.ptr(col_t), code_t=_lib.ptr(code_t), workspace=_lib.ptr(ws), workspace_bytes=ws.numel() * 4)
            _lib.check(L.gnan_csr_transpose(a, _lib.stream_of(self.rowptr)), "gnan_csr_transpose")
            t = HopGraph(n_rows=self.n_cols, n_cols=self.n_rows, n_codes=self.n_codes, code=code_t, cnt=self.cnt, rowptr=rowptr_t, col=col_t)
        else:
            deg = (self.rowptr[1:] - self.rowptr[:-1]).long()
            row_of_edge = torch.repeat_interleave(torch.arange(self.n_rows, device=self.device), deg)
            order = torch.argsort(self.col.long(), stable=True)
            counts = torch.bincount(self.col.long(), minlength=self.n_cols)
            rowptr_t = torch.zeros(self.n_cols + 1, dtype=torch.int64, device=self.device)
            rowptr_t[1:] = torch.cumsum(counts, 0)
            t = HopGraph(n_rows=self.n_cols, n_cols=self.n_rows, n_codes=self.n_codes,
                         code=self.code[order].contiguous(), cnt=self.cnt,
                         rowptr=rowptr_t.to(self.rowptr.dtype), col=row_of_edge[order].to(torch.int32).contiguous())
        t._cnt_by_col = True
        self._transposed = t
        return t


KHOP_WORKGROUPS = 1024       # persistent workgroups of gnan_bfs_khop (each owns an N-bit bitmap and one queue)
KHOP_MAX_PAIRS = 1 << 33     # refuse K-hop lists beyond this many pairs (5 B each)
KHOP_QUEUE_START = 1 << 16   # first guess of the largest K-hop ball; grows x8 while some row overflows


def _khop_csr(src: torch.Tensor, dst: torch.Tensor, n: int, K: int, rows: Optional[Tuple[int, int]]) -> "HopGraph":
    """K-hop truncated hop-coded CSR from the coalesced, (src, dst)-sorted edge list (``gnan_bfs_khop``)."""
    dev = src.device
    lo, hi = (0, n) if rows is None else (int(rows[0]), int(rows[1]))
    if not 0 <= lo <= hi <= n:
        raise ValueError(f"rows={rows} outside [0, {n}]")
    if not 2 <= K <= _lib.MAX_CODES - 2:
        raise ValueError(f"max_hops must be in [2, {_lib.MAX_CODES - 2}]")
    n_rows = hi - lo
    adj_ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    adj_ptr[1:] = torch.cumsum(torch.bincount(src, minlength=n), 0)
    adj_col = dst.to(torch.int32).contiguous()
    wgs = max(1, min(KHOP_WORKGROUPS, n_rows))
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    level_cnt = torch.zeros((n_rows, K + 1), dtype=torch.int32, device=dev)
    cap = min(n, KHOP_QUEUE_START)
    while True:                                   # count pass; the queue capacity grows until every ball fits
        need = _lib.lib().gnan_bfs_khop_workspace_bytes(n, cap, wgs)
        ws = torch.empty(need // 4 + 1, dtype=torch.int32, device=dev)
        a = _lib.BfsKhopArgs(rowptr=_lib.ptr(adj_ptr), rowptr_is64=1, max_hops=K, col=_lib.ptr(adj_col), n=n, row_lo=lo,
                             row_hi=hi, level_cnt=_lib.ptr(level_cnt), out_rowptr=None, out_col=None, out_code=None,
                             queue_cap=cap, n_workgroups=wgs, status=_lib.ptr(status), workspace=_lib.ptr(ws),
                             workspace_bytes=ws.numel() * 4)
        _lib.check(_lib.lib().gnan_bfs_khop(a, _lib.stream_of(ws)), "gnan_bfs_khop")      # count pass
        if not int(status.item()) & 1:
            break
        if cap >= n:
            raise _lib.GnanHipError("gnan_bfs_khop: queue overflow at full capacity")   # cannot happen: a ball has <= n nodes
        cap = min(n, cap * 8)
        status.zero_()
    per_row = level_cnt.sum(1, dtype=torch.int64)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(per_row, 0)
    nnz = int(rowptr[-1])
    if nnz > KHOP_MAX_PAIRS:
        raise _lib.GnanHipError(f"{K}-hop lists of this graph hold {nnz:.3g} pairs (> {KHOP_MAX_PAIRS:.3g}); lower max_hops")
    col = torch.empty(nnz, dtype=torch.int32, device=dev)
    code = torch.empty(nnz, dtype=torch.uint8, device=dev)
    a.level_cnt, a.out_rowptr, a.out_col, a.out_code = None, _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(code)
    _lib.check(_lib.lib().gnan_bfs_khop(a, _lib.stream_of(ws)), "gnan_bfs_khop")          # fill pass
    # hop levels are contiguous inside a row but the order inside a level depends on scheduling: fix it by node id,
    # so that the same graph always gives the same lists (and the same floating-point summation order downstream)
    seg = torch.repeat_interleave(torch.arange(n_rows, device=dev), per_row) * (K + 1) + code.long()
    by_col = torch.sort(col.long(), stable=True).indices
    order = by_col[torch.sort(seg[by_col], stable=True).indices]
    col = col[order].contiguous()                 # `code` is unchanged by the permutation (constant per segment)
    cnt = torch.cat([level_cnt, (n - per_row).to(torch.int32).unsqueeze(1)], dim=1)
    if nnz < 2 ** 31:
        rowptr = rowptr.to(torch.int32)
    return HopGraph.from_csr(rowptr, col, code, n_cols=n, n_codes=K + 2, cnt=cnt)


def shell_counts_csr(rowptr: torch.Tensor, code: torch.Tensor, n_cols: int, n_codes: int) -> torch.Tensor:
    """``cnt[i, d]`` for a hop-coded CSR: listed pairs per code, rest bucket = ``n_cols - listed``
    (the reference's counting rule, pre_process_datasets.py:136-140, per shell)."""
    n_rows = rowptr.numel() - 1
    deg = (rowptr[1:] - rowptr[:-1]).long()
    row_of_edge = torch.repeat_interleave(torch.arange(n_rows, device=rowptr.device), deg)
    flat = torch.bincount(row_of_edge * n_codes + code.long(), minlength=n_rows * n_codes)
    cnt = flat.view(n_rows, n_codes)
    cnt[:, n_codes - 1] = n_cols - deg
    return cnt.to(torch.int32)


def hop_inputs(n_codes: int, device) -> torch.Tensor:
    """The distinct values ``node_distances`` takes: ``float32(1/(1+d))`` for ``d < D-1``, then 0
    (pre_process_datasets.py:112-114) — the only points rho is ever evaluated at."""
    key = (int(n_codes), str(device))
    hit = _HOP_INPUTS.get(key)
    if hit is None:                                         # D host divisions: bit-identical to the reference's
        u = torch.zeros(n_codes, dtype=torch.float32)      # CPU values; uploaded once per (D, device) — a per-call
        u[: n_codes - 1] = 1.0 / (torch.arange(n_codes - 1, dtype=torch.float32) + 1.0)   # H2D copy would be a sync
        hit = _HOP_INPUTS[key] = u.to(device)
    return hit


_HOP_INPUTS = {}
