"""ctypes binding of ``libgnan_hip.so`` (the C ABI declared in ``include/gnan_hip.h``).

The library is the product: there is no CPU or eager-PyTorch fallback.  If the
shared object is missing or a call fails, the error is raised to the caller.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GNAN_HIP_LIB: development aid for same-box A/B runs of two builds of the library (tools/ab_lib.sh)
LIB_PATH = os.environ.get("GNAN_HIP_LIB") or os.path.join(_HERE, "libgnan_hip.so")
ABI_VERSION = 45
ERR_BAD_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_WORKSPACE = -1, -2, -3, -4      # enum gnan_status

GNAN_F32, GNAN_BF16 = 0, 1
FMLP_AUTO, FMLP_LANE, FMLP_MFMA, FMLP_PWL = 0, 1, 2, 3   # PWL is host-side only (gnan_fpwl_fwd)
MAX_CODES = 256


class GnanHipError(RuntimeError):
    pass


class FmlpArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("n", C.c_int64), ("x_stride", C.c_int64),
        ("F", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("C", C.c_int32),
        ("w_first", C.c_void_p), ("b_first", C.c_void_p), ("w_mid", C.c_void_p), ("b_mid", C.c_void_p),
        ("w_last", C.c_void_p), ("b_last", C.c_void_p),
        ("sum_features", C.c_int32),
        ("out", C.c_void_p), ("out_stride", C.c_int64),
        ("algo", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64),
    ]


class FpwlArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("n", C.c_int64), ("x_stride", C.c_int64), ("F", C.c_int32), ("C", C.c_int32),
        ("off", C.c_void_p), ("anchor", C.c_void_p), ("val", C.c_void_p), ("slope", C.c_void_p),
        ("max_pieces", C.c_int32), ("features_per_group", C.c_int32), ("max_group_pieces", C.c_int32),
        ("sum_features", C.c_int32), ("out", C.c_void_p), ("out_stride", C.c_int64),
        ("out_dtype", C.c_int32), ("total", C.c_void_p), ("total_workspace", C.c_void_p), ("total_workspace_bytes", C.c_size_t),
        ("total_rows", C.c_int64), ("piece_out", C.c_void_p), ("piece_in", C.c_void_p), ("flags", C.c_int32),
        ("index_table", C.c_void_p), ("index_key", C.c_void_p), ("index_buckets", C.c_int32),
        ("sum_workspace", C.c_void_p), ("sum_workspace_bytes", C.c_size_t),
        ("sum_total", C.c_void_p), ("sum_total_workspace", C.c_void_p), ("sum_total_workspace_bytes", C.c_size_t),
        ("sum_total_arrive", C.c_void_p),
    ]


class FpwlIndexArgs(C.Structure):
    _fields_ = [
        ("off", C.c_void_p), ("anchor", C.c_void_p), ("F", C.c_int32), ("buckets", C.c_int32),
        ("range", C.c_void_p), ("table", C.c_void_p), ("key", C.c_void_p), ("stats", C.c_void_p),
    ]


FPWL_MOMENTS_GENERAL, FPWL_LOCATE_SORTED, FPWL_INDEX_HALF_LINES, FPWL_INDEX_BS512, FPWL_INDEX_BS1024 = 1, 2, 4, 8, 16   # gnan_fpwl_args.flags
FPWL_ROWS_MOMENTS_LANE_PER_CHANNEL = 32


class PwlBuildArgs(C.Structure):
    _fields_ = [
        ("w_first", C.c_void_p), ("b_first", C.c_void_p), ("w_mid", C.c_void_p), ("b_mid", C.c_void_p),
        ("w_last", C.c_void_p), ("b_last", C.c_void_p),
        ("F", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("C", C.c_int32), ("cap", C.c_int32),
        ("anchor", C.c_void_p), ("val", C.c_void_p), ("slope", C.c_void_p), ("off", C.c_void_p),
        ("overflow", C.c_void_p), ("scratch", C.c_void_p), ("scratch_bytes", C.c_size_t),
        ("index_range", C.c_void_p), ("index_table", C.c_void_p), ("index_key", C.c_void_p), ("index_buckets", C.c_int32),
    ]


class FmlpBwdArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("n", C.c_int64), ("x_stride", C.c_int64),
        ("F", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("C", C.c_int32),
        ("w_first", C.c_void_p), ("b_first", C.c_void_p), ("w_mid", C.c_void_p), ("b_mid", C.c_void_p),
        ("w_last", C.c_void_p), ("b_last", C.c_void_p),
        ("sum_features", C.c_int32), ("grad", C.c_void_p), ("grad_stride", C.c_int64),
        ("d_w_first", C.c_void_p), ("d_b_first", C.c_void_p), ("d_w_mid", C.c_void_p), ("d_b_mid", C.c_void_p),
        ("d_w_last", C.c_void_p), ("d_b_last", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64),
    ]


class FpwlGradArgs(C.Structure):
    _fields_ = [
        ("off", C.c_void_p), ("anchor", C.c_void_p), ("moments", C.c_void_p), ("moments_fixed", C.c_void_p),
        ("scales", C.c_void_p),
        ("w_first", C.c_void_p), ("b_first", C.c_void_p), ("w_mid", C.c_void_p), ("b_mid", C.c_void_p),
        ("w_last", C.c_void_p), ("b_last", C.c_void_p),
        ("F", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("C", C.c_int32), ("max_pieces", C.c_int32),
        ("d_w_first", C.c_void_p), ("d_b_first", C.c_void_p), ("d_w_mid", C.c_void_p), ("d_b_mid", C.c_void_p),
        ("d_w_last", C.c_void_p), ("d_b_last", C.c_void_p),
    ]


class RhoLutArgs(C.Structure):
    _fields_ = [
        ("cnt", C.c_void_p), ("cnt_stride", C.c_int64), ("n_rows", C.c_int64), ("D", C.c_int32), ("C", C.c_int32),
        ("u", C.c_void_p), ("anchor", C.c_void_p), ("val", C.c_void_p), ("slope", C.c_void_p), ("n_pieces", C.c_void_p),
        ("max_pieces", C.c_int32), ("lut", C.c_void_p), ("arg", C.c_void_p),
    ]


class SpmmArgs(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int64), ("n_cols", C.c_int64),
        ("rowptr", C.c_void_p), ("rowptr_is64", C.c_int32),
        ("col", C.c_void_p), ("code", C.c_void_p), ("row_ids", C.c_void_p),
        ("S", C.c_void_p), ("s_dtype", C.c_int32), ("W", C.c_int32), ("s_stride", C.c_int64),
        ("lut", C.c_void_p), ("lut_row_stride", C.c_int64), ("D", C.c_int32), ("Cw", C.c_int32),
        ("cnt", C.c_void_p), ("cnt_stride", C.c_int64),
        ("s_total", C.c_void_p), ("weight_by_col", C.c_int32), ("minus_rest", C.c_int32),
        ("reduce_cr", C.c_int32), ("scatter_out", C.c_int32), ("Y", C.c_void_p), ("y_stride", C.c_int64),
        ("long_threshold", C.c_int64), ("long_rows", C.c_void_p), ("long_slice_ptr", C.c_void_p),
        ("n_long", C.c_int32), ("n_slices", C.c_int32), ("slice_edges", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("s_by_code", C.c_int32), ("nnz", C.c_int64), ("packed_index", C.c_int32),
        ("hot_lo", C.c_int64), ("hot_rows", C.c_int32), ("shell_out", C.c_void_p),
    ]


class SortedCsrArgs(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int64), ("nnz", C.c_int64), ("rowptr", C.c_void_p), ("rowptr_is64", C.c_int32), ("col", C.c_void_p),
        ("code", C.c_void_p), ("cnt", C.c_void_p), ("D", C.c_int32), ("pack_shift", C.c_int32), ("order", C.c_void_p),
        ("rowptr_s", C.c_void_p), ("col_s", C.c_void_p), ("code_s", C.c_void_p), ("colp_s", C.c_void_p), ("cnt_s", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class CsrTransposeArgs(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int64), ("n_cols", C.c_int64), ("nnz", C.c_int64), ("rowptr", C.c_void_p), ("rowptr_is64", C.c_int32),
        ("col", C.c_void_p), ("code", C.c_void_p), ("long_rows", C.c_void_p), ("n_long", C.c_int32), ("rowptr_t", C.c_void_p),
        ("col_t", C.c_void_p), ("code_t", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class PbKeysArgs(C.Structure):
    _fields_ = [
        ("rowptr", C.c_void_p), ("rowptr_is64", C.c_int32), ("col", C.c_void_p), ("code", C.c_void_p), ("n_rows", C.c_int64),
        ("self_pos", C.c_void_p), ("code_base", C.c_int32), ("n_acc", C.c_int32), ("slot_ptr", C.c_void_p), ("bin_of_row", C.c_void_p),
        ("bin_slot0", C.c_void_p), ("n_cb", C.c_int32), ("cb_width", C.c_int32), ("n_tiles", C.c_uint32), ("key", C.c_void_p),
        ("val", C.c_void_p), ("tmp_src", C.c_void_p), ("tmp_dst", C.c_void_p), ("tile_cnt", C.c_void_p), ("long_rows", C.c_void_p),
        ("n_long", C.c_int32),
    ]


class PbFillArgs(C.Structure):
    _fields_ = [
        ("nnz", C.c_int64), ("n_kept", C.c_int64), ("n_tiles", C.c_uint32), ("n_bins", C.c_int32), ("n_cb", C.c_int32),
        ("dummy", C.c_int32), ("key", C.c_void_p), ("val", C.c_void_p), ("tmp_src", C.c_void_p), ("tmp_dst", C.c_void_p),
        ("tile_ptr", C.c_void_p), ("tile_start", C.c_void_p), ("tile_cnt", C.c_void_p), ("chunk_first", C.c_void_p),
        ("src16", C.c_void_p), ("dst16", C.c_void_p), ("chunk_q", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class SpmmPbArgs(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int64), ("n_cols", C.c_int64), ("S", C.c_void_p), ("s_stride", C.c_int64), ("W", C.c_int32),
        ("D", C.c_int32), ("lut", C.c_void_p), ("cnt", C.c_void_p), ("cnt_stride", C.c_int64), ("s_total", C.c_void_p),
        ("Y", C.c_void_p), ("y_stride", C.c_int64), ("n_entries", C.c_int64), ("src", C.c_void_p), ("dst", C.c_void_p),
        ("cb_width", C.c_int32), ("n_cblocks", C.c_int32), ("chunk_q", C.c_void_p), ("cb_chunk_ptr", C.c_void_p),
        ("n_bins", C.c_int32), ("acc_per_bin", C.c_int32), ("bin_order", C.c_void_p), ("bin_entry_ptr", C.c_void_p),
        ("bin_row_ptr", C.c_void_p), ("slot_ptr", C.c_void_p), ("n_acc", C.c_int32), ("code_base", C.c_int32),
        ("self_col", C.c_void_p), ("headroom_bits", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("flags", C.c_int32), ("self_is_row", C.c_int32),
        ("shell_out", C.c_void_p), ("S_self", C.c_void_p), ("out_add", C.c_void_p), ("out_add_scale", C.c_void_p),
    ]


class PackZArgs(C.Structure):
    _fields_ = [
        ("n", C.c_int64), ("dY", C.c_void_p), ("dy_stride", C.c_int64), ("cnt", C.c_void_p), ("cnt_stride", C.c_int64),
        ("D", C.c_int32), ("with_rest", C.c_int32), ("lut", C.c_void_p), ("shell", C.c_void_p), ("s_total", C.c_void_p),
        ("Z", C.c_void_p), ("q", C.c_void_p), ("dlut", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class PbPack1Args(C.Structure):
    _fields_ = [
        ("n", C.c_int64), ("dY", C.c_void_p), ("dy_stride", C.c_int64), ("cnt", C.c_void_p), ("cnt_stride", C.c_int64),
        ("D", C.c_int32), ("d1", C.c_int32), ("with_rest", C.c_int32), ("self_is_row", C.c_int32), ("self_col", C.c_void_p),
        ("lut", C.c_void_p), ("S", C.c_void_p), ("shell", C.c_void_p), ("s_total", C.c_void_p), ("c", C.c_void_p), ("e", C.c_void_p),
        ("q", C.c_void_p), ("dlut", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class SpmmPbBwdArgs(C.Structure):
    _fields_ = [
        ("pb", SpmmPbArgs), ("v_self", C.c_void_p), ("s_rows", C.c_void_p), ("s_rows_stride", C.c_int64),
        ("with_rest", C.c_int32), ("dS", C.c_void_p), ("ds_stride", C.c_int64), ("dlut", C.c_void_p), ("ds_add", C.c_void_p),
        ("ds_add_scale", C.c_void_p), ("rest_total", C.c_void_p), ("rest_q", C.c_void_p),
    ]


class MomentScalesArgs(C.Structure):
    _fields_ = [
        ("grad", C.c_void_p), ("n", C.c_int64), ("width", C.c_int32), ("bits", C.c_int32), ("grad_stride", C.c_int64),
        ("anchor", C.c_void_p), ("T", C.c_int64), ("n_anchors", C.c_void_p), ("x_abs_max", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("scales", C.c_void_p),
        ("zero", C.c_void_p), ("zero_bytes", C.c_size_t), ("arrive_counter", C.c_void_p),
    ]


MOMENT_SCALES_WORKSPACE_BYTES = 8192      # GNAN_MOMENT_SCALES_WORKSPACE_BYTES


class SpmmLutGradArgs(C.Structure):
    _fields_ = [
        ("spmm", SpmmArgs), ("dY", C.c_void_p), ("dy_stride", C.c_int64), ("dy_channels", C.c_int32),
        ("reduce_rows", C.c_int32), ("dwt", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class PackBwdRowsArgs(C.Structure):
    _fields_ = [
        ("dY", C.c_void_p), ("dy_stride", C.c_int64), ("W", C.c_int32), ("D", C.c_int32), ("cnt", C.c_void_p),
        ("cnt_stride", C.c_int64), ("n", C.c_int64), ("with_rest", C.c_int32), ("half", C.c_int32), ("V", C.c_void_p),
        ("hot", C.c_void_p), ("n_hot", C.c_int64), ("q_sum", C.c_void_p), ("q_workspace", C.c_void_p),
        ("q_workspace_bytes", C.c_size_t), ("q_arrive", C.c_void_p),
    ]


class SpmmBwdNarrowArgs(C.Structure):
    _fields_ = [
        ("spmm", SpmmArgs), ("s_rows", C.c_void_p), ("s_rows_stride", C.c_int64), ("w_real", C.c_int32),
        ("with_rest", C.c_int32), ("dS", C.c_void_p), ("ds_stride", C.c_int64), ("dlut", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("ds_add", C.c_void_p),
        ("hot_code_lo", C.c_int32), ("hot_codes", C.c_int32), ("ds_add_scale", C.c_void_p), ("rest_total", C.c_void_p),
        ("rest_q", C.c_void_p),
    ]


class SmallMlp(C.Structure):
    _fields_ = [
        ("L", C.c_int32), ("H", C.c_int32), ("C", C.c_int32), ("w_first", C.c_void_p), ("b_first", C.c_void_p),
        ("w_mid", C.c_void_p), ("b_mid", C.c_void_p), ("w_last", C.c_void_p), ("b_last", C.c_void_p),
    ]


class SmallGraphArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_stride", C.c_int64), ("n", C.c_int32), ("F", C.c_int32), ("f", SmallMlp), ("rho", SmallMlp),
        ("code", C.c_void_p), ("D", C.c_int32), ("pre_rho", C.c_int32), ("cnt", C.c_void_p), ("cnt_stride", C.c_int64),
        ("S", C.c_void_p), ("lut", C.c_void_p), ("Y", C.c_void_p), ("Ysum", C.c_void_p), ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_size_t),
    ]


class SmallMlpGrads(C.Structure):
    _fields_ = [("w_first", C.c_void_p), ("b_first", C.c_void_p), ("w_mid", C.c_void_p), ("b_mid", C.c_void_p),
                ("w_last", C.c_void_p), ("b_last", C.c_void_p)]


class SmallGraphBwdArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_stride", C.c_int64), ("n", C.c_int32), ("F", C.c_int32), ("f", SmallMlp), ("rho", SmallMlp),
        ("code", C.c_void_p), ("D", C.c_int32), ("pre_rho", C.c_int32), ("cnt", C.c_void_p), ("cnt_stride", C.c_int64),
        ("S", C.c_void_p), ("lut", C.c_void_p), ("dY", C.c_void_p), ("dYsum", C.c_void_p), ("df", SmallMlpGrads),
        ("drho", SmallMlpGrads), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class SmallGraphNamArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_stride", C.c_int64), ("n", C.c_int32), ("F", C.c_int32), ("f", SmallMlp), ("rho", SmallMlp),
        ("nam", SmallMlp), ("code", C.c_void_p), ("D", C.c_int32), ("reserved", C.c_int32), ("cnt", C.c_void_p),
        ("cnt_stride", C.c_int64), ("fx", C.c_void_p), ("lut", C.c_void_p), ("hidden", C.c_void_p), ("out", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class SmallGraphNamBwdArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_stride", C.c_int64), ("n", C.c_int32), ("F", C.c_int32), ("f", SmallMlp), ("rho", SmallMlp),
        ("nam", SmallMlp), ("code", C.c_void_p), ("D", C.c_int32), ("reserved", C.c_int32), ("cnt", C.c_void_p),
        ("cnt_stride", C.c_int64), ("fx", C.c_void_p), ("lut", C.c_void_p), ("hidden", C.c_void_p), ("d_out", C.c_void_p),
        ("df", SmallMlpGrads), ("drho", SmallMlpGrads), ("dnam", SmallMlpGrads), ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_size_t),
    ]


class LossArgs(C.Structure):
    _fields_ = [
        ("logits", C.c_void_p), ("n_rows", C.c_int64), ("C", C.c_int32), ("kind", C.c_int32), ("stride", C.c_int64),
        ("index", C.c_void_p), ("n", C.c_int64), ("labels", C.c_void_p), ("loss", C.c_void_p), ("hits", C.c_void_p),
        ("grad", C.c_void_p), ("grad_stride", C.c_int64), ("loss_sum", C.c_void_p), ("hits_sum", C.c_void_p),
        ("skip_sums", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("label_flag", C.c_void_p),
    ]


LOSS_BCE_LOGITS, LOSS_CROSS_ENTROPY = 0, 1


class BfsDenseArgs(C.Structure):
    _fields_ = [
        ("rowptr", C.c_void_p), ("col", C.c_void_p), ("n", C.c_int32), ("max_hops", C.c_int32), ("code", C.c_void_p),
        ("cnt", C.c_void_p), ("status", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class BfsKhopArgs(C.Structure):
    _fields_ = [
        ("rowptr", C.c_void_p), ("rowptr_is64", C.c_int32), ("max_hops", C.c_int32), ("col", C.c_void_p), ("n", C.c_int64),
        ("row_lo", C.c_int64), ("row_hi", C.c_int64), ("level_cnt", C.c_void_p), ("out_rowptr", C.c_void_p),
        ("out_col", C.c_void_p), ("out_code", C.c_void_p), ("queue_cap", C.c_int32), ("n_workgroups", C.c_int32),
        ("status", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class RestTermArgs(C.Structure):
    _fields_ = [
        ("Y", C.c_void_p), ("y_stride", C.c_int64), ("n", C.c_int64), ("total", C.c_void_p), ("W", C.c_int32),
        ("lut", C.c_void_p), ("lut_row_stride", C.c_int64), ("D", C.c_int32), ("Cw", C.c_int32), ("cnt", C.c_void_p),
        ("cnt_stride", C.c_int64), ("row_ids", C.c_void_p), ("reduce_cr", C.c_int32),
    ]


class SmallBatchArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_stride", C.c_int64), ("total_nodes", C.c_int64), ("F", C.c_int32), ("n_graphs", C.c_int32),
        ("max_nodes", C.c_int32), ("f", SmallMlp), ("rho", SmallMlp), ("code", C.c_void_p), ("node_off", C.c_void_p),
        ("code_off", C.c_void_p), ("D", C.c_int32), ("rho_raw_hops", C.c_int32), ("rest_zero", C.c_int32),
        ("S", C.c_void_p), ("lut", C.c_void_p), ("Y", C.c_void_p), ("Ysum", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("cnt", C.c_void_p), ("cnt_stride", C.c_int64),
    ]


class SmallBatchBwdArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_stride", C.c_int64), ("total_nodes", C.c_int64), ("F", C.c_int32), ("n_graphs", C.c_int32),
        ("max_nodes", C.c_int32), ("f", SmallMlp), ("rho", SmallMlp), ("code", C.c_void_p), ("node_off", C.c_void_p),
        ("code_off", C.c_void_p), ("D", C.c_int32), ("rho_raw_hops", C.c_int32), ("rest_zero", C.c_int32),
        ("S", C.c_void_p), ("lut", C.c_void_p), ("dY", C.c_void_p), ("dYsum", C.c_void_p), ("df", SmallMlpGrads),
        ("drho", SmallMlpGrads), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("cnt", C.c_void_p),
        ("cnt_stride", C.c_int64),
    ]


_lib: Optional[C.CDLL] = None

# every symbol include/gnan_hip.h declares: (name, restype, argtypes)
SYMBOLS = {
    "gnan_abi_version": (C.c_int, []),
    "gnan_last_error": (C.c_char_p, []),
    "gnan_fmlp_fwd_workspace_bytes": (C.c_size_t, [C.POINTER(FmlpArgs)]),
    "gnan_fmlp_fwd": (C.c_int, [C.POINTER(FmlpArgs), C.c_void_p]),
    "gnan_fmlp_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(FmlpBwdArgs)]),
    "gnan_fmlp_bwd": (C.c_int, [C.POINTER(FmlpBwdArgs), C.c_void_p]),
    "gnan_dropout_mask": (C.c_int, [C.c_uint64, C.c_float, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "gnan_pwl_build_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "gnan_pwl_build": (C.c_int, [C.POINTER(PwlBuildArgs), C.c_void_p]),
    "gnan_pwl_check_fit": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "gnan_rho_row_lut": (C.c_int, [C.POINTER(RhoLutArgs), C.c_void_p]),
    "gnan_fpwl_total_workspace_bytes": (C.c_size_t, [C.POINTER(FpwlArgs)]),
    "gnan_fpwl_sum_workspace_bytes": (C.c_size_t, [C.POINTER(FpwlArgs)]),
    "gnan_fpwl_fwd": (C.c_int, [C.POINTER(FpwlArgs), C.c_void_p]),
    "gnan_fpwl_index_build": (C.c_int, [C.POINTER(FpwlIndexArgs), C.c_void_p]),
    "gnan_feature_range": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gnan_fpwl_moments": (C.c_int, [C.POINTER(FpwlArgs), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gnan_fpwl_moments_fixed": (C.c_int, [C.POINTER(FpwlArgs), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "gnan_fpwl_locate_bytes": (C.c_size_t, [C.POINTER(FpwlArgs)]),
    "gnan_fpwl_locate": (C.c_int, [C.POINTER(FpwlArgs), C.c_void_p, C.c_void_p, C.c_void_p]),
    "gnan_fpwl_rows_fwd": (C.c_int, [C.POINTER(FpwlArgs), C.c_void_p, C.c_void_p, C.c_void_p]),
    "gnan_fpwl_rows_moments_fixed": (C.c_int, [C.POINTER(FpwlArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                               C.c_void_p, C.c_void_p]),
    "gnan_fpwl_moment_scales": (C.c_int, [C.POINTER(MomentScalesArgs), C.c_void_p]),
    "gnan_fpwl_param_grads": (C.c_int, [C.POINTER(FpwlGradArgs), C.c_void_p]),
    "gnan_graph_replace_memsets": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "gnan_graph_node_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gnan_small_graph_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "gnan_small_graph_fwd": (C.c_int, [C.POINTER(SmallGraphArgs), C.c_void_p]),
    "gnan_small_graph_bwd": (C.c_int, [C.POINTER(SmallGraphBwdArgs), C.c_void_p]),
    "gnan_small_graph_bwd_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "gnan_small_graph_nam_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "gnan_small_graph_nam_fwd": (C.c_int, [C.POINTER(SmallGraphNamArgs), C.c_void_p]),
    "gnan_small_graph_nam_bwd": (C.c_int, [C.POINTER(SmallGraphNamBwdArgs), C.c_void_p]),
    "gnan_small_batch_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int64, C.c_int32, C.c_int32]),
    "gnan_small_batch_fwd": (C.c_int, [C.POINTER(SmallBatchArgs), C.c_void_p]),
    "gnan_small_batch_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(SmallBatchBwdArgs)]),
    "gnan_small_batch_bwd": (C.c_int, [C.POINTER(SmallBatchBwdArgs), C.c_void_p]),
    "gnan_hops_to_code": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gnan_dense_blocks_to_code": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    "gnan_multi_copy": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p]),
    "gnan_loss_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gnan_loss_step": (C.c_int, [C.POINTER(LossArgs), C.c_void_p]),
    "gnan_spmm_fwd_workspace_bytes": (C.c_size_t, [C.POINTER(SpmmArgs)]),
    "gnan_spmm_fwd": (C.c_int, [C.POINTER(SpmmArgs), C.c_void_p]),
    "gnan_degree_sorted_csr_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gnan_degree_sorted_csr": (C.c_int, [C.POINTER(SortedCsrArgs), C.c_void_p]),
    "gnan_long_row_plan_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gnan_long_row_plan_count": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "gnan_long_row_plan_fill": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "gnan_csr_transpose_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "gnan_csr_transpose": (C.c_int, [C.POINTER(CsrTransposeArgs), C.c_void_p]),
    "gnan_pb_plan_long_row_threshold": (C.c_int32, []),
    "gnan_pb_plan_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "gnan_pb_plan_keys": (C.c_int, [C.POINTER(PbKeysArgs), C.c_void_p]),
    "gnan_pb_plan_fill_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_uint32]),
    "gnan_pb_plan_fill": (C.c_int, [C.POINTER(PbFillArgs), C.c_void_p]),
    "gnan_spmm_pb_workspace_bytes": (C.c_size_t, [C.POINTER(SpmmPbArgs)]),
    "gnan_spmm_pb_fwd": (C.c_int, [C.POINTER(SpmmPbArgs), C.c_void_p]),
    "gnan_spmm_pb_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(SpmmPbBwdArgs)]),
    "gnan_spmm_pb_bwd": (C.c_int, [C.POINTER(SpmmPbBwdArgs), C.c_void_p]),
    "gnan_weight_table": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "gnan_colsum_weighted": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    "gnan_spmm_pack_z_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gnan_spmm_pack_z": (C.c_int, [C.POINTER(PackZArgs), C.c_void_p]),
    "gnan_spmm_pb_pack1_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gnan_spmm_pb_pack1": (C.c_int, [C.POINTER(PbPack1Args), C.c_void_p]),
    "gnan_spmm_shell_sums": (C.c_int, [C.POINTER(SpmmArgs), C.c_void_p]),
    "gnan_spmm_lut_grad_workspace_bytes": (C.c_size_t, [C.POINTER(SpmmLutGradArgs)]),
    "gnan_spmm_lut_grad": (C.c_int, [C.POINTER(SpmmLutGradArgs), C.c_void_p]),
    "gnan_spmm_pack_bwd_rows": (C.c_int, [C.POINTER(PackBwdRowsArgs), C.c_void_p]),
    "gnan_spmm_pack_bwd_rows_workspace_bytes": (C.c_size_t, [C.POINTER(PackBwdRowsArgs)]),
    "gnan_spmm_bwd_narrow_workspace_bytes": (C.c_size_t, [C.POINTER(SpmmBwdNarrowArgs)]),
    "gnan_spmm_bwd_narrow": (C.c_int, [C.POINTER(SpmmBwdNarrowArgs), C.c_void_p]),
    "gnan_colsum_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "gnan_colsum": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t,
                              C.c_void_p]),
    "gnan_rest_term_add": (C.c_int, [C.POINTER(RestTermArgs), C.c_void_p]),
    "gnan_segment_sum": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "gnan_feature_sum": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "gnan_gather_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "gnan_bfs_dense_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "gnan_bfs_dense": (C.c_int, [C.POINTER(BfsDenseArgs), C.c_void_p]),
    "gnan_bfs_khop_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "gnan_bfs_khop": (C.c_int, [C.POINTER(BfsKhopArgs), C.c_void_p]),
    "gnan_colsum_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.c_void_p]),
    "gnan_dense_to_code": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}


def lib() -> C.CDLL:
    """Load the HIP library (once).  Raises ``GnanHipError`` if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GnanHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for this path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.gnan_abi_version() != ABI_VERSION:
            raise GnanHipError(f"ABI mismatch: library {handle.gnan_abi_version()} vs binding {ABI_VERSION}")
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().gnan_last_error()
        raise GnanHipError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def multi_copy(pairs) -> None:
    """``dst.copy_(src)`` for up to eight (dst, src) pairs of contiguous same-shape, same-dtype device tensors in one launch."""
    pairs = [(d, s) for d, s in pairs if d.numel()]
    if not pairs:
        return
    for d, s in pairs:
        if d.shape != s.shape or d.dtype != s.dtype or not (d.is_contiguous() and s.is_contiguous() and d.is_cuda and s.is_cuda):
            raise GnanHipError("multi_copy: contiguous device tensors of equal shape and dtype")
    n = len(pairs)
    src = (C.c_void_p * n)(*[s.data_ptr() for _, s in pairs])
    dst = (C.c_void_p * n)(*[d.data_ptr() for d, _ in pairs])
    nbytes = (C.c_int64 * n)(*[d.numel() * d.element_size() for d, _ in pairs])
    check(lib().gnan_multi_copy(n, src, dst, nbytes, stream_of(pairs[0][0])), "gnan_multi_copy")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_of(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def require_device(*tensors: Optional[torch.Tensor]) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise GnanHipError(
                "gnan_amd's kernels run on the MI355X only: got a CPU tensor. Move the module and its inputs to the GPU "
                "(`model.to('cuda')`, `data.to('cuda')`); there is deliberately no CPU fallback behind the HIP path "
                "(a module that lives on the CPU is evaluated by gnan_amd.cpu_route when it is called with CPU inputs).")
