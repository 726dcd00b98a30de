/*
 * gnan_hip.h — C ABI of libgnan_hip.so: the MI355X (gfx950) kernels behind GNAN's
 * distance-weighted additive aggregation (TensorGNAN.forward / GNAN.forward).
 *
 * The reference has no FFI layer for this path (it is stock ATen ops issued from Python);
 * each entry point below names the reference lines whose arithmetic it replaces.
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer into caller-owned memory (row-major, contiguous rows);
 *    the library never allocates or frees caller-visible memory; scratch is passed in;
 *  - every call enqueues work on `stream` and returns immediately (no device sync);
 *  - return value: 0 on success, negative gnan_status on failure; the message of the last
 *    failure on the calling thread is available through gnan_last_error();
 *  - no C++ exception crosses the boundary; no mutable global state.
 */
#ifndef GNAN_HIP_H
#define GNAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly the functions declared in this header are exported */
#pragma GCC visibility push(default)

#define GNAN_ABI_VERSION 45

typedef void* gnan_stream_t; /* hipStream_t */

enum gnan_status {
  GNAN_OK = 0,
  GNAN_ERR_BAD_ARG = -1,     /* null pointer, negative size, misaligned stride ... */
  GNAN_ERR_UNSUPPORTED = -2, /* shape outside what the kernels cover */
  GNAN_ERR_HIP = -3,         /* a HIP runtime call failed */
  GNAN_ERR_WORKSPACE = -4    /* workspace too small */
};

enum gnan_dtype { GNAN_F32 = 0, GNAN_BF16 = 1 };

/* Largest number of hop codes (shells incl. the rest bucket) a uint8 code can address. */
#define GNAN_MAX_CODES 256

int gnan_abi_version(void);
const char* gnan_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Per-feature shape functions  fx[n, k, :] = f_k(x[n, k])
 * replaces the Python feature loop GNAN.py:57-62 (= GNAN.py:150-155, models.py:360-365,
 * models.py:292-297), i.e. F x (L addmm + (L-1) relu + strided copy_).
 *
 * Weights are the reference's per-feature nn.Linear tensors stacked over the feature axis:
 *   L == 1 : w_last [F, C] (Linear(1, C).weight[:, 0]), b_last [F, C]
 *   L >= 2 : w_first [F, H], b_first [F, H]                       Linear(1, H)
 *            w_mid [L-2, F, H, H] (out, in), b_mid [L-2, F, H]    Linear(H, H)
 *            w_last [F, C, H], b_last [F, C]                      Linear(H, C)
 * Bias pointers may be NULL (bias=False).  ReLU after every layer but the last; training-mode Dropout
 * (GNAN.py:28,32) runs inside the lane kernel when dropout_p > 0 (see the struct), else none is applied.
 *
 * sum_features == 0 : out[n, k*C + c]               ("reference order", fx of GNAN.py:57)
 * sum_features == 1 : out[n, c] = sum_k fx[n, k, c]  (f_sums of GNAN.py:157)
 * ------------------------------------------------------------------------------------------- */
typedef struct gnan_fmlp_args {
  const float* x;        /* [n, F], row stride x_stride floats */
  int64_t n;
  int64_t x_stride;
  int32_t F, L, H, C;
  const float* w_first;
  const float* b_first;
  const float* w_mid;
  const float* b_mid;
  const float* w_last;
  const float* b_last;
  int32_t sum_features;
  float* out;            /* [n, out_stride] */
  int64_t out_stride;
  int32_t algo;          /* gnan_fmlp_algo; AUTO picks the matrix-core kernel when the shape allows, and for n * F <= 64
                            evaluations (L in {2, 3}, H <= 64, no Dropout) a wave per evaluation: no workspace then */
  void* workspace;       /* packed weights for the matrix-core kernel, 16-byte aligned */
  size_t workspace_bytes;
  float dropout_p;       /* > 0: training-mode Dropout behind every hidden ReLU (GNAN.py:28,32), kept units scaled by    */
  uint64_t dropout_seed; /* 1 / (1 - p); the mask is a counter-based hash of (seed, node, feature, layer, unit) —        */
                         /* gnan_dropout_mask writes it out.  Lane kernel only (algo AUTO picks it; MFMA is refused)     */
} gnan_fmlp_args;

enum gnan_fmlp_algo { GNAN_FMLP_AUTO = 0, GNAN_FMLP_LANE = 1, GNAN_FMLP_MFMA = 2 };

size_t gnan_fmlp_fwd_workspace_bytes(const gnan_fmlp_args* a);
int gnan_fmlp_fwd(const gnan_fmlp_args* a, gnan_stream_t stream);

/* Parameter gradients of the shape functions for small batches — what autograd computes behind GNAN.py:57-62 when
 * trainer.py:66 calls backward() on a graph of a few thousand nodes (the forward was gnan_fmlp_fwd):
 *   d_w_first[k, j] = sum_n dLoss/df_k(x[n,k]) . df_k/dw1_j   etc., one workgroup per feature, no atomics (reproducible).
 * grad is [n, C] with sum_features (every feature sees the same upstream gradient) or [n, F*C].  Covers L in {2, 3}
 * (L == 2: w_mid / b_mid / d_w_mid / d_b_mid are ignored), H <= 64, C <= 8; d_b_* must be NULL exactly where the bias is
 * NULL.  Large batches use gnan_fpwl_moments instead. */
typedef struct gnan_fmlp_bwd_args {
  const float* x;        /* [n, F], row stride x_stride floats */
  int64_t n;
  int64_t x_stride;
  int32_t F, L, H, C;
  const float* w_first;  /* [F, H] */
  const float* b_first;  /* [F, H] or NULL */
  const float* w_mid;    /* [F, H, H] */
  const float* b_mid;    /* [F, H] or NULL */
  const float* w_last;   /* [F, C, H] */
  const float* b_last;   /* [F, C] or NULL */
  int32_t sum_features;
  const float* grad;
  int64_t grad_stride;
  float* d_w_first;
  float* d_b_first;
  float* d_w_mid;
  float* d_b_mid;
  float* d_w_last;
  float* d_b_last;
  void* workspace;       /* gnan_fmlp_bwd_workspace_bytes(): partial gradients when a feature's nodes are cut into ranges
                            (few features, many nodes), added in range order */
  size_t workspace_bytes;
  float dropout_p;       /* the forward's Dropout (gnan_fmlp_args): the same masks are recomputed from the same seed */
  uint64_t dropout_seed;
} gnan_fmlp_bwd_args;

size_t gnan_fmlp_bwd_workspace_bytes(const gnan_fmlp_bwd_args* a);
int gnan_fmlp_bwd(const gnan_fmlp_bwd_args* a, gnan_stream_t stream);

/* The Dropout keep-mask gnan_fmlp_fwd / gnan_fmlp_bwd apply for (dropout_p, dropout_seed), written out:
 * mask[n, k, l, j] in {0, 1} (uint8, [n_nodes, F, n_hidden_layers, H]) for hidden layer l, unit j of feature k at node n.
 * For tests and for callers that evaluate a shape outside the kernels' coverage with the same masks. */
int gnan_dropout_mask(uint64_t seed, float p, int64_t n_nodes, int32_t F, int32_t n_hidden_layers, int32_t H, uint8_t* mask,
                      gnan_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Shape functions by exact piecewise-linear table look-up — same outputs as gnan_fmlp_fwd.
 * Each f_k (a ReLU MLP of a scalar, GNAN.py:24-34) is exactly piecewise linear; the caller tabulates it
 * from the current weights (gnan_amd/pwl.py) and this kernel evaluates
 *     f_k(x)[c] = val[i, c] + slope[i, c] * (x - anchor[i]),   i = off[k] + #{ j in 1..P_k : anchor[off[k]+j] <= x }
 * for every (node, feature): replaces GNAN.py:57-62 by N*F binary searches in LDS.
 *   off [F+1] int32 piece offsets (feature k has P_k + 1 = off[k+1]-off[k] pieces, P_k breakpoints
 *   anchor[off[k]+1 .. off[k]+P_k], ascending); anchor [T]; val, slope [T, C].
 *   max_pieces       = max_k (off[k+1]-off[k])
 *   features_per_group in {1,2,4,8,16}: consecutive features whose tables share one LDS image;
 *   max_group_pieces = max over groups of the pieces in the group (sizes the LDS image).
 * ------------------------------------------------------------------------------------------- */
typedef struct gnan_fpwl_args {
  const float* x;          /* [n, F] */
  int64_t n;
  int64_t x_stride;
  int32_t F, C;
  const int32_t* off;
  const float* anchor;
  const float* val;
  const float* slope;
  int32_t max_pieces;
  int32_t features_per_group;
  int32_t max_group_pieces;
  int32_t sum_features;    /* as gnan_fmlp_args */
  void* out;
  int64_t out_stride;      /* in elements */
  int32_t out_dtype;       /* gnan_dtype of `out`: GNAN_BF16 stores bf16 rows (per-feature mode, C == 1, whole groups) —
                              the operand format of the bf16-storage aggregation; `total` then sums the rounded values */
  float* total;            /* optional [F]: column sums of the per-feature output (the aggregation's s_total),
                              produced in the same pass; needs C == 1, whole groups, 16-B aligned rows */
  void* total_workspace;   /* gnan_fpwl_total_workspace_bytes() */
  size_t total_workspace_bytes;
  int64_t total_rows;      /* `total` covers rows [0, total_rows) only (a rank's owned rows ahead of its halo rows);
                              <= 0 or >= n: all rows */
  uint8_t* piece_out;      /* gnan_fpwl_fwd, optional [ceil(F / features_per_group), n, features_per_group] bytes (group-major, 4-byte
                              aligned): the piece of every (node, feature) within its feature, kept
                              for the backward pass.  Needs the fast feature-sum kernel (C == 1, sum_features, feature quads) and
                              max_pieces <= 256; GNAN_ERR_UNSUPPORTED otherwise (call again without it) */
  const uint8_t* piece_in; /* gnan_fpwl_moments_fixed, optional: those bytes — the C == 1 kernel then skips the search (ignored
                              by the other moment kernels and above 256 pieces per feature) */
  int32_t flags;             /* gnan_fpwl_flags: kernel selection switches (A/B measurements, tests); 0 = the library's choice */
  const uint16_t* index_table; /* gnan_fpwl_fwd, optional: the direct-index tables gnan_fpwl_index_build wrote for THESE tables   */
  const float* index_key;      /* ([F, buckets] entries and [F, 2] key coefficients).  With C == 1, whole 16-feature groups and   */
  int32_t index_buckets;       /* 16-byte aligned rows the look-up then finds a value's piece by arithmetic + one to three         */
                               /* comparisons instead of a search (csrc/fpwl_index.hip); same results bit for bit                  */
  void* sum_workspace;         /* gnan_fpwl_fwd, optional (ABI 42): gnan_fpwl_sum_workspace_bytes() bytes.  With it a feature-sum     */
  size_t sum_workspace_bytes;  /* look-up over several feature groups on a MEDIUM batch gives every (node block, group) a workgroup */
                               /* of its own and adds the groups' partial sums in group order afterwards (same bits as the walk    */
                               /* of all groups inside one workgroup, which leaves most of the chip idle below ~10^6 nodes)         */
  float* sum_total;            /* gnan_fpwl_fwd, optional (ABI 44), ONLY with sum_workspace in use (gnan_fpwl_sum_workspace_bytes() > 0, */
  void* sum_total_workspace;   /* C == 1): sum_total[0] = sum of out over rows [0, total_rows) out of the group-sum pass (float64 per  */
  size_t sum_total_workspace_bytes; /* workgroup, fixed order) — the rest bucket's operand without a gnan_colsum over the result;   */
                               /* workspace: ceil(n / 256) * 8 bytes, 8-byte aligned                                              */
  uint32_t* sum_total_arrive;  /* optional arrival counter (see gnan_moment_scales_args): sum_total out of the group-sum launch      */
} gnan_fpwl_args;

/* gnan_fpwl_args.flags (the library reads no environment variables: switches are the caller's, passed per call) */
enum gnan_fpwl_flags {
  GNAN_FPWL_MOMENTS_GENERAL = 1, /* gnan_fpwl_moments_fixed: the general kernel also where the one-channel kernel applies */
  GNAN_FPWL_LOCATE_SORTED = 2,   /* gnan_fpwl_locate: the sorted-array search also where the tree search applies */
  GNAN_FPWL_INDEX_HALF_LINES = 4, /* direct-index look-up: 16-feature groups (half lines of x per workgroup) where 32 would fit */
  GNAN_FPWL_INDEX_BS512 = 8,     /* direct-index look-up with 32-feature groups: 512-thread workgroups whatever the mode */
  GNAN_FPWL_INDEX_BS1024 = 16,   /* ... 1024-thread workgroups whatever the mode */
  GNAN_FPWL_ROWS_MOMENTS_LANE_PER_CHANNEL = 32 /* gnan_fpwl_rows_moments_fixed, 33..42 channels: a lane per channel (one node per step)
                                  * instead of a pair of channels per lane and three nodes per step (A/B; same bits) */
};

size_t gnan_fpwl_total_workspace_bytes(const gnan_fpwl_args* a);
size_t gnan_fpwl_sum_workspace_bytes(const gnan_fpwl_args* a);   /* 0: the call would not use one */
int gnan_fpwl_fwd(const gnan_fpwl_args* a, gnan_stream_t stream);

/* Direct-index acceleration of the one-channel look-up (csrc/fpwl_index.hip; no counterpart in the reference, whose
 * GNAN.py:57-62 evaluates the MLPs).  Per feature a uniform grid of `buckets` cells over [range[k][0], range[k][1]] —
 * the values the feature actually takes (gnan_feature_range: column minima / maxima, one pass per feature matrix) — maps
 * x to a cell by one fused multiply-add; table[k][cell] = 4 * (breakpoints of f_k below the cell) | code << 14 with
 * code 0 / 1 / 3 for at most one / two or three / more breakpoints inside the cell.  The range is a HINT: the look-up
 * compares x with the breakpoints inside its cell and searches code-3 cells (among them the two end cells, which take
 * everything outside the range), so every x is looked up exactly whatever the range says.  stats[k] (optional) = code-3
 * cells strictly inside the range.
 * gnan_feature_range: range [F][2] out; workspace F * 8 bytes; NaNs are skipped. */
typedef struct gnan_fpwl_index_args {
  const int32_t* off;      /* [F + 1] piece offsets (gnan_pwl_build / gnan_amd.pwl) */
  const float* anchor;     /* [T] */
  int32_t F;
  int32_t buckets;         /* 256, 512, 1024 or 2048 */
  const float* range;      /* [F][2] (lo, hi) */
  uint16_t* table;         /* [F][buckets] out, 16-byte aligned */
  float* key;              /* [F][2] out: cell(x) = (int) clamp(x * key[k][0] + key[k][1], 0, buckets - 1) */
  int32_t* stats;          /* optional [F] out */
} gnan_fpwl_index_args;
int gnan_fpwl_index_build(const gnan_fpwl_index_args* a, gnan_stream_t stream);
int gnan_feature_range(const float* x, int64_t n, int64_t x_stride, int32_t F, float* range, void* workspace,
                       size_t workspace_bytes, gnan_stream_t stream);

/* Backward of the table look-up (autograd through GNAN.py:57-62 w.r.t. the f_k parameters): per-piece
 * moments of the upstream gradient `grad` ([n, F*C], or [n, C] when sum_features),
 *   moments[t, 0, c] = sum_{(n,k) in piece t} grad[n,k,c]
 *   moments[t, 1, c] = sum_{(n,k) in piece t} grad[n,k,c] * (x[n,k] - anchor[t])
 * accumulated with atomics into `moments` [T, 2, C] (ZEROED by the caller).  On a piece both f_k and
 * d f_k / d theta are affine in x, so these moments determine the parameter gradient exactly
 * (gnan_amd/pwl.py finishes it on two points per piece).  `val`, `slope`, `out` of the args are unused. */
int gnan_fpwl_moments(const gnan_fpwl_args* a, const float* grad, int64_t grad_stride, float* moments,
                      gnan_stream_t stream);
/* The same moments in 64-bit fixed point: every term v is added as round(v * scales[m]) (m = 0 / 1 for the two
 * moments; scales: two doubles in DEVICE memory, powers of two chosen by the caller such that n * max|v| * scale
 * < 2^62 AND max|v| * scale < 2^51 — what gnan_fpwl_moment_scales produces: a term is converted by one fused
 * multiply-add onto 1.5 * 2^52, which is exact only below 2^51) into `moments` [T, 2, C] int64 (ZEROED by the caller);
 * the caller divides by the scales afterwards.
 * Integer LDS atomics run ~12x faster than float ones on gfx950 and the sums are bit-reproducible. */
int gnan_fpwl_moments_fixed(const gnan_fpwl_args* a, const float* grad, int64_t grad_stride, const double* scales,
                            int64_t* moments, gnan_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Table look-up with SEVERAL output channels in two phases (csrc/fpwl_rows.hip; node classification: C = classes).
 * The piece a value falls into does not depend on the channel, so it is found once per (node, feature) and the
 * channel work runs with lane = channel (coalesced table / gradient rows, no node-strided accesses):
 *   gnan_fpwl_locate            piece[n, k] = row of the stacked tables (off[k] + piece of x[n, k]), dx[n, k] = x[n, k] -
 *                               anchor[piece[n, k]]; both [n, F] row-major (gnan_fpwl_locate_bytes(a) bytes EACH); of the
 *                               args x, n, x_stride, F, off, anchor, max_pieces are read;
 *   gnan_fpwl_rows_fwd          out as gnan_fpwl_fwd (fp32; sum_features: [n, C], else [n, F*C]) from piece / dx and
 *                               val / slope; args read: n, F, C, val, slope, sum_features, out, out_stride, out_dtype;
 *   gnan_fpwl_rows_moments_fixed  the moments of gnan_fpwl_moments_fixed (same integers) from piece / dx and the upstream
 *                               gradient; args read: n, F, C, off, max_pieces, max_group_pieces, sum_features.  The
 *                               channels are cut into equal chunks of at most 64 whose bins — min(max_pieces,
 *                               max_group_pieces) * (2 chunk + 1) * 8 bytes per feature — fit 150 KiB of LDS.
 * ------------------------------------------------------------------------------------------- */
size_t gnan_fpwl_locate_bytes(const gnan_fpwl_args* a);
int gnan_fpwl_locate(const gnan_fpwl_args* a, int32_t* piece, float* dx, gnan_stream_t stream);
int gnan_fpwl_rows_fwd(const gnan_fpwl_args* a, const int32_t* piece, const float* dx, gnan_stream_t stream);
int gnan_fpwl_rows_moments_fixed(const gnan_fpwl_args* a, const int32_t* piece, const float* dx, const float* grad,
                                 int64_t grad_stride, const double* scales, int64_t* moments, gnan_stream_t stream);

/* The two scales of gnan_fpwl_moments_fixed, computed on the device:
 *   scales[0] = 2^floor(bits - log2(max|grad|)),  scales[1] = 2^floor(bits - log2(max|grad| * (x_abs_max + max|anchor|)))
 * (exponents clamped to +-1000) with grad [n, width] (row stride grad_stride), anchor [T] — of which only the first
 * *n_anchors are read when n_anchors (one int32 in DEVICE memory, e.g. &off[F] of tables held in a buffer of full
 * capacity) is not NULL —, x_abs_max one double in DEVICE memory (max |x| of the feature matrix) and
 * bits = 61 - ceil(log2 n), so that n terms cannot overflow 62 bits (values above 50 are treated as 50: the kernels
 * convert a term with one fused multiply-add onto 1.5 * 2^52, exact below 2^51).
 * workspace: GNAN_MOMENT_SCALES_WORKSPACE_BYTES bytes (a pair of maxima per workgroup of the pass; nothing to initialise).
 * zero / zero_bytes (optional): a buffer the same pass sets to zero — the int64 moment accumulators the following
 * gnan_fpwl_moments_fixed adds into (a separate fill launch otherwise).  One pass over grad + a one-workgroup kernel; no
 * host round trip. */
#define GNAN_MOMENT_SCALES_WORKSPACE_BYTES 8192
/* Arrival counters (ABI 45): several passes end in a sum over their workgroups' partial results.  With an `*_arrive` counter — 4 bytes
 * of device memory that are ZERO before the first launch that uses them; the pass leaves them zero — the last workgroup to finish
 * takes that sum itself, bit-identical to the second launch it saves (5 us of a 0.24-ms step); NULL keeps the two launches.  A
 * counter serves one pass at a time (launches on one stream may share it).  Passes of more than a few thousand workgroups ignore it. */
typedef struct gnan_moment_scales_args {
  const float* grad;         /* [n, width], row stride grad_stride */
  int64_t n;
  int32_t width;
  int32_t bits;              /* terms stay below 2^bits */
  int64_t grad_stride;
  const float* anchor;       /* [T] */
  int64_t T;
  const int32_t* n_anchors;  /* optional device pointer: only the first *n_anchors anchors are real */
  const double* x_abs_max;   /* device scalar max |x| */
  void* workspace;           /* >= GNAN_MOMENT_SCALES_WORKSPACE_BYTES bytes, 4-byte aligned */
  size_t workspace_bytes;
  double* scales;            /* [2] out */
  void* zero;                /* optional: zero_bytes bytes (8-byte aligned, a multiple of 8) set to zero by the same pass */
  size_t zero_bytes;
  uint32_t* arrive_counter;  /* optional arrival counter (see above): the scales out of the one launch */
} gnan_moment_scales_args;
int gnan_fpwl_moment_scales(const gnan_moment_scales_args* a, gnan_stream_t stream);

/* Parameter gradients of the shape functions from the per-piece moments — the last step of the table path's backward
 * pass (autograd through GNAN.py:57-62 w.r.t. every fs[k] parameter, trainer.py:66).  On a piece the network is affine
 * with a fixed activation pattern, so  d/dtheta sum_n <g_n, f_k(x_n)>  =  sum over pieces of
 * d/dtheta ( <M0, f(anchor)> + <M1, slope> ): one reverse pass for the value and one for the slope per piece, float64,
 * one workgroup per feature, no atomics (bit-reproducible).  Moments as gnan_fpwl_moments wrote them (`moments`) or in
 * fixed point (`moments_fixed` + the `scales` they were accumulated with); tables as gnan_pwl_build wrote them (only
 * off / anchor are read).  Weight and gradient layouts as gnan_fmlp_bwd_args.  Covers L in {2, 3}, H <= 64 (L == 3) /
 * 128 (L == 2), C <= 64. */
typedef struct gnan_fpwl_grad_args {
  const int32_t* off;           /* [F+1] */
  const float* anchor;          /* [T] */
  const float* moments;         /* [T, 2, C] float32, or NULL */
  const int64_t* moments_fixed; /* [T, 2, C] int64, or NULL */
  const double* scales;         /* [2], device memory (with moments_fixed) */
  const float* w_first;         /* [F, H] */
  const float* b_first;         /* [F, H] or NULL */
  const float* w_mid;           /* [F, H, H] (L == 3) */
  const float* b_mid;           /* [F, H] or NULL */
  const float* w_last;          /* [F, C, H] */
  const float* b_last;          /* [F, C] or NULL */
  int32_t F, L, H, C;
  int32_t max_pieces;           /* >= max_k (off[k+1] - off[k]); sizes the list of non-empty pieces in LDS */
  float* d_w_first;
  float* d_b_first;
  float* d_w_mid;
  float* d_b_mid;
  float* d_w_last;
  float* d_b_last;
} gnan_fpwl_grad_args;

int gnan_fpwl_param_grads(const gnan_fpwl_grad_args* a, gnan_stream_t stream);

/* Build the look-up tables on the device: one workgroup per feature finds the kinks of f_k (zero crossings
 * of its hidden pre-activations, float64) and tabulates the network at them.  Covers L in {2, 3}, H <= 128.
 * Outputs are the compact tables gnan_fpwl_fwd reads (features back to back, off[k] = first piece of feature
 * k, capacity F*(cap+1) pieces); *overflow is set if a feature has more than `cap` kinks (the caller then
 * falls back).  Weight layout as gnan_fmlp_args. */
typedef struct gnan_pwl_build_args {
  const float* w_first;   /* [F, H] */
  const float* b_first;   /* [F, H] or NULL */
  const float* w_mid;     /* [F, H, H] when L == 3 */
  const float* b_mid;     /* [F, H] or NULL */
  const float* w_last;    /* [F, C, H] */
  const float* b_last;    /* [F, C] or NULL */
  int32_t F, L, H, C;
  int32_t cap;            /* <= 1024 */
  float* anchor;          /* [F*(cap+1)]      compact */
  float* val;             /* [F*(cap+1), C] */
  float* slope;           /* [F*(cap+1), C] */
  int32_t* off;           /* [F+1] */
  int32_t* overflow;      /* [1]: written 0 / 1 by every build (nothing to initialise) */
  void* scratch;          /* gnan_pwl_build_scratch_bytes(F, C, cap) */
  size_t scratch_bytes;
  /* optional (ABI 45), all four or none: the direct-index tables of the look-up (gnan_fpwl_index_build) out of the build's own
   * compaction pass — one launch less per forward.  Same tables, bit for bit. */
  const float* index_range;  /* [F, 2] value range of every feature (gnan_feature_range) */
  uint16_t* index_table;     /* out [F, index_buckets] */
  float* index_key;          /* out [F, 2] */
  int32_t index_buckets;     /* 256, 512, 1024 or 2048 */
} gnan_pwl_build_args;

size_t gnan_pwl_build_scratch_bytes(int32_t F, int32_t C, int32_t cap);
int gnan_pwl_build(const gnan_pwl_build_args* a, gnan_stream_t stream);
/* The guard of a captured look-up (hipGraph replay: no read-back between build and look-up): `meta` = off[F + 1] | overflow as
 * gnan_pwl_build left them; flag[0] = 1.0f if the build overflowed, a feature has more than max_pieces pieces or a group of
 * features_per_group consecutive features more than max_group_pieces — i.e. the look-up sized with those numbers did not see the
 * whole tables.  Never cleared here: the caller zeroes it once per step, hands it to the optimizer update as its skip flag
 * (torch's found_inf) and reads it after the replay. */
int gnan_pwl_check_fit(const int32_t* meta, int32_t F, int32_t features_per_group, int32_t max_pieces,
                       int32_t max_group_pieces, float* flag, gnan_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Per-row weight table of the pre-rho normalisation:  lut[i, d, :] = rho(u[d] / max(cnt[i, d], 1))
 * replaces GNAN.py:65-67 — torch.div(node_distances, normalization_matrix) followed by the rho MLP on all N^2 pairs —
 * by D table look-ups per row: rho : R -> R^C is a ReLU MLP of a scalar, tabulated by gnan_pwl_build as a ONE-feature
 * table (F = 1: off = {0, T}); u[d] = float32(1 / (1 + d)), u[D-1] = 0 are the values node_distances takes
 * (pre_process_datasets.py:112-114), cnt the shell sizes (pre_process_datasets.py:136-140).  Same look-up arithmetic as
 * gnan_fpwl_fwd.  `arg` (optional) receives the arguments u[d] / cnt[i, d]: the backward pass bins the gradient of the
 * table by their pieces (gnan_fpwl_moments* with x = arg, F = 1, then gnan_fpwl_param_grads).
 * ------------------------------------------------------------------------------------------- */
typedef struct gnan_rho_lut_args {
  const int32_t* cnt;        /* [n_rows, cnt_stride] shell counts */
  int64_t cnt_stride;
  int64_t n_rows;
  int32_t D;                 /* hop codes incl. the rest bucket */
  int32_t C;                 /* output channels of rho */
  const float* u;            /* [D] */
  const float* anchor;       /* [T] tables of rho as gnan_pwl_build wrote them */
  const float* val;          /* [T, C] */
  const float* slope;        /* [T, C] */
  const int32_t* n_pieces;   /* optional device pointer to T (= off[1]) when the tables sit in a buffer of full capacity */
  int32_t max_pieces;        /* host-side bound on T (LDS bytes = 4 * max_pieces) */
  float* lut;                /* [n_rows, D, C] */
  float* arg;                /* optional [n_rows, D] */
} gnan_rho_lut_args;

int gnan_rho_row_lut(const gnan_rho_lut_args* a, gnan_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * rho(distance)-weighted neighbourhood sum over a hop-coded adjacency
 *
 *   Y[i, w] = sum_{e in row i} wt(i, code_e, w) * S[col_e, w]
 *           + wt(i, D-1, w) * ( s_total[w] - sum_{e in row i} S[col_e, w] )      (if s_total)
 *   wt(i, d, w) = lut[i*lut_row_stride + d*Cw + (w % Cw)] / max(cnt[i*cnt_stride + d], 1)
 *
 * replaces rho on all N^2 pairs + normalisation + bmm + feature sum:
 *   GNAN.py:65-73 / models.py:368-376 (TensorGNAN) and the per-node loop GNAN.py:159-170 /
 *   models.py:464-475 (GNAN).  It relies on node_distances being piecewise constant over
 *   hop shells (pre_process_datasets.py:112-121): rho is evaluated once per shell by the
 *   caller (`lut`), never per pair.
 *
 * Adjacency, two layouts:
 *   CSR   : rowptr (int32 or int64, n_rows+1), col int32, code uint8 (hop index < D-1)
 *   dense : rowptr == col == NULL; code is [n_rows, n_cols] row-major, neighbour j of row i
 *           is column j; code D-1 marks "unreachable" pairs (they still count as listed).
 * row_ids (optional) selects / reorders output rows: output row q aggregates adjacency row
 * row_ids[q] (GNAN.forward(node_ids), GNAN.py:159).  The adjacency row index is also the
 * index into cnt / a per-row lut.
 *
 * Normalisation orders:
 *   post-rho (models.py:369-370)  : global lut (lut_row_stride = 0) + cnt
 *   none                          : global lut, cnt = NULL
 *   pre-rho  (GNAN.py:65-67)      : per-row lut (lut_row_stride = D*Cw), cnt = NULL
 *
 * Transposed use (gradient w.r.t. S, i.e. autograd through GNAN.py:70): run the same kernel on the
 * transposed adjacency with weight_by_col = 1 (the weight table is indexed by the neighbour, which
 * is the forward pass's output row) and minus_rest = 1; s_total must be NULL.
 *
 * Rows longer than `long_threshold` edges are not touched by the main kernel; the caller lists
 * them in long_rows and the library splits each into slices of `slice_edges` edges reduced in a
 * fixed order (deterministic), using `workspace`.
 * ------------------------------------------------------------------------------------------- */
typedef struct gnan_spmm_args {
  int64_t n_rows;            /* output rows */
  int64_t n_cols;            /* rows of S (nodes that can be neighbours) */
  const void* rowptr;        /* NULL => dense layout */
  int32_t rowptr_is64;
  const int32_t* col;
  const uint8_t* code;
  const int32_t* row_ids;    /* optional [n_rows] */
  const void* S;             /* [n_cols, W] operand rows */
  int32_t s_dtype;           /* gnan_dtype */
  int32_t W;
  int64_t s_stride;          /* elements between operand rows */
  const float* lut;          /* [D, Cw] or [n_adj_rows, D, Cw] */
  int64_t lut_row_stride;    /* 0 => global table */
  int32_t D;                 /* number of codes incl. the rest bucket (index D-1) */
  int32_t Cw;                /* 1 or a divisor pattern of W: weight channel = w % Cw */
  const int32_t* cnt;        /* optional [n_adj_rows, cnt_stride] shell counts */
  int64_t cnt_stride;
  const float* s_total;      /* optional [W]: column sums of S; enables the rest-bucket term */
  int32_t weight_by_col;     /* transposed use: table row = neighbour col_e instead of output row */
  int32_t minus_rest;        /* transposed use: wt(., d) - wt(., D-1) per listed pair */
  int32_t reduce_cr;         /* fused read-out: 0 = store all W columns; c in {1,2,4} = store only
                                Y[q, c'] = sum over columns w = c' (mod c)  (the feature sum of GNAN.py:72-73) */
  int32_t scatter_out;       /* 1: rows are PROCESSED in row_ids order (a schedule, e.g. by degree, so that the
                                rows sharing a wavefront have similar lengths) but STORED at Y[row_ids[q]] —
                                row_ids must then be a permutation of the adjacency rows.
                                2: the adjacency (rowptr / col / code / cnt) is ITSELF stored in processing order (e.g. a
                                degree-sorted copy of the CSR): slot q reads adjacency row q and stores at Y[row_ids[q]].
                                Same result as 1 on the un-permuted adjacency, fewer scattered index requests */
  float* Y;                  /* [n_rows, W] fp32  ([n_rows, reduce_cr] with the fused read-out) */
  int64_t y_stride;
  /* long-row plan (CSR only; n_long == 0 => every row goes through the main kernel) */
  int64_t long_threshold;
  const int32_t* long_rows;      /* [n_long] output-row indices q of the long rows */
  const int32_t* long_slice_ptr; /* [n_long+1] prefix sum of slices per long row */
  int32_t n_long;
  int32_t n_slices;
  int32_t slice_edges;
  void* workspace;
  size_t workspace_bytes;
  int32_t s_by_code;             /* gnan_spmm_fwd only: S has n_cols * D rows and pair (i, c, d) reads S[c * D + d] — the
                                    backward w.r.t. a narrow S folds the per-pair weight into a pre-weighted operand
                                    Z[i, d, :] = (wt(i, d) - wt(i, D-1)) * dY[i, :] and gathers it with unit weights.
                                    CSR layout, fp32 rows, no s_total, W <= 32 (GNAN_ERR_UNSUPPORTED beyond) */
  int64_t nnz;                   /* listed pairs = readable length of col / code (CSR); 0 = unknown.  Known, the kernels read a
                                  * lane's run of index entries with wide loads (which may reach past the row end, never past nnz) */
  int32_t packed_index;          /* gnan_spmm_fwd only: every col entry is  column | hop code << 29  and `code` is not read (may be
                                  * NULL): one index stream instead of two.  CSR layout, n_cols <= 2^29, D <= 4, Cw == 1, no
                                  * weight_by_col / minus_rest / s_by_code (GNAN_ERR_UNSUPPORTED otherwise) */
  int64_t hot_lo;                /* gnan_spmm_fwd, narrow fp32 rows (W in {1, 2, 4}, packed index): operand rows [hot_lo, hot_lo +  */
  int32_t hot_rows;              /* hot_rows) are the most listed neighbours' rows, most listed first, and the column ids of their
                                  * pairs point there (a compact copy behind the operand).  hot_rows > 0 lets the library serve them
                                  * from LDS (persistent workgroups, hot_rows * W <= 32768 floats); 0 = off.  Same output bits */
  float* shell_out;              /* gnan_spmm_fwd, optional (ABI 45): [n_rows, D - 1] the row's RAW sums of S over its listed pairs, per
                                  * hop code (before any weight) — what a training forward keeps so that its backward is a pass over
                                  * rows (gnan_spmm_pack_z).  One-column operand (W == 1), global one-channel table with D <= 4, CSR,
                                  * no hub-row slices (n_slices == 0), no fused read-out, no hot rows; GNAN_ERR_UNSUPPORTED otherwise */
} gnan_spmm_args;

size_t gnan_spmm_fwd_workspace_bytes(const gnan_spmm_args* a);
int gnan_spmm_fwd(const gnan_spmm_args* a, gnan_stream_t stream);

/* The degree-sorted copy of a hop-coded CSR, the layout gnan_spmm_fwd walks with scatter_out = 2 (gnan_amd.graph.
 * HopGraph.degree_sorted_copy): order = the rows sorted by their number of listed pairs, ascending, ties by row id (a STABLE
 * radix sort); rowptr_s / col_s / code_s = the CSR with its rows in that order (pairs of a row in their original order);
 * cnt_s[q, :] = cnt[order[q], :] (optional: [n_rows, D] contiguous); colp_s = col_s | code_s << pack_shift (optional).  Index
 * work only, bit-exact.  workspace: gnan_degree_sorted_csr_workspace_bytes(n_rows), 256-byte aligned. */
typedef struct gnan_sorted_csr_args {
  int64_t n_rows;
  int64_t nnz;
  const void* rowptr;
  int32_t rowptr_is64;
  const int32_t* col;
  const uint8_t* code;
  const int32_t* cnt;        /* optional */
  int32_t D;
  int32_t pack_shift;        /* with colp_s */
  int32_t* order;            /* out [n_rows] */
  void* rowptr_s;            /* out [n_rows + 1], the width of rowptr */
  int32_t* col_s;            /* out [nnz] */
  uint8_t* code_s;           /* out [nnz] */
  int32_t* colp_s;           /* out [nnz], optional */
  int32_t* cnt_s;            /* out [n_rows, D], optional */
  void* workspace;
  size_t workspace_bytes;
} gnan_sorted_csr_args;

size_t gnan_degree_sorted_csr_workspace_bytes(int64_t n_rows);
int gnan_degree_sorted_csr(const gnan_sorted_csr_args* a, gnan_stream_t stream);

/* The hub-row plan of gnan_spmm_fwd (long_rows / long_slice_ptr of gnan_spmm_args; gnan_amd.graph.HopGraph.long_row_plan): the rows
 * with more than `threshold` pairs in ascending order and the prefix sums of their slice counts ceil(pairs / slice_edges).  Two calls
 * around ONE read-back: _count leaves the number of hub rows in total[0] (device memory), the caller sizes long_rows [n_long] and
 * slice_ptr [n_long + 1] and calls _fill with the SAME workspace (gnan_long_row_plan_workspace_bytes(n_rows), 256-byte aligned).
 * Index work only, bit-exact. */
size_t gnan_long_row_plan_workspace_bytes(int64_t n_rows);
int gnan_long_row_plan_count(const void* rowptr, int32_t rowptr_is64, int64_t n_rows, int64_t threshold, void* workspace,
                             size_t workspace_bytes, int32_t* total, gnan_stream_t stream);
int gnan_long_row_plan_fill(const void* rowptr, int32_t rowptr_is64, int64_t n_rows, int64_t threshold, int64_t slice_edges, int64_t n_long,
                            void* workspace, size_t workspace_bytes, int32_t* long_rows, int32_t* slice_ptr, gnan_stream_t stream);

/* The transposed adjacency of a hop-coded CSR (gnan_amd.graph.HopGraph.transposed: what the backward walks): row j of the result
 * lists the rows i that list neighbour j, in ascending i (a stable sort of the pairs by column id), with the pairs' hop codes;
 * rowptr_t has the width of rowptr.  long_rows: the rows with more than gnan_pb_plan_long_row_threshold() pairs.  Bit-exact.
 * workspace: gnan_csr_transpose_workspace_bytes(nnz, n_cols), 256-byte aligned. */
typedef struct gnan_csr_transpose_args {
  int64_t n_rows;
  int64_t n_cols;
  int64_t nnz;
  const void* rowptr;
  int32_t rowptr_is64;
  const int32_t* col;
  const uint8_t* code;
  const int32_t* long_rows;
  int32_t n_long;
  void* rowptr_t;            /* out [n_cols + 1] */
  int32_t* col_t;            /* out [nnz] */
  uint8_t* code_t;           /* out [nnz] */
  void* workspace;
  size_t workspace_bytes;
} gnan_csr_transpose_args;

size_t gnan_csr_transpose_workspace_bytes(int64_t nnz, int64_t n_cols);
int gnan_csr_transpose(const gnan_csr_transpose_args* a, gnan_stream_t stream);

/* The pair-level work of building the bucketed copy gnan_spmm_pb_fwd walks (gnan_amd.graph.HopGraph.pb_plan; the row-level
 * arrays — slots, bins — are n_rows-sized scans the caller makes).  Index work only, bit-exact:
 *   gnan_pb_plan_rows   c0[i] = pairs of row i with hop code 0; self_col[i] / self_pos[i] = column / position inside the row of the
 *                       last such pair (-1: none)
 *   gnan_pb_plan_keys   per listed pair e (rows in CSR order): key[e] = bin_of_row[i] * n_cb + col / cb_width (n_tiles for the pair
 *                       at self_pos[i], which is left out), val[e] = e, tmp_src[e] = col % cb_width, tmp_dst[e] = (slot of the pair
 *                       inside its bin) * n_acc + (code - code_base), the row's slots dealt round-robin over its kept pairs;
 *                       tile_cnt[key] += 1 (tile_cnt [n_tiles + 1] zeroed by the caller)
 *   gnan_pb_plan_fill   stable radix sort of (key, val); entry s of the sorted list goes to tile_ptr[key] + (s - tile_start[key])
 *                       of src16 / dst16 (tile_ptr: prefix of the tiles' sizes padded to 16, tile_start: of their real sizes);
 *                       pads get src 0 / dst `dummy`; chunk_q[chunk_first[cb * n_bins + b] + k] = tile_ptr[b * n_cb + cb] + 16 k.
 * workspace of the fill: gnan_pb_plan_fill_workspace_bytes(nnz, n_tiles), 256-byte aligned. */
typedef struct gnan_pb_keys_args {
  const void* rowptr;
  int32_t rowptr_is64;
  const int32_t* col;
  const uint8_t* code;
  int64_t n_rows;
  const int32_t* self_pos;   /* optional [n_rows] */
  int32_t code_base;
  int32_t n_acc;
  const int32_t* slot_ptr;   /* [n_rows + 1] */
  const int32_t* bin_of_row; /* [n_rows] */
  const int32_t* bin_slot0;  /* [n_bins] */
  int32_t n_cb;
  int32_t cb_width;
  uint32_t n_tiles;
  uint32_t* key;             /* out [nnz] */
  uint32_t* val;             /* out [nnz] */
  uint16_t* tmp_src;         /* out [nnz] */
  uint16_t* tmp_dst;         /* out [nnz] */
  uint32_t* tile_cnt;        /* in/out [n_tiles + 1] */
  const int32_t* long_rows;  /* [n_long]: every row with more than gnan_pb_plan_long_row_threshold() pairs (walked by a workgroup) */
  int32_t n_long;
} gnan_pb_keys_args;

typedef struct gnan_pb_fill_args {
  int64_t nnz;
  int64_t n_kept;            /* pairs that enter the tiles (nnz - those left out) */
  uint32_t n_tiles;
  int32_t n_bins;
  int32_t n_cb;
  int32_t dummy;             /* dst of pad entries */
  const uint32_t* key;
  const uint32_t* val;
  const uint16_t* tmp_src;
  const uint16_t* tmp_dst;
  const int32_t* tile_ptr;   /* [n_tiles + 1] */
  const int32_t* tile_start; /* [n_tiles + 1] */
  const uint32_t* tile_cnt;  /* [n_tiles + 1] */
  const int32_t* chunk_first;/* [n_tiles + 1], column-block-major */
  uint16_t* src16;           /* out [n_entries] */
  uint16_t* dst16;           /* out [n_entries] */
  int32_t* chunk_q;          /* out [n_chunks] */
  void* workspace;
  size_t workspace_bytes;
} gnan_pb_fill_args;

int32_t gnan_pb_plan_long_row_threshold(void);
int gnan_pb_plan_rows(const void* rowptr, int32_t rowptr_is64, const int32_t* col, const uint8_t* code, int64_t n_rows,
                      const int32_t* long_rows, int32_t n_long, int32_t* c0, int32_t* self_col, int32_t* self_pos, gnan_stream_t stream);
int gnan_pb_plan_keys(const gnan_pb_keys_args* a, gnan_stream_t stream);
size_t gnan_pb_plan_fill_workspace_bytes(int64_t nnz, uint32_t n_tiles);
int gnan_pb_plan_fill(const gnan_pb_fill_args* a, gnan_stream_t stream);

/* -------------------------------------------------------------------------------------------
 * gnan_spmm_pb_fwd — the same neighbourhood sum as gnan_spmm_fwd for NARROW fp32 operand rows (W in {1, 2, 4}: the
 * sum-first order of GNAN.py:157-170, S = f_sums), global weight table (Cw == 1, D <= 4), CSR layout, every output row, from
 * a bucketed copy of the adjacency the caller builds once per graph (gnan_amd.graph.HopGraph.pb_plan — index work only):
 *
 *   entries      the listed pairs that carry hop codes [code_base, code_base + n_acc), grouped into tiles
 *                (row bin, column block), bin-major, every tile padded to whole chunks of 16 entries;
 *                src[q] = operand row inside its column block (cb_width rows per block, cb_width * W * 4 <= 65536),
 *                dst[q] = accumulator inside its bin = (slot of the row inside the bin) * n_acc + (code - code_base);
 *                pad entries: src 0, dst acc_per_bin - 1 (a dummy the plan never assigns)
 *   chunk_q      entry offset of every chunk, COLUMN-BLOCK-major (cb_chunk_ptr [n_cblocks + 1] delimits the blocks)
 *   bins         bin b owns entries [bin_entry_ptr[b], bin_entry_ptr[b + 1]) and rows [bin_row_ptr[b], bin_row_ptr[b + 1]);
 *                row i owns the accumulator slots [slot_ptr[i], slot_ptr[i + 1]) (a hub row several: its entries are dealt
 *                over them), numbered from slot_ptr[first row of the bin]; at most (acc_per_bin - 1) / n_acc slots per
 *                bin, acc_per_bin * W * 8 <= 65536; bin_order lists the bins in launch order (largest first)
 *   self_col     optional [n_rows]: the operand row of the row's ONE pair with hop code 0 (-1: none) when those pairs are
 *                left out of the entries (code_base = 1)
 *   headroom_bits  ceil(log2(most entries any single output row receives)): sizes the 64-bit fixed point
 *
 * Y[i, :] = sum_d (wt(i,d) - wt(i,D-1)) * sum_{pairs of row i with code d} S[col, :]  +  wt(i,D-1) * s_total
 * (wt(i,d) = lut[d] / max(cnt[i,d], 1); without s_total the rest weight is zero), i.e. gnan_spmm_fwd's result with the
 * per-code sums exact to 2^-40 of max |S| (integer accumulation: order-independent, bit-reproducible) instead of a
 * float32 chain.  No listed pair costs a memory request of its own: see csrc/spmm_pb.hip.
 * workspace: gnan_spmm_pb_workspace_bytes(a) bytes, 16-byte aligned (the expanded operand, n_entries * W floats).
 * A non-finite operand value yields NaN in every output row.
 * ------------------------------------------------------------------------------------------- */
enum gnan_pb_flags {
  GNAN_PB_EXPAND_PER_ITERATION = 1,  /* phase 1 reads a chunk's offset inside every wave-iteration (default: 64 offsets per load) */
  GNAN_PB_FIXED_VIA_DOUBLE = 2,      /* phase 2 converts v * 2^shift through float64 (default: from the float's mantissa / exponent) */
  GNAN_PB_REDUCE_UNROLL4 = 4         /* phase 2 keeps four rounds of loads in flight per thread (default: two) */
};
typedef struct gnan_spmm_pb_args {
  int64_t n_rows;
  int64_t n_cols;
  const float* S;            /* [n_cols, W] fp32, contiguous rows */
  int64_t s_stride;          /* == W */
  int32_t W;
  int32_t D;                 /* hop codes incl. the rest bucket */
  const float* lut;          /* [D] */
  const int32_t* cnt;        /* optional [n_rows, cnt_stride] */
  int64_t cnt_stride;
  const float* s_total;      /* optional [W] */
  float* Y;                  /* [n_rows, y_stride] */
  int64_t y_stride;
  int64_t n_entries;
  const uint16_t* src;       /* [n_entries] */
  const uint16_t* dst;       /* [n_entries], 8-byte aligned */
  int32_t cb_width;
  int32_t n_cblocks;
  const int32_t* chunk_q;
  const int32_t* cb_chunk_ptr;
  int32_t n_bins;
  int32_t acc_per_bin;
  const int32_t* bin_order;
  const int32_t* bin_entry_ptr;
  const int32_t* bin_row_ptr;
  const int32_t* slot_ptr;   /* [n_rows + 1] */
  int32_t n_acc;
  int32_t code_base;
  const int32_t* self_col;
  int32_t headroom_bits;
  void* workspace;
  size_t workspace_bytes;
  int32_t flags;             /* A/B switches (enum gnan_pb_flags): 0 = the defaults; bits 8-15 = workgroups per column block */
  int32_t self_is_row;       /* the hop-code-0 pair of EVERY row i lists operand row i (a hop-coded graph's self pairs, n_cols >=
                                n_rows): self_col is not read and may be NULL */
  /* optional, W == 1 (ABI 45) — what lets the SAME two phases serve the one-column backward without moving a second column
   * (gnan_spmm_pb_pack1 below): */
  float* shell_out;          /* [n_rows]: the row's raw sum over its listed pairs of the accumulated hop code (before any weight) —
                                kept by a training forward: the table gradient is then a sum over ROWS, not over pairs */
  const float* S_self;       /* [n_cols]: the operand of the hop-code-0 (self) pairs, when it is not S */
  const float* out_add;      /* [1], with out_add_scale [1]: out_add[0] * out_add_scale[0] is added to every output row */
  const float* out_add_scale;
} gnan_spmm_pb_args;

size_t gnan_spmm_pb_workspace_bytes(const gnan_spmm_pb_args* a);
int gnan_spmm_pb_fwd(const gnan_spmm_pb_args* a, gnan_stream_t stream);

/* Backward of the ONE-column aggregation through the same machinery, over the bucketed copy of the TRANSPOSED adjacency
 * (row j = operand node j, entries = the output rows i that list j): both gradients from one pass, like
 * gnan_spmm_bwd_narrow.  `pb` describes the transposed graph's plan for W = 2 with ONE accumulated hop code d1 = code_base
 * (n_acc == 1: the 1-hop-truncated graphs of configs 3-5); pb.S = V[d1], the packed rows [dY_i / cnt(i, d1) | dY_i / cnt(i, D-1)]
 * of gnan_spmm_pack_bwd_rows (half = 1) — pb.cnt / pb.s_total / pb.Y are not read; v_self = V[0] (the rows of hop code 0) when
 * the plan serves the self pairs from self_col.  With t1[j], tr[j] the sums of the two halves over j's entries (+ the self
 * pair's rest half in tr):
 *   dS[j]    = lut[0] * V[0][self].x + lut[d1] * t1 - lut[D-1] * tr  (+ ds_add[0] * ds_add_scale[0])
 *   dlut[d1] = sum_j s_rows[j] * t1[j],  dlut[0] = sum_j s_rows[j] * V[0][self].x,
 *   dlut[D-1] = - sum_j s_rows[j] * tr[j] + rest_total[0] * rest_q[0]        (with_rest; other entries 0)
 * The sums over an operand node's entries are exact to 2^-40 of max |V| (integer accumulation); the table gradient adds
 * float64 partials per bin in bin order: bit-reproducible.  workspace: gnan_spmm_pb_bwd_workspace_bytes(g), 16-byte aligned. */
typedef struct gnan_spmm_pb_bwd_args {
  gnan_spmm_pb_args pb;
  const float* v_self;       /* optional [n_fwd_rows, 2] */
  const float* s_rows;       /* [pb.n_rows] forward operand, one column */
  int64_t s_rows_stride;
  int32_t with_rest;
  float* dS;                 /* [pb.n_rows] */
  int64_t ds_stride;
  float* dlut;               /* [pb.D] */
  const float* ds_add;       /* optional [1] */
  const float* ds_add_scale; /* optional [1] */
  const float* rest_total;   /* optional [1] */
  const float* rest_q;       /* optional [1] */
} gnan_spmm_pb_bwd_args;

size_t gnan_spmm_pb_bwd_workspace_bytes(const gnan_spmm_pb_bwd_args* g);
int gnan_spmm_pb_bwd(const gnan_spmm_pb_bwd_args* g, gnan_stream_t stream);

/* The one-column backward WITHOUT a second column (round 6; 0.74 -> 0.45 ms on the 10M-node graph).  With the forward's shell sums
 * kept (gnan_spmm_pb_args.shell_out: shell[i] = sum of S over row i's listed pairs of hop code d1), every term of the table
 * gradient is a sum over ROWS, and the operand gradient needs ONE number per row — the two per-pair factors of gnan_spmm_pb_bwd,
 * dY_i / cnt(i, d1) and dY_i / cnt(i, D-1), enter dS only through  c_i = lut[d1] a1_i - lut[D-1] ar_i:
 *   a0 = dY_i / max(cnt(i, 0), 1), a1 = dY_i / max(cnt(i, d1), 1), ar = with_rest ? dY_i / max(cnt(i, D-1), 1) : 0  (cnt NULL: 1)
 *   c[i] = lut[d1] a1 - lut[D-1] ar          e[i] = lut[0] a0 - lut[D-1] ar  (the self pair's share; 0 for a row without one)
 *   q    = sum_i ar                          dlut[0] = sum_i self_i a0,   dlut[d1] = sum_i shell[i] a1
 *   dlut[D-1] = - sum_i (self_i + shell[i]) ar + s_total[0] q            (self_i = S[self column of row i], 0 without one)
 * float64 per workgroup, fixed order.  Then  dS = gnan_spmm_pb_fwd over the TRANSPOSED adjacency's W = 1 plan with S = c, S_self = e,
 * lut = ones, cnt = NULL, s_total = NULL, out_add = q, out_add_scale = &lut[D-1]:  dS[j] = sum_{i lists j} c_i + e_j + lut[D-1] q.
 * workspace: gnan_spmm_pb_pack1_workspace_bytes(n) bytes, 8-byte aligned, nothing to initialise. */
typedef struct gnan_pb_pack1_args {
  int64_t n;                 /* rows of the forward */
  const float* dY;           /* [n], stride dy_stride */
  int64_t dy_stride;
  const int32_t* cnt;        /* optional [n, cnt_stride] */
  int64_t cnt_stride;
  int32_t D, d1, with_rest;
  int32_t self_is_row;       /* the forward plan's: row i's self pair lists operand row i */
  const int32_t* self_col;   /* the forward plan's self column per row (-1: none), or NULL (no self pairs unless self_is_row) */
  const float* lut;          /* [D] */
  const float* S;            /* forward operand, one column, contiguous */
  const float* shell;        /* [n] */
  const float* s_total;      /* optional [1] (with_rest) */
  float* c;                  /* out [n] */
  float* e;                  /* out [n] */
  float* q;                  /* out [1] */
  float* dlut;               /* out [D] */
  void* workspace;
  size_t workspace_bytes;
} gnan_pb_pack1_args;

/* The same row pass for the row-parallel kernels (graphs the bucketed route declines; any D <= 4): from the forward's per-code shell
 * sums T [n, D - 1] (gnan_spmm_args.shell_out) and dY [n]:
 *   a_d = dY_i / max(cnt(i, d), 1) (cnt NULL: dY_i),  ar = with_rest ? a_{D-1} : 0
 *   Z[i, d] = lut[d] a_d - lut[D-1] ar  (d < D - 1),  Z[i, D-1] = 0         q = sum_i ar
 *   dlut[d] = sum_i a_d T[i, d]  (d < D - 1),          dlut[D-1] = - sum_i ar sum_d T[i, d] + s_total[0] q   (with_rest, else 0)
 * Then dS = gnan_spmm_fwd over the transposed adjacency with S = Z ([n * D] rows, s_by_code = 1), a table of ones, no counts, plus
 * lut[D-1] q on every row.  float64 per workgroup, fixed order.  workspace: gnan_spmm_pack_z_workspace_bytes(n), 16-byte aligned. */
typedef struct gnan_pack_z_args {
  int64_t n;
  const float* dY;           /* [n], stride dy_stride */
  int64_t dy_stride;
  const int32_t* cnt;        /* optional [n, cnt_stride] */
  int64_t cnt_stride;
  int32_t D, with_rest;
  const float* lut;          /* [D] */
  const float* shell;        /* [n, D - 1] */
  const float* s_total;      /* optional [1] */
  float* Z;                  /* out [n, D] */
  float* q;                  /* out [1] */
  float* dlut;               /* out [D] */
  void* workspace;
  size_t workspace_bytes;
} gnan_pack_z_args;
size_t gnan_spmm_pack_z_workspace_bytes(int64_t n);
int gnan_spmm_pack_z(const gnan_pack_z_args* a, gnan_stream_t stream);

size_t gnan_spmm_pb_pack1_workspace_bytes(int64_t n);
int gnan_spmm_pb_pack1(const gnan_pb_pack1_args* a, gnan_stream_t stream);

/* Shell sums for the backward pass (autograd through GNAN.py:67-70 w.r.t. rho's parameters):
 *   T[q, d, w] = sum_{e in row, code_e == d} S[col_e, w]            d < D-1
 *   T[q, D-1, w] (+)= s_total[w] - sum_{e in row} S[col_e, w]       (if s_total; dense layout
 *                                                                    accumulates rest-coded pairs)
 * Same arguments as gnan_spmm_fwd; `Y` is T [n_rows, D, W] fp32, ZEROED by the caller; lut, cnt,
 * y_stride and the long-row plan are ignored.  The gradient of the weight table is then
 *   dwt[q, d, c] = sum_{w % Cw == c} dY[q, w] * T[q, d, w]. */
int gnan_spmm_shell_sums(const gnan_spmm_args* a, gnan_stream_t stream);

/* Column sums total[w] = sum_j S[j, w] of the operand (the `s_total` argument of gnan_spmm_fwd):
 * one streaming pass, float64 across threads, fixed order.  Workspace: gnan_colsum_workspace_bytes(W). */
size_t gnan_colsum_workspace_bytes(int32_t W);
int gnan_colsum(const float* S, int64_t n, int32_t W, int64_t stride, float* total, void* workspace,
                size_t workspace_bytes, gnan_stream_t stream);
/* Y[q, c] += (rest-bucket weight of row i_q) * (the column sums read-out channel c collects): the term of gnan_spmm_fwd
 * that depends on the operand only through `s_total` — gnan_spmm_fwd(s_total = total) == gnan_spmm_fwd(s_total = zeros) followed
 * by this, up to rounding — so that a multi-rank forward can aggregate while the all-reduce of the column sums is in flight.
 * lut [D, Cw] (lut_row_stride == 0) or per row [n, D, Cw]; cnt optional [n, cnt_stride] shell sizes; row_ids optional [n]
 * (the rows of lut / cnt the outputs belong to); reduce_cr as in gnan_spmm_args (0: Y has W columns). */
typedef struct gnan_rest_term_args {
  float* Y;
  int64_t y_stride;
  int64_t n;
  const float* total;        /* [W] */
  int32_t W;
  const float* lut;
  int64_t lut_row_stride;
  int32_t D, Cw;
  const int32_t* cnt;
  int64_t cnt_stride;
  const int32_t* row_ids;
  int32_t reduce_cr;
} gnan_rest_term_args;
int gnan_rest_term_add(const gnan_rest_term_args* a, gnan_stream_t stream);
/* Per-graph read-out of a batch of graphs: out[g, :] = sum of the rows node_off[g] .. node_off[g + 1] - 1 of Y [N, C]
 * (batched_pyg_main.py:173-181; C <= 64).  One wave per graph, fixed order. */
int gnan_segment_sum(const float* Y, int64_t y_stride, int32_t C, const int32_t* node_off, int32_t n_graphs, float* out,
                     gnan_stream_t stream);
/* Feature sum of kept per-feature rows: out[i, c] = sum_k fx[i, k * C + c] (GNAN.py:157 applied to the rows of
 * models.py:360-365; C in {1, 2, 4}, W = F * C with W % 4 == 0, 16-byte aligned rows).  One streaming pass. */
int gnan_feature_sum(const float* fx, int64_t n, int32_t W, int64_t stride, int32_t C, float* out, int64_t out_stride,
                     gnan_stream_t stream);
/* dst[k, :] = src[ids[k], :] (W fp32 columns, dst rows contiguous; ids int64 in device memory): the compact second copy of
 * the most listed nodes' operand rows behind the operand, which gnan_spmm_args.hot_lo / hot_rows describe. */
int gnan_gather_rows(const float* src, int64_t src_stride, const int64_t* ids, int64_t k, int32_t W, float* dst,
                     gnan_stream_t stream);

/* -------------------------------------------------------------------------------------------
 * The loss step of the epoch loops (trainer.py:53-71 / :125-147) in one launch (two past 512 rows, one more to zero the
 * gradient rows a mask leaves out): rows index[0..n) of the logits (all rows 0..n-1 without index),
 *   GNAN_LOSS_BCE_LOGITS   (C == 1, labels float32 [n]): mean_i (1 - t_i) x_i - logsigmoid(x_i)   = nn.BCEWithLogitsLoss()
 *   GNAN_LOSS_CROSS_ENTROPY (C >= 2, labels int64 [n]):  mean_i logsumexp(x_i) - x_i[t_i]          = nn.CrossEntropyLoss()
 * (default options: mean reduction, no weights, no label smoothing; every label in [0, C): rows torch would IGNORE
 * (ignore_index) are not supported — the caller keeps its own loss for such labels), float32 terms as torch forms them,
 * float64 across rows in a fixed order.  Optional outputs: hits = #{(sigmoid(x_i) > 0.5) == t_i} or #{argmax x_i == t_i}
 * (trainer.py:5-20); grad [n_rows, C] = d loss / d logits (rows outside index: zero; index entries must be distinct);
 * loss_sum += loss and hits_sum += hits (the epoch's running totals, trainer.py:67-71, device scalars).
 * Replaces the 20-25 element-wise / reduction launches stock torch needs for the same numbers.
 * ------------------------------------------------------------------------------------------- */
enum gnan_loss_kind { GNAN_LOSS_BCE_LOGITS = 0, GNAN_LOSS_CROSS_ENTROPY = 1 };
typedef struct gnan_loss_args {
  const float* logits;       /* [n_rows, C], row stride `stride` */
  int64_t n_rows;
  int32_t C;
  int32_t kind;              /* enum gnan_loss_kind */
  int64_t stride;
  const int64_t* index;      /* optional [n] rows that count (device memory) */
  int64_t n;                 /* >= 1 */
  const void* labels;        /* [n] float32 (BCE) / int64 (cross entropy), aligned with index */
  float* loss;               /* [1] */
  int64_t* hits;             /* optional [1] */
  float* grad;               /* optional [n_rows, C], row stride grad_stride */
  int64_t grad_stride;
  float* loss_sum;           /* optional [1] accumulators */
  float* hits_sum;
  const float* skip_sums;    /* optional [1] device flag: non-zero = do not touch loss_sum / hits_sum (a captured step whose
                                guard tripped is re-run eagerly and counted then) */
  void* workspace;           /* gnan_loss_workspace_bytes(n) */
  size_t workspace_bytes;
  float* label_flag;         /* optional [1] device flag (ABI 42): set to 1 by a cross-entropy row whose class label lies
                                outside [0, C) — torch leaves ignore_index rows out of the mean, this kernel averages over all n
                                rows: a caller that cannot look at the labels on the host (a replayed step) reads the flag later */
} gnan_loss_args;
size_t gnan_loss_workspace_bytes(int64_t n);
int gnan_loss_step(const gnan_loss_args* a, gnan_stream_t stream);

/* -------------------------------------------------------------------------------------------
 * The whole forward of a SMALL dense-coded graph in one launch (GNAN.py:146-172 / models.py:358-384 with the post-rho
 * normalisation of models.py:368-370; graph-level tasks feed one ~30-node graph per step, trainer.py:23-86):
 *   S[j, :]   = sum_k f_k(x[j, k])                                   (features added in order k = 0 .. F-1)
 *   lut[d, :] = rho(u_d),  u_d = float32(1 / (1 + d)) for d < D-1, u_{D-1} = 0
 *   Y[i, c]   = sum_j lut[code[i, j], c or 0] / max(cnt[i, code[i, j]], 1) * S[j, c]       (cnt == NULL: no division)
 *   Ysum[c]   = sum_i Y[i, c]                                          (the graph read-out, GNAN.py:75-79)
 * One workgroup per feature + one for rho; the last to finish aggregates (a counter in the workspace).  Covers n <= 128,
 * D <= 256, L in {2, 3}, H <= 64, C <= 8, rho.C in {1, f.C}; GNAN_ERR_UNSUPPORTED otherwise (gnan_fmlp_fwd + gnan_spmm_fwd
 * compute the same).  S and lut are outputs too: gnan_spmm_fwd on the transposed codes, gnan_spmm_lut_grad and gnan_fmlp_bwd
 * take them for the backward pass.  The MLP weights are stacked as in gnan_fmlp_args (rho: one "feature").
 * workspace: gnan_small_graph_workspace_bytes(n, F, f.C) bytes whose first 4 are ZERO before the first launch (the kernel
 * leaves them zero); one workspace per stream.
 * ------------------------------------------------------------------------------------------- */
typedef struct gnan_small_mlp {
  int32_t L, H, C;
  const float* w_first;      /* [F, H] */
  const float* b_first;      /* [F, H] or NULL */
  const float* w_mid;        /* [F, H, H] (L == 3) */
  const float* b_mid;        /* [F, H] or NULL */
  const float* w_last;       /* [F, C, H] */
  const float* b_last;       /* [F, C] or NULL */
} gnan_small_mlp;
typedef struct gnan_small_graph_args {
  const float* x;            /* [n, F], row stride x_stride */
  int64_t x_stride;
  int32_t n, F;
  gnan_small_mlp f;          /* the F shape functions */
  gnan_small_mlp rho;        /* one scalar MLP */
  const uint8_t* code;       /* [n, n] hop codes (gnan_dense_to_code) */
  int32_t D;
  int32_t pre_rho;           /* != 0: the pre-rho normalisation of GNAN.py:65-67 — weight of pair (i, j) = rho(u_d / max(cnt[i, d], 1)),
                                d = code[i, j]: rho at n * D arguments (64 per extra workgroup); needs cnt, rho.C == 1, D <= 64 */
  const int32_t* cnt;        /* optional [n, cnt_stride] shell sizes */
  int64_t cnt_stride;
  float* S;                  /* [n, f.C] */
  float* lut;                /* [D, rho.C]; with pre_rho the rows' table [n, D] */
  float* Y;                  /* optional [n, f.C] */
  float* Ysum;               /* optional [f.C] */
  void* workspace;
  size_t workspace_bytes;
} gnan_small_graph_args;
size_t gnan_small_graph_workspace_bytes(int32_t n, int32_t F, int32_t C);
int gnan_small_graph_fwd(const gnan_small_graph_args* a, gnan_stream_t stream);

/* MANY small graphs in one launch — the batched variant of the reference (batched_pyg_main.py:133-184): graph g owns nodes
 * node_off[g] .. node_off[g + 1] - 1 of x / S / Y and the [n_g, n_g] block of hop codes at code + code_off[g] (what the
 * reference's collate function lays out as one block-diagonal matrix, batched_pyg_main.py:54-91); Ysum [n_graphs, f.C] is the
 * per-graph read-out (the scatter_add_ of batched_pyg_main.py:173-181), lut [n_graphs, D, rho.C] a rho table per graph.
 * rho_raw_hops: rho is evaluated on the raw hop count d (batched_pyg_main.py:151) instead of 1 / (1 + d); rest_zero: codes
 * beyond D - 2 (the -1 mask of :155-156) carry weight 0.  No shell normalisation.  Covers graphs of <= 128 nodes (max_nodes).
 * workspace: gnan_small_batch_workspace_bytes(...) bytes whose first 128 * n_graphs are ZERO before the first launch (the
 * kernel leaves them zero).  gnan_hops_to_code / gnan_dense_blocks_to_code make the packed codes from the hop matrices:
 * status[0] |= 1 for an entry that is no integer in [0, 254] or negative, |= 2 for a listed pair outside the diagonal blocks
 * (dense form), status[1] = the largest hop (ZEROED by the caller). */
typedef struct gnan_small_batch_args {
  const float* x;            /* [total_nodes, F], row stride x_stride */
  int64_t x_stride;
  int64_t total_nodes;
  int32_t F;
  int32_t n_graphs;
  int32_t max_nodes;         /* largest graph */
  gnan_small_mlp f;
  gnan_small_mlp rho;
  const uint8_t* code;       /* packed [n_g, n_g] blocks */
  const int32_t* node_off;   /* [n_graphs + 1] */
  const int64_t* code_off;   /* [n_graphs + 1] */
  int32_t D;
  int32_t rho_raw_hops;
  int32_t rest_zero;
  float* S;                  /* [total_nodes, f.C] */
  float* lut;                /* [n_graphs, D, rho.C] */
  float* Y;                  /* optional [total_nodes, f.C] */
  float* Ysum;               /* optional [n_graphs, f.C] */
  void* workspace;
  size_t workspace_bytes;
  const int32_t* cnt;        /* optional [total_nodes, cnt_stride] shell sizes (ABI 42): pair (i, j) of a graph then weighs
                                lut[d] / max(cnt[i, d], 1) as in gnan_small_graph_fwd — the post-rho normalisation of
                                models.py:368-370 for a "batch" whose graphs (often just one: trainer.py's batch_size = 1) sit
                                in slots whose sizes only the device knows; needs D <= 64 and a one-channel rho */
  int64_t cnt_stride;        /* >= D */
} gnan_small_batch_args;
size_t gnan_small_batch_workspace_bytes(int32_t n_graphs, int64_t total_nodes, int32_t F, int32_t C);
int gnan_small_batch_fwd(const gnan_small_batch_args* a, gnan_stream_t stream);
int gnan_hops_to_code(const float* hops, int64_t count, uint8_t* code, int32_t* status, gnan_stream_t stream);
int gnan_dense_blocks_to_code(const float* dist, int64_t stride, int64_t n, const int32_t* graph_of, const int32_t* node_off,
                              const int64_t* code_off, uint8_t* code, int32_t* status, gnan_stream_t stream);

/* ... and its backward pass in one launch: the gradients of every stacked parameter tensor of f and rho (autograd through
 * GNAN.py:146-172 / models.py:358-384 + trainer.py:66) from dY [n, f.C] — or dYsum [f.C], the gradient of the graph read-out,
 * which every row shares — and the S and lut the forward left behind.  Workgroup k < F forms the operand gradient
 * dS[j, :] = sum_i lut[code[i, j]] / max(cnt[i, code[i, j]], 1) * dY[i, :] itself and runs gnan_fmlp_bwd's body on feature k;
 * workgroup F forms dlut[d] = sum_i 1 / max(cnt[i, d], 1) * sum_{j : code[i, j] == d} <dY[i, :], S[j, :]> and runs it on rho.
 * Covers n <= 128, D <= 64, a one-channel rho, L in {2, 3}, H <= 64, C <= 8; GNAN_ERR_UNSUPPORTED otherwise (the general
 * kernels then: gnan_spmm_fwd on the transposed codes, gnan_spmm_lut_grad, gnan_fmlp_bwd twice).  No atomics on data; a workspace
 * only to split the pre-rho variant's rho work (below). */
typedef struct gnan_small_mlp_grads {
  float* w_first;
  float* b_first;            /* NULL exactly where the MLP has no such bias */
  float* w_mid;
  float* b_mid;
  float* w_last;
  float* b_last;
} gnan_small_mlp_grads;
typedef struct gnan_small_graph_bwd_args {
  const float* x;
  int64_t x_stride;
  int32_t n, F;
  gnan_small_mlp f;
  gnan_small_mlp rho;
  const uint8_t* code;
  int32_t D;
  int32_t pre_rho;           /* as in gnan_small_graph_args: lut is [n, D], rho's gradients come from its n * D arguments */
  const int32_t* cnt;
  int64_t cnt_stride;
  const float* S;            /* [n, f.C] node sums of the forward */
  const float* lut;          /* [D] rho table of the forward ([n, D] with pre_rho) */
  const float* dY;           /* [n, f.C] or NULL */
  const float* dYsum;        /* [f.C] or NULL (used when dY is NULL) */
  gnan_small_mlp_grads df;
  gnan_small_mlp_grads drho;
  void* workspace;           /* optional (pre_rho): gnan_small_graph_bwd_workspace_bytes(n, D) bytes whose first 16 are ZERO before the
                                first launch (the kernel leaves them zero) — rho's n * D arguments are then split over up to 12
                                workgroups whose partial gradients the last one adds in order; without it one workgroup walks them all */
  size_t workspace_bytes;
} gnan_small_graph_bwd_args;
size_t gnan_small_graph_bwd_workspace_bytes(int32_t n, int32_t D);
int gnan_small_graph_bwd(const gnan_small_graph_bwd_args* a, gnan_stream_t stream);

/* The backward pass of gnan_small_batch_fwd: launch 1 — blockIdx.y = graph, the workgroups of gnan_small_graph_bwd per graph — leaves
 * every graph's contribution to the gradients of f and rho in its slab of the workspace; launch 2 adds the slabs in graph
 * order into df / drho.  dY [total_nodes, f.C], or dYsum [n_graphs, f.C] (the gradient of the per-graph read-out).  Covers
 * graphs of <= 128 nodes, D <= 64, a rho of one channel or one per output channel (lut [n_graphs, D, rho.C]), L in {2, 3},
 * H <= 64, C <= 8; GNAN_ERR_UNSUPPORTED otherwise (the
 * general kernels on the blocks' CSR then).  workspace: gnan_small_batch_bwd_workspace_bytes(args) bytes, no initialisation. */
typedef struct gnan_small_batch_bwd_args {
  const float* x;
  int64_t x_stride;
  int64_t total_nodes;
  int32_t F;
  int32_t n_graphs;
  int32_t max_nodes;
  gnan_small_mlp f;
  gnan_small_mlp rho;
  const uint8_t* code;
  const int32_t* node_off;
  const int64_t* code_off;
  int32_t D;
  int32_t rho_raw_hops;
  int32_t rest_zero;
  const float* S;            /* [total_nodes, f.C] node sums of the forward */
  const float* lut;          /* [n_graphs, D, rho.C] rho tables of the forward */
  const float* dY;           /* [total_nodes, f.C] or NULL */
  const float* dYsum;        /* [n_graphs, f.C] or NULL (used when not NULL) */
  gnan_small_mlp_grads df;
  gnan_small_mlp_grads drho;
  void* workspace;
  size_t workspace_bytes;
  const int32_t* cnt;        /* optional shell sizes, as in gnan_small_batch_args (ABI 42) */
  int64_t cnt_stride;
} gnan_small_batch_bwd_args;
size_t gnan_small_batch_bwd_workspace_bytes(const gnan_small_batch_bwd_args* a);
int gnan_small_batch_bwd(const gnan_small_batch_bwd_args* a, gnan_stream_t stream);

/* The same small graph-level task with a NAM read-out over the per-feature aggregates (models.py:358-384 with is_graph_task
 * and readout_n_layers > 0; models.py:259-300 for the read-out):
 *   out[c] = sum_k nam_k(hidden[k])[c],   hidden[k] = sum_i sum_j w(i, j) f_k(x[j, k]),
 *   w(i, j) = rho(u_d) / max(cnt[i, d], 1) (cnt NULL: no normalisation), d = min(code[i, j], D - 1).
 * f and rho are one-wide (models.py:320-321), L in {2, 3}, H <= 64; nam: L in {1, 2, 3} (L == 1: w_last [F, C], b_last [F, C]
 * or NULL — Linear(1, C) per feature), C <= 8; n <= 128, D <= 64; GNAN_ERR_UNSUPPORTED otherwise.  Forward: ONE launch of F
 * workgroups (each evaluates rho on the D distances, the column sums of w, f_k, hidden[k] and nam_k; the last to arrive adds the
 * features in order); it leaves fx [F, n] (feature-major), lut [D] and hidden [F] for the backward pass.  Backward: ONE
 * launch of F + 1 workgroups: every parameter gradient of f, rho and nam from d_out [C].  workspace:
 * gnan_small_graph_nam_workspace_bytes(F, nam.C) bytes whose first 16 are ZERO before the first launch (both kernels leave
 * them zero).  Fixed summation orders: bit-reproducible. */
typedef struct gnan_small_graph_nam_args {
  const float* x;            /* [n, F], row stride x_stride */
  int64_t x_stride;
  int32_t n, F;
  gnan_small_mlp f;          /* C == 1 */
  gnan_small_mlp rho;        /* C == 1 */
  gnan_small_mlp nam;        /* the read-out's F shape functions */
  const uint8_t* code;       /* [n, n] */
  int32_t D;
  int32_t reserved;          /* 0 */
  const int32_t* cnt;        /* optional [n, cnt_stride] */
  int64_t cnt_stride;
  float* fx;                 /* [F, n] */
  float* lut;                /* [D] */
  float* hidden;             /* [F] */
  float* out;                /* [nam.C] */
  void* workspace;
  size_t workspace_bytes;
} gnan_small_graph_nam_args;
typedef struct gnan_small_graph_nam_bwd_args {
  const float* x;
  int64_t x_stride;
  int32_t n, F;
  gnan_small_mlp f;
  gnan_small_mlp rho;
  gnan_small_mlp nam;
  const uint8_t* code;
  int32_t D;
  int32_t reserved;
  const int32_t* cnt;
  int64_t cnt_stride;
  const float* fx;           /* what the forward left */
  const float* lut;
  const float* hidden;
  const float* d_out;        /* [nam.C] */
  gnan_small_mlp_grads df;
  gnan_small_mlp_grads drho;
  gnan_small_mlp_grads dnam;
  void* workspace;
  size_t workspace_bytes;
} gnan_small_graph_nam_bwd_args;
size_t gnan_small_graph_nam_workspace_bytes(int32_t F, int32_t C);
int gnan_small_graph_nam_fwd(const gnan_small_graph_nam_args* a, gnan_stream_t stream);
int gnan_small_graph_nam_bwd(const gnan_small_graph_nam_bwd_args* a, gnan_stream_t stream);

/* Up to eight small device-to-device copies in one launch (host arrays of `count` device pointers and byte counts; ranges
 * must not overlap): the input slots of a captured graph-task step are refilled with it. */
int gnan_multi_copy(int32_t count, const void* const* src, void* const* dst, const int64_t* bytes, gnan_stream_t stream);

/* same for bf16 operand rows (stride in elements, W % 4 == 0, 8-B aligned rows) */
/* The wide backward's two per-row expressions as kernels (framework element-wise chains before; ABI 45):
 *   gnan_weight_table:    wt[i, d, c] = lut[d, c] / max(cnt[i, d], 1) - (with_rest ? lut[D-1, c] / max(cnt[i, D-1], 1) : 0)   (cnt NULL: 1)
 *   gnan_colsum_weighted: total[w] = scale[0] * sum_r S[r, w] / max(cnt[r * cnt_stride], 1)  (scale NULL: 1) — with cnt pointing at the
 *                         rest column of the count table and scale at lut[D-1]: d/dS_j of the rest bucket's column-sum term.
 * workspace of gnan_colsum_weighted: gnan_colsum_workspace_bytes(W). */
int gnan_weight_table(const float* lut, const int32_t* cnt, int64_t cnt_stride, int64_t n, int32_t D, int32_t Cw, int32_t with_rest,
                      float* wt, gnan_stream_t stream);
int gnan_colsum_weighted(const float* S, int64_t n, int32_t W, int64_t stride, const int32_t* cnt, int64_t cnt_stride, const float* scale,
                         float* total, void* workspace, size_t workspace_bytes, gnan_stream_t stream);
int gnan_colsum_bf16(const void* S, int64_t n, int32_t W, int64_t stride, float* total, void* workspace,
                     size_t workspace_bytes, gnan_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Dense inputs -> hop codes + shell counts
 * replaces nothing in the model file: it re-derives, on the GPU, the shell structure that
 * pre_process_datasets.py:112-121 baked into the two dense matrices, and checks it.
 *
 *   code[i, j] = round(1 / nd[i, j]) - 1        (exact for nd = float32(1/(1+hop)))
 *              = 255                            for nd == 0 (unreachable)
 *   cnt[i, d]  = #{ j : code[i, j] == d },  d < 255;   cnt[i, 255] = #{ j : nd[i, j] == 0 }
 *
 * status[0] |= 1 if some nd is not of the form float32(1/(1+hop)) with hop <= 254
 * status[0] |= 2 if norm != NULL and norm[i, j] != cnt[i, code[i, j]] somewhere
 * status[1]  = max hop seen (atomicMax).          status must be zeroed by the caller.
 * ------------------------------------------------------------------------------------------- */
int gnan_dense_to_code(const float* nd, const float* norm, int64_t n_rows, int64_t n_cols,
                       int64_t in_stride, uint8_t* code, int32_t* cnt /* [n_rows, 256] */,
                       int32_t* status /* [2] */, gnan_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Gradient of gnan_spmm_fwd w.r.t. its weight table, fused.  Two cases: the truncated-hop case (CSR layout, D <= 4,
 * Cw == 1, fp32 operand rows), and the DENSE layout (rowptr NULL: every pair listed, D <= 256, Cw == 1, a global table,
 * reduce_rows != 0, no s_total — one wave per row, the pairs' dot products binned by hop code in LDS);
 * anything else returns GNAN_ERR_UNSUPPORTED — use gnan_spmm_shell_sums then.
 *   dwt[q, d] = inv(q, d) * sum_w dY[q, w % dy_channels] * T[q, d, w]
 * with T the per-shell sums of the operand over row q's listed pairs (rest bucket: s_total - lower shells when
 * a->s_total is set) and inv = 1 / max(cnt, 1) when a->cnt is set.  dy_channels == W, or the fused read-out's
 * reduce_cr (its gradient is broadcast over the feature columns).  Rows are addressed as in the forward
 * (row_ids / scatter_out / hub-row plan); a->lut is only consulted for D and Cw, a->Y is ignored.
 * reduce_rows == 0: dwt is [n_rows, D];  != 0: dwt is [D] = the sum over rows (fixed-order float64 partials).
 * This is the backward of models.py:368-371 w.r.t. rho's outputs at the D distinct distances.
 * ------------------------------------------------------------------------------------------- */
typedef struct gnan_spmm_lut_grad_args {
  gnan_spmm_args spmm;       /* graph, operand and row addressing as in the forward */
  const float* dY;           /* [n_rows, dy_channels] */
  int64_t dy_stride;
  int32_t dy_channels;
  int32_t reduce_rows;
  float* dwt;                /* [n_rows, D] or, with reduce_rows, [D] */
  void* workspace;
  size_t workspace_bytes;
} gnan_spmm_lut_grad_args;
size_t gnan_spmm_lut_grad_workspace_bytes(const gnan_spmm_lut_grad_args* a);
int gnan_spmm_lut_grad(const gnan_spmm_lut_grad_args* a, gnan_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Backward of gnan_spmm_fwd for NARROW operands, both gradients from ONE pass over the transposed adjacency (CSR layout,
 * D <= 4, one global weight channel, fp32) — autograd through GNAN.py:67-73 / models.py:368-376 w.r.t. f_sums and rho's
 * outputs, for the sum-first evaluation (W = out_channels).
 * `a` describes the TRANSPOSED adjacency (row j = a node as neighbour, its pairs (i, code) = the forward rows that list it);
 * a->S holds one row per (hop code d, forward row i) — row d * a->n_cols + i, code-major: the rows of one code are contiguous —,
 * a->W = 2 * half floats wide (half a power of two >= w_real):
 *     [ dY_i / cnt(i, d)  (w_real floats, zero padded to half) | dY_i / cnt(i, D-1)  (likewise; zeros without a rest bucket) ]
 * a->lut is the global table [D] (a->Cw = 1, a->lut_row_stride = 0, a->cnt = NULL: the counts are folded into a->S).
 *   dS[j, :]   = sum_{d} lut[d] * A_d[j] - lut[D-1] * Q[j]       A_d[j] = sum of the first halves over j's code-d pairs,
 *                                                                 Q[j]  = sum of the second halves over all of j's pairs
 *   dlut[d]    = sum_j < s_rows[j, :], A_d[j] >   (d < D-1, or every d without a rest bucket)
 *   dlut[D-1]  = - sum_j < s_rows[j, :], Q[j] >   (with_rest; the caller adds < s_total, sum_i dY_i / cnt(i, D-1) >)
 * s_rows [n_rows, w_real] are the operand rows of the forward pass (S itself).  Fixed-order float64 partials: bit-reproducible.
 * It replaces gnan_spmm_fwd(s_by_code) + gnan_spmm_lut_grad — two traversals of the same pairs — by one.
 * One-channel operands (a->W == 2) may come with a->packed_index = 1 (col entries = column | code << 29, a->code unread):
 * spmm_bwd_hot_kernel then walks the rows with persistent workgroups, and with a->hot_rows > 0 the packed rows
 * [a->hot_lo, a->hot_lo + a->hot_rows) of the code blocks [hot_code_lo, hot_code_lo + hot_codes) — a->hot_rows * hot_codes
 * * 8 bytes <= 64 KB — are served from LDS.  Ordinary rows give the same dS bits as the generic kernel; hub-row slices and the
 * table gradient's float64 partials are added wave by wave (fixed order, bit-reproducible).
 * ------------------------------------------------------------------------------------------- */
/* builds that packed operand, code-major: V[d*(n + n_hot) + i, :] = [ dY[i, :W] / max(cnt[i, d], 1) | dY[i, :W] /
 * max(cnt[i, D-1], 1) ], each half zero padded to `half` floats (cnt == NULL: no division; with_rest == 0: second halves
 * zero).  With n_hot > 0 every code block has n + n_hot rows and row n + k repeats node hot[k] (int64 ids in DEVICE
 * memory): the compact second copy of the most listed nodes' rows that a column array remapped to n + k reads
 * (HopGraph.hot_columns). */
typedef struct gnan_pack_bwd_rows_args {
  const float* dY;           /* [n, W], row stride dy_stride */
  int64_t dy_stride;
  int32_t W, D;
  const int32_t* cnt;        /* optional [n, cnt_stride] */
  int64_t cnt_stride;
  int64_t n;
  int32_t with_rest;
  int32_t half;              /* floats per half row: a power of two >= W */
  float* V;                  /* [D, n + n_hot, 2 * half] */
  const int64_t* hot;        /* optional [n_hot] node ids (device memory) */
  int64_t n_hot;
  float* q_sum;              /* optional [1] (W == 1, with_rest): sum_i dY_i / max(cnt[i, D-1], 1) over the n real nodes — the column
                                sum of the packed rows' second halves that gnan_spmm_bwd_narrow's rest terms need (rest_q), out of the
                                same pass (float64 per workgroup, fixed order) instead of a gnan_colsum over the strided halves */
  void* q_workspace;         /* with q_sum: gnan_spmm_pack_bwd_rows_workspace_bytes(a) bytes, 8-byte aligned */
  size_t q_workspace_bytes;
  uint32_t* q_arrive;        /* optional arrival counter (see gnan_moment_scales_args): q_sum out of the packing launch */
} gnan_pack_bwd_rows_args;
size_t gnan_spmm_pack_bwd_rows_workspace_bytes(const gnan_pack_bwd_rows_args* a);
int gnan_spmm_pack_bwd_rows(const gnan_pack_bwd_rows_args* a, gnan_stream_t stream);

typedef struct gnan_spmm_bwd_narrow_args {
  gnan_spmm_args spmm;       /* the TRANSPOSED adjacency and the packed operand, as described above */
  const float* s_rows;       /* [n_rows, w_real] operand rows of the forward pass */
  int64_t s_rows_stride;
  int32_t w_real;
  int32_t with_rest;
  float* dS;                 /* [n_rows, w_real] */
  int64_t ds_stride;
  float* dlut;               /* [D] */
  void* workspace;
  size_t workspace_bytes;
  const float* ds_add;       /* optional [w_real]: a vector added to every row of dS — d/dS_j of the rest bucket's
                                wt(i, rest) * total term is the same for every j (the caller forms it from the packed rows) */
  int32_t hot_code_lo;       /* with spmm.hot_rows > 0: first code block whose hot rows sit in LDS ... */
  int32_t hot_codes;         /* ... and how many blocks (>= 1) */
  const float* ds_add_scale; /* optional device scalar: ds_add is multiplied by it on the way (rho(0) = lut[D - 1]: the caller then
                                passes the bare column sums of the packed rows' rest halves — a framework launch less) */
  const float* rest_total;   /* optional [w_real], with rest_q [w_real]: dlut[D - 1] += <rest_total, rest_q>, the table gradient */
  const float* rest_q;       /*   of the same term (float64, in the final pass) — three framework launches less */
} gnan_spmm_bwd_narrow_args;
size_t gnan_spmm_bwd_narrow_workspace_bytes(const gnan_spmm_bwd_narrow_args* a);
int gnan_spmm_bwd_narrow(const gnan_spmm_bwd_narrow_args* a, gnan_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * All-pairs hop distances on the GPU (graphs small enough for N x N bytes)
 * replaces the host preprocessing pre_process_datasets.py:104-142 (scipy Dijkstra over the directed
 * unit-weight adjacency + N^2 Python lambda calls counting equal entries):
 *   code[i, j] = hop(i -> j), 255 if unreachable or beyond max_hops;   cnt[i, d] = #{ j : code[i, j] == d }
 * Adjacency: CSR of the directed edge list (rowptr int32 [n+1], col int32), duplicates harmless.
 * status[0] |= 1 if a hop distance >= 255 occurred; status[1] = max hop seen.  status zeroed by the caller.
 * ------------------------------------------------------------------------------------------- */
size_t gnan_bfs_dense_workspace_bytes(int32_t n);
typedef struct gnan_bfs_dense_args {
  const int32_t* rowptr;     /* [n + 1] */
  const int32_t* col;
  int32_t n;
  int32_t max_hops;
  uint8_t* code;             /* [n, n] */
  int32_t* cnt;              /* [n, 256] */
  int32_t* status;           /* [2] */
  void* workspace;
  size_t workspace_bytes;
} gnan_bfs_dense_args;
int gnan_bfs_dense(const gnan_bfs_dense_args* a, gnan_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K-hop truncated hop-coded CSR, for graphs too large for N x N bytes (pre_process_datasets.py:104-142 truncated at
 * max_hops: pairs farther apart join the rest bucket, i.e. the dense matrices with every entry beyond max_hops zeroed).
 * For each source row r in [row_lo, row_hi): all nodes within max_hops directed hops and their hop counts.
 * Call twice with the same graph arguments:
 *   count pass (out_rowptr == NULL): level_cnt[r - row_lo, d] = #{nodes at hop d}, d = 0..max_hops (d = 0: r itself)
 *   fill pass  (out_rowptr = exclusive prefix sum of the row totals, int64): out_col / out_code hold row r's nodes
 *              at out_rowptr[r - row_lo] in BFS order (hop 0 first); the order inside one hop level is unspecified.
 * Adjacency: CSR of the directed edge list, duplicates and self loops harmless.  queue_cap bounds the nodes one row
 * may list; status[0] |= 1 if a row exceeded it (outputs invalid: retry with a larger queue_cap).  status zeroed by
 * the caller.  workspace: gnan_bfs_khop_workspace_bytes(n, queue_cap, n_workgroups).
 * ------------------------------------------------------------------------------------------- */
size_t gnan_bfs_khop_workspace_bytes(int64_t n, int32_t queue_cap, int32_t n_workgroups);
typedef struct gnan_bfs_khop_args {
  const void* rowptr;        /* [n + 1] int32 or int64 */
  int32_t rowptr_is64;
  int32_t max_hops;
  const int32_t* col;
  int64_t n;
  int64_t row_lo, row_hi;
  int32_t* level_cnt;        /* count pass: [rows, max_hops + 1] */
  const int64_t* out_rowptr; /* fill pass: exclusive prefix sum of the row totals; NULL = count pass */
  int32_t* out_col;
  uint8_t* out_code;
  int32_t queue_cap;
  int32_t n_workgroups;
  int32_t* status;           /* [1] */
  void* workspace;
  size_t workspace_bytes;
} gnan_bfs_khop_args;
int gnan_bfs_khop(const gnan_bfs_khop_args* a, gnan_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * hipGraph hygiene for captured steps (gnan_amd/graphed.py; nothing comparable in the reference, which issues every
 * epoch's launches from Python, trainer.py:23-86).  `graph` is a hipGraph_t obtained by stream capture and not yet
 * instantiated: every memset node is replaced by a kernel node performing the same fill, with the same dependencies
 * and dependents.  On ROCm 7.2 a captured hipMemsetAsync replays correctly only once (garbage fill values afterwards);
 * the framework's multi-block reductions zero their semaphores that way.  *n_replaced (optional, host memory) receives
 * the number of nodes swapped.
 * ------------------------------------------------------------------------------------------- */
int gnan_graph_replace_memsets(void* graph, int32_t* n_replaced);
/* How many nodes a captured hipGraph_t has, and how many of them are kernel launches (either pointer may be NULL; host
 * memory): what "a training step is six launches" is checked with (tools/muta_epoch.py, tests). */
int gnan_graph_node_count(void* graph, int32_t* n_kernels, int32_t* n_nodes);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* GNAN_HIP_H */
