"""ctypes binding of libseqrec_hip.so -- the C ABI declared in include/seqrec_hip.h.

There is NO fallback: if the shared object is missing or does not load, every
entry point raises.  (The CPU oracle lives in ``oracle/`` and is test
infrastructure; the product never imports it.)
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SEQREC_LIB lets a developer point at an experimental build of the SAME library (tools/); it is
# never a fallback: if it is set and missing, loading fails.
LIB_PATH = os.environ.get("SEQREC_LIB") or os.path.join(HERE, "libseqrec_hip.so")

CELL = {"simplernn": 0, "lstm": 1, "gru": 2}
ACT = {"relu": 0, "tanh": 1, "linear": 2,
       # Keras 2.0's other element-wise activations: step-wise form of the scans only (SEQREC_ACT_SIGMOID ... include/seqrec_hip.h)
       "sigmoid": 3, "hard_sigmoid": 4, "softplus": 5, "softsign": 6, "elu": 7}
N_GATES = {"simplernn": 1, "lstm": 4, "gru": 3}

# RNG stream ids (specification: oracle/rng.py)
STREAM_NEG = 1
STREAM_DROP_IN = 2
STREAM_DROP_OUT = 3
STREAM_DROP_REC = 4

ERRORS = {-1: "SEQREC_E_ARG", -2: "SEQREC_E_SHAPE", -3: "SEQREC_E_UNSUPPORTED"}
# bits of a device status word (SEQREC_STATUS_* of include/seqrec_hip.h)
STATUS_BITS = {1: "gradient norm not finite (SEQREC_STATUS_BAD_NORM): an overflowing, NaN or uninitialised gradient value -- update refused",
               2: "gradient divisor / token count not a finite number > 0 (SEQREC_STATUS_BAD_DIVISOR) -- update refused",
               4: "clip scale 0 or not finite (SEQREC_STATUS_BAD_SCALE): the step would have been a silent no-op -- update refused",
               8: "row index outside its table (SEQREC_STATUS_BAD_INDEX): read as a zero row"}

P = C.c_void_p
I = C.c_int
L = C.c_int64
F = C.c_float
D = C.c_double
U64 = C.c_uint64

ABI_VERSION = 5          # SEQREC_ABI_VERSION of include/seqrec_hip.h this binding was written against

# name -> argtypes   (restype is int unless listed in _RESTYPES)
_SIGS = {
    "seqrec_abi_version": [],
    "seqrec_build_arch": [],
    "seqrec_gather_rows": [P, P, P, L, I, P, P, I, P],
    "seqrec_gather_rows_bounded": [P, L, P, P, L, I, P, P, I, P, P],
    "seqrec_gemm_f32": [I, I, L, L, L, P, L, P, L, P, L, P, I, I, P, P],
    "seqrec_gemm_f32_slabs": [I, I, L, L, L, P, L, P, L, I, P, P, P],
    "seqrec_gemm_f32_fused": [I, I, L, L, L, P, L, P, L, P, L, P, I, I, P, P, P],
    "seqrec_gemm_workspace_floats": [L, L, I],
    "seqrec_gemm_f32_grouped": [I, I, I, P, I, P, P],
    "seqrec_gemm_f32_grouped_slabs": [I, I, I, P, I, P, P, P],
    "seqrec_debug_gemm_tile": [I, I],
    "seqrec_debug_scan_cluster": [I],
    "seqrec_cluster_scan_errors": [P],
    "seqrec_cluster_scan_errors_reset": [P],
    "seqrec_debug_cluster_spin_limit": [I],
    "seqrec_release_stream": [P],
    "seqrec_rnn_upack_floats": [I, I],
    "seqrec_rnn_pack_u": [I, I, P, P, P],
    "seqrec_rnn_fwd": [I, I, I, I, I, I, P, P, P, P, P, P, P],
    "seqrec_rnn_bwd": [I, I, I, I, I, I, P, P, P, P, P, P, P, P],
    "seqrec_rnn_pack_u_stepwise": [I, I, P, P, P],
    "seqrec_rnn_fwd_stepwise": [I, I, I, I, I, I, P, P, P, P, P, P, P, P, I, P],
    "seqrec_rnn_bwd_stepwise": [I, I, I, I, I, I, P, P, L, P, P, P, P, P, P, P, P, I, P],
    "seqrec_rnn_bwd_stepwise_parts": [I, I, I, I, I, I, P, P, L, P, P, P, P, P, P, P, P, P, I, P],
    "seqrec_graph_cache_clear": [],
    "seqrec_full_softmax_ce": [P, L, P, L, I, F, P, P, P],
    "seqrec_sampled_softmax_ce": [P, L, P, I, P, P, P, P, P, P, L, I, F, P, P, P],
    "seqrec_sampled_softmax_ce_rows": [P, L, P, I, P, P, P, P, P, L, I, F, P, P, P],
    "seqrec_sampled_softmax_ce_rows_idx": [P, L, P, I, P, L, P, P, P, P, P, L, I, F, P, P, P],
    "seqrec_exchange_pack": [P, L, I, P, P, L, L, U64, U64, I, P, P, I, I, P, P, I, I, I, I, P, P, P, P, P, P, P],
    "seqrec_exchange_unpack": [P, I, P, P, I, P, L, P, P, P, P, P],
    "seqrec_train_cell": [P, P],
    "seqrec_gemm_f32_pair": [P, P],
    "seqrec_cell_plan_bytes": [],
    "seqrec_exchange_grad_pack": [P, L, I, I, I, P, I, L, P, P, P, I, L, P, P, P, P],
    "seqrec_route_count_host": [P, P, L, I, P],
    "seqrec_route_fill_host": [P, P, L, I, L, I, P, P, P, P],
    "seqrec_route_blob_host": [P, I, P, P, P, L, I, I, I, I, P, P, P, P, F, P, P, L],
    "seqrec_reduce_sum": [P, L, P, I, P],
    "seqrec_colsum": [P, L, I, L, P, I, P, P],
    "seqrec_mul": [P, P, P, L, P],
    "seqrec_fill_f32": [P, F, L, P],
    "seqrec_fill_i32": [P, I, L, P],
    "seqrec_topk_merge": [P, L, L, I, I, P, P, P, P],
    "seqrec_topk_finish": [P, P, L, I, P, P, P],
    "seqrec_opt_sqnorm": [I, P, P, P, I, P, P, L, P, P],
    "seqrec_opt_sqnorm_slabs": [I, P, P, I, P, I, P, P, I, P, P, L, P, P],
    "seqrec_loss_reduce": [P, L, P, P],
    "seqrec_opt_apply": [I, P, P, P, P, P, I, P, F, F, F, P, P, P, P, P, P],
    "seqrec_prior_grad": [P, P, L, C.c_float, P, P, P],
    "seqrec_pack_batch": [P, P, P, P, I, I, P, P, P, P],
    "seqrec_pack_batch_host": [P, P, P, P, I, I, P, P, P, P, P, P],
    "seqrec_history_features": [P, P, P, P, I, I, I, L, I, P, P],
    "seqrec_index_affine_i32": [P, P, P, P, L, I, I, P],
    "seqrec_rows_scatter_add": [P, P, P, P, L, P, L, I, I, P],
    "seqrec_rows_sqnorm": [P, P, P, L, I, I, P, P],
    "seqrec_rows_adagrad": [P, P, P, P, P, L, I, I, F, F, P, P],
    "seqrec_rows_scatter_add_multi": [P, I, P],
    "seqrec_rows_merge_workspace_bytes": [L, I],
    "seqrec_rows_merge_sorted": [P, I, P, L, P],
    "seqrec_opt_sqnorm_ordered_floats": [I, I, L],
    "seqrec_opt_sqnorm_ordered": [I, P, P, P, I, P, L, P, I, P, L, P, P],
    "seqrec_rows_sqnorm_multi": [P, I, P, P],
    "seqrec_rows_adagrad_multi": [P, I, F, F, P, P],
    "seqrec_sqnorm_multi": [I, P, P, P, P],
    "seqrec_adagrad_dense_multi": [I, P, P, P, P, F, F, P, P],
    "seqrec_sqnorm": [P, L, P, P],
    "seqrec_clip_scale": [P, F, P, P],
    "seqrec_adagrad_dense": [P, P, P, L, F, F, P, P],
    "seqrec_sample_negatives": [U64, U64, I, P, P, I, P, P],
    "seqrec_sample_gather": [C.c_uint64, C.c_uint64, I, P, P, I, P, I, P, P, P, P, P],
    "seqrec_rnn_pack_u_sample": [I, I, P, P, C.c_uint64, C.c_uint64, I, P, P, I, P, I, P, P, P, P, P],
    "seqrec_rnn_pack_u_sample_batch": [I, I, P, P, C.c_uint64, C.c_uint64, I, P, P, I, P, I, P, P, P, P, P, P, P, P, I, I, P, P, P, P, P, P],
    "seqrec_dropout_mask": [U64, U64, P, L, I, L, D, P, P],
    "seqrec_rank_count": [P, I, P, P, P, L, I, P, P, P],
    "seqrec_rank_count_thr": [P, I, P, P, P, P, L, I, P, P],
    "seqrec_target_score": [P, I, P, P, P, L, P, P],
}
_RESTYPES = {
    "seqrec_debug_gemm_tile": None,
    "seqrec_debug_scan_cluster": None,
    "seqrec_debug_cluster_spin_limit": None,
    "seqrec_build_arch": C.c_char_p,
    "seqrec_gemm_workspace_floats": L,
    "seqrec_rnn_upack_floats": L,
    "seqrec_rows_merge_workspace_bytes": L,
    "seqrec_opt_sqnorm_ordered_floats": L,
    "seqrec_route_blob_host": L,
    "seqrec_cell_plan_bytes": L,
}
EXPORTS = sorted(_SIGS)


class RowsJob(C.Structure):
    """seqrec_rows_job (include/seqrec_hip.h)."""
    _fields_ = [("table", P), ("accum", P), ("gtab", P), ("slot", P), ("rows", P), ("vals", P), ("ldv", L),
                ("row_scale", P), ("n", L), ("width", C.c_int32), ("base", C.c_int32),
                ("n_slabs", C.c_int32), ("reserved_", C.c_int32), ("slab_stride", L)]


def rows_jobs(jobs):
    """list of dicts(table, accum, gtab, slot, rows, vals, ldv, row_scale, n, width, base) of torch tensors / ints
    -> (ctypes array, count).  Keep the tensors alive until the launches are enqueued."""
    arr = (RowsJob * len(jobs))()
    for i, j in enumerate(jobs):
        for k in ("table", "accum", "gtab", "slot", "rows", "vals", "row_scale"):
            t = j.get(k)
            setattr(arr[i], k, None if t is None else t.data_ptr())
        arr[i].ldv, arr[i].n, arr[i].width, arr[i].base = int(j["ldv"]), int(j["n"]), int(j["width"]), int(j["base"])
        arr[i].n_slabs, arr[i].slab_stride = int(j.get("n_slabs", 0)), int(j.get("slab_stride", 0))
    return arr, len(jobs)


PACK_HOST_MAX = 960      # SEQREC_PACK_HOST_MAX (include/seqrec_hip.h)
PACK_MERGED_MAX = 640    # SEQREC_PACK_MERGED_MAX: a batch that rides in the step's prologue launch


class GemmDesc(C.Structure):
    """seqrec_gemm_desc (include/seqrec_hip.h)."""
    _fields_ = [("M", L), ("N", L), ("K", L), ("A", P), ("lda", L), ("B", P), ("ldb", L), ("C", P), ("ldc", L),
                ("bias", P), ("accumulate", C.c_int32), ("a_index", P)]


def gemm_descs(items):
    """items: list of (M, N, K, A, lda, B, ldb, C, ldc[, a_index]) with torch tensors -> ctypes array."""
    arr = (GemmDesc * len(items))()
    for i, it in enumerate(items):
        M, N, K, A, lda, B, ldb, Cm, ldc = it[:9]
        arr[i].M, arr[i].N, arr[i].K = int(M), int(N), int(K)
        arr[i].A, arr[i].lda, arr[i].B, arr[i].ldb = A.data_ptr(), int(lda), B.data_ptr(), int(ldb)
        arr[i].C, arr[i].ldc, arr[i].bias, arr[i].accumulate = Cm.data_ptr(), int(ldc), None, 0
        arr[i].a_index = it[9].data_ptr() if len(it) > 9 and it[9] is not None else None
    return arr


class GemmPair(C.Structure):
    """seqrec_gemm_pair (include/seqrec_hip.h)."""
    _fields_ = [("a_kc0", C.c_int32), ("b_kc0", C.c_int32), ("a_kc1", C.c_int32), ("b_kc1", C.c_int32),
                ("M0", L), ("N0", L), ("K0", L), ("A0", P), ("lda0", L), ("B0", P), ("ldb0", L), ("C0", P), ("ldc0", L),
                ("splitk0", C.c_int32), ("reserved0_", C.c_int32), ("ws0", P),
                ("add_table", P), ("add_index", P), ("add_scale", P), ("add_ld", L),
                ("M1", L), ("N1", L), ("K1", L), ("A1", P), ("lda1", L), ("B1", P), ("ldb1", L),
                ("splitk1", C.c_int32), ("reserved1_", C.c_int32), ("ws1", P),
                ("n_slabs1", C.c_int32), ("together", C.c_int32)]


class CellPlan(C.Structure):
    """seqrec_cell_plan (include/seqrec_hip.h): the arguments of seqrec_train_cell."""
    _fields_ = [("stages", C.c_int32), ("cell", C.c_int32), ("act", C.c_int32), ("Hp", C.c_int32), ("H_real", C.c_int32), ("G", C.c_int32),
                ("K", C.c_int32), ("Dp", C.c_int32), ("T", C.c_int32), ("B", C.c_int32), ("use_graph", C.c_int32), ("reserved0_", C.c_int32),
                ("n", L), ("step_off_host", P),
                ("pack_u", C.c_int32), ("sample", C.c_int32), ("seed", U64), ("step", U64),
                ("U", P), ("upack", P), ("thresh", P), ("alias", P), ("V", C.c_int32), ("reserved1_", C.c_int32), ("sample_table", P),
                ("sample_logq", P), ("neg_out", P), ("Eneg_out", P), ("lq_neg_out", P),
                ("batch", C.c_int32), ("reserved3_", C.c_int32), ("flat", P), ("starts", P), ("sess_host", P),
                ("sess_out", P), ("step_off_out", P), ("ids_out", P), ("tgt_out", P), ("prev_out", P),
                ("x_table", P), ("x_ld", L), ("x_index", P), ("W", P), ("bias", P),
                ("XW", P), ("Hout", P), ("gates", P), ("aux", P),
                ("Eneg", P), ("neg", P), ("lq_neg", P),
                ("ln", P), ("dlt", P), ("loss_rows", P), ("inv_denom", F), ("reserved2_", C.c_int32),
                ("tgt_table", P), ("tgt_ld", L), ("tgt_index", P), ("tgt_ids", P), ("lq_tgt", P), ("logq_table", P),
                ("dHd", P), ("gemm_ws", P), ("sk_dh", C.c_int32), ("deneg_mode", C.c_int32), ("sk_deneg", C.c_int32), ("sk_wgrad", C.c_int32),
                ("wgrad_slabs", C.c_int32), ("sk_dx", C.c_int32),
                ("dEneg_slabs", P), ("dPre", P), ("scan_ws", P), ("prev", P),
                ("dU", P), ("dW", P), ("db", P), ("ones", P), ("wgrad_ws", P), ("dX_slabs", P),
                ("ns_deneg", C.c_int32), ("ns_wgrad", C.c_int32), ("ns_dx", C.c_int32), ("n_descs", C.c_int32), ("deneg_off", L),
                ("descs_out", GemmDesc * 6)]


class GemmFuse(C.Structure):
    """seqrec_gemm_fuse (include/seqrec_hip.h)."""
    _fields_ = [("a_index", P), ("add_table", P), ("add_index", P), ("add_scale", P), ("add_ld", L)]


def gemm_fuse(a_index=None, add_table=None, add_index=None, add_scale=None, add_ld=0):
    f = GemmFuse()
    f.a_index = None if a_index is None else a_index.data_ptr()
    f.add_table = None if add_table is None else add_table.data_ptr()
    f.add_index = None if add_index is None else add_index.data_ptr()
    f.add_scale = None if add_scale is None else add_scale.data_ptr()
    f.add_ld = int(add_ld)
    return f


class DhParts(C.Structure):
    """seqrec_dh_parts (include/seqrec_hip.h)."""
    _fields_ = [("slabs", P), ("n_slabs", C.c_int32), ("reserved_", C.c_int32), ("slab_stride", L), ("add_table", P),
                ("add_index", P), ("add_scale", P), ("add_ld", L)]


def dh_parts(slabs, n_slabs, slab_stride, add_table=None, add_index=None, add_scale=None, add_ld=0):
    f = DhParts()
    f.slabs, f.n_slabs, f.slab_stride = slabs.data_ptr(), int(n_slabs), int(slab_stride)
    f.add_table = None if add_table is None else add_table.data_ptr()
    f.add_index = None if add_index is None else add_index.data_ptr()
    f.add_scale = None if add_scale is None else add_scale.data_ptr()
    f.add_ld = int(add_ld)
    return f


def ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def i64_array(vals):
    return (C.c_int64 * len(vals))(*[int(v) for v in vals])

_lib = None


class SeqrecError(RuntimeError):
    pass


def load():
    """Load the shared object (once).  Raises if it is absent -- build it with
    ``python -m`` ``seq-recommendations_amd/build.py`` or ``__graft_entry__.build()``."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime (libamdhip64); it must be in the process BEFORE this library
    # is loaded, otherwise the system copy under /opt/rocm gets loaded first and the two runtimes do
    # not share devices or allocations (kernel launches then fail with hipErrorNoDevice).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise SeqrecError(
            "libseqrec_hip.so not found at %s: the HIP extension is required (no CPU fallback). "
            "Run __graft_entry__.build()." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, I)
    if lib.seqrec_abi_version() != ABI_VERSION:      # struct layouts below are those of this version (seqrec_rows_job grew in 3)
        raise SeqrecError("%s has ABI version %d, this binding needs %d: rebuild it (__graft_entry__.build())"
                          % (LIB_PATH, lib.seqrec_abi_version(), ABI_VERSION))
    if lib.seqrec_cell_plan_bytes() != C.sizeof(CellPlan):
        raise SeqrecError("seqrec_cell_plan is %d bytes in %s and %d in this binding" % (lib.seqrec_cell_plan_bytes(), LIB_PATH, C.sizeof(CellPlan)))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise SeqrecError("%s failed: %s" % (what, ERRORS.get(rc, "hipError %d" % rc)))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return t.data_ptr()             # a plain int: ctypes converts it for the c_void_p parameters


_FN = {}


def call(name, *args):
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        check(rc, name)
