"""ctypes loader for libmgl_sw_hip.so (the C ABI declared in include/mgl_sw.h).

There is no fallback: if the library is missing it is built in-tree with hipcc, and if that
fails the import error is raised.  Compute entry points fail with MGL_SW_ERR_DEVICE when no
GPU is visible.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
# MGL_SW_LIB points the loader at a diagnostic build (scripts/ablate.sh); never set in production
LIB_PATH = os.environ.get("MGL_SW_LIB") or os.path.join(HERE, "libmgl_sw_hip.so")
CSRC = os.path.join(HERE, "csrc")

FLAG_UNIFORM_GEOMETRY = 1
FLAG_BINARY_CIGAR = 2
FLAG_GROUPED_GEOMETRY = 4
FLAG_SCORE_ONLY = 8
FLAG_SHARED_TARGET = 0x10
OK, ERR_BAD_ARG, ERR_CIGAR_OVERFLOW, ERR_NOMEM, ERR_DEVICE, ERR_UNSUPPORTED = range(6)

# every symbol include/mgl_sw.h declares (tests check that the library exports them all)
SYMBOLS = (
    "mgl_sw_version", "mgl_sw_strerror", "mgl_sw_device_count", "mgl_sw_max_query_len", "mgl_sw_max_lds_query_len", "mgl_sw_ctx_set_carry_memory", "mgl_sw_ctx_set_stripe_rows", "mgl_sw_ctx_set_cooperative", "mgl_sw_ctx_set_strip_kernel", "mgl_sw_ctx_set_lane_kernel", "mgl_sw_ctx_set_lane_checkpoint", "mgl_sw_ctx_set_small_kernel", "mgl_sw_set_service", "mgl_sw_service_stats", "mgl_sw_ctx_create",
    "mgl_sw_ctx_destroy", "mgl_sw_last_error", "mgl_sw_ctx_set_workspace", "mgl_sw_ctx_set_profiling", "mgl_sw_ctx_set_precision",
    "mgl_sw_ctx_get_timing", "mgl_sw_normalize_params", "mgl_sw_align", "mgl_sw_align_batch", "mgl_sw_align_batch_status",
    "mgl_sw_align_batch_device", "mgl_sw_align_batch_device_2bit", "mgl_sw_align_batch_device_matrix", "mgl_sw_align_batch_device_indexed", "mgl_sw_backtrack_matrix", "mgl_sw_ctx_expand_slot", "mgl_sw_ctx_slot_layout",
    "mgl_sw_cigar_from_backtrack", "mgl_sw_band_fill", "mgl_sw_set_coalescing", "mgl_sw_coalescing_stats", "mgl_sw_group_by_geometry",
    "mgl_sw_multi_create", "mgl_sw_multi_destroy", "mgl_sw_multi_device_count", "mgl_sw_multi_ctx", "mgl_sw_multi_set_workspace",
    "mgl_sw_multi_last_error", "mgl_sw_align_batch_multi", "mgl_sw_multi_last_shards", "mgl_sw_shard_by_cells",
    "mgl_sw_align_batch_2bit", "mgl_sw_register_host_buffer", "mgl_sw_unregister_host_buffer", "mgl_sw_explain",
    "mgl_sw_explain_sized", "mgl_sw_ctx_check",
)
# MGL_SW_VERSION of the include/mgl_sw.h this mirror was written against: the structs below (Plan, Timing) are that header's, and the
# library writes sizeof(ITS struct) through the pointers it is given -- so a library of another version is refused at load time
ABI_VERSION = 102


class Score(C.Structure):
    """mgl_sw_score == ScoreMax (sw_common.h:36-40)."""
    _fields_ = [(n, C.c_int32) for n in ("mqe", "mqe_t", "max", "max_t", "max_q", "seg_length")]


class Timing(C.Structure):
    _fields_ = [("dp_ms", C.c_float), ("tb_ms", C.c_float), ("dp_launches", C.c_int32), ("tb_launches", C.c_int32),
                ("cells", C.c_int64), ("tb_bytes", C.c_int64), ("packed16", C.c_int32), ("clock_mhz", C.c_int32),
                ("fill_kernel", C.c_int32), ("reserved", C.c_int32)]


class Plan(C.Structure):
    """mgl_sw_plan: what the library would do with a batch (mgl_sw_explain)."""
    _fields_ = [(n, C.c_int32) for n in ("fill_kernel", "precision_bits", "rows", "waves_per_block", "waves_per_pair", "traceback", "fused_walk",
                                         "sorted_by_library", "fill_streams", "workspace_halves")] + [
        (n, C.c_int64) for n in ("chunk_pairs", "chunks", "workspace_bytes_per_pair", "workspace_bytes", "workspace_fixed_bytes", "resident_waves")]


def explain(n, max_tl, max_ql, parameters=(200, -150, 260, 11), strategy=1, flags=0, packed2=False, entry=0, workspace=0, ctx=None):
    """The planner's decisions for a batch (no GPU needed when ctx is None): a Plan."""
    p = Plan()
    rc = lib().mgl_sw_explain(ctx, int(workspace), int(n), int(max_tl), int(max_ql), *[int(x) for x in parameters], int(strategy), int(flags),
                              int(bool(packed2)), int(entry), C.byref(p))
    if rc != OK:
        raise MglSwError(rc)
    return p


FILL_KERNEL_NAMES = ("sw_dp_kernel", "sw_dp16_kernel", "sw_dp64_kernel", "sw_dp_coop_kernel", "sw_dp16_lane_kernel", "sw_dp_coop16_kernel", "sw_dp16_strip_kernel", "sw_dp16_lane_ck_kernel", "sw_small_kernel", "sw_dp16_lane_matrix_kernel")


def _sources_newer():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h"))]
    srcs.append(os.path.join(HERE, "..", "include", "mgl_sw.h"))
    return any(os.path.getmtime(s) > t for s in srcs if os.path.exists(s))


def build(force=False):
    """Compile csrc/ for gfx950 into mgl_amd/libmgl_sw_hip.so (hipcc cross-compiles without a GPU)."""
    if force or _sources_newer():
        subprocess.check_call(["make", "-s", "-j8", "-C", CSRC] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    # RTLD_NOW: an unresolved symbol in the library must fail here, not at the first call
    L = C.CDLL(LIB_PATH, mode=os.RTLD_NOW)
    vp, i32p, i64p, cp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.c_char_p
    L.mgl_sw_version.restype = C.c_int
    if L.mgl_sw_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} is MGL_SW_VERSION {L.mgl_sw_version()}, this mirror of include/mgl_sw.h is {ABI_VERSION}: rebuild (make -C mgl_amd/csrc)")
    L.mgl_sw_strerror.restype = C.c_char_p
    L.mgl_sw_strerror.argtypes = [C.c_int]
    L.mgl_sw_device_count.restype = C.c_int
    L.mgl_sw_max_query_len.restype = C.c_int
    L.mgl_sw_max_lds_query_len.restype = C.c_int
    L.mgl_sw_ctx_set_carry_memory.argtypes = [vp, C.c_int]
    L.mgl_sw_ctx_set_stripe_rows.argtypes = [vp, C.c_int]
    L.mgl_sw_ctx_set_cooperative.argtypes = [vp, C.c_int]
    L.mgl_sw_ctx_set_strip_kernel.argtypes = [vp, C.c_int]
    L.mgl_sw_ctx_set_lane_kernel.argtypes = [vp, C.c_int]
    L.mgl_sw_ctx_set_lane_checkpoint.argtypes = [vp, C.c_int]
    L.mgl_sw_ctx_set_small_kernel.argtypes = [vp, C.c_int]
    L.mgl_sw_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.mgl_sw_ctx_destroy.argtypes = [vp]
    L.mgl_sw_ctx_destroy.restype = None
    L.mgl_sw_last_error.argtypes = [vp]
    L.mgl_sw_last_error.restype = C.c_char_p
    L.mgl_sw_ctx_set_workspace.argtypes = [vp, C.c_int64]
    L.mgl_sw_ctx_set_profiling.argtypes = [vp, C.c_int]
    L.mgl_sw_ctx_get_timing.argtypes = [vp, C.POINTER(Timing)]
    L.mgl_sw_normalize_params.argtypes = [C.POINTER(C.c_int)] * 4
    L.mgl_sw_normalize_params.restype = None
    L.mgl_sw_align.argtypes = [cp, C.c_int, cp, C.c_int] + [C.c_int] * 5 + [
        cp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(Score)]
    L.mgl_sw_align_batch.argtypes = [vp, C.c_int64, vp, vp, vp, vp] + [C.c_int] * 5 + [vp, vp, vp, C.c_int, vp]
    L.mgl_sw_align_batch_device.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, C.c_int, C.c_int] + [C.c_int] * 5 + [
        vp, vp, vp, C.c_int, vp, vp, C.c_int]
    L.mgl_sw_ctx_set_precision.argtypes = [vp, C.c_int]
    L.mgl_sw_align_batch_device_2bit.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int] + [
        C.c_int] * 5 + [vp, vp, vp, C.c_int, vp, vp, C.c_int]
    L.mgl_sw_align_batch_device_indexed.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int] + [
        C.c_int] * 5 + [vp, vp, vp, C.c_int, vp, vp, C.c_int]
    L.mgl_sw_align_batch_device_matrix.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int,
                                                   C.c_int, vp, vp, vp, C.c_int, vp, vp, C.c_int]
    L.mgl_sw_backtrack_matrix.argtypes = [cp, C.c_int, cp, C.c_int] + [C.c_int] * 5 + [i32p, C.POINTER(Score)]
    L.mgl_sw_cigar_from_backtrack.argtypes = [i32p, C.c_int, C.c_int, C.c_int, C.POINTER(Score), cp, C.c_int,
                                              C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mgl_sw_set_coalescing.argtypes = [C.c_int, C.c_int]
    L.mgl_sw_band_fill.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp] + [C.c_int] * 5 + [C.POINTER(Score)]
    L.mgl_sw_group_by_geometry.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    L.mgl_sw_coalescing_stats.argtypes = [i64p, i64p]
    L.mgl_sw_set_service.argtypes = [C.c_int, C.c_int]
    L.mgl_sw_service_stats.argtypes = [i64p, i64p]
    L.mgl_sw_ctx_expand_slot.argtypes = [vp, C.c_int64, C.c_int, C.c_int, i32p]
    L.mgl_sw_ctx_slot_layout.argtypes = [vp, C.c_int64, C.POINTER(C.c_int)]
    L.mgl_sw_align_batch_status.argtypes = [vp, C.c_int64, vp, vp, vp, vp] + [C.c_int] * 5 + [vp, vp, vp, C.c_int, vp, vp]
    L.mgl_sw_multi_create.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]
    L.mgl_sw_multi_destroy.argtypes = [vp]
    L.mgl_sw_multi_destroy.restype = None
    L.mgl_sw_multi_device_count.argtypes = [vp]
    L.mgl_sw_multi_ctx.argtypes = [vp, C.c_int]
    L.mgl_sw_multi_ctx.restype = vp
    L.mgl_sw_multi_set_workspace.argtypes = [vp, C.c_int64]
    L.mgl_sw_multi_last_error.argtypes = [vp]
    L.mgl_sw_multi_last_error.restype = C.c_char_p
    L.mgl_sw_align_batch_multi.argtypes = [vp, C.c_int64, vp, vp, vp, vp] + [C.c_int] * 5 + [vp, vp, vp, C.c_int, vp, vp]
    L.mgl_sw_multi_last_shards.argtypes = [vp, vp]
    L.mgl_sw_shard_by_cells.argtypes = [C.c_int64, vp, vp, C.c_int, C.c_int64, vp]
    L.mgl_sw_align_batch_2bit.argtypes = [vp, C.c_int64, vp, C.c_int64, vp, vp, vp, C.c_int64, vp, vp, C.c_int, C.c_int] + [C.c_int] * 5 + [
        vp, vp, vp, C.c_int, vp, vp, C.c_int]
    L.mgl_sw_explain.argtypes = [vp, C.c_int64, C.c_int64] + [C.c_int] * 10 + [C.POINTER(Plan)]
    L.mgl_sw_explain_sized.argtypes = [vp, C.c_int64, C.c_int64] + [C.c_int] * 10 + [vp, C.c_size_t]
    L.mgl_sw_ctx_check.argtypes = [vp]
    L.mgl_sw_register_host_buffer.argtypes = [vp, vp, C.c_size_t]
    L.mgl_sw_unregister_host_buffer.argtypes = [vp, vp]
    _lib = L
    return L


class MglSwError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = lib().mgl_sw_strerror(status).decode()
        super().__init__(f"mgl_sw status {status} ({msg})" + (f": {detail}" if detail else ""))
