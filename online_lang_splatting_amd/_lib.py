"""Loads libolsr.so (the HIP kernels + C-ABI of include/olsr.h) and declares its prototypes.

There is no fallback: if the library is missing or a symbol is absent this raises, loudly.
"""
import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# (OLSR_LIB: another build of the same library, for kernel experiments; the compiled torch binding always links libolsr.so)
LIB_PATH = os.environ.get("OLSR_LIB") or os.path.join(_HERE, "libolsr.so")

# every symbol include/olsr.h declares
EXPORTS = (
    "olsr_geometry_bytes", "olsr_image_bytes", "olsr_binning_bytes", "olsr_backward_scratch_bytes", "olsr_last_forward_token", "olsr_live_rows", "olsr_forward", "olsr_forward_async", "olsr_forward_async_loss", "olsr_fused_loss_scratch_bytes",
    "olsr_backward", "olsr_accumulate_gradients", "olsr_sparse_exchange_mask", "olsr_sparse_exchange_scratch_ints", "olsr_sparse_exchange_pack", "olsr_sparse_exchange_unpack", "olsr_mapping_loss", "olsr_mapping_loss_scratch_bytes", "olsr_tracking_loss", "olsr_pose_step", "olsr_pose_step_gated", "olsr_knn_mean_dist2", "olsr_knn_scratch_bytes", "olsr_adam_step", "olsr_adam_step_sum", "olsr_adam_step_masked", "olsr_bucket_add", "olsr_mark_visible", "olsr_geometry_field", "olsr_binning_field", "olsr_image_field",
    "olsr_set_profiling", "olsr_get_stage_times", "olsr_debug_sort_timing", "olsr_debug_sort_plan", "olsr_debug_sort_knobs", "olsr_debug_sort_small", "olsr_debug_sort_compact", "olsr_debug_sort_threads", "olsr_debug_composite_stamps", "olsr_debug_sync_fault", "olsr_debug_backward_ordered", "olsr_debug_backward_ordered_scratch_bytes", "olsr_live_rows_wait", "olsr_live_rows_overwritten", "olsr_backward_rows", "olsr_last_error", "olsr_version",
)

_lib = None


class OlsrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"olsr error {code}: {msg}")
        self.code = code


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m online_lang_splatting_amd.build` "
            "(hipcc, gfx950).  There is no CPU or PyTorch fallback for the rasterizer.")
    L = C.CDLL(LIB_PATH)
    missing = [s for s in EXPORTS if not hasattr(L, s)]
    if missing:
        raise ImportError(f"{LIB_PATH} lacks symbols {missing}; rebuild it")
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    scene_p = C.POINTER(_abi.OlsrScene)
    L.olsr_geometry_bytes.argtypes, L.olsr_geometry_bytes.restype = [i32, i32], sz
    L.olsr_image_bytes.argtypes, L.olsr_image_bytes.restype = [i32, i32, i32], sz
    L.olsr_binning_bytes.argtypes, L.olsr_binning_bytes.restype = [i64, i32], sz
    L.olsr_forward.argtypes = [scene_p, _abi.ALLOC_FN, vp, _abi.ALLOC_FN, vp, _abi.ALLOC_FN, vp,
                               vp, vp, vp, vp, vp, vp, C.POINTER(i32), vp]
    L.olsr_forward.restype = C.c_int
    L.olsr_forward_async.argtypes = [scene_p, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.olsr_forward_async.restype = C.c_int
    L.olsr_forward_async_loss.argtypes = [scene_p, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                          C.POINTER(_abi.OlsrLossFusion), vp]
    L.olsr_forward_async_loss.restype = C.c_int
    L.olsr_fused_loss_scratch_bytes.argtypes, L.olsr_fused_loss_scratch_bytes.restype = [i32, i32, i32], sz
    L.olsr_backward_scratch_bytes.argtypes, L.olsr_backward_scratch_bytes.restype = [i64, i32], sz
    L.olsr_last_forward_token.argtypes, L.olsr_last_forward_token.restype = [], i32
    L.olsr_live_rows.argtypes, L.olsr_live_rows.restype = [i32, i32], i64
    L.olsr_live_rows_wait.argtypes, L.olsr_live_rows_wait.restype = [i32, i32, i32], i64
    L.olsr_live_rows_overwritten.argtypes, L.olsr_live_rows_overwritten.restype = [i32], i32
    L.olsr_backward_rows.argtypes, L.olsr_backward_rows.restype = [i32, i32, i64, i32], i64
    L.olsr_backward.argtypes = ([scene_p, vp, vp, i32, vp, vp, _abi.ALLOC_FN, vp, vp, i64] + [vp] * 3 + [vp] * 13
                                + [C.POINTER(_abi.OlsrGradBucket), vp, vp])
    L.olsr_backward.restype = C.c_int
    L.olsr_accumulate_gradients.argtypes = [i32, i32, i32, i32] + [vp] * 12
    L.olsr_accumulate_gradients.restype = C.c_int
    L.olsr_sparse_exchange_mask.argtypes, L.olsr_sparse_exchange_mask.restype = [i32, i32] + [vp] * 5, C.c_int
    L.olsr_sparse_exchange_scratch_ints.argtypes, L.olsr_sparse_exchange_scratch_ints.restype = [i32], i64
    L.olsr_sparse_exchange_pack.argtypes, L.olsr_sparse_exchange_pack.restype = [i32, i32, i32] + [vp] * 10, C.c_int
    L.olsr_sparse_exchange_unpack.argtypes, L.olsr_sparse_exchange_unpack.restype = [i32, i32, i32] + [vp] * 5, C.c_int
    L.olsr_mapping_loss_scratch_bytes.argtypes, L.olsr_mapping_loss_scratch_bytes.restype = [i32, i32], sz
    L.olsr_mapping_loss.argtypes = [C.POINTER(_abi.OlsrLossParams)] + [vp] * 14
    L.olsr_mapping_loss.restype = C.c_int
    L.olsr_tracking_loss.argtypes = [C.POINTER(_abi.OlsrLossParams)] + [vp] * 13
    L.olsr_tracking_loss.restype = C.c_int
    L.olsr_pose_step.argtypes = [C.POINTER(_abi.OlsrPoseParams)] + [vp] * 6
    L.olsr_pose_step.restype = C.c_int
    L.olsr_pose_step_gated.argtypes = [C.POINTER(_abi.OlsrPoseParams)] + [vp] * 7
    L.olsr_pose_step_gated.restype = C.c_int
    L.olsr_adam_step.argtypes = [i32, i32, i32, C.POINTER(_abi.OlsrAdamParams)] + [vp] * 10
    L.olsr_adam_step.restype = C.c_int
    L.olsr_adam_step_sum.argtypes = [i32, i32, i32, C.POINTER(_abi.OlsrAdamParams), i32, C.POINTER(vp)] + [vp] * 9
    L.olsr_adam_step_sum.restype = C.c_int
    L.olsr_adam_step_masked.argtypes = [i32, i32, i32, C.POINTER(_abi.OlsrAdamParams), i32, C.POINTER(vp), C.POINTER(vp)] + [vp] * 9
    L.olsr_adam_step_masked.restype = C.c_int
    L.olsr_bucket_add.argtypes, L.olsr_bucket_add.restype = [i32, i32] + [vp] * 9, C.c_int
    L.olsr_knn_scratch_bytes.argtypes, L.olsr_knn_scratch_bytes.restype = [i32], sz
    L.olsr_knn_mean_dist2.argtypes, L.olsr_knn_mean_dist2.restype = [i32, vp, vp, vp, vp], C.c_int
    L.olsr_mark_visible.argtypes, L.olsr_mark_visible.restype = [i32, vp, vp, vp, vp, vp], C.c_int
    L.olsr_geometry_field.argtypes, L.olsr_geometry_field.restype = [vp, i32, i32, C.c_char_p], vp
    L.olsr_binning_field.argtypes, L.olsr_binning_field.restype = [vp, i64, i32, C.c_char_p], vp
    L.olsr_image_field.argtypes, L.olsr_image_field.restype = [vp, i32, i32, i32, C.c_char_p], vp
    L.olsr_set_profiling.argtypes, L.olsr_set_profiling.restype = [C.c_int], None
    L.olsr_get_stage_times.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
    L.olsr_get_stage_times.restype = C.c_int
    L.olsr_debug_sort_plan.argtypes = [i64, C.c_int, C.POINTER(i32), C.POINTER(i32)]
    L.olsr_debug_sort_plan.restype = C.c_int
    L.olsr_debug_sort_knobs.argtypes = [C.c_int, C.c_int, C.c_int]
    L.olsr_debug_sort_knobs.restype = None
    L.olsr_debug_sort_small.argtypes, L.olsr_debug_sort_small.restype = [C.c_int], None
    L.olsr_debug_sort_compact.argtypes, L.olsr_debug_sort_compact.restype = [C.c_int], None
    L.olsr_debug_sort_threads.argtypes, L.olsr_debug_sort_threads.restype = [C.c_int], C.c_int
    L.olsr_debug_composite_stamps.argtypes, L.olsr_debug_composite_stamps.restype = [vp, C.c_int], None
    L.olsr_debug_sync_fault.argtypes = [C.c_int, C.c_int]
    L.olsr_debug_sync_fault.restype = None
    L.olsr_debug_backward_ordered_scratch_bytes.argtypes, L.olsr_debug_backward_ordered_scratch_bytes.restype = [i64, i32], sz
    L.olsr_debug_backward_ordered.argtypes = [scene_p, vp, i32, vp, vp] + [vp] * 3 + [vp] + [vp] * 6 + [i32, vp]
    L.olsr_debug_backward_ordered.restype = C.c_int
    L.olsr_last_error.argtypes, L.olsr_last_error.restype = [], C.c_char_p
    L.olsr_version.argtypes, L.olsr_version.restype = [], C.c_char_p
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise OlsrError(rc, lib().olsr_last_error().decode())


def stage_times(max_entries=1 << 16):
    """[(stage name, milliseconds)] for every stage issued on this thread since
    olsr_set_profiling(1), in issue order."""
    names = (C.c_char_p * max_entries)()
    ms = (C.c_float * max_entries)()
    n = lib().olsr_get_stage_times(names, ms, max_entries)
    return [(names[i].decode(), float(ms[i])) for i in range(n)]


def set_profiling(enable):
    lib().olsr_set_profiling(1 if enable else 0)
