"""ctypes binding of libpointrcnn_hip.so (the C ABI declared in include/pointrcnn_hip.h).

The library is the product: there is NO CPU or eager-PyTorch fallback.  If the shared
object is missing, `lib()` raises and every op fails loudly.
"""
import ctypes
import os

# torch must be imported BEFORE the library is loaded: torch bundles its own HIP runtime
# (SONAME libamdhip64.so.7), and the dynamic loader resolves our NEEDED entry to that
# already-loaded copy.  Loaded the other way round, the process would hold two HIP/HSA
# runtimes and the second one finds no device.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRCNN_LIB") or os.path.join(_HERE, "libpointrcnn_hip.so")    # PRCNN_LIB: A/B builds of the same ABI

_c = ctypes
_vp, _i, _f, _d, _i64, _sz = _c.c_void_p, _c.c_int, _c.c_float, _c.c_double, _c.c_int64, _c.c_size_t

# name -> (restype, argtypes); mirrors include/pointrcnn_hip.h one to one
SIGNATURES = {
    "prcnn_status_string": (_c.c_char_p, [_i]),
    "prcnn_version": (_i, []),
    "prcnn_last_hip_error_string": (_c.c_char_p, []),
    "prcnn_farthest_point_sample": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "prcnn_fps_temp_bytes": (_sz, [_i, _i]),
    "prcnn_fps_shared_temp_bytes": (_sz, [_i, _i]),
    "prcnn_farthest_point_sample_shared": (_i, [_i, _i, _i, _vp, _vp, _sz, _vp, _vp]),
    "prcnn_gather_point": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "prcnn_query_ball_point": (_i, [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "prcnn_query_ball_point_msg": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "prcnn_group_point": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "prcnn_three_nn": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "prcnn_three_interpolate": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "prcnn_fp_weights": (_i, [_i64, _vp, _vp, _vp]),
    "prcnn_three_nn_interpolate": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "prcnn_nms_rotate_workspace_bytes": (_sz, [_i]),
    "prcnn_nms_rotate": (_i, [_i, _vp, _vp, _d, _i, _vp, _vp, _vp, _sz, _vp]),
    "prcnn_nms_rotate_noalloc": (_i, [_i, _vp, _vp, _d, _i, _vp, _vp, _vp]),
    "prcnn_nms_rotate_batched_workspace_bytes": (_sz, [_i, _i]),
    "prcnn_nms_rotate_batched": (_i, [_i, _i, _vp, _vp, _d, _i, _vp, _vp, _vp, _sz, _vp]),
    "prcnn_mlp_layer": (_i, [_i64, _i, _i, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _vp]),
    "prcnn_sa_group_add_layer": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _i, _i, _i,
                                      _vp, _i64, _vp]),
    "prcnn_sa_group_add_mlp2_pool": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _i, _i, _i] + [_vp] * 9 +
                                     [_vp, _i64, _vp]),
    "prcnn_sa_group_mlp_layer": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp,
                                      _i, _i, _i, _vp, _i64, _vp]),
    "prcnn_sa_group_mlp3_pool": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i] + [_vp] * 9 +
                                 [_vp, _i64, _vp]),
    "prcnn_ball_count_classes": (_i, [_i64, _vp, _vp, _vp, _vp]),
    "prcnn_sa_group_mlp3_pool_classes": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i] + [_vp] * 9 +
                                         [_vp, _i64, _vp, _vp, _vp]),
    "prcnn_sa_group_add_mlp2_pool_classes": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _i, _i, _i] + [_vp] * 9 +
                                             [_vp, _i64, _vp, _vp, _vp]),
    "prcnn_ball_compact_rows": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "prcnn_sa_rows_add_layer": (_i, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _i, _i, _vp, _i64, _vp]),
    "prcnn_mlp_layer_n": (_i, [_i64, _vp, _i, _i, _vp, _i64, _vp, _vp, _vp, _i, _i, _vp, _i64, _vp]),
    "prcnn_segment_max": (_i, [_i64, _i, _vp, _i64, _vp, _vp, _i64, _vp]),
    "prcnn_max_pool": (_i, [_i64, _i, _i, _vp, _i64, _vp, _i64, _vp]),
    "prcnn_three_nn_sorted_workspace_bytes": (_sz, [_i, _i, _i]),
    "prcnn_three_nn_sorted": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "prcnn_three_nn_interpolate_sorted": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "prcnn_three_nn_interpolate_affine": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i64, _i,
                                               _vp, _sz, _vp]),
    "prcnn_three_nn_interpolate_add": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _i, _vp, _sz, _vp]),
    "prcnn_three_interpolate_add": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _i, _vp]),
    "prcnn_three_interpolate_affine": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i64, _i, _vp]),
    "prcnn_query_ball_point_sorted_workspace_bytes": (_sz, [_i, _i, _i]),
    "prcnn_query_ball_point_msg_sorted": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "prcnn_grid_order": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "prcnn_three_nn_ordered": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "prcnn_three_nn_interpolate_affine_ordered": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp,
                                                       _i64, _i, _vp, _sz, _vp]),
    "prcnn_query_ball_point_msg_ordered": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "prcnn_selection_sort": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "prcnn_knn_point": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "prcnn_rpn_decode_proposals": (_i, [_i, _i, _i, _i, _i, _f, _f, _vp] + [_vp] * 9 + [_vp]),
    "prcnn_bev_group_keep": (_i, [_i, _i, _vp, _vp, _vp, _d, _vp, _vp, _vp, _vp, _sz, _vp]),
    "prcnn_nms_aabb_batched": (_i, [_i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _sz, _vp]),
    "prcnn_pack_detections": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "prcnn_gather_point_grad_workspace_bytes": (_sz, [_i, _i, _i]),
    "prcnn_gather_point_grad": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "prcnn_group_point_grad_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "prcnn_group_point_grad": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "prcnn_three_interpolate_grad_workspace_bytes": (_sz, [_i, _i, _i]),
    "prcnn_three_interpolate_grad": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
}

_lib = None


class PointRCNNHipError(RuntimeError):
    pass


def lib():
    """Load the shared library once; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PointRCNNHipError(
                "libpointrcnn_hip.so not found at %s -- build it with "
                "`python -m pointrcnn_amd.build` (there is no fallback path)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)       # AttributeError if the ABI and the header disagree
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, what):
    """Map the C status to the exception the reference's OP_REQUIRES would raise."""
    if status == 0:
        return
    msg = lib().prcnn_status_string(status).decode()
    if status in (-1, -2):
        raise ValueError("%s: %s" % (what, msg))
    if status == -3:
        msg += " [%s]" % lib().prcnn_last_hip_error_string().decode()
    raise PointRCNNHipError("%s: %s" % (what, msg))
