"""The C-ABI library loads and exports every symbol include/pointrcnn_hip.h declares
(no compute calls: runs without a GPU)."""
import ctypes
import os
import re

from pointrcnn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "pointrcnn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(prcnn_[a-z0-9_]+)\s*\(", text)))


def test_library_exists_and_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build with `python -m pointrcnn_amd.build`"
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), "missing export %s" % n


def test_python_binding_covers_the_header():
    assert sorted(_lib.SIGNATURES) == _declared()
    L = _lib.lib()
    assert L.prcnn_version() >= 100
    assert L.prcnn_status_string(0) == b"ok"
    assert L.prcnn_nms_rotate_workspace_bytes(9000) > 9000 * 141 * 8
    assert L.prcnn_fps_temp_bytes(4, 16384) == 0
    assert L.prcnn_fps_temp_bytes(4, 131072) == 256 + 4 * 2 * 8 * 8


def test_argument_checks_need_no_gpu():
    """The OP_REQUIRES-style checks run before any HIP call."""
    L = _lib.lib()
    assert L.prcnn_farthest_point_sample(1, 16, 0, None, None, None, None) == -1      # npoint > 0
    assert L.prcnn_query_ball_point(1, 16, 4, 0.0, 8, None, None, None, None, None) == -1   # radius > 0
    assert L.prcnn_query_ball_point(1, 16, 4, 0.5, 0, None, None, None, None, None) == -1   # nsample > 0
    assert L.prcnn_three_nn(1, 4, 0, None, None, None, None, None) == -2              # m >= 1
    assert L.prcnn_mlp_layer(64, 8, 8, None, 8, None, None, None, 1, 24, 0, None, 8, None) == -2
