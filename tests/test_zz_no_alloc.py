"""GPU: the allocation-free entry point prcnn_nms_rotate_noalloc (static device scratch inside the
library, for callers that cannot hipMalloc) against prcnn_nms_rotate with a caller workspace,
through the C ABI: identical keep lists (index work: bit-exact), on two calls in a row -- the
second with a smaller input, so stale scratch contents of the larger call must not leak -- and
with no device allocation between the calls.  Collected last (file name) so that a failure here
cannot hide another test under `pytest -x`."""
import numpy as np
import pytest
import torch

from pointrcnn_amd import _lib, synthetic as S

pytestmark = pytest.mark.gpu

MAX_N = 9216          # PRCNN_NMS_NOALLOC_MAX_N (include/pointrcnn_hip.h)


def _buffers(cuda, n, mx):
    L = _lib.lib()
    wsb = L.prcnn_nms_rotate_workspace_bytes(n)
    ws = torch.empty(wsb + 256, dtype=torch.uint8, device=cuda)
    return {"ws": ws, "wsb": wsb, "off": (-ws.data_ptr()) % 256,
            "keep_a": torch.full((mx,), -7, dtype=torch.int64, device=cuda),
            "keep_b": torch.full((mx,), -7, dtype=torch.int64, device=cuda),
            "nk_a": torch.full((1,), -1, dtype=torch.int32, device=cuda),
            "nk_b": torch.full((1,), -1, dtype=torch.int32, device=cuda)}


def test_noalloc_entry_point_equals_workspace_entry_point(cuda):
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    cases = [(9000, 0.8, 100, 9000), (2048, 0.5, 300, 2048), (9216, 0.3, 9216, 77), (130, 0.7, 64, 130), (1, 0.5, 5, 1)]
    # every buffer of every case exists before the first call: nothing is allocated between calls
    data = []
    for n, thr, mx, seed in cases:
        boxes, scores = S.nms_boxes(seed=seed, N=n, clusters=max(2, n // 60))
        data.append((torch.as_tensor(boxes).to(cuda), torch.as_tensor(scores).to(cuda), _buffers(cuda, n, mx)))
    torch.cuda.synchronize()
    mem0 = torch.cuda.memory_allocated(cuda)
    for (n, thr, mx, _), (b, s, bufs) in zip(cases, data):
        rc = L.prcnn_nms_rotate_noalloc(n, b.data_ptr(), s.data_ptr(), thr, mx, bufs["keep_a"].data_ptr(),
                                        bufs["nk_a"].data_ptr(), st)
        assert rc == 0, L.prcnn_status_string(rc)
    assert torch.cuda.memory_allocated(cuda) == mem0
    for (n, thr, mx, _), (b, s, bufs) in zip(cases, data):
        rc = L.prcnn_nms_rotate(n, b.data_ptr(), s.data_ptr(), thr, mx, bufs["keep_b"].data_ptr(),
                                bufs["nk_b"].data_ptr(), bufs["ws"].data_ptr() + bufs["off"], bufs["wsb"], st)
        assert rc == 0, L.prcnn_status_string(rc)
    torch.cuda.synchronize()
    for (n, thr, mx, _), (_, _, bufs) in zip(cases, data):
        ka, kb = int(bufs["nk_a"].item()), int(bufs["nk_b"].item())
        assert ka == kb and 0 < ka <= min(n, mx)
        assert torch.equal(bufs["keep_a"][:ka], bufs["keep_b"][:kb])
    assert int(data[0][2]["nk_a"].item()) == 100


def test_noalloc_rejects_sizes_beyond_the_static_scratch(cuda):
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    n = MAX_N + 1
    b = torch.zeros((n, 5), dtype=torch.float64, device=cuda)
    s = torch.zeros((n,), dtype=torch.float64, device=cuda)
    keep = torch.zeros((10,), dtype=torch.int64, device=cuda)
    nk = torch.full((1,), -1, dtype=torch.int32, device=cuda)
    rc = L.prcnn_nms_rotate_noalloc(n, b.data_ptr(), s.data_ptr(), 0.5, 10, keep.data_ptr(), nk.data_ptr(), st)
    assert rc == -5 and b"not supported" in L.prcnn_status_string(rc)            # PRCNN_UNSUPPORTED
    assert L.prcnn_nms_rotate_noalloc(-1, None, None, 0.5, 10, None, None, st) == -2
    # empty input / max_output_size 0 behave as the workspace form does: n_keep = 0
    rc = L.prcnn_nms_rotate_noalloc(0, None, None, 0.5, 10, keep.data_ptr(), nk.data_ptr(), st)
    assert rc == 0 and int(nk.item()) == 0
