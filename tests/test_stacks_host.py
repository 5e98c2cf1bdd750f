"""Host-side helpers of pointrcnn_amd.stacks that carry precomputed geometry between streams /
graph replays (CPU tensors; no kernel is called)."""
import torch

from pointrcnn_amd import stacks


def _fake_geometry(b=3):
    balls, nn = [], []
    for (m, _, nsamples, _) in stacks.RPN_SA:
        balls.append([(torch.randint(0, 100, (b, m, ns), dtype=torch.int32), torch.randint(0, ns + 1, (b, m), dtype=torch.int32))
                      for ns in nsamples])
    for n in (256, 1024, 4096, 16384):
        nn.append((torch.randint(0, 64, (b, n, 3), dtype=torch.int32), torch.rand((b, n, 3))))
    return {"balls": balls, "nn": nn}


def test_geometry_tensor_list_round_trip_and_slice():
    geo = _fake_geometry()
    flat = stacks.geometry_tensors(geo)
    assert len(flat) == 2 * sum(len(lv[1]) for lv in stacks.RPN_SA) + 2 * 4
    back = stacks.geometry_from_tensors(geo, flat)
    for a, b in zip(stacks.geometry_tensors(back), flat):
        assert a is b
    part = stacks.geometry_slice(geo, 1, 3)
    for a, b in zip(stacks.geometry_tensors(part), flat):
        assert a.shape[0] == 2 and torch.equal(a, b[1:3]) and a.is_contiguous()
    clone = stacks.geometry_clone(geo)
    for a, b in zip(stacks.geometry_tensors(clone), flat):
        assert a is not b and torch.equal(a, b)


def test_flat_copy_set_views_share_one_buffer():
    tens = stacks.geometry_tensors(_fake_geometry(2)) + [torch.rand((2, 64, 3))]
    flat_a, va = stacks.flat_copy_set(tens)
    flat_b, vb = stacks.flat_copy_set(tens)
    assert flat_a.dtype == torch.uint8 and flat_a.numel() == flat_b.numel() and flat_a.numel() % 256 == 0
    for t, v in zip(tens, va):
        assert v.dtype == t.dtype and v.shape == t.shape and torch.equal(v, t)
        assert (v.data_ptr() - flat_a.data_ptr()) % 256 == 0
    for v in vb:
        v.zero_()
    flat_b.copy_(flat_a)                      # ONE copy moves the whole set
    for t, v in zip(tens, vb):
        assert torch.equal(v, t)
