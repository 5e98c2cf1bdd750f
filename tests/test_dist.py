"""Multi-rank path on CPU: world_size 2 over gloo (the GPU path uses the same code over
RCCL).  Scene sharding is a partition; the all-gather of detection records returns, on
every rank, exactly the records a single process would have produced, in scene order."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pointrcnn_amd import dist as pdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene_records(scene_ids):
    """Deterministic fake detections of a scene (what NMS + the heads would emit)."""
    S, N = len(scene_ids), 50
    if S == 0:
        return torch.empty((0, pdist.RECORD_LEN), dtype=torch.float32)
    boxes = torch.stack([torch.arange(N * 7, dtype=torch.float64).reshape(N, 7) + 1000 * s for s in scene_ids])
    scores = torch.stack([torch.linspace(1, 0, N, dtype=torch.float64) + s for s in scene_ids])
    keep = torch.stack([torch.arange(pdist.MAX_DET) % N for _ in scene_ids]).to(torch.int64)
    n_keep = torch.tensor([(3 * s + 1) % 40 for s in scene_ids], dtype=torch.int32)
    return pdist.pack_detections(torch.tensor(scene_ids, dtype=torch.float32), boxes, scores, keep, n_keep)


def _worker(rank, world, port, n_scenes, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = pdist.init_from_env("cpu")
    mine = pdist.shard_scenes(n_scenes, r, w)
    rec = _scene_records(mine)
    if n_scenes % w:
        # unequal shards: every rank pads to ceil(n/world) rows with sentinel records
        out = pdist.all_gather_detections(rec, n_scenes=n_scenes)
        q.put((rank, out.numpy().copy()))
        dist.barrier()
        dist.destroy_process_group()
        return
    out = pdist.all_gather_detections(rec)
    # the asynchronous form (bench.py: records of several replays in one collective, completed later)
    two = torch.cat([rec, rec + 0.0], 0)
    raw, work = pdist.all_gather_detections(two, async_op=True)
    work.wait()
    assert torch.equal(pdist.sort_by_scene(raw)[::2], out)
    q.put((rank, out.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_scenes_is_a_partition():
    for world in (1, 2, 4, 8):
        parts = [pdist.shard_scenes(128, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(128))
        assert len({len(p) for p in parts}) == 1
        assert all(s % world == r for r, p in enumerate(parts) for s in p)


def test_pack_unpack_roundtrip():
    rec = _scene_records([4, 9])
    assert rec.shape == (2, pdist.RECORD_LEN)
    un = pdist.unpack_detections(rec)
    assert [u[0] for u in un] == [4, 9]
    assert un[0][1].shape == ((3 * 4 + 1) % 40, 7) and un[1][2].shape == ((3 * 9 + 1) % 40,)
    assert float(rec[0, 2 + 8 * 13:].abs().sum()) == 0.0          # entries past n_keep are zero


import pytest


@pytest.mark.parametrize("n_scenes", [8, 7, 1])
def test_all_gather_world2_equals_single_process(n_scenes):
    """8: equal shards; 7 and 1: n_scenes % world != 0 (one rank holds one scene fewer, or none)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_scenes, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, "rank failed (see its traceback above)"
    got = dict(q.get(timeout=10) for _ in range(world))
    want = _scene_records(list(range(n_scenes))).numpy()
    for r in range(world):
        assert np.array_equal(got[r], want)
