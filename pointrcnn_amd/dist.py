"""Scene sharding across the GPUs of one node + the one collective of the path.

The reference is single-GPU.  Every op is independent per scene, so
scenes are dealt round-robin to ranks (one process per GPU) and the only exchange is an
all-gather of fixed-size per-scene detection records over RCCL/xGMI (backend "nccl" on
ROCm; "gloo" for CPU tests).  Record layout (float32, RECORD_LEN values):
    [scene_id, n_keep, MAX_DET x (x, y, z, l, h, w, ry, score)]
~3.2 KB per scene: the collective is latency-bound, one all-gather per batch is enough.
"""
import contextlib
import os
import sys

import torch
import torch.distributed as dist

MAX_DET = 100
RECORD_LEN = 2 + MAX_DET * 8


@contextlib.contextmanager
def _stdout_to_stderr():
    """RCCL prints a version banner on file descriptor 1 when a communicator is created; callers
    (bench.py) promise ONE JSON line on stdout, so the group is formed with fd 1 pointed at fd 2."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def init_from_env(device_type="cuda"):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun).
    Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # PRCNN_DIST_FORCE=1: form the (RCCL) group even for one rank, so that a 1-GPU box exercises
    # the collective exactly as the N>1 job issues it
    force = os.environ.get("PRCNN_DIST_FORCE") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # PRCNN_DIST_BACKEND=gloo: rehearsal of the N>1 path on a box with fewer GPUs than ranks
        # (ranks then share devices, local rank modulo the device count); RCCL is the product path
        backend = os.environ.get("PRCNN_DIST_BACKEND", "nccl") if device_type == "cuda" else "gloo"
        if device_type == "cuda":
            if backend != "nccl":
                local = local % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local)
        with _stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local))
                # create the communicator now (not inside the first timed collective)
                dist.barrier(device_ids=[local])
                torch.cuda.synchronize()
            else:
                dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_scenes(n_scenes, rank, world):
    """Scene s is owned by rank s mod world."""
    return list(range(rank, n_scenes, world))


def pack_detections(scene_ids, boxes3d, scores, keep, n_keep):
    """Build the (S, RECORD_LEN) float32 record tensor on the device, without host syncs.
    boxes3d (S,N,7), scores (S,N), keep (S,MAX_DET) int64 (entries past n_keep ignored),
    n_keep (S,) int."""
    S = boxes3d.shape[0]
    dev = boxes3d.device
    if boxes3d.is_cuda:
        # one HIP launch (prcnn_pack_detections); the torch composition below serves host tensors
        from . import _lib
        ids = torch.as_tensor(scene_ids, dtype=torch.float32, device=dev).contiguous()
        b3 = boxes3d.to(torch.float64).contiguous()
        sc = scores.to(torch.float64).contiguous()
        kp = keep.to(torch.int64).contiguous()
        nk = n_keep.to(torch.int32).contiguous()
        if kp.shape != (S, MAX_DET):
            raise ValueError("keep must be (S, %d)" % MAX_DET)
        rec = torch.empty((S, RECORD_LEN), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().prcnn_pack_detections(S, b3.shape[1], MAX_DET, ids.data_ptr(), b3.data_ptr(),
                                                    sc.data_ptr(), kp.data_ptr(), nk.data_ptr(), rec.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream), "pack_detections")
        return rec
    valid = torch.arange(MAX_DET, device=dev).unsqueeze(0) < n_keep.view(S, 1)
    k = torch.where(valid, keep, torch.zeros_like(keep))
    b = torch.gather(boxes3d, 1, k.unsqueeze(-1).expand(S, MAX_DET, 7)).to(torch.float32)
    s = torch.gather(scores, 1, k).to(torch.float32).unsqueeze(-1)
    det = torch.cat([b, s], dim=-1) * valid.unsqueeze(-1)
    rec = torch.empty((S, RECORD_LEN), dtype=torch.float32, device=dev)
    rec[:, 0] = torch.as_tensor(scene_ids, dtype=torch.float32, device=dev)
    rec[:, 1] = n_keep.to(torch.float32)
    rec[:, 2:] = det.reshape(S, MAX_DET * 8)
    return rec


def sort_by_scene(gathered):
    """Rank-major gathered records -> ordered by scene id (column 0)."""
    return gathered[torch.argsort(gathered[:, 0], stable=True)]


SENTINEL_SCENE = -1.0        # scene id of padding records (dropped after the sort)
MAX_SCENE_ID = 1 << 24       # scene ids travel as float32: exact up to 2**24


def pad_records(records, rows):
    """records (S_local, RECORD_LEN) padded to `rows` rows with sentinel records (scene id -1): every
    rank must hand the same number of rows to all_gather_into_tensor, and round-robin shards differ by
    one scene when n_scenes % world != 0."""
    s = records.shape[0]
    if s > rows:
        raise ValueError("pad_records: %d records do not fit %d rows" % (s, rows))
    if s == rows:
        return records
    pad = torch.zeros((rows - s, records.shape[1]), dtype=records.dtype, device=records.device)
    pad[:, 0] = SENTINEL_SCENE
    return torch.cat([records, pad], 0)


def all_gather_detections(records, async_op=False, n_scenes=None):
    """records (S_local, RECORD_LEN) -> (world*S_local, RECORD_LEN) ordered by scene id.
    One all-gather; without a process group it is the identity.
    n_scenes: total number of scenes over all ranks when the shards may be unequal
    (shard_scenes with n_scenes % world != 0): every rank pads to ceil(n_scenes / world) rows with
    sentinel records, which sort first and are cut off, so (n_scenes, RECORD_LEN) comes back.
    Without it all ranks must pass the same S_local (checked against the gathered size by torch).
    async_op=True: returns (gathered in rank-major order, work handle) without making the current
    stream wait for the collective; call work.wait() before reading, sort_by_scene() to order
    (and drop the first world*rows - n_scenes rows if n_scenes was given)."""
    if not dist.is_initialized():
        return (records, None) if async_op else records
    world = dist.get_world_size()
    if n_scenes is not None:
        if n_scenes > MAX_SCENE_ID:
            raise ValueError("scene ids travel as float32 and are exact only up to 2**24")
        records = pad_records(records, (n_scenes + world - 1) // world)
    flat = records.contiguous().reshape(-1)
    out = torch.empty(world * flat.numel(), dtype=records.dtype, device=records.device)
    if async_op:
        work = dist.all_gather_into_tensor(out, flat, async_op=True)
        return out.reshape(world * records.shape[0], records.shape[1]), work
    dist.all_gather_into_tensor(out, flat)          # rank-major concatenation
    res = sort_by_scene(out.reshape(world * records.shape[0], records.shape[1]))
    if n_scenes is not None:
        res = res[res.shape[0] - n_scenes:]         # sentinel records (scene id -1) sort first
    return res


def unpack_detections(records):
    """Host-side view: list of (scene_id, boxes (k,7), scores (k,))."""
    r = records.detach().cpu()
    res = []
    for row in r:
        k = int(row[1].item())
        d = row[2:].reshape(MAX_DET, 8)[:k]
        res.append((int(row[0].item()), d[:, :7].numpy(), d[:, 7].numpy()))
    return res
