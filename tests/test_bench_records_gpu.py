"""GPU: bench.py's detection-record path with and without a process group.

`PRCNN_DIST_FORCE=1 WORLD_SIZE=1` forms a one-rank RCCL group, so the records of the timed steps
go through the same all-gather the N > 1 job issues (one collective at the closing fence of the
timed region).  The gathered records must equal, bit for bit, the records of the same steps
without a group -- SURVEY.md 8(e): "gathered detections must equal the single-GPU run's
detections scene by scene".  bench.py itself checks every replay against an eager step
(`replay_equals_eager`), so a stale or clobbered graph buffer fails both runs."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(tmp_path, name, extra_env):
    out = str(tmp_path / (name + ".npy"))
    env = dict(os.environ)
    env.update(extra_env)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "3", "--no-cpu-baseline",
           "--no-extras", "--dump-records", out]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=170)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    return line, np.load(out)


def test_records_through_one_rank_rccl_group_equal_the_ungrouped_run(cuda, tmp_path):
    plain, rec0 = _bench(tmp_path, "plain", {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    group, rec1 = _bench(tmp_path, "group", {"PRCNN_DIST_FORCE": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0",
                                             "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29631"})
    assert plain["config"]["replay_equals_eager"] and group["config"]["replay_equals_eager"]
    assert plain["config"]["collective"] is None and "all-gather" in group["config"]["collective"]
    # 6 steps x 16 scenes, 802 floats each; the grouped run returns them ordered by scene id
    assert rec0.shape == (6 * 16, 802) and rec1.shape == rec0.shape
    order = np.argsort(rec0[:, 0], kind="stable")
    assert np.array_equal(rec1, rec0[order])
    assert (rec1[:, 1] == 100).all()                       # every scene kept max_output_size boxes
    # 6 steps with K % 4 != 0: units of 4 + eager remainder steps, and still one record per scene and step
    ids, counts = np.unique(rec1[:, 0], return_counts=True)
    assert list(ids) == list(range(16)) and (counts == 6).all()


def test_two_rank_rehearsal_on_one_gpu_gathers_both_ranks_records(cuda, tmp_path):
    """The N = 2 job as the driver launches it (torch.distributed.run, one process per rank), rehearsed on
    ONE GPU: both ranks compute on cuda:0 and the group is gloo (RCCL wants a device per rank; the
    collective call and everything around it are the same code).  Rank r owns scenes 16 r .. 16 r + 15;
    the gathered records of each rank's scenes equal, bit for bit, the ungrouped run of that rank."""
    out = str(tmp_path / "w2.npy")
    env = dict(os.environ)
    env.update({"PRCNN_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PRCNN_DIST_FORCE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29633", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6",
           "--warmup", "3", "--no-cpu-baseline", "--no-extras", "--dump-records", out]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=170)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["replay_equals_eager"]
    assert "gloo" in line["config"]["collective"]
    rec = np.load(out)
    assert rec.shape == (2 * 6 * 16, 802)
    assert np.array_equal(rec[:, 0], np.sort(rec[:, 0], kind="stable"))          # ordered by scene id
    for rank in (0, 1):
        _, alone = _bench(tmp_path, "alone%d" % rank, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0",
                                                                 "PRCNN_BENCH_INPUT_RANK": str(rank)})
        mine = rec[(rec[:, 0] >= 16 * rank) & (rec[:, 0] < 16 * rank + 16)]
        assert mine.shape == alone.shape
        assert np.array_equal(mine, alone[np.argsort(alone[:, 0], kind="stable")])
