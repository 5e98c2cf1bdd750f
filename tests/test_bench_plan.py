"""Host logic of bench.py's round schedule: any K is dealt into rounds exactly (no eager remainder)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("streams", [1, 2, 3, 4])
def test_round_plan_runs_exactly_k_steps(bench, streams):
    for k in list(range(0, 60)) + [96, 100, 144, 1000]:
        plan = bench.round_plan(k, streams)
        assert sum(sum(r) for r in plan) == k
        for r in plan:
            assert len(r) == streams and max(r) <= 4 and min(r) >= 0
            assert max(r) - min(r) <= 1                       # units of a round within one batch of each other
            assert sum(r) > 0
        # full rounds first, one smaller round last
        full = min(12, 4 * streams)
        assert all(sum(r) == full for r in plan[:-1])


def test_round_plan_driver_command(bench):
    assert bench.round_plan(20, 3) == [[4, 4, 4], [3, 3, 2]]   # --steps 20
    assert bench.round_plan(5, 3) == [[2, 2, 1]]               # --warmup 5
    assert bench.round_plan(48, 3) == [[4, 4, 4]] * 4          # the defaults
