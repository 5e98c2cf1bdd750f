#!/bin/bash
# The three bench lines kept under profiles/ (default, the driver's command, bf16 tier); results in gpurun_out/.
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err
timeout -k 10 400 python bench.py --precision bf16 --no-cpu-baseline > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err
python - <<'PY'
import json
for n in ("default", "driver", "bf16"):
    d = json.load(open("gpurun_out/bench_%s.json" % n))
    r = d["roofline"]
    print(n, d["value"], d["ms_per_step"], r["kernel"], r.get("shape"), r["frac"], d.get("single_scene_latency_ms"), d.get("value_no_dedup"))
PY
