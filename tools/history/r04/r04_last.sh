#!/bin/bash
# the bench line with arena and input placed, and the tests that run bench.py
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_last
RB_ALLOC_LOG=1 python bench.py 2>gpurun_out/r04_last/bench.err | tail -1 > gpurun_out/r04_last/bench.json
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r04_last/bench.json"))
    print("value %.4g" % d["value"], "ms/step", round(d["ms_per_step"],3), d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["kernel_ms_steps"])
    print({k:v for k,v in d["config"]["out_arena_placement"].items() if k!="note"}, d["config"].get("ops_placement"))
    print("parity", d.get("parity_sample"), "|", d.get("parity_full")); print(d.get("output_digest"), d["roofline"].get("traffic"))
except Exception as e:
    print("failed", e); print(open("gpurun_out/r04_last/bench.err").read()[-1500:])
PY
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_bench_launch.py -x -q -m gpu 2>&1 | tail -3
