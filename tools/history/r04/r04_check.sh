#!/bin/bash
# parity subset for the clip kernels + bench line + SQ counters of the stream kernel (one box)
cd $GRAFT_REPO_ROOT
tag=${1:-r04_chk}
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_break_onewalk.py tests/test_gpu_digest.py tests/test_gpu_fullsize.py tests/test_long_ops.py -x -q -m gpu > gpurun_out/$tag/pytest.txt 2>&1
tail -5 gpurun_out/$tag/pytest.txt
python bench.py --steps 20 --no-cpu-baseline --e2e-records 0 2>gpurun_out/$tag/bench.err | tail -1 > gpurun_out/$tag/bench.json
python bench.py --steps 20 --no-cpu-baseline --e2e-records 0 --op break 2>/dev/null | tail -1 > gpurun_out/$tag/bench_break.json
bash tools/prof_sq.sh ${tag}_sq "--e2e-records 0" > gpurun_out/$tag/sq.txt 2>&1
python - $tag <<'PY'
import json,sys
t=sys.argv[1]
for f in ("bench.json","bench_break.json"):
    d=json.load(open(f"gpurun_out/{t}/"+f)); print(f, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("output_digest"))
print(open(f"gpurun_out/{t}/sq.txt").read())
PY
