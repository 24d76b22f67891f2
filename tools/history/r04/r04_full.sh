#!/bin/bash
# what the driver does at round end, on one box: pytest -m gpu, smoke, the default bench line
cd $GRAFT_REPO_ROOT
tag=${1:-r04_full}
mkdir -p gpurun_out/$tag
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/$tag/pytest.txt 2>&1
tail -4 gpurun_out/$tag/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
RB_ALLOC_LOG=1 python bench.py 2>gpurun_out/$tag/bench.err | tail -1 > gpurun_out/$tag/bench.json
python - $tag <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/{sys.argv[1]}/bench.json"))
print("value %.4g" % d["value"], "ms/step", round(d["ms_per_step"],3), d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["kernel_ms_steps"])
print("box", json.dumps(d.get("box")))
print("parity", d.get("parity_sample"), "|", d.get("parity_full"))
print("cpu", {k:(v if not isinstance(v,dict) else '...') for k,v in d.get("cpu_baseline",{}).items()})
print("e2e", d.get("e2e_paf_records_per_s"), d["config"].get("batch_buffers_chunked"))
PY
grep -h "rb_dev_alloc" gpurun_out/$tag/bench.err | head
