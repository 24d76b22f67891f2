#!/bin/bash
# a row of bench processes on one box, each with its `box` block: does the memory-mix probe on the SAME buffers drift with the kernel?
cd $GRAFT_REPO_ROOT
tag=${1:-r04_drift2}; n=${2:-8}
mkdir -p gpurun_out/$tag
for i in $(seq $n); do
  timeout 300 python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 2>gpurun_out/$tag/err_$i.txt | tail -1 > gpurun_out/$tag/b_$i.json
  python - $tag $i <<'PY'
import json,sys
t,i=sys.argv[1],sys.argv[2]
try:
    d=json.load(open(f"gpurun_out/{t}/b_{i}.json")); b=d.get("box",{})
    print(i, "kernel", d["roofline"]["kernel_ms"], d["roofline"]["kernel_ms_steps"], "clock", b.get("kernel_clock_mhz"), "probe", b.get("probe_ms"), b.get("probe_scattered_ms"), "k/p", b.get("kernel_over_probe"), "chunked", d["config"].get("batch_buffers_chunked"), b.get("error"))
except Exception as e:
    print(i, "failed", e, open(f"gpurun_out/{t}/err_{i}.txt").read()[-800:])
PY
done 2>&1 | tee gpurun_out/$tag/log.txt
