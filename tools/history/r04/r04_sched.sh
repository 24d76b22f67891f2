#!/bin/bash
# the launch order of the records (RB_SCHED) on one box, with the box block each time
cd $GRAFT_REPO_ROOT
tag=${1:-r04_sched}; shift
mkdir -p gpurun_out/$tag
for round in 1 2; do
for sc in "$@"; do
  if [ "$sc" = longest ]; then unset RB_SCHED; else export RB_SCHED=$sc; fi
  timeout 300 python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 $SCHED_ARGS 2>gpurun_out/$tag/err.txt | tail -1 > gpurun_out/$tag/b.json
  python - $tag "$sc" <<'PY'
import json,sys
t,i=sys.argv[1],sys.argv[2]
try:
    d=json.load(open(f"gpurun_out/{t}/b.json")); b=d.get("box",{})
    print(i, "kernel", d["roofline"]["kernel_ms"], "step", round(d["ms_per_step"],3), "clock", b.get("kernel_clock_mhz"), "probe", b.get("probe_ms"), b.get("probe_scattered_ms"), "k/p", b.get("kernel_over_probe"), "chunked", d["config"].get("batch_buffers_chunked"), "tail95", (b.get("launch_tail") or {}).get("ms_after_95pct_of_waves_done"), d.get("output_digest"), b.get("error"))
except Exception as e:
    print(i, "failed", e, open(f"gpurun_out/{t}/err.txt").read()[-800:])
PY
done; done 2>&1 | tee gpurun_out/$tag/log.txt
