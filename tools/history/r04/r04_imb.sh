#!/bin/bash
# SURVEY 8(d)'s imbalance shapes of config 2 at full size + config 2 itself, with the box block
cd $GRAFT_REPO_ROOT
tag=${1:-r04_imb}
mkdir -p gpurun_out/$tag
run() { name=$1; shift
  timeout 600 python bench.py --steps 20 --no-cpu-baseline --e2e-records 0 "$@" 2>gpurun_out/$tag/$name.err | tail -1 > gpurun_out/$tag/$name.json
  python - $tag $name <<'PY'
import json,sys
t,n=sys.argv[1],sys.argv[2]
try:
    d=json.load(open(f"gpurun_out/{t}/{n}.json")); b=d.get("box",{})
    print(n, "ops/s %.4g" % d["value"], "ms/step", round(d["ms_per_step"],4), "kernel", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "hits", d["hits_per_gpu"], "generic", d["generic_hits_per_gpu"], "tail", b.get("launch_tail"), "probe", b.get("probe_ms"), b.get("error"))
except Exception as e:
    print(n, "failed", e, open(f"gpurun_out/{t}/{n}.err").read()[-600:])
PY
}
run config2 --workload config2
run config2_lognormal --workload config2-lognormal
run config2_uniform --workload config2 --placement uniform
run config2_lognormal_uniform --workload config2-lognormal --placement uniform
run config2_lognormal_1e6 --workload config2-lognormal --records 1000000
