#!/bin/bash
# round-4 baseline on one box: bench line, memory-mix probe, SQ counters of the stream kernel, rocm-smi state
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_base
rocm-smi --showmemorypartition --showcomputepartition --showpower --showclocks --showmaxpower > gpurun_out/r04_base/smi.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/st_probe tools/st_probe.hip 2>/dev/null
python bench.py --steps 20 --no-cpu-baseline --e2e-records 0 2>gpurun_out/r04_base/bench.err | tail -1 > gpurun_out/r04_base/bench.json
/tmp/st_probe chunks > gpurun_out/r04_base/probe.txt 2>&1
python bench.py --steps 20 --no-cpu-baseline --e2e-records 0 2>/dev/null | tail -1 > gpurun_out/r04_base/bench2.json
bash tools/prof_sq.sh r04_base_sq "--e2e-records 0" > gpurun_out/r04_base/sq.txt 2>&1
python - <<'PY'
import json
for f in ("bench.json","bench2.json"):
    d=json.load(open("gpurun_out/r04_base/"+f)); print(f, d["ms_per_step"], d["roofline"])
print(open("gpurun_out/r04_base/probe.txt").read())
print(open("gpurun_out/r04_base/sq.txt").read())
PY
