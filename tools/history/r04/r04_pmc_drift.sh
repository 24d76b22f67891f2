#!/bin/bash
# wave-cycle accounting of the clip kernel in the first and in a late process of a row on one box (SQ counters only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r04_pmcd}
mkdir -p gpurun_out/$tag
rocprofv3 -L 2>/dev/null | grep -iE "Counter_Name.*(ICACHE|IFETCH|SQC_|SQ_WAIT|SQ_INST_LEVEL|SQ_BUSY)" | awk '{print $3}' | tr '\n' ' ' > gpurun_out/$tag/sq_counters.txt
pmc() { # label, counters...
  lab=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$tag/pmc_$lab -o c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --e2e-records 0 --no-box > gpurun_out/$tag/pmc_$lab.log 2>&1
  python3 - "$tag" "$lab" <<'PY'
import csv, glob, sys, collections
tag, lab = sys.argv[1], sys.argv[2]
per = collections.defaultdict(dict)
for f in sorted(glob.glob(f"gpurun_out/{tag}/pmc_{lab}/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "liftover_stream" in row["Kernel_Name"]:
            per[row["Dispatch_Id"]][row["Counter_Name"]] = per[row["Dispatch_Id"]].get(row["Counter_Name"], 0) + float(row["Counter_Value"])
if per:
    best = max(per.values(), key=lambda d: d.get("SQ_WAVE_CYCLES", 0))
    print(lab, {k: f"{v:.4g}" for k, v in sorted(best.items())})
else:
    print(lab, "no counters", open(f"gpurun_out/{tag}/pmc_{lab}.log").read()[-300:])
PY
}
b() { python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 --no-box 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench kernel %.3f' % d['roofline']['kernel_ms'])"; }
{
cat gpurun_out/$tag/sq_counters.txt; echo
b
pmc a1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_BUSY_CYCLES
pmc b1 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR
for i in 1 2 3 4 5 6 7 8; do b; done
pmc a2 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_BUSY_CYCLES
pmc b2 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR
b
} 2>&1 | tee gpurun_out/$tag/log.txt
