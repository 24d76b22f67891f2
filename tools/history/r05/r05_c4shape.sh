#!/bin/bash
# config 4's record shape through the clip kernels: timings, then SQ counters (bench.py starts no child under a profiler)
#   tools/r05_c4shape.sh <tag> [records] [extra bench args]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r05_c4shape}
rec=${2:-10000000}
extra=${3:-}
out=gpurun_out/$tag
mkdir -p $out
for op in break liftover; do
  timeout 900 python3 bench.py --workload config4-shape --op $op --records $rec --steps 5 --warmup 1 --no-cpu-baseline --no-box --placement-tries 1 $extra > $out/${op}.json 2> $out/${op}.err
  tail -c 1500 $out/${op}.err
  python3 - $out/${op}.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(d["metric"][:60], "| ms/step", round(d["ms_per_step"], 3), "kernel_ms", r["kernel_ms"], "frac", r["frac"], "hits", d["hits_per_gpu"], "generic", d["generic_hits_per_gpu"], "out_ops", d["out_ops_per_gpu"])
except Exception as e:
    print("no line:", e)
PY
done
if [ "${SQ:-1}" = "1" ]; then
prec=$(( rec < 2000000 ? rec : 2000000 ))
for op in break liftover; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $out/sq_$op -o a -- python3 bench.py --workload config4-shape --op $op --records $prec --steps 2 --warmup 1 --no-cpu-baseline --no-box --placement-tries 1 $extra > $out/sq_${op}_a.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d $out/sq_$op -o b -- python3 bench.py --workload config4-shape --op $op --records $prec --steps 2 --warmup 1 --no-cpu-baseline --no-box --placement-tries 1 $extra > $out/sq_${op}_b.log 2>&1
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for op in ("break", "liftover"):
    per = collections.defaultdict(dict)
    for f in sorted(glob.glob(f"{out}/sq_{op}/**/*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            if "liftover_stream" in row["Kernel_Name"] or "liftover_tile" in row["Kernel_Name"]:
                per[(f.split("/")[-1][0], row["Kernel_Name"][:40], row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"])
    # the last dispatch of each file
    last = {}
    for (fl, kn, did), v in per.items():
        k = (fl, kn)
        if k not in last or int(did) > last[k][0]:
            last[k] = (int(did), v)
    for k, (did, v) in sorted(last.items()):
        print(op, k, {a: f"{b:.4g}" for a, b in v.items()})
PY
fi
