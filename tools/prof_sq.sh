#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-sq}
args=${2:-}
mkdir -p gpurun_out/$tag
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/$tag -o sq -- python3 bench.py --no-box --e2e-records 0 --steps 2 --warmup 1 --no-cpu-baseline $args > gpurun_out/${tag}_sq.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
per = collections.defaultdict(dict)
for f in sorted(glob.glob(f"gpurun_out/{tag}/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "liftover_stream" in row["Kernel_Name"]:
            per[row["Dispatch_Id"]][row["Counter_Name"]] = float(row["Counter_Value"])
best = max(per.values(), key=lambda d: d.get("SQ_INSTS_VALU", 0))
print({k: f"{v:.4g}" for k, v in best.items()})
PY
