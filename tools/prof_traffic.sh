#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the clip kernel's full-size launches (separate --pmc passes), printed per launch
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-traffic}
args=${2:-}
mkdir -p gpurun_out/$tag
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$tag -o fetch -- python3 bench.py --no-box --e2e-records 0 --steps 2 --warmup 1 --no-cpu-baseline $args > gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$tag -o write -- python3 bench.py --no-box --e2e-records 0 --steps 2 --warmup 1 --no-cpu-baseline $args > gpurun_out/${tag}_write.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
out = {}
for f in sorted(glob.glob(f"gpurun_out/{tag}/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "liftover_stream" in row["Kernel_Name"]:
            out.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
f, w = max(out["FETCH_SIZE"]), max(out["WRITE_SIZE"])
print(f"per launch: FETCH_SIZE {f:.4g} KB x2 = {f*2*1024/1e9:.2f} GB, WRITE_SIZE {w:.4g} KB = {w*1024/1e9:.2f} GB, total {(f*2+w)*1024/1e9:.2f} GB")
PY
