#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats) of a bench.py command; the stats file lands in gpurun_out/<tag>/kt_kernel_stats.csv
#   tools/r05_kt.sh <tag> "<bench args>"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o kt -- python3 bench.py --no-cpu-baseline --no-box --e2e-records 0 "$@" > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python3 - gpurun_out/$tag <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/kt_kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:9.4f} ms  min {float(r['MinNs'])/1e6:9.4f}  max {float(r['MaxNs'])/1e6:9.4f}  {float(r['Percentage']):5.1f} %")
PY
