#!/bin/bash
# rocprofv3 runs behind profiles/r<round>_*: kernel trace + stats, FETCH_SIZE and WRITE_SIZE in separate --pmc passes, SQ counters
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r03_a}
args=${2:-}
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o kt -- python3 bench.py --no-box --e2e-records 0 --steps 10 --warmup 2 --no-cpu-baseline $args > gpurun_out/${tag}_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$tag -o fetch -- python3 bench.py --no-box --e2e-records 0 --steps 2 --warmup 1 --no-cpu-baseline $args > gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$tag -o write -- python3 bench.py --no-box --e2e-records 0 --steps 2 --warmup 1 --no-cpu-baseline $args > gpurun_out/${tag}_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$tag -o sq -- python3 bench.py --no-box --e2e-records 0 --steps 2 --warmup 1 --no-cpu-baseline $args > gpurun_out/${tag}_sq.log 2>&1

# per-kernel sums of every counter file
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
for f in sorted(glob.glob(f"gpurun_out/{tag}/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:40]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, d in acc.items():
        if "liftover_stream" in k or "liftover_tile" in k or "copy_clips" in k:
            print(f.split("/")[-1], k, {c: f"{v:.4g}" for c, v in d.items()})
PY
tail -1 gpurun_out/${tag}_kt.log
