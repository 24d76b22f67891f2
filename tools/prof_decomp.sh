#!/bin/bash
# VALU / SALU of rb_k_liftover_stream with phases of the kernel switched off (bench.py --debug-skip: 1 no emission, 2 no
# resolution, 4 no streaming, 64 no store masks): which part of the record's instruction stream costs what.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-decomp}
mkdir -p gpurun_out/$tag
# (the production kernel has its diagnostics compiled out: this script wants the -DRB_DIAG=1 build, tools/mkvariant.sh diag -DRB_DIAG=1)
export RB_VARIANT=diag  # (rustybam_amd.capi loads variants/<name>.so; the product library is never overwritten)
for skip in 0 2 1 3 4 64; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/$tag/s$skip -o sq -- python3 bench.py --no-box --e2e-records 0 --steps 2 --warmup 1 --no-cpu-baseline --debug-skip $skip > gpurun_out/$tag/s${skip}.log 2>&1
  python3 - "$tag" "$skip" <<'PY' >> gpurun_out/$tag/summary.txt
import csv, glob, sys, collections
tag, skip = sys.argv[1], sys.argv[2]
per = collections.defaultdict(dict)
for f in sorted(glob.glob(f"gpurun_out/{tag}/s{skip}/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "liftover_stream" in row["Kernel_Name"]:
            per[row["Dispatch_Id"]][row["Counter_Name"]] = float(row["Counter_Value"])
best = max(per.values(), key=lambda d: d.get("SQ_INSTS_VALU", 0)) if per else {}
print("debug-skip", skip, {k: f"{v:.4g}" for k, v in sorted(best.items())})
PY
  rm -rf gpurun_out/$tag/s$skip
done
cat gpurun_out/$tag/summary.txt
