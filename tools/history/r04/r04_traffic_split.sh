#!/bin/bash
# where the clip kernel's bytes beyond the algorithmic ones come from: FETCH_SIZE / WRITE_SIZE (separate passes) and the launch time
# of the diagnostics build with parts switched off (512: no end groups, 64: no speculative stores), and of the line-rounding variant
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r04_split}
mkdir -p gpurun_out/$tag
cp rustybam_amd/librustybam_amd.so /tmp/keep.so
one() { # label, debug-skip
  lab=$1; skip=$2
  k=$(python bench.py --steps 5 --no-cpu-baseline --e2e-records 0 --no-box --debug-skip $skip 2>/dev/null | tail -1 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/$tag/${lab}_$c -o c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-records 0 --no-box --debug-skip $skip > gpurun_out/$tag/${lab}_$c.log 2>&1
  done
  python3 - "$tag" "$lab" "$k" <<'PY'
import csv, glob, sys, collections
tag, lab, k = sys.argv[1:4]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(float)
    for f in glob.glob(f"gpurun_out/{tag}/{lab}_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "liftover_stream" in row["Kernel_Name"] and row["Counter_Name"] == c:
                per[row["Dispatch_Id"]] += float(row["Counter_Value"])
    out[c] = max(per.values()) if per else None
fe, wr = out["FETCH_SIZE"], out["WRITE_SIZE"]
print(lab, "kernel ms", k, "fetch GB (x2)", None if fe is None else round(fe * 1024 * 2 / 1e9, 3), "write GB", None if wr is None else round(wr * 1024 / 1e9, 3))
PY
}
{
one product 0
one diag 2048
one no_endgroups 512
one no_spec_stores 64
one neither 576
if [ -f rustybam_amd/variants/lineround.so ]; then
  cp rustybam_amd/variants/lineround.so rustybam_amd/librustybam_amd.so
  one lineround 0
  one lineround_no_endgroups 512
  cp /tmp/keep.so rustybam_amd/librustybam_amd.so
fi
} 2>&1 | tee gpurun_out/$tag/log.txt
cp /tmp/keep.so rustybam_amd/librustybam_amd.so
