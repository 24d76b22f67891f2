#!/bin/bash
# what changes on a box while bench processes come and go: probe + TLB-side counters before and after a run of processes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r04_drift}
mkdir -p gpurun_out/$tag
rocprofv3 -L 2>/dev/null | grep -iE "utcl|tlb|latency|TCP_|TCC_EA0_RD|TCC_EA0_WR|TCC_BUBBLE|TCC_TAG_STALL|MALL|GRBM_GUI" > gpurun_out/$tag/counters.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/st_probe tools/st_probe.hip 2>/dev/null
probe() { /tmp/st_probe chunks 2>&1 | grep -E "read \+ write pair \(1.2x, 2 slots\)|^read  pair|^write pair \(1 slot" | tr '\n' ';'; echo; }
pmc() { # $1 = label
  rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES --output-format csv -d gpurun_out/$tag/pmc_$1 -o c -- python3 bench.py --no-box --e2e-records 0 --steps 3 --warmup 1 --no-cpu-baseline --e2e-records 0 > gpurun_out/$tag/pmc_$1.log 2>&1
  python3 - "$tag" "$1" <<'PY'
import csv, glob, sys, collections
tag, lab = sys.argv[1], sys.argv[2]
per = collections.defaultdict(dict)
for f in sorted(glob.glob(f"gpurun_out/{tag}/pmc_{lab}/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "liftover_stream" in row["Kernel_Name"]:
            per[row["Dispatch_Id"]][row["Counter_Name"]] = per[row["Dispatch_Id"]].get(row["Counter_Name"], 0) + float(row["Counter_Value"])
if per:
    best = max(per.values(), key=lambda d: d.get("TCP_UTCL1_REQUEST", 0))
    print(lab, {k: f"{v:.4g}" for k, v in sorted(best.items())})
else:
    print(lab, "no counters")
PY
}
b() { python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench kernel %.3f' % d['roofline']['kernel_ms'])"; }
{
echo "probe0: $(probe)"
b
pmc fresh
for i in 1 2 3 4 5 6; do b; done
echo "probe1: $(probe)"
pmc drifted
b
echo "probe2: $(probe)"
} > gpurun_out/$tag/log.txt 2>&1
cat gpurun_out/$tag/log.txt; head -50 gpurun_out/$tag/counters.txt
