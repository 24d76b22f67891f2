#!/bin/bash
# rocprofv3 passes over tools/bench_nucfreq.py: kernel trace + stats, HBM traffic, instruction mix / LDS counters of rb_k_nf_tiles
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r02_nf}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o kt -- python3 tools/bench_nucfreq.py --steps 5 > gpurun_out/${tag}_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$tag -o fetch -- python3 tools/bench_nucfreq.py --steps 2 > gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$tag -o write -- python3 tools/bench_nucfreq.py --steps 2 > gpurun_out/${tag}_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$tag -o sq -- python3 tools/bench_nucfreq.py --steps 2 > gpurun_out/${tag}_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/$tag -o lds -- python3 tools/bench_nucfreq.py --steps 2 > gpurun_out/${tag}_lds.log 2>&1
tail -1 gpurun_out/${tag}_kt.log | cut -c1-300
python3 - <<'PY'
import csv, glob, collections, os
tag = os.environ.get("TAG", "")
for f in sorted(glob.glob("gpurun_out/%s/**/*counter_collection.csv" % (tag or "*"), recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "nf_tiles" in r["Kernel_Name"]:
            a = acc[r["Kernel_Name"].split("(")[0][:40] + " " + r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in acc.items():
        print(os.path.basename(f)[:12], k, "per launch: %.4g" % (v / max(n, 1) * (1 if True else 1)), "rows", n)
PY
