#!/bin/bash
# SQ counters of the trim pair kernel on tools/bench_config4.py
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/c4sq
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/c4sq -o sq -- python3 tools/bench_config4.py --records 2000000 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_WR FETCH_SIZE --output-format csv -d gpurun_out/c4sq -o sq2 -- python3 tools/bench_config4.py --records 2000000 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/c4sq/*counter_collection.csv")):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "overlap_split_wave<192>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(k, "%.4g" % v)
PY
