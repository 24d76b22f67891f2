#!/bin/bash
# SQ counters of the text kernels on the text form of the bench workload
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
N=${1:-100000}
D=/tmp/rb_e2e_prof; mkdir -p $D
rustybam_amd/rb synth-paf 0x5EED0003 0 $N > $D/w.paf; rustybam_amd/rb synth-bed 3000 > $D/w.bed
rm -rf gpurun_out/text_sq
RB_FULL_EXIT=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/text_sq -o sq -- rustybam_amd/rb liftover --bed $D/w.bed $D/w.paf > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/text_sq/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "cigars" in k:
            acc[k[:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in acc.items():
        cyc = d["GRBM_GUI_ACTIVE"] / 8
        print(k, {c: "%.3g" % v for c, v in d.items()}, "VALU busy %.0f%%" % (100 * d["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024)), "LDS busy %.0f%%" % (100 * d["SQ_ACTIVE_INST_LDS"] * 4 / (cyc * 256)))
PY
rm -rf $D
