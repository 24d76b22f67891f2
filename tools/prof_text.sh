#!/bin/bash
# kernel trace of `rb liftover` (text in -> text out) on the text form of the bench workload
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
N=${1:-100000}
D=/tmp/rb_e2e_prof; mkdir -p $D
rustybam_amd/rb synth-paf 0x5EED0003 0 $N > $D/w.paf; rustybam_amd/rb synth-bed 3000 > $D/w.bed
RB_NO_PIPELINE=1 RB_FULL_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/text_prof -o kt -- rustybam_amd/rb liftover --bed $D/w.bed $D/w.paf > $D/out.paf 2> gpurun_out/text_prof.log
head -12 gpurun_out/text_prof/kt_kernel_stats.csv | cut -c1-140
ls -la $D/w.paf $D/out.paf | awk '{print $5, $9}'
rm -rf $D
