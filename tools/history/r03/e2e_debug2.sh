#!/bin/bash
cd $GRAFT_REPO_ROOT
n=${1:-300000}
d=/dev/shm/rb_dbg_$$
mkdir -p $d gpurun_out/e2e_debug
RB=rustybam_amd/rb
$RB synth-paf 0x5EED0003 0 $n > $d/w.paf
$RB synth-bed 3000 > $d/w.bed
RB_DEBUG_TEXT_CHECK=1 RB_PIPE_WORKERS=1 RB_CHUNK_MB=256 $RB liftover --bed $d/w.bed $d/w.paf > $d/o.paf 2> gpurun_out/e2e_debug/check.err
grep "text check" gpurun_out/e2e_debug/check.err
rm -rf $d
