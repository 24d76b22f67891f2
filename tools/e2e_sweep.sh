#!/bin/bash
# pipelined route at the headline size: workers x chunk size x host threads (see tools/e2e_big.sh).  usage: tools/e2e_sweep.sh [records] [tag]
cd $GRAFT_REPO_ROOT
n=${1:-1000000}
tag=${2:-e2e_sweep}
d=/dev/shm/rb_e2e_$$
mkdir -p $d gpurun_out/$tag
RB=rustybam_amd/rb
$RB synth-paf 0x5EED0003 0 $n > $d/w.paf
$RB synth-bed 3000 > $d/w.bed
run() {
  name=$1; shift
  for rep in 1 2; do
    rm -f $d/out.paf
    s=$(date +%s.%N)
    env RB_TIMING=1 "$@" $RB liftover --bed $d/w.bed $d/w.paf > $d/out.paf 2> gpurun_out/$tag/${name}_$rep.err
    rc=$?
    e=$(date +%s.%N)
    echo "$name run $rep: rc $rc, $(awk "BEGIN{printf \"%.3f s, %.0f records/s\", $e - $s, $n / ($e - $s)}"), md5 $(md5sum < $d/out.paf | cut -c1-8)" | tee -a gpurun_out/$tag/summary.txt
  done
}
for w in 1 2 3; do for c in 256 512 1024; do run w${w}_c${c} RB_PIPE_WORKERS=$w RB_CHUNK_MB=$c; done; done
run w2_c512_t16 RB_PIPE_WORKERS=2 RB_THREADS=16
run w2_c512_t32 RB_PIPE_WORKERS=2 RB_THREADS=32
run w3_c512_t16 RB_PIPE_WORKERS=3 RB_THREADS=16
run w2_c384 RB_PIPE_WORKERS=2 RB_CHUNK_MB=384
run w2_c768 RB_PIPE_WORKERS=2 RB_CHUNK_MB=768
rm -rf $d
