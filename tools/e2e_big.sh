#!/bin/bash
# end to end at the headline size (PAF-records/s = input records / wall time from the first byte read to the last byte
# written): `rb liftover` on the text of config 3 -- N records (default 1e6: 14.7 GB in, 19 GB out) x 3000 windows -- in /dev/shm,
# pipelined route vs whole-file route, RB_TIMING phases on stderr.  usage: tools/e2e_big.sh [records] [tag]
cd $GRAFT_REPO_ROOT
n=${1:-1000000}
tag=${2:-e2e_big}
d=/dev/shm/rb_e2e_$$
mkdir -p $d gpurun_out/$tag
RB=rustybam_amd/rb
t0=$(date +%s.%N)
$RB synth-paf 0x5EED0003 0 $n > $d/w.paf
$RB synth-bed 3000 > $d/w.bed
t1=$(date +%s.%N)
echo "synth: $(awk "BEGIN{print $t1 - $t0}") s, $(stat -c %s $d/w.paf) bytes" | tee gpurun_out/$tag/summary.txt
run() { # name, env..., -- args
  name=$1; shift
  for rep in 1 2; do
    rm -f $d/out_$name.paf   # (the shell truncating 19 GB of an earlier run would be timed as part of this one)
    s=$(date +%s.%N)
    env RB_TIMING=1 "$@" > $d/out_$name.paf 2> gpurun_out/$tag/${name}_$rep.err
    rc=$?
    e=$(date +%s.%N)
    echo "$name run $rep: rc $rc, $(awk "BEGIN{printf \"%.3f s, %.0f records/s\", $e - $s, $n / ($e - $s)}"), out $(stat -c %s $d/out_$name.paf) bytes" | tee -a gpurun_out/$tag/summary.txt
  done
}
run pipelined $RB liftover --bed $d/w.bed $d/w.paf
run whole RB_NO_PIPELINE=1 $RB liftover --bed $d/w.bed $d/w.paf
run pipe_prealloc RB_PREALLOC=1 $RB liftover --bed $d/w.bed $d/w.paf
run pipe_wr32 RB_WRITE_THREADS=32 $RB liftover --bed $d/w.bed $d/w.paf
run pipe_w4_512 RB_PIPE_WORKERS=4 $RB liftover --bed $d/w.bed $d/w.paf
run pipe_w5_512 RB_PIPE_WORKERS=5 $RB liftover --bed $d/w.bed $d/w.paf
run pipe_w6_384 RB_PIPE_WORKERS=6 RB_CHUNK_MB=384 $RB liftover --bed $d/w.bed $d/w.paf
run pipe_w4_first RB_PIPE_WORKERS=4 RB_FIRST_CHUNK_MB=64 $RB liftover --bed $d/w.bed $d/w.paf
run pipe_w2 RB_PIPE_WORKERS=2 $RB liftover --bed $d/w.bed $d/w.paf
run pipe_w4 RB_PIPE_WORKERS=4 RB_CHUNK_MB=256 $RB liftover --bed $d/w.bed $d/w.paf
run gpus2_same RB_GPUS_SAME_DEVICE=1 $RB --gpus 2 liftover --bed $d/w.bed $d/w.paf
run break_pipelined $RB break-paf --max-size 100 $d/w.paf
for f in $d/out_*.paf; do cmp -s $d/out_whole.paf $f && echo "$(basename $f): same bytes as the whole-file route" || echo "$(basename $f): DIFFERENT from the whole-file route"; done | tee -a gpurun_out/$tag/summary.txt
md5sum $d/out_whole.paf $d/out_break_pipelined.paf | tee -a gpurun_out/$tag/summary.txt
rm -rf $d
