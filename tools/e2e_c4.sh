#!/bin/bash
# BASELINE.json config 4 as the config states it, as TEXT and at size: `rb trim-paf | rb break-paf --max-size 100` (README.md:22-23;
# src/paf.rs:210-305, src/main.rs:218-230, :271-281) on N records of SURVEY 8(d)'s whole-genome PAF (`rb synth-paf config4`: 25 contigs,
# 4 records per query, ~500 ops a record), 1 GPU and `--gpus 2 / 3` (workers on one device here: a 1-GPU lease), byte-identity between N.
# usage: tools/e2e_c4.sh [records] [tag]
cd $GRAFT_REPO_ROOT
n=${1:-10000000}
tag=${2:-e2e_c4}
d=/dev/shm/rb_c4_$$
mkdir -p $d gpurun_out/$tag
RB=rustybam_amd/rb
S=gpurun_out/$tag/summary.txt
t0=$(date +%s.%N)
$RB synth-paf config4 $n > $d/w.paf
t1=$(date +%s.%N)
echo "synth: $(awk "BEGIN{print $t1 - $t0}") s, $(stat -c %s $d/w.paf) bytes, $(wc -l < $d/w.paf) records" | tee $S
run() { # name, then the pipeline as a shell string using $d
  name=$1; shift
  rm -f $d/out_$name.paf
  s=$(date +%s.%N)
  bash -c "$*" > $d/out_$name.paf 2> gpurun_out/$tag/$name.err
  rc=$?
  e=$(date +%s.%N)
  echo "$name: rc $rc, $(awk "BEGIN{printf \"%.3f s, %.0f records/s\", $e - $s, $n / ($e - $s)}"), out $(stat -c %s $d/out_$name.paf) bytes, $(wc -l < $d/out_$name.paf) lines" | tee -a $S
}
run trim1 "$RB trim-paf $d/w.paf"
run pipe1 "$RB trim-paf $d/w.paf | $RB break-paf --max-size 100 -"
run pipe2 "RB_GPUS_SAME_DEVICE=1 $RB --gpus 2 trim-paf $d/w.paf | RB_GPUS_SAME_DEVICE=1 $RB --gpus 2 break-paf --max-size 100 -"
run pipe3 "RB_GPUS_SAME_DEVICE=1 $RB --gpus 3 trim-paf $d/w.paf | RB_GPUS_SAME_DEVICE=1 $RB --gpus 3 break-paf --max-size 100 -"
run trim3 "RB_GPUS_SAME_DEVICE=1 $RB --gpus 3 trim-paf $d/w.paf"
for f in pipe2 pipe3; do cmp -s $d/out_pipe1.paf $d/out_$f.paf && echo "$f: same bytes as one GPU" || echo "$f: DIFFERENT from one GPU"; done | tee -a $S
cmp -s $d/out_trim1.paf $d/out_trim3.paf && echo "trim3: same bytes as one GPU" | tee -a $S || echo "trim3: DIFFERENT from one GPU" | tee -a $S
md5sum $d/out_pipe1.paf $d/out_trim1.paf | sed "s#$d/##" | tee -a $S
# the oracle CLI on the first 2000 queries (8000 records): the same bytes as the head of the GPU pipeline's output restricted to them
head -8000 $d/w.paf > $d/head.paf
oracle/rb_oracle trim-paf $d/head.paf 2>/dev/null | oracle/rb_oracle break-paf --max-size 100 - 2>/dev/null > $d/o_head.paf
$RB trim-paf $d/head.paf 2>/dev/null | $RB break-paf --max-size 100 - 2>/dev/null > $d/g_head.paf
cmp -s $d/o_head.paf $d/g_head.paf && echo "first 8000 records: rb pipeline = oracle CLI pipeline ($(wc -l < $d/o_head.paf) lines)" | tee -a $S || echo "first 8000 records: rb pipeline DIFFERS from the oracle CLI" | tee -a $S
tail -3 gpurun_out/$tag/pipe1.err | tee -a $S
rm -rf $d
