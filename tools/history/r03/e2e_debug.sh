#!/bin/bash
# which factor makes a pipelined run differ from the whole-file run (chunk size? workers?), and where the first difference lies
cd $GRAFT_REPO_ROOT
n=${1:-300000}
d=/dev/shm/rb_dbg_$$
mkdir -p $d gpurun_out/e2e_debug
RB=rustybam_amd/rb
$RB synth-paf 0x5EED0003 0 $n > $d/w.paf
$RB synth-bed 3000 > $d/w.bed
RB_NO_PIPELINE=1 $RB liftover --bed $d/w.bed $d/w.paf > $d/ref.paf
for cfg in "1 256" "2 256" "3 256" "4 256" "4 512" "4 128" "6 128"; do
  set -- $cfg
  RB_PIPE_WORKERS=$1 RB_CHUNK_MB=$2 $RB liftover --bed $d/w.bed $d/w.paf > $d/o.paf; rc=$?
  if cmp -s $d/ref.paf $d/o.paf; then echo "workers $1 chunk $2: rc $rc same"; else
    echo "workers $1 chunk $2: rc $rc DIFFERENT: $(cmp $d/ref.paf $d/o.paf | head -1)"
    cp $d/o.paf $d/bad.paf
  fi
done 2>&1 | tee gpurun_out/e2e_debug/summary.txt
if [ -f $d/bad.paf ]; then
  python3 - $d/ref.paf $d/bad.paf <<'PY' | tee -a gpurun_out/e2e_debug/summary.txt
import sys
a, b = open(sys.argv[1], 'rb'), open(sys.argv[2], 'rb')
n = 0
bad = 0
for la, lb in zip(a, b):
    n += 1
    if la != lb:
        bad += 1
        if bad <= 5:
            fa, fb = la.split(b'\t'), lb.split(b'\t')
            print('line', n, 'ref', fa[:13], len(la), 'got', fb[:13], len(lb))
            ca, cb = fa[-1], fb[-1]
            k = next((i for i in range(min(len(ca), len(cb))) if ca[i] != cb[i]), None)
            print('   cigar differs at byte', k, ca[max(0, (k or 0) - 20):(k or 0) + 20], cb[max(0, (k or 0) - 20):(k or 0) + 20])
print('lines compared', n, 'different', bad)
PY
fi
rm -rf $d
