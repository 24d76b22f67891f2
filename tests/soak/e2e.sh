#!/bin/bash
# End-to-end check of the metric's second half: PAF-records/s from first byte read to last byte written,
# `rb liftover` (C++ host + MI355X) vs the oracle CLI (CPU) on the text form of the bench workload.
# usage: tests/soak/e2e.sh <n_records> <n_records_for_oracle>
set -e
N=${1:-20000}; NO=${2:-300}
D=${TMPDIR:-/tmp}/rb_e2e; mkdir -p $D
RB=$(dirname $0)/../../rustybam_amd/rb; OR=$(dirname $0)/../../oracle/rb_oracle
$RB synth-paf 0x5EED0003 0 $N > $D/w.paf; $RB synth-bed 3000 > $D/w.bed; head -n $NO $D/w.paf > $D/s.paf
ls -la $D/w.paf | awk '{print "paf bytes", $5}'
t0=$(date +%s.%N); RB_TIMING=1 $RB liftover --bed $D/w.bed $D/w.paf > $D/out.paf; t1=$(date +%s.%N)
python3 -c "print('rb liftover: %d records in %.2f s = %.0f records/s end to end; output %d lines' % ($N, $t1-$t0, $N/($t1-$t0), sum(1 for _ in open('$D/out.paf'))))"
t0=$(date +%s.%N); $RB liftover --bed $D/w.bed $D/s.paf > $D/s_rb.paf; t1=$(date +%s.%N)
t2=$(date +%s.%N); $OR liftover --bed $D/w.bed $D/s.paf > $D/s_or.paf 2>/dev/null; t3=$(date +%s.%N)
python3 -c "print('subset of %d records: rb %.2f s, oracle CLI (1 thread, per-base) %.2f s = %.0f records/s' % ($NO, $t1-$t0, $t3-$t2, $NO/($t3-$t2)))"
cmp $D/s_rb.paf $D/s_or.paf && echo "subset outputs byte-identical"
rm -rf $D
