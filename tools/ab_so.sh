#!/bin/bash
# same-box A/B of prebuilt library variants (rustybam_amd/variants/<name>.so, tools/mkvariant.sh): bench.py kernel time, AB_ROUNDS
# interleaved rounds (default 4); the same binary differs by +-0.7 ms between two processes on one box (where the driver places the
# 20 / 30 GB buffers), so min and median over the rounds are what to compare
cd $GRAFT_REPO_ROOT
rounds=${AB_ROUNDS:-4}
rm -f /tmp/ab_times.txt
# (odd rounds run the variants in the order given, even rounds in reverse: a box whose times creep from process to process -- some
#  do, by 10 % over a dozen processes -- then favours nobody)
fwd="$*"; rev=""; for n in "$@"; do rev="$n $rev"; done
for round in $(seq $rounds); do
  if [ $((round % 2)) -eq 1 ]; then order="$fwd"; else order="$rev"; fi
  for n in $order; do
    export RB_VARIANT=$n  # (rustybam_amd.capi loads variants/<name>.so; the product library is never overwritten)
    python bench.py --steps 10 --no-cpu-baseline $AB_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n', 'step %.3f' % d['ms_per_step'], 'kernel %.3f' % d['roofline']['kernel_ms'], d.get('output_digest'))" | tee -a /tmp/ab_times.txt
  done
done
python - <<'PY'
import collections, statistics
t = collections.defaultdict(list)
for ln in open('/tmp/ab_times.txt'):
    f = ln.split()
    t[f[0]].append(float(f[4]))
for k, v in t.items():
    print(f"{k:>14}: kernel ms min {min(v):.3f}  median {statistics.median(v):.3f}  max {max(v):.3f}  ({len(v)} runs)")
PY
