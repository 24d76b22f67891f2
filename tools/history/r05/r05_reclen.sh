#!/bin/bash
# the clip kernels over the record length: mean ops per record in {250, 500, 1k, 2k, 5k} (uniform in 0.6 .. 1.4 of the mean, 5e9 ops in all),
# liftover and break-paf, tile kernel on (RB_SHORT_MAX as given) and off -- profiles/r05_reclen_summary.md
#   tools/r05_reclen.sh <tag> ["<short_max values>"] ["<means>"]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r05_reclen}
smx=${2:-"0 2048"}
means=${3:-"250 500 1000 2000 5000"}
out=gpurun_out/$tag
mkdir -p $out
echo "mean_ops records op short_max ms_per_step kernel_ms frac unplaced digest" > $out/table.txt
for mean in $means; do
  lo=$(( mean * 6 / 10 )); hi=$(( mean * 14 / 10 )); rec=$(( 5000000000 / mean ))
  for op in liftover break; do
    for sm in $smx; do
      if [ "$sm" = "0" ]; then export RB_TILE=0; unset RB_SHORT_MAX; else unset RB_TILE; export RB_SHORT_MAX=$sm; fi
      f=$out/${op}_${mean}_${sm}.json
      timeout -k 5 600 python3 bench.py --workload config4-shape --op $op --records $rec --ops-lo $lo --ops-hi $hi --steps 5 --warmup 1 --no-cpu-baseline --no-box --placement-tries 1 > $f 2> $out/${op}_${mean}_${sm}.err
      python3 - $f $mean $rec $op $sm >> $out/table.txt <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], round(d["ms_per_step"], 3), r["kernel_ms"], r["frac"], (r.get("unplaced") or {}).get("frac"), d["output_digest"])
except Exception as e:
    print(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], "failed:", e)
PY
    done
  done
done
unset RB_TILE RB_SHORT_MAX
cat $out/table.txt
