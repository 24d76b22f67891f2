#!/bin/bash
# same-box A/B of library variants on an arbitrary command that prints one JSON line with "ms_per_step" (tools/bench_nucfreq.py ...)
# or on the text kernels (kernel trace of `rb liftover`, whole-file route).  usage: ab_generic.sh nf|text <variant> ...
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
what=$1; shift
if [ "$what" = "text" ]; then
  D=/tmp/rb_ab_text; mkdir -p $D
  rustybam_amd/rb synth-paf 0x5EED0003 0 100000 > $D/w.paf; rustybam_amd/rb synth-bed 3000 > $D/w.bed
fi
for round in 1 2 3; do
  for n in "$@"; do
    export RB_VARIANT=$n  # (rustybam_amd.capi loads variants/<name>.so; the product library is never overwritten)
    if [ "$what" = "nf" ]; then
      python tools/bench_nucfreq.py --steps 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n', 'ms_per_step %.3f' % d['ms_per_step'])"
    else
      rm -rf gpurun_out/ab_text_$n
      LD_PRELOAD=$GRAFT_REPO_ROOT/rustybam_amd/variants/$n.so RB_NO_PIPELINE=1 RB_FULL_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_text_$n -o kt -- rustybam_amd/rb liftover --bed $D/w.bed $D/w.paf > $D/out.paf 2>/dev/null
      python - $n <<'PY'
import csv, sys
n = sys.argv[1]
rows = {r["Name"]: r for r in csv.DictReader(open(f"gpurun_out/ab_text_{n}/kt_kernel_stats.csv"))}
print(n, {k.split("(")[0].replace("void ", ""): round(float(v["AverageNs"]) / 1e6, 3) for k, v in rows.items() if "cigars" in k})
PY
    fi
  done
done
