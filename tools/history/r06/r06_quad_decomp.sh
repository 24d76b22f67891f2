#!/bin/bash
# where the four-pairs-per-wavefront kernel's time goes: library variants that end a pair early (k_trim4.hip RB_Q4_STOP = 1 .. 4;
# tools/mkvariant.sh q4stop<k> --src k_trim4.hip -DRB_Q4_STOP=<k>; q4full = the file as it is), selected with RB_VARIANT, same box,
# first pass only (the later passes of a variant with wrong rows are meaningless).  Prints the large launches of the quad kernel.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
recs=${1:-4000000}
for n in ${VARIANTS:-q4full q4stop1 q4stop2 q4stop3 q4stop4 q4full}; do
  rm -rf gpurun_out/q4d_$n
  RB_VARIANT=$n RB_TRIM_QUAD=${RB_TRIM_QUAD:-8} RB_C4_ONE_PASS=1 RB_DEBUG_TRIM_NO_SERIAL=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/q4d_$n -o kt -- python3 tools/bench_config4.py --records $recs > /dev/null 2>&1
  f=$(find gpurun_out/q4d_$n -name "*kernel_trace.csv" | head -1)
  python3 - "$n" "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[2])) if "overlap_split_quad" in r["Kernel_Name"]]
d = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows), reverse=True)
print(sys.argv[1], "quad launches ms", [round(x, 3) for x in d[:3]])
PY
done
