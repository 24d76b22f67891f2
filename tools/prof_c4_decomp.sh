#!/bin/bash
# where the pair kernel's time goes: variants that end a pair early (k_trim.hip RB_TW_STOP = 1..5; tools/mkvariant.sh stop<k> --src
# k_trim.hip -DRB_TW_STOP=<k>; stop0 = the library as it is), same box, first pass only (the later passes of a variant with wrong rows
# are meaningless).  Prints the first launch's duration of rb_k_overlap_split_wave<192>.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in stop0 stop1 stop2 stop3 stop4 stop5 stop0; do
  export RB_VARIANT=$n  # (rustybam_amd.capi loads variants/<name>.so; the product library is never overwritten)
  rm -rf gpurun_out/c4d_$n
  RB_C4_ONE_PASS=1 RB_DEBUG_TRIM_NO_SERIAL=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c4d_$n -o kt -- python3 tools/bench_config4.py --records 4000000 > /dev/null 2>&1
  f=$(find gpurun_out/c4d_$n -name "*kernel_trace.csv" | head -1)
  python3 - "$n" "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[2])) if "overlap_split_wave<192>" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(sys.argv[1], "first launch ms", round((int(rows[0]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6, 3) if rows else None, "launches", len(rows))
PY
done
