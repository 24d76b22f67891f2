#!/bin/bash
# same-box A/B of library variants on the config-4 pair kernel (kernel trace of tools/bench_config4.py at 4e6 records)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for n in "$@"; do
    export RB_VARIANT=$n  # (rustybam_amd.capi loads variants/<name>.so; the product library is never overwritten)
    rm -rf gpurun_out/ab_c4_$n
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_c4_$n -o kt -- python3 tools/bench_config4.py --records 4000000 > /dev/null 2>&1
    f=$(find gpurun_out/ab_c4_$n -name "*kernel_stats.csv" | head -1)
    echo "$n $(grep 'overlap_split_wave<192>' $f | cut -d, -f2-4)"
  done
done
