#!/bin/bash
# nucfreq after the 16-bit build's list (tests, the call's time, per-kernel times) and config 4's device stages unprofiled (the stage's wall)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_nucfreq.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
  python3 tools/bench_config4.py --records 10000000 2>/dev/null | tail -1 > gpurun_out/r05_c4_plain_$i.json
  python3 - gpurun_out/r05_c4_plain_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
k = d["trim_kernels_ms"]
print("c4 unprofiled: trim_wall_s", d["trim_wall_s"], "kernels", round(k["selection"] + k["pair_kernels"] + k["apply_and_check"], 2), "ratio", d["trim_wall_over_kernels"],
      "break_wall_s", d["break_wall_s"], "pieces", d["break_pieces"])
PY
done
python3 tools/bench_nucfreq.py 2>/dev/null | tail -1 | cut -c1-900
mkdir -p gpurun_out/r05_nf
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_nf -o kt -- python3 tools/bench_nucfreq.py > gpurun_out/r05_nf.json 2> gpurun_out/r05_nf.err
grep -E "rb_k_nf|Name" gpurun_out/r05_nf/kt_kernel_stats.csv | cut -d, -f1-6
