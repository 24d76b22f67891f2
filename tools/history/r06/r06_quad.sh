#!/bin/bash
# round 6: the four-pairs-per-wavefront pair kernel (k_trim4.hip) -- parity first, then the pair kernels' times per setting of RB_TRIM_QUAD
# (0 = the wave-per-pair kernel for every pair, 8 / 4 = ops per lane of a region), same box, config 4's shape at $1 records
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
recs=${1:-4000000}
mkdir -p gpurun_out
if [ -z "$SKIP_PARITY" ]; then
  timeout 900 python3 -m pytest tests/test_gpu_trim.py -x -q 2>&1 | tail -5
  timeout 600 python3 tests/soak/soak_trim.py ${SOAK:-30} 2>&1 | tail -3
fi
for q in ${QUADS:-0 8 4}; do
  rm -rf gpurun_out/q4_$q
  RB_TRIM_QUAD=$q timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q4_$q -o kt -- python3 tools/bench_config4.py --records $recs > gpurun_out/q4_$q.json 2> gpurun_out/q4_$q.err
  echo "RB_TRIM_QUAD=$q"
  python3 - gpurun_out/q4_$q.json <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: r[k] for k in ("trim_passes", "trim_pairs", "trim_wall_s", "pairs_by_wave_kernel")}, r["trim_kernels_ms"]["per_pass"])
except Exception as e:
    print("no json:", e)
PY
  grep -E "overlap_split|Name" gpurun_out/q4_$q/kt_kernel_stats.csv | cut -d, -f1-6
done
