#!/bin/bash
# on a box in the slow state (clip kernel > 10.0 ms) compare library variants; on a fast box say so and leave
cd $GRAFT_REPO_ROOT
tag=${1:-r04_hs}; shift
mkdir -p gpurun_out/$tag
k=$(python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 --no-box 2>/dev/null | tail -1 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
echo "first process: kernel $k ms" | tee gpurun_out/$tag/log.txt
if python -c "import sys; sys.exit(0 if float('$k') > 10.0 else 1)"; then
  AB_ROUNDS=${AB_ROUNDS:-3} bash tools/r04_ab.sh ${tag}_ab "$@" 2>&1 | tail -5 | tee -a gpurun_out/$tag/log.txt
else
  echo "fast box: nothing to do" | tee -a gpurun_out/$tag/log.txt
fi
