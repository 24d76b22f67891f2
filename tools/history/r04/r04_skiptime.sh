#!/bin/bash
# wall time of the clip kernel's diagnostics build with phases switched off (bench.py --debug-skip; results invalid, times meaningful)
cd $GRAFT_REPO_ROOT
tag=${1:-r04_skip}
mkdir -p gpurun_out/$tag
for skip in 0 2 1 3 64 66 0; do
  python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 --no-box --debug-skip $skip 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('skip $skip', 'kernel %.3f' % d['roofline']['kernel_ms'], d['roofline']['kernel_ms_steps'])"
done 2>&1 | tee gpurun_out/$tag/log.txt
