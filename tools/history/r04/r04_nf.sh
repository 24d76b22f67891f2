#!/bin/bash
# nucfreq: parity tests, then config 5's call with the persistent byte-counter kernel off / 2 / 3 / 4 workgroups per CU
cd $GRAFT_REPO_ROOT
tag=${1:-r04_nf}
mkdir -p gpurun_out/$tag
timeout 1200 python -m pytest tests/test_gpu_nucfreq.py -x -q -m gpu 2>&1 | tail -3
for pw in 0 3 2 4 3 0; do
  RB_NF_PERSIST=$pw timeout 300 python tools/bench_nucfreq.py --steps 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('persist $pw', {k: d[k] for k in d if 'ms' in k or 'frac' in k or 'checks' in k or 'digest' in k or 'roofline' in k})"
done 2>&1 | tee gpurun_out/$tag/log.txt
