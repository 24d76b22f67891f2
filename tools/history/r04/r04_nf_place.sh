#!/bin/bash
# does the placement of nucfreq's counts array (4 GB written per call) matter?  alternating processes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_nf_place
{
for i in 1 2 3; do for t in 1 4; do
  python tools/bench_nucfreq.py --steps 10 --placement-tries $t 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tries $t', 'ms/call %.3f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'])"
done; done
} 2>&1 | tee gpurun_out/r04_nf_place/log.txt
