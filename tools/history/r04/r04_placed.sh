#!/bin/bash
# rb_dev_alloc_placed: its test, then bench processes with and without the placement of the output arena, alternating
cd $GRAFT_REPO_ROOT
tag=${1:-r04_placed}
mkdir -p gpurun_out/$tag
timeout 600 python -m pytest tests/test_gpu_alloc.py -x -q -m gpu 2>&1 | tail -3
{
for i in 1 2; do
for t in 4 1 "4 --placement-by sweep"; do
  RB_ALLOC_LOG=1 python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 --no-box --placement-tries $t 2>>gpurun_out/$tag/err.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tries $t', 'kernel %.3f' % d['roofline']['kernel_ms'], 'sizing', d.get('sizing_ms'), d['config'].get('out_arena_placement') and {k: v for k, v in d['config']['out_arena_placement'].items() if k != 'note'}, d.get('output_digest'))"
done
done
} 2>&1 | tee gpurun_out/$tag/log.txt
