#!/bin/bash
# does it matter where slot 1 lies relative to slot 0 inside a 2 MB physical chunk?  (RB_SLOT_PAD: ops added to the slot stride)
cd $GRAFT_REPO_ROOT
tag=${1:-r04_slotpad}
mkdir -p gpurun_out/$tag
{
for round in 1 2; do
for pad in 0 1024 8192 65536 131072 262144 393216 520192; do
  RB_SLOT_PAD=$pad python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 --no-box 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pad $pad', 'kernel %.3f' % d['roofline']['kernel_ms'], d.get('output_digest'))"
done
done
} 2>&1 | tee gpurun_out/$tag/log.txt
