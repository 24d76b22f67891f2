#!/bin/bash
# the read side of the clip kernel (diagnostics build: 576 = no store of clipped ops at all, 64 = no speculative stores, 2048 = everything)
# for the load-ring variants, one box
cd $GRAFT_REPO_ROOT
tag=${1:-r04_readside}; shift
mkdir -p gpurun_out/$tag
cp rustybam_amd/librustybam_amd.so /tmp/keep.so
{
for round in 1 2; do
for v in "$@"; do
  cp rustybam_amd/variants/$v.so rustybam_amd/librustybam_amd.so
  for skip in 576 64 2048; do
  python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 --no-box --debug-skip $skip 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v skip $skip', 'kernel %.3f' % d['roofline']['kernel_ms'])"
  done
done
done
} 2>&1 | tee gpurun_out/$tag/log.txt
cp /tmp/keep.so rustybam_amd/librustybam_amd.so
