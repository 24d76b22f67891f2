#!/bin/bash
# does WHERE the runtime keeps the kernel's arguments matter?  HIP_FORCE_DEV_KERNARG = 0 / 1 (host-coherent system memory / device memory), alternating
cd $GRAFT_REPO_ROOT
tag=${1:-r04_env}
mkdir -p gpurun_out/$tag
for round in 1 2 3; do
for kv in "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "NONE=1"; do
  env $kv timeout 300 python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d.get('box',{}); print('$kv', 'kernel', d['roofline']['kernel_ms'], 'probe', b.get('probe_ms'), b.get('probe_scattered_ms'), 'flat', b.get('probe_flat_ms'), b.get('probe_flat_scattered_ms'), 'clock', b.get('kernel_clock_mhz'))"
done; done 2>&1 | tee gpurun_out/$tag/log.txt
