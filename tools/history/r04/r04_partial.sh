#!/bin/bash
# do partly written lines cost time?  library variants (speculative stores rounded to whole 32 / 64 / 128-byte granules), each with and
# without the end-group patches (diagnostics build, --debug-skip 2048 = nothing off, 512 = no end groups), interleaved on one box
cd $GRAFT_REPO_ROOT
tag=${1:-r04_partial}; shift
mkdir -p gpurun_out/$tag
cp rustybam_amd/librustybam_amd.so /tmp/keep.so
run() { # variant, skip
  cp rustybam_amd/variants/$1.so rustybam_amd/librustybam_amd.so
  python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 --no-box --debug-skip $2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 skip $2', 'kernel %.3f' % d['roofline']['kernel_ms'], d.get('output_digest'))"
}
{
for round in 1 2; do
  for v in "$@"; do for s in ${SKIPS:-2048 512}; do run $v $s; done; done
done
} 2>&1 | tee gpurun_out/$tag/log.txt
cp /tmp/keep.so rustybam_amd/librustybam_amd.so
