#!/bin/bash
# on a box whose write side is slow (the probe's writes alone > 4.2 ms) compare the allocation routes of the batch's buffers
cd $GRAFT_REPO_ROOT
tag=${1:-r04_hw}
mkdir -p gpurun_out/$tag
one() { # label, env...
  lab=$1; shift
  env "$@" python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 2>/dev/null | tail -1 > gpurun_out/$tag/$lab.json
  python - gpurun_out/$tag/$lab.json $lab <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); b=d.get('box',{})
print(sys.argv[2], 'kernel', d['roofline']['kernel_ms'], 'probe mix', b.get('probe_scattered_ms'), 'reads', b.get('probe_read_only_ms'), 'writes', b.get('probe_write_only_ms'), 'k reads', b.get('kernel_reads_only_ms'), d['config'].get('batch_memory'))
PY
}
{
one first RB_X=1
w=$(python -c "import json; print(json.load(open('gpurun_out/$tag/first.json'))['box'].get('probe_write_only_ms', 0))")
if python -c "import sys; sys.exit(0 if float('$w') > 4.2 else 1)"; then
  one default RB_ALLOC_MODE=default
  one contiguous RB_ALLOC_MODE=contiguous
  one chunks RB_ALLOC_MODE=chunks
  one scatter RB_ALLOC_MODE=scatter
  one default2 RB_ALLOC_MODE=default
  one chunks2 RB_ALLOC_MODE=chunks
else
  echo "writes fast on this box: nothing to do"
fi
} 2>&1 | tee gpurun_out/$tag/log.txt
