#!/bin/bash
# diagnostics: time the stream kernel with phases skipped (results invalid; timing only)
for d in "$@"; do
  python bench.py --no-cpu-baseline --steps 5 --warmup 1 --debug-skip $d 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('skip=$d kernel_ms', d['roofline']['kernel_ms'], 'step_ms', round(d['ms_per_step'],3))"
done
