#!/bin/bash
# bench.py after a change to the script alone: its tests, then the default line
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
d=gpurun_out/${1:-r05_bench_check}; mkdir -p $d
timeout -k 5 1500 python3 -m pytest tests/test_gpu_multi.py -x -q > $d/tests.log 2>&1; echo "tests rc=$?"; tail -3 $d/tests.log
timeout -k 5 900 python3 bench.py > $d/bench.json 2> $d/bench.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('$d/bench.json').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('ms/step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'unplaced', r.get('unplaced'), 'traffic', r.get('traffic'))
print('arena', c.get('out_arena_placement')); print('ops', c.get('ops_placement'))" | cut -c1-900
