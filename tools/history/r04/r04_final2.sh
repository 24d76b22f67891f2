#!/bin/bash
# final binary, output arena placed: the imbalance shapes, break-paf, the irregular workload
cd $GRAFT_REPO_ROOT
bash tools/r04_imb.sh r04_imb4
mkdir -p gpurun_out/r04_other2
for w in "break --op break" "break_irregular1pct --op break --irregular-frac 0.01"; do
  set -- $w; name=$1; shift
  timeout 600 python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 "$@" 2>gpurun_out/r04_other2/$name.err | tail -1 > gpurun_out/r04_other2/$name.json
  python -c "
import json; d=json.load(open('gpurun_out/r04_other2/$name.json')); print('$name', 'ms/step', round(d['ms_per_step'],3), 'frac', d['roofline']['frac'], d.get('output_digest'), d['config'].get('out_arena_placement') and {k: v for k, v in d['config']['out_arena_placement'].items() if k != 'note'})" || tail -3 gpurun_out/r04_other2/$name.err
done
