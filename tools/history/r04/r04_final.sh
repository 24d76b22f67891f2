#!/bin/bash
# end of round 4: the tests added last, the committed profile of the headline step, the imbalance shapes, break-paf and the irregular workload
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_long_ops.py tests/test_gpu_cli.py -x -q -m gpu -k "2_32 or panic or pipelined" 2>&1 | tail -4
bash tools/r04_prof.sh r04_a
bash tools/r04_imb.sh r04_imb3
mkdir -p gpurun_out/r04_other
for w in "break --op break" "irregular --workload irregular" "break_irregular --op break --workload irregular"; do
  set -- $w; name=$1; shift
  timeout 600 python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 "$@" 2>gpurun_out/r04_other/$name.err | tail -1 > gpurun_out/r04_other/$name.json
  python -c "
import json; d=json.load(open('gpurun_out/r04_other/$name.json')); print('$name', 'ms/step', round(d['ms_per_step'],3), 'frac', d['roofline']['frac'], d.get('output_digest'))" || tail -3 gpurun_out/r04_other/$name.err
done
