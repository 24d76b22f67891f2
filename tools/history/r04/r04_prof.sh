#!/bin/bash
# the round's committed profile of the headline step + the unprofiled bench line of the same box
cd $GRAFT_REPO_ROOT
tag=${1:-r04_a}
bash tools/prof_round.sh $tag "--e2e-records 0 --no-box" > gpurun_out/${tag}_prof.log 2>&1
tail -6 gpurun_out/${tag}_prof.log
python bench.py --no-cpu-baseline --e2e-records 0 2>/dev/null | tail -1 > gpurun_out/${tag}_bench.json
python -c "
import json; d=json.load(open('gpurun_out/${tag}_bench.json')); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['box'].get('kernel_clock_mhz'), d['box'].get('probe_ms'), d['box'].get('probe_scattered_ms'), d['config'].get('batch_buffers_chunked'))"
