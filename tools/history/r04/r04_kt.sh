#!/bin/bash
# the kernel-trace pass of the round's profile again (tools/prof_round.sh, first pass), after bench.py's last change
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r04_b}
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o kt -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --e2e-records 0 --no-box > gpurun_out/${tag}_kt.log 2>&1
tail -1 gpurun_out/${tag}_kt.log | cut -c1-300
