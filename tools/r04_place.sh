#!/bin/bash
# within one process: the clip kernel with its output arena in K different sets of physical pages; three processes
cd $GRAFT_REPO_ROOT
tag=${1:-r04_place}
mkdir -p gpurun_out/$tag
for i in 1 2 3; do
  RB_BENCH_PLACEMENTS=${K:-4} python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 --no-box 2>&1 >/dev/null | grep placement | tee -a gpurun_out/$tag/log.txt
  echo "--" | tee -a gpurun_out/$tag/log.txt
done
