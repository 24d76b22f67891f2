#!/bin/bash
# within one process: the clip kernel with its output arena, then its input ops, then its rows + workspace in K different sets of physical pages
cd $GRAFT_REPO_ROOT
tag=${1:-r04_place}
mkdir -p gpurun_out/$tag
for i in $(seq ${PROCS:-3}); do
  RB_BENCH_PLACEMENTS=${K:-4} python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 --no-box --placement-tries ${TRIES:-1} 2>gpurun_out/$tag/err$i.txt >/dev/null; grep placement gpurun_out/$tag/err$i.txt | cut -c1-150 | tee -a gpurun_out/$tag/log.txt; tail -5 gpurun_out/$tag/err$i.txt | cut -c1-300
  echo "--" | tee -a gpurun_out/$tag/log.txt
done
