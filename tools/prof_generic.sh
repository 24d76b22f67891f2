#!/bin/bash
# kernel stats of the generic route: break-paf with one record in a hundred irregular, with and without the generic kernel's checkpoints
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in cp nocp; do
  if [ $v = nocp ]; then export RB_DEBUG_NO_GEN_CP=1; else unset RB_DEBUG_NO_GEN_CP; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_generic_$v -o kt -- python3 bench.py --no-box --e2e-records 0 --irregular-frac 0.01 --op break --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/prof_generic_$v.log 2>&1
  echo "== $v"; grep -h "generic\|break_pieces\|break_declined\|scan_records\|liftover_stream\|break_list" $(find gpurun_out/prof_generic_$v -name "*kernel_stats.csv") | cut -c1-150
done
