#!/bin/bash
# round 6: config 4 as one device pipeline (tools/bench_config4.py) -- the plain line, then the same command under rocprofv3 for the
# per-kernel split; $1 = records (default 1e7)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
recs=${1:-10000000}
mkdir -p gpurun_out/r06_c4
timeout 900 python3 tools/bench_config4.py --records $recs --gather > gpurun_out/r06_c4_bench.json 2> gpurun_out/r06_c4_bench.err
tail -1 gpurun_out/r06_c4_bench.json | cut -c1-3000
tail -3 gpurun_out/r06_c4_bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_c4 -o kt -- python3 tools/bench_config4.py --records $recs > gpurun_out/r06_c4_prof_bench.json 2> gpurun_out/r06_c4_prof.err
cut -d, -f1-6 gpurun_out/r06_c4/kt_kernel_stats.csv | head -40
