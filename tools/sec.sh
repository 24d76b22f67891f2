export RB_BENCH_VERBOSE=1
python bench.py --no-cpu-baseline --steps 5 --warmup 1 --op break 2>&1 | tail -2 | cut -c1-600
python bench.py --no-cpu-baseline --steps 5 --warmup 1 --workload config2 2>&1 | tail -2 | cut -c1-900
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/brk -o kt -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --op break > /dev/null 2>&1
head -8 gpurun_out/brk/kt_kernel_stats.csv | cut -c1-150
