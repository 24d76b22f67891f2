#!/bin/bash
# round 5's committed measurements behind profiles/r05_*: the record-length sweep, per-kernel times on config 4's shape, config 4 itself
# (trim-paf passes + break-paf, device resident), and the secondary workloads of the bench
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/r05_reclen.sh r05_reclen2 > gpurun_out/r05_reclen2.log 2>&1; tail -22 gpurun_out/r05_reclen2.log
bash tools/r05_kt.sh r05_kt_c4_break --workload config4-shape --op break --steps 5 --warmup 1 --placement-tries 1
bash tools/r05_kt.sh r05_kt_c4_lift --workload config4-shape --op liftover --steps 5 --warmup 1 --placement-tries 1
mkdir -p gpurun_out/r05_c4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_c4 -o kt -- python3 tools/bench_config4.py --records 10000000 > gpurun_out/r05_c4.json 2> gpurun_out/r05_c4.err
tail -1 gpurun_out/r05_c4.json | cut -c1-1500
grep -E "overlap_split|trim_select|trim_place|trim_check|liftover_stream|liftover_tile|scan_records|apply|gather|Name" gpurun_out/r05_c4/kt_kernel_stats.csv | cut -d, -f1-6
for args in "--op break" "--op break --irregular-frac 0.01" "--workload irregular --records 100000" "--workload config2" "--workload config2-lognormal"; do
  timeout -k 5 600 python3 bench.py $args --steps 10 --no-cpu-baseline --no-box --e2e-records 0 2> /dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$args', '| ms/step %.3f' % d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'unplaced', (r.get('unplaced') or {}).get('frac'), 'generic', d['generic_hits_per_gpu'])" | tee -a gpurun_out/r05_secondary.txt
done
