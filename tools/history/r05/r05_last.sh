#!/bin/bash
# the round's last run: the driver's sequence on the final tree, then the profile of the headline (traffic hash-keyed to these sources), the secondary lines
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/r05_full.sh r05_last
bash tools/prof_round.sh r05_c > gpurun_out/r05_c_round.log 2>&1; tail -2 gpurun_out/r05_c_round.log | cut -c1-200
rm -f gpurun_out/r05_secondary2.txt
for args in "--op break" "--op break --irregular-frac 0.01" "--workload irregular --records 100000" "--workload config4-shape" "--workload config4-shape --op break"; do
  timeout -k 5 600 python3 bench.py $args --steps 10 --no-cpu-baseline --no-box --e2e-records 0 2> /dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$args', '| ms/step %.3f' % d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'unplaced', (r.get('unplaced') or {}).get('frac'), 'generic', d['generic_hits_per_gpu'])" | tee -a gpurun_out/r05_secondary2.txt
done
