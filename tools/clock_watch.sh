#!/bin/bash
# diagnostics: bench.py five times in a row with rocm-smi sampled alongside (does the kernel time follow clocks / temperatures?)
cd $GRAFT_REPO_ROOT
( while true; do date +%s.%N; rocm-smi --showclocks --showtemp --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|junction|memory|Power"; sleep 0.4; done ) > gpurun_out/clock_watch_smi.log 2>&1 &
W=$!
for i in 1 2 3 4 5; do
  echo "bench $i start $(date +%s.%N)"
  python bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('kernel_ms', d['roofline']['kernel_ms'])"
  echo "bench $i end $(date +%s.%N)"
  if [ $i = 3 ]; then echo "sleep 90"; sleep 90; fi
done
kill $W
