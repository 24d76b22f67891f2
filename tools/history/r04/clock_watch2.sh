#!/bin/bash
# diagnostics: what do the clocks and the power read WHILE the clip kernel runs (400 launches back to back = 4 s)?
cd $GRAFT_REPO_ROOT
( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|junction" | tr '\n' ' '; echo; sleep 0.15; done ) > gpurun_out/clock_watch2_smi.log 2>&1 &
W=$!
python bench.py --no-cpu-baseline --steps 400 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('kernel_ms', d['roofline']['kernel_ms'])"
kill $W
grep -c . gpurun_out/clock_watch2_smi.log
# the samples with the highest power = taken while the kernels ran
sort -t'(' -k5 gpurun_out/clock_watch2_smi.log | awk '{print}' | grep -E "Power \(W\): [0-9]{3,4}" | sed -E 's/.*sclk clock level: [0-9]+: \(([0-9]+)Mhz\).*mclk[^(]*\(([0-9]+)Mhz\).*junction\) \(C\): ([0-9.]+).*Power \(W\): ([0-9.]+).*/sclk \1 mclk \2 junction \3 power \4/' | sort -k8 -n | tail -12
