#!/bin/bash
# the randomised soaks against the oracle (the tile kernel takes most of their records), then the round's last profile of the headline
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_soak
timeout -k 5 900 python3 tests/soak/soak.py 150 > gpurun_out/r05_soak/soak.log 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r05_soak/soak.log
timeout -k 5 900 python3 tests/soak/soak_break.py > gpurun_out/r05_soak/soak_break.log 2>&1; echo "soak_break rc=$?"; tail -2 gpurun_out/r05_soak/soak_break.log
timeout -k 5 600 python3 tests/soak/soak_long.py > gpurun_out/r05_soak/soak_long.log 2>&1; echo "soak_long rc=$?"; tail -2 gpurun_out/r05_soak/soak_long.log
timeout -k 5 600 python3 -m pytest tests/test_gpu_tile.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
bash tools/prof_round.sh r05_b > gpurun_out/r05_b_round.log 2>&1; tail -3 gpurun_out/r05_b_round.log | cut -c1-300
timeout -k 5 900 python3 bench.py > gpurun_out/r05_b_bench.json 2> gpurun_out/r05_b_bench.err; echo "bench rc=$?"
