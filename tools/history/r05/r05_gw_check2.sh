#!/bin/bash
# the generic wave kernel of the LIBRARY (the plain step of pass 3, groups of two, four waves): parity (wild / mixed batches, long ops, the
# irregular paths of break-paf, the soaks), then same-box timing against the named variants
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 5 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_long_ops.py tests/test_gpu_break_onewalk.py tests/test_gpu_tile.py tests/test_gpu_imbalance.py -x -q -m gpu 2>&1 | tail -4
timeout -k 5 600 python3 tests/soak/soak.py 100 2>&1 | tail -1
timeout -k 5 600 python3 tests/soak/soak_long.py 2>&1 | tail -1
timeout -k 5 600 python3 tests/soak/soak_break.py 2>&1 | tail -1
timeout -k 5 600 python3 tests/soak/soak_tile.py 100 5000 2>&1 | tail -1
bash tools/r05_gw_ab.sh "$@"
AB_ROUNDS=1 AB_ARGS="--op break --irregular-frac 0.01 --no-box --placement-tries 1 --e2e-records 0" bash tools/ab_so.sh "$@" 2>&1 | grep "step" | tail -4
