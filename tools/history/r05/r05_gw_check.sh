#!/bin/bash
# the generic wave kernel's variants: parity first (wild / mixed batches, long ops, the irregular paths of break-paf), then same-box A/B on the irregular workload
cd $GRAFT_REPO_ROOT
export RB_VARIANT=${1:-gw4}  # (rustybam_amd.capi loads variants/<name>.so; the product library is never overwritten)
timeout -k 5 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_long_ops.py tests/test_gpu_break_onewalk.py tests/test_gpu_tile.py -x -q -m gpu 2>&1 | tail -4
timeout -k 5 600 python3 tests/soak/soak.py 60 2>&1 | tail -1
shift
AB_ROUNDS=2 AB_ARGS="--workload irregular --records 100000 --no-box --placement-tries 1 --e2e-records 0" bash tools/ab_so.sh "$@" 2>&1 | tail -8
AB_ROUNDS=1 AB_ARGS="--op break --irregular-frac 0.01 --no-box --placement-tries 1 --e2e-records 0" bash tools/ab_so.sh cur gw4 2>&1 | grep "step" | tail -3
