#!/bin/bash
# a library variant (rustybam_amd/variants/<name>.so) through the parity tests (RB_TILE=0: every record through the per-record kernel), then same-box A/B
#   tools/r05_variant_check.sh <name> "<AB args>" [rounds]
cd $GRAFT_REPO_ROOT
v=$1
export RB_VARIANT=$v  # (rustybam_amd.capi loads variants/<name>.so; the product library is never overwritten)
RB_TILE=0 timeout -k 5 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_imbalance.py tests/test_gpu_digest.py -x -q -m gpu 2>&1 | tail -6
timeout -k 5 900 python3 -m pytest tests/test_gpu_tile.py -x -q -m gpu 2>&1 | tail -3
AB_ROUNDS=${3:-3} AB_ARGS="$2" bash tools/ab_so.sh cur $v 2>&1 | tail -8
