#!/bin/bash
# the record scan after a change: every test that goes through it, the soaks, then timing of the named variants (tools/r05_scan_ab.sh)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export RB_VARIANT=$1  # (rustybam_amd.capi loads variants/<name>.so; the product library is never overwritten)
timeout -k 5 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_trim.py tests/test_long_ops.py tests/test_gpu_digest.py -x -q -m gpu 2>&1 | tail -3
timeout -k 5 600 python3 tests/soak/soak.py 100 2>&1 | tail -1
timeout -k 5 600 python3 tests/soak/soak_long.py 2>&1 | tail -1
timeout -k 5 600 python3 tests/soak/soak_trim.py 2>&1 | tail -1
bash tools/r05_scan_ab.sh "$@"
