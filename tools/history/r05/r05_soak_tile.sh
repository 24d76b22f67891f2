#!/bin/bash
# the tile kernel's randomised soak (tests/soak/soak_tile.py): $1 cases from seed $2
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_soak_tile
timeout -k 5 ${3:-1200} python3 tests/soak/soak_tile.py ${1:-60} ${2:-0} > gpurun_out/r05_soak_tile/soak_${2:-0}.log 2>&1; echo "soak_tile rc=$?"; tail -5 gpurun_out/r05_soak_tile/soak_${2:-0}.log | cut -c1-400
