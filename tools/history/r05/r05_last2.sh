#!/bin/bash
# after the generic kernel's changes: soaks on the final library, then the round's last sequence (tools/r05_last.sh)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_last2
timeout -k 5 600 python3 tests/soak/soak.py 100 2>&1 | tail -1 | tee gpurun_out/r05_last2/soaks.txt
timeout -k 5 600 python3 tests/soak/soak_long.py 2>&1 | tail -1 | tee -a gpurun_out/r05_last2/soaks.txt
timeout -k 5 600 python3 tests/soak/soak_break.py 2>&1 | tail -1 | tee -a gpurun_out/r05_last2/soaks.txt
timeout -k 5 600 python3 tests/soak/soak_tile.py 200 9000 2>&1 | tail -1 | tee -a gpurun_out/r05_last2/soaks.txt
bash tools/r05_last.sh
