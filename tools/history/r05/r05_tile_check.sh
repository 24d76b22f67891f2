#!/bin/bash
# first contact of the tile kernel with the GPU: smoke, the core parity files, then config 4's shape (every step under a timeout)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r05_tile_check}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 5 300 python3 __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $out/smoke.log
timeout -k 5 ${TEST_TIMEOUT:-1500} python3 -m pytest ${TESTS:-tests/test_gpu_parity.py tests/test_gpu_break_onewalk.py} -x -q -m gpu > $out/tests.log 2>&1; echo "tests rc=$?"; tail -15 $out/tests.log
if [ "${BENCH:-1}" = "1" ]; then
SQ=0 bash tools/r05_c4shape.sh $tag/c4 ${REC:-10000000}
fi
