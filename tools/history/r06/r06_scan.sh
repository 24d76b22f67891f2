#!/bin/bash
# round 6: the row form of the record scan (rb_k_scan_rows, four records per wavefront) -- parity tests first, then tools/scan_time.py
# with and without it (RB_SCAN_ROWS=0: the wave-per-record kernel for everything), same box; the rows' CRCs must agree
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_PARITY" ]; then
  timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_long_ops.py tests/test_gpu_trim.py -x -q 2>&1 | tail -4
  timeout 600 python3 -m pytest tests/test_gpu_cli.py -x -q -k stats 2>&1 | tail -3
fi
for v in 0 1 0 1; do
  echo "RB_SCAN_ROWS=$v"; RB_SCAN_ROWS=$v timeout 300 python3 tools/scan_time.py 2>&1 | tail -3
done
