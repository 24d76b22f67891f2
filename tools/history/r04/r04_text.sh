#!/bin/bash
# the format kernel with an op's text as one 8-byte LDS store: its tests, then the two library variants on one box
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_text.py tests/test_long_ops.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -4
bash tools/ab_generic.sh text fw0 fw1 2>&1 | tail -8
