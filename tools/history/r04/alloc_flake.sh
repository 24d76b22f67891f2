#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== nucfreq + alloc"; python -m pytest tests/test_gpu_nucfreq.py tests/test_gpu_alloc.py -x -q -m gpu 2>&1 | tail -3
echo "== cli + alloc"; python -m pytest tests/test_gpu_cli.py tests/test_gpu_alloc.py -x -q -m gpu 2>&1 | tail -3
echo "== alloc alone"; python -m pytest tests/test_gpu_alloc.py -x -q -m gpu 2>&1 | tail -3
echo "== alloc twice"; python -m pytest tests/test_gpu_alloc.py tests/test_gpu_alloc.py -x -q -m gpu 2>&1 | tail -3
