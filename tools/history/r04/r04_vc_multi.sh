#!/bin/bash
# parity tests of the clip kernels with a library variant in place of the product library, then the variants against each other
cd $GRAFT_REPO_ROOT
v=$1; tag=$2; shift 2
mkdir -p gpurun_out/$tag
cp rustybam_amd/librustybam_amd.so /tmp/keep_vc.so
cp rustybam_amd/variants/$v.so rustybam_amd/librustybam_amd.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_break_onewalk.py tests/test_gpu_digest.py tests/test_long_ops.py tests/test_gpu_imbalance.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -4
RB_FULLSIZE_RECORDS=1000000 timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "liftover or break_paf_integrity or config2" 2>&1 | tail -3
cp /tmp/keep_vc.so rustybam_amd/librustybam_amd.so
SKIPS="0" bash tools/r04_partial.sh ${tag}_ab "$@" 2>&1 | tail -20
