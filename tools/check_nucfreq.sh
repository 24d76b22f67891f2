#!/bin/bash
# round 6, nucfreq: parity of the pipelined tile kernel (tests + soak), then same-box A/B against the serial loop (variants nf_old / nf_new)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_nf
timeout 900 python -m pytest tests/test_gpu_nucfreq.py -x -q -m gpu > gpurun_out/r06_nf/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r06_nf/tests.log
timeout 900 python tests/soak/soak_nucfreq.py ${SOAK:-40} > gpurun_out/r06_nf/soak.log 2>&1; echo "soak rc=$?"; tail -3 gpurun_out/r06_nf/soak.log
bash tools/ab_generic.sh nf "$@" 2>&1 | tee gpurun_out/r06_nf/ab.log
