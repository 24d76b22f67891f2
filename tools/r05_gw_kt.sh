#!/bin/bash
# generic wave kernel: its own time (kernel trace) for the named variants on the irregular workload
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp rustybam_amd/librustybam_amd.so /tmp/keep_kt.so
for v in "$@"; do
  cp rustybam_amd/variants/$v.so rustybam_amd/librustybam_amd.so
  echo "== $v"; bash tools/r05_kt.sh r05_kt_gw_$v --workload irregular --records 100000 --steps 5 --warmup 1 --placement-tries 1 2>&1 | grep -E "generic|stream\(|checkpoints"
done
cp /tmp/keep_kt.so rustybam_amd/librustybam_amd.so
