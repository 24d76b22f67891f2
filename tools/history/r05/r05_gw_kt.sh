#!/bin/bash
# generic wave kernel: its own time (kernel trace) for the named variants on the irregular workload
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export RB_VARIANT=$v  # (rustybam_amd.capi loads variants/<name>.so; the product library is never overwritten)
  echo "== $v"; bash tools/r05_kt.sh r05_kt_gw_$v --workload irregular --records 100000 --steps 5 --warmup 1 --placement-tries 1 2>&1 | grep -E "generic|stream\(|checkpoints"
done
