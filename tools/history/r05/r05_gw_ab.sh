#!/bin/bash
# generic wave kernel, timing only: variants against the product on the irregular workload, same box
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
AB_ROUNDS=2 AB_ARGS="--workload irregular --records 100000 --no-box --placement-tries 1 --e2e-records 0" bash tools/ab_so.sh "$@" 2>&1 | tail -12
