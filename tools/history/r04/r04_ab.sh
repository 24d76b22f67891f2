#!/bin/bash
# same-box A/B of library variants + the memory-mix probe of the box: tools/r04_ab.sh <tag> <variant>...
cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/$tag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/st_probe tools/st_probe.hip 2>/dev/null
/tmp/st_probe chunks 2>&1 | grep -E "read \+ write (pair|flat) \(1.2x, 2 slots\)|^read  pair" > gpurun_out/$tag/probe.txt
AB_ROUNDS=${AB_ROUNDS:-3} AB_ARGS="--e2e-records 0 --no-box" bash tools/ab_so.sh "$@" > gpurun_out/$tag/ab.txt 2>&1
cat gpurun_out/$tag/probe.txt; tail -8 gpurun_out/$tag/ab.txt
