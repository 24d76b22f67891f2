#!/bin/bash
# experiments on the GPU box: rebuild k_liftover.hip / capi.hip with extra defines and time the headline step
# usage: tools/ms_sweep.sh "<defines A>" "<defines B>" ...   (run from the repo root)
set -e
for D in "$@"; do
  make -s -C rustybam_amd/csrc clean >/dev/null
  make -s -j8 -C rustybam_amd/csrc all CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -ffp-contract=off $D" >/dev/null 2>&1
  echo "== $D"
  RB_BENCH_VERBOSE=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $BENCH_ARGS 2>&1 | grep -E "kernel ms|output_digest" | sed -E 's/.*("output_digest": "[0-9a-fx]+").*("kernel_ms": [0-9.]+).*/\1 \2/'
done
