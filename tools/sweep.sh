#!/bin/bash
# diagnostics: rebuild k_liftover.hip with extra -D flags and time the streaming kernel (full and descriptor mode)
export RB_BENCH_VERBOSE=1
cd $(dirname $0)/../rustybam_amd/csrc
for cfg in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=off $cfg -c k_liftover.hip -o k_liftover.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librustybam_amd.so capi.o k_records.o k_liftover.o k_misc.o k_trim.o k_text.o k_nucfreq.o || { echo "build failed: $cfg"; continue; }
  for mode in "" "--descriptors"; do
    echo -n "[$cfg] ${mode:-full} "; (cd ../.. && python bench.py --no-cpu-baseline --steps 10 --warmup 2 $mode 2>&1 | grep "kernel ms")
  done
done
