#!/bin/bash
# diagnostics: rebuild the stream kernel with a given prefetch depth / emission batch and time both modes
for cfg in "$@"; do
  pf=${cfg%%:*}; eb=${cfg##*:}
  (cd $(dirname $0)/../rustybam_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=off -DRB_PF=$pf -DRB_EB=$eb -c k_liftover.hip -o k_liftover.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librustybam_amd.so capi.o k_records.o k_liftover.o k_misc.o k_trim.o) || exit 1
  for mode in "" "--descriptors"; do
    python bench.py --no-cpu-baseline --steps 5 --warmup 1 $mode 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('PF=$pf EB=$eb mode=${mode:-full} kernel_ms', d['roofline']['kernel_ms'])"
  done
done
