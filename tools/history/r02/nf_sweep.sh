#!/bin/bash
# usage: tools/nf_sweep.sh "<-D flags>" ...   -- rebuild k_nucfreq.hip with each flag set, run the config-5 bench (on the GPU box)
cd "$(dirname "$0")/../rustybam_amd/csrc"
for cfg in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=off $cfg -c k_nucfreq.hip -o k_nucfreq.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librustybam_amd.so capi.o k_records.o k_liftover.o k_misc.o k_trim.o k_text.o k_nucfreq.o || { echo "build failed: $cfg"; continue; }
  echo "== $cfg: $(python ../../tools/bench_nucfreq.py --steps 5 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],3),"ms")')"
done
