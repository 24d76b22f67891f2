#!/bin/bash
# diagnostics (on the GPU box): one kernel file rebuilt under each set of -D flags as a library VARIANT (tools/mkvariant.sh: rustybam_amd/variants/,
# never the product library) and timed through RB_VARIANT.
#   tools/sweep.sh k_liftover.hip "-DA=1" "-DA=2 -DB"     the headline step (bench.py --placement-tries 1: kernel ms + output digest)
#   tools/sweep.sh k_nucfreq.hip "-DNF_TILE=2048" ...     config 5 (tools/bench_nucfreq.py: ms per call)
# BENCH_ARGS adds flags to the bench.
cd "$(dirname "$0")/.."
src=$1; shift
i=0
for cfg in "$@"; do
  i=$((i + 1)); name=sweep_$i
  bash tools/mkvariant.sh $name --src $src $cfg > /dev/null 2>&1 || { echo "build failed: $cfg"; continue; }
  if [ "$src" = "k_nucfreq.hip" ]; then
    echo "== $cfg: $(RB_VARIANT=$name python tools/bench_nucfreq.py --steps 5 $BENCH_ARGS 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],3),"ms")')"
  else
    echo "== $cfg: $(RB_VARIANT=$name python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-box --e2e-records 0 --placement-tries 1 $BENCH_ARGS 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["roofline"]["kernel_ms"], "ms", d.get("output_digest"))')"
  fi
done
