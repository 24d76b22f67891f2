#!/bin/bash
# one default-shaped bench line (arena and ops placed) per call: what a box gives with the placement, and what it would have given without
cd $GRAFT_REPO_ROOT
tag=${1:-r04_box}
mkdir -p gpurun_out/$tag
python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 --no-box 2>/dev/null | tail -1 > gpurun_out/$tag/b.json
python - $tag <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/{sys.argv[1]}/b.json")); c=d["config"]
print("kernel", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "arena", c["out_arena_placement"]["launch_ms"], c["out_arena_placement"]["kept"], "ops", c["ops_placement"] and (c["ops_placement"].get("incumbent_ms"), c["ops_placement"]["launch_ms"], c["ops_placement"]["kept"]))
PY
