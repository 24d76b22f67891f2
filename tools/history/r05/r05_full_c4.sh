#!/bin/bash
# the driver's sequence, then config 4's device stages once more (unprofiled)
cd $GRAFT_REPO_ROOT
bash tools/r05_full.sh ${1:-r05_full3}
python3 tools/bench_config4.py --records 10000000 2>/dev/null | tail -1 > gpurun_out/r05_c4_plain_3.json
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_c4_plain_3.json").read()); k = d["trim_kernels_ms"]
print("c4:", d["trim_wall_s"], round(k["selection"] + k["pair_kernels"] + k["apply_and_check"], 2), d["trim_wall_over_kernels"], d["break_wall_s"])
PY
