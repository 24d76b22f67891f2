#!/bin/bash
# the driver's own sequence on the final tree: the whole GPU suite, smoke, the default bench line
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-driver_sequence}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 5 2400 python3 -m pytest tests -q -m gpu -x > $out/tests.log 2>&1; echo "tests rc=$?"; tail -4 $out/tests.log
timeout -k 5 300 python3 __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout -k 5 900 python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -c 300 $out/bench.err
python3 - $out/bench.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("ms/step", round(d["ms_per_step"], 3), "kernel_ms", r["kernel_ms"], "frac", r["frac"], "unplaced", r.get("unplaced"), "traffic", r.get("traffic"), "|", d.get("parity_sample"), "|", (d.get("parity_full") or "")[:80], "| e2e", d.get("e2e_paf_records_per_s"))
except Exception as e:
    print("no line:", e)
PY
