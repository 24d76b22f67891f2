#!/bin/bash
# one bench process with the box block (phases of a record included); kept per call to compare fast and slow boxes
cd $GRAFT_REPO_ROOT
tag=${1:-r04_hunt}
mkdir -p gpurun_out/$tag
for i in 1 2; do
python bench.py --steps 10 --no-cpu-baseline --e2e-records 0 2>gpurun_out/$tag/err$i.txt | tail -1 > gpurun_out/$tag/b$i.json
python - gpurun_out/$tag/b$i.json <<'PY' | tee -a gpurun_out/$tag/log.txt
import json,sys
d=json.loads(open(sys.argv[1]).read()); b=d['box']
print("kernel", d["roofline"]["kernel_ms"], "probe", b.get("probe_ms"), b.get("probe_scattered_ms"), "r/w only", b.get("probe_read_only_ms"), b.get("probe_write_only_ms"), "k nospec/readsonly", b.get("kernel_without_speculative_stores_ms"), b.get("kernel_reads_only_ms"), 'phases', b.get('phase_cycles_per_record'), b.get('error'))
PY
done
