# round 6, the tile kernel's cut around a record it cannot take: tile / trim / break tests, the tile soak, the starts soak, then the config-4 shape with and
# without 1 % irregular records through two library variants interleaved -- tile_old = k_tile.hip of the commit before (tools/mkvariant.sh tile_old --src
# k_tile.hip on that tree), tile_new = the tree's own (profiles/r06_alloc_summary.md has nothing of this; DESIGN.md section 0, the advisor's paragraph, has the numbers)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_tilecut
timeout 1200 python -m pytest tests/test_gpu_tile.py tests/test_gpu_trim.py tests/test_gpu_break_onewalk.py -x -q -m gpu > gpurun_out/r06_tilecut/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r06_tilecut/tests.log
timeout 900 python tests/soak/soak_tile.py 100 > gpurun_out/r06_tilecut/soak_tile.log 2>&1; echo "soak_tile rc=$?"; tail -1 gpurun_out/r06_tilecut/soak_tile.log
timeout 600 python tests/soak/soak_starts.py 20 > gpurun_out/r06_tilecut/soak_starts.log 2>&1; echo "soak_starts rc=$?"; tail -1 gpurun_out/r06_tilecut/soak_starts.log
for v in tile_old tile_new tile_old tile_new; do for f in 0 0.01; do for op in liftover break; do RB_VARIANT=$v python bench.py --workload config4-shape --op $op --irregular-frac $f --steps 10 --no-cpu-baseline --no-box --e2e-records 0 --placement-tries 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(\"$v\", \"$op\", $f, round(d[\"ms_per_step\"],3), d.get(\"tile_records_handed_back_per_gpu\"), d.get(\"output_digest\"))"; done; done; done 2>&1 | tee gpurun_out/r06_tilecut/ab.log
