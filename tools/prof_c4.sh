#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r03_c4}
recs=${2:-10000000}
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o kt -- python3 tools/bench_config4.py --records $recs > gpurun_out/${tag}.json 2> gpurun_out/${tag}.err
tail -1 gpurun_out/${tag}.json
grep -E "overlap_split|trim_select|trim_place|trim_check|break_pieces|liftover_stream|scan_records|apply|gather|Name" gpurun_out/$tag/kt_kernel_stats.csv | cut -d, -f1-8
python3 tools/gen_config4_paf.py 200000 > /tmp/c4.paf
export RB_TIMING=1
python3 - <<'PY'
import subprocess, time
for cmd, out in (("rustybam_amd/rb trim-paf /tmp/c4.paf", "/tmp/c4_t.paf"),) * 3 + (("rustybam_amd/rb break-paf --max-size 100 /tmp/c4_t.paf", "/tmp/c4_b.paf"),):
    t = time.perf_counter()
    subprocess.check_call(cmd.split(), stdout=open(out, "wb"))
    print(f"{cmd}: {time.perf_counter() - t:.3f} s")
PY
