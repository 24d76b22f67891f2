#!/bin/bash
# round 6's committed measurements behind profiles/r06_*: the secondary workloads of the bench (one line each), config 4's shape through both
# commands, the scan over three batch shapes, nucfreq, the writer probe and `rb liftover` end to end at the headline size
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r06_secondary.txt
for args in "--op break" "--op break --irregular-frac 0.01" "--workload irregular --records 100000" "--workload config2" "--workload config2 --placement uniform" "--workload config2-lognormal" "--workload config4-shape --op break" "--workload config4-shape --op liftover"; do
  tagname=$(echo "$args" | tr -d '-' | tr ' .' '__')
  timeout -k 5 900 python3 bench.py $args --steps 10 --no-cpu-baseline --no-box --e2e-records 0 2> /dev/null | tail -1 > gpurun_out/r06_${tagname}_bench.json
  python3 - "$args" gpurun_out/r06_${tagname}_bench.json <<'PY' | tee -a gpurun_out/r06_secondary.txt
import json, sys
d = json.loads(open(sys.argv[2]).read()); r = d["roofline"]
print(sys.argv[1], "| ms/step %.3f" % d["ms_per_step"], "kernel_ms", r["kernel_ms"], "frac", r["frac"], "unplaced", (r.get("unplaced") or {}).get("frac"), "generic", d["generic_hits_per_gpu"],
      "tiles", d.get("tiles_per_gpu"), "handed back", d.get("tile_records_handed_back_per_gpu"), "digest", d.get("output_digest"))
PY
done
echo "--- scan" | tee -a gpurun_out/r06_secondary.txt
timeout 300 python3 tools/scan_time.py 2>&1 | tail -3 | tee -a gpurun_out/r06_secondary.txt
echo "--- nucfreq" | tee -a gpurun_out/r06_secondary.txt
timeout 600 python3 tools/bench_nucfreq.py 2>/dev/null | tail -1 > gpurun_out/r06_nf_bench.json; python3 -c "
import json; d=json.load(open('gpurun_out/r06_nf_bench.json')); print({k: d[k] for k in ('ms_per_step','value','unit') if k in d}, d.get('roofline', {}).get('frac'))" | tee -a gpurun_out/r06_secondary.txt
echo "--- the writer alone (tools/shm_write_probe: 16 GB into one file of /dev/shm)" | tee -a gpurun_out/r06_secondary.txt
gcc -O2 -pthread -o /tmp/shm_write_probe tools/shm_write_probe.c && for t in 1 2 4 8 32; do /tmp/shm_write_probe /dev/shm/rb_probe.bin 16 $t 2>&1 | grep -i "pwrite" | head -2; done | tee -a gpurun_out/r06_secondary.txt
rm -f /dev/shm/rb_probe.bin
echo "--- end to end" | tee -a gpurun_out/r06_secondary.txt
n=1000000; d=/dev/shm/rb_e2e_$$; mkdir -p $d; RB=rustybam_amd/rb
$RB synth-paf 0x5EED0003 0 $n > $d/w.paf; $RB synth-bed 3000 > $d/w.bed
run() { name=$1; shift; for rep in 1 2; do rm -f $d/out_$name.paf; s=$(date +%s.%N); env RB_TIMING=1 "$@" > $d/out_$name.paf 2> gpurun_out/r06_e2e_${name}_$rep.err; rc=$?; e=$(date +%s.%N)
  echo "$name run $rep: rc $rc, $(awk "BEGIN{printf \"%.3f s, %.0f records/s\", $e - $s, $n / ($e - $s)}"), out $(stat -c %s $d/out_$name.paf) bytes" | tee -a gpurun_out/r06_secondary.txt; done; }
run pipelined $RB liftover --bed $d/w.bed $d/w.paf
run break_pipelined $RB break-paf --max-size 100 $d/w.paf
run whole RB_NO_PIPELINE=1 $RB liftover --bed $d/w.bed $d/w.paf
cmp -s $d/out_whole.paf $d/out_pipelined.paf && echo "pipelined = whole-file route, byte for byte" | tee -a gpurun_out/r06_secondary.txt
grep -h "rb timing" gpurun_out/r06_e2e_pipelined_1.err | awk '{a[$3" "$4" "$5]+=$(NF-1)} END {for (k in a) print k, a[k]}' | sort -k2 -n -r | head -12 | tee -a gpurun_out/r06_secondary.txt
rm -rf $d
