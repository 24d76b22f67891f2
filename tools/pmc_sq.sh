cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export RB_DEBUG_NO_RR=1
for mode in full desc; do
  if [ $mode = desc ]; then F=--descriptors; else F=; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_$mode -o a -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 $F > gpurun_out/pmc_$mode.log 2>&1
done
python3 - <<'PY'
import csv, collections
for mode in ("full","desc"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f"gpurun_out/pmc_{mode}/a_counter_collection.csv")):
        k=row["Kernel_Name"]
        if "liftover_stream" not in k: continue
        agg[k.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,c in agg.items():
        print(mode, {n: f"{v[-1]:.4g}" for n,v in sorted(c.items())})
PY
