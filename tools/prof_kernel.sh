#!/bin/bash
# rocprofv3 runs behind profiles/: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r01_c}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o kt -- python3 bench.py --no-box --e2e-records 0 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$tag -o fetch -- python3 bench.py --no-box --e2e-records 0 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$tag -o write -- python3 bench.py --no-box --e2e-records 0 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$tag -o sq -- python3 bench.py --no-box --e2e-records 0 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_sq.log 2>&1
tail -1 gpurun_out/${tag}_kt.log
python3 bench.py --steps 10 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -1 gpurun_out/${tag}_bench.json
