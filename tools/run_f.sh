export RB_BENCH_VERBOSE=1
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -2 | python -c "
import sys,json
L=sys.stdin.read().strip().split('\n'); print(L[0][:110]); d=json.loads(L[-1]); print('fused  ', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('parity_sample'))"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --unfused 2>&1 | tail -2 | python -c "
import sys,json
L=sys.stdin.read().strip().split('\n'); print(L[0][:110]); d=json.loads(L[-1]); print('unfused', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
python bench.py --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('full bench', d['ms_per_step'], d['value'], d['parity_sample'], d['cpu_baseline']['value'])"
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload config2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('config2', d['ms_per_step'], d['value'], d['roofline']['frac'])"
