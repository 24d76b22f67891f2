"""bench.py as the driver runs it, on a small batch: `--gpus N` starts N ranks, `--scaling strong` cuts one batch into N record
ranges, and the digest of the gathered output is the same for every N ("1/2/4/8-GPU outputs byte-identical to each
other").  With fewer visible GPUs than ranks, the N-rank legs still run -- every rank on GPU 0 (RB_BENCH_SAME_DEVICE=1; RCCL refuses
two ranks on one device, so the barrier / max / gather of that mode go over gloo) -- and the RCCL path itself is exercised with a
world of one; with N GPUs visible the ranks are real RCCL ranks on their own devices."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--records", "20000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--scaling", "strong"]


def _bench(gpus, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus)] + ARGS, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 prints the one JSON line
    return json.loads(lines[0])


def test_strong_scaling_digest_is_the_same_for_every_gpu_count():
    import torch
    one = _bench(1)
    assert one["n_gpus"] == 1 and one["scaling"] == "strong" and one["job_records"] == 20000 and one["job_hits"] > 200000
    # the output arena and the ops array are placed by measurement (set-up): every candidate's time is in the line, and which one was kept
    for key in ("out_arena_placement", "ops_placement"):
        pl = one["config"][key]
        if pl["kept"] == "incumbent":                       # (the ops array stays where it is unless a candidate is faster)
            assert key == "ops_placement" and pl["incumbent_ms"] <= min(pl["launch_ms"])
        else:
            assert len(pl["launch_ms"]) >= 1 and 0 <= pl["kept"] < len(pl["launch_ms"]) and pl["launch_ms"][pl["kept"]] == min(pl["launch_ms"]), key
    forced = _bench(1, {"RB_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533", "RANK": "0", "WORLD_SIZE": "1",
                        "LOCAL_RANK": "0"})   # the RCCL init / barrier / gather path with a world of one
    assert forced["output_digest"] == one["output_digest"]
    n_dev = torch.cuda.device_count()
    for n in (2, 3, 4, 8):
        if n > n_dev and n > 3:
            break
        many = _bench(n, {"RB_BENCH_SAME_DEVICE": "1"} if n > n_dev else None)
        assert many["n_gpus"] == n and many["job_records"] == 20000
        assert many["job_hits"] == one["job_hits"] and many["output_digest"] == one["output_digest"], n


def test_weak_scaling_two_ranks_sum_their_shards():
    """--scaling weak (the driver's default): rank r owns records [r * R, (r + 1) * R); the job's record / hit counts are the sum"""
    import torch
    extra = None if torch.cuda.device_count() >= 2 else {"RB_BENCH_SAME_DEVICE": "1"}
    args = [a for a in ARGS if a not in ("--scaling", "strong")]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra or {})
    outs = []
    for n in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + args, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0]))
    assert outs[0]["scaling"] == outs[1]["scaling"] == "weak"
    assert outs[1]["job_records"] == 2 * outs[0]["job_records"] and outs[1]["job_hits"] > outs[0]["job_hits"]
