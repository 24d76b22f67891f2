"""bench.py --gpus N without a launcher starts N ranks itself, as fresh child processes, before anything touches the GPU
(VERDICT r01 item 1).  The dry-run flag makes every rank report its environment and leave, so this runs without a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=120)


def test_gpus_flag_spawns_that_many_ranks():
    r = _run(["--gpus", "3", "--launch-dry-run"])
    assert r.returncode == 0, r.stderr
    ranks = sorted((json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")), key=lambda d: d["rank"])
    assert [d["rank"] for d in ranks] == [0, 1, 2] and [d["local_rank"] for d in ranks] == [0, 1, 2]
    assert all(d["world"] == 3 and d["addr"] == "127.0.0.1" and d["ipc_legacy"] == "0" for d in ranks)
    assert len({d["port"] for d in ranks}) == 1


def test_launcher_world_size_is_taken_as_is_and_must_match():
    r = _run(["--gpus", "2", "--launch-dry-run"], {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"})
    assert r.returncode == 0 and json.loads(r.stdout)["rank"] == 1      # under torch.distributed.run: no second spawn
    r = _run(["--gpus", "8", "--launch-dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_a_failing_rank_fails_the_job():
    r = _run(["--gpus", "2", "--records", "1000", "--steps", "1", "--warmup", "0"],
             {"HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})   # no GPU: every rank exits non-zero
    assert r.returncode != 0


def test_defaults_place_the_arena_by_the_step_itself(monkeypatch):
    """the driver's `python bench.py` (no flags): N = 1, a K / W that finish in minutes, the output arena and the ops array placed by
    measurement (five candidates, ranked by the step itself) -- and `--placement-tries 1` is the way back to plain rb_dev_alloc"""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.op, a.workload) == (1, "liftover", "config3") and 1 <= a.steps <= 100 and a.warmup >= 1
    assert a.placement_tries == 5 and a.placement_by == "launch"
    monkeypatch.setattr(sys, "argv", ["bench.py", "--placement-tries", "1", "--placement-by", "sweep"])
    a = bench.parse()
    assert a.placement_tries == 1 and a.placement_by == "sweep"


def test_an_ended_rank_leaves_nothing_in_shared_memory():
    """bench.py's end-to-end leg keeps its files in /dev/shm (memory).  launch_ranks ends the other ranks with SIGTERM when one fails: the
    leg's directory must go with the process -- and a directory an earlier run left after being killed outright is taken away too."""
    import glob
    import signal
    import time
    if not os.path.isdir("/dev/shm") or not os.path.exists(os.path.join(ROOT, "rustybam_amd", "rb")):
        pytest.skip("no /dev/shm or the front end is not built")
    stale = "/dev/shm/rb_e2e_stale_for_the_test"
    os.makedirs(stale, exist_ok=True)
    open(os.path.join(stale, "w.paf"), "wb").write(b"x" * 1000)
    old = time.time() - 3 * 3600
    os.utime(stale, (old, old))
    before = set(glob.glob("/dev/shm/rb_e2e_*")) - {stale}
    code = f"import sys; sys.path.insert(0, {ROOT!r}); import bench; bench.e2e_leg(100000, 3000)"
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        new = set()
        for _ in range(300):  # (until the leg has made its directory and is writing the input)
            new = set(glob.glob("/dev/shm/rb_e2e_*")) - before - {stale}
            if new and any(os.path.exists(os.path.join(d, "w.paf")) for d in new):
                break
            if p.poll() is not None:
                break
            time.sleep(0.1)
        assert new and p.poll() is None, "the leg did not get as far as writing its input"
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=60)
    finally:
        if p.poll() is None:
            p.kill()
    assert p.returncode != 0
    assert set(glob.glob("/dev/shm/rb_e2e_*")) - before == set()
    assert not os.path.exists(stale)
