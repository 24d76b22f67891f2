"""Does RCCL accept two ranks on ONE device?  (bench.py's RB_BENCH_SAME_DEVICE mode uses gloo for its three control collectives
because the answer on this image is no: "Duplicate GPU detected".)  Usage: python tools/rccl_same_device_probe.py"""
import os
import subprocess
import sys

if "RANK" not in os.environ:
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rc = [p.wait(timeout=240) for p in procs]
    print("rccl two ranks on one device:", "works" if rc == [0, 0] else f"refused (exit codes {rc})")
    sys.exit(0)
import torch
import torch.distributed as dist
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda:0") * (int(os.environ["RANK"]) + 1)
dist.all_reduce(t)
torch.cuda.synchronize()
assert float(t[0]) == 3.0
dist.destroy_process_group()
