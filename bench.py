#!/usr/bin/env python3
"""bench.py -- liftover over 100 kb sliding windows (BASELINE.json configs[2]) on N MI355X.

One "step" = one full liftover pass of the hot path over one resident batch: hit counting, scan,
the streaming clip kernel, the generic kernel and the summary kernel, through the C ABI
(rb_dev_liftover), inputs already in HBM.  Per GPU: 1e6 synthetic PAF records (n_ops uniform
[1000, 9000], ~5e9 CIGAR ops, 20 GB packed) placed uniformly on a chr1-sized target x 3000 windows
(st = i * 82,796, width 100 kb).

N > 1 (`--gpus N`): one process per GPU.  Under torch.distributed.run (WORLD_SIZE set) the ranks are
the launcher's; otherwise this script starts them itself, as fresh child processes, BEFORE anything
here touches the GPU.  Records shard by contiguous range, no data-path collective; torch.distributed
(RCCL) only carries the barrier, the max-over-ranks time and the verification digests.
  --scaling weak   (default) every rank owns its own 1e6 records: rank r = records [r * 1e6, (r + 1) * 1e6)
  --scaling strong the one 1e6-record batch of configs[2] is cut into N op-balanced record ranges
                   (rustybam_amd.shard.shard_bounds): "records sharded 1 -> 8 GPUs"
`output_digest` is an order-sensitive digest of every hit row and clipped CIGAR of the whole job in
canonical (gathered) order; with --scaling strong it must be the same for N = 1, 2, 4, 8.

Prints ONE JSON line on rank 0: metric CIGAR-ops/s (whole job), plus `roofline` for the streaming
kernel (HIP events on the launch stream) and `cpu_baseline` (the oracle, a faithful per-base port of
the reference, timed on this host's cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--records", type=int, default=1_000_000, help="records per GPU")
    ap.add_argument("--windows", type=int, default=3000)
    ap.add_argument("--workload", default="config3", choices=["config3", "config2", "config2-lognormal", "irregular", "config4-shape"],
                    help="config3: the headline; config2: one 1 Mbp window; config2-lognormal: SURVEY 8(d)'s imbalance shape of config 2 "
                         "(op counts log-normal(ln 2000, 1.35) clipped to [31, 80000], the fixture's own range; --records defaults to 1e5 x "
                         "mean 4.1k ops); irregular: config 3's records made irregular (adjacent ops of one "
                         "type at the start, an N / H op at an end of two thirds of them) so that every hit takes the generic wave-per-hit kernel; "
                         "config4-shape: BASELINE config 4's record shape through the same two ops (--records defaults to 1e7 records of 300-700 "
                         "ops, seed 0x5EED0004, placed uniformly under the 3000 sliding windows; --op liftover | break): the short-record regime")
    ap.add_argument("--ops-lo", type=int, default=0, help="config4-shape: op counts uniform in [--ops-lo, --ops-hi] (default 300 .. 700)")
    ap.add_argument("--ops-hi", type=int, default=0)
    ap.add_argument("--irregular-frac", type=float, default=0.0,
                    help="config3 / --op break: this fraction of the records made irregular as in --workload irregular (0.01: what the "
                         "one-walk break path must take record by record instead of redoing the batch)")
    ap.add_argument("--placement", default="default", choices=["default", "uniform", "overlap"],
                    help="where the records lie on the target: config 3 = uniform; config 2 = overlap (every record overlaps the window); "
                         "`--workload config2 --placement uniform` is SURVEY 8(d)'s second imbalance case: about 0.8 %% of the records overlap "
                         "the window, every record is still walked (liftover.rs:119-121)")
    ap.add_argument("--no-box", action="store_true", help="skip the `box` block (memory-mix probe, in-kernel clock, tail of the launch)")
    ap.add_argument("--box-smi", action="store_true",
                    help="the `box` block also asks rocm-smi (two child processes, started before this one touches the GPU; never under a profiler)")
    ap.add_argument("--legacy", action="store_true", help="RB_BSEARCH_LEGACY (rustc 1.52 .. 1.81 binary search): duplicates resolved by probe replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--early-exit", action="store_true", help="allow the kernel to stop a record early (off: full walk)")
    ap.add_argument("--unfused", action="store_true", help="run rb_dev_scan_records as its own pass inside the step instead of the fused scan")
    ap.add_argument("--placement-tries", type=int, default=5,
                    help="candidates rb_dev_alloc_placed may allocate for the output arena (a store sweep over each, the fastest kept: set-up, "
                         "not timed); 1 = plain rb_dev_alloc")
    ap.add_argument("--placement-by", default="launch", choices=["launch", "sweep"],
                    help="what ranks the candidates: the step itself on each (rb_dev_alloc_placed_by) or the library's store sweep (rb_dev_alloc_placed)")
    ap.add_argument("--debug-skip", type=int, default=0, help="diagnostics: skip kernel phases (invalid results)")
    ap.add_argument("--op", default="liftover", choices=["liftover", "break"],
                    help="liftover (headline) or break-paf --max-size 100 on the same records (secondary measurement)")
    ap.add_argument("--two-walk", action="store_true", help="--op break: collect the pieces in a pass of its own, then clip (the round-1 path)")
    ap.add_argument("--descriptors", action="store_true",
                    help="RB_LIFT_DESCRIPTORS: return which ops each clip keeps instead of copying them (not the headline mode)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --records per GPU; strong: --records in all, cut into one op-balanced record range per GPU")
    ap.add_argument("--e2e-records", type=int, default=1_000_000,
                    help="records of the text-in -> text-out leg (`rb [--gpus N] liftover` on the PAF text of the same workload, first byte "
                         "read to last byte written; 1e6 = the headline size, 14.7 GB in / 19 GB out in /dev/shm -- cut to 1e5 when the "
                         "host lacks the memory); 0 = skip")
    ap.add_argument("--launch-dry-run", action="store_true", help=argparse.SUPPRESS)  # tests: ranks report their environment and exit
    return ap.parse_args()


def launch_ranks(args):
    """`bench.py --gpus N` without a launcher: start N ranks as fresh child processes (one per GPU) and wait for them.
    Nothing in this process has touched the GPU (no torch, no HIP): a process that has must never fork + exec."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                try:
                    code = p.wait(timeout=0.2)
                except subprocess.TimeoutExpired:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:  # one rank failed: the others would wait in a barrier forever
                    rc = code
                    for q in pending:
                        q.terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def e2e_leg(n_rec, n_win, gpus=1):
    """end-to-end PAF-records/s = input records / wall time from the first byte read to the last byte written, for the
    `rb liftover` front end (C++ host + this GPU; `rb --gpus N` = N worker processes, one per GPU, host-side gather) on the text
    form of the same workload.  Runs as child processes, before this process has touched the GPU.  Returns a dict for the JSON
    line, or None when the front end is not built."""
    import shutil
    import subprocess
    import tempfile
    rb = os.path.join(ROOT, "rustybam_amd", "rb")
    if not os.path.exists(rb):
        return None
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    try:  # 35 KB of text per record in + out, and the front end holds its output in memory once more
        st = os.statvfs(shm or "/tmp")
        avail = int([ln for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")][0].split()[1]) * 1024
        if min(st.f_bavail * st.f_frsize, avail) < n_rec * 120_000:
            n_rec = min(n_rec, 100_000)
    except Exception:
        n_rec = min(n_rec, 100_000)
    # The files live in memory (/dev/shm): a run that is ended from outside must not leave gigabytes behind.  SIGTERM (what launch_ranks sends
    # the other ranks when one fails) becomes an exception for the length of this leg, so that the `finally` below runs; what an earlier
    # run that was killed outright left is taken away here (directories of this name older than an hour).
    import signal
    for name in (os.listdir(shm) if shm else []):
        q = os.path.join(shm, name)
        try:
            if name.startswith("rb_e2e_") and os.path.isdir(q) and time.time() - os.stat(q).st_mtime > 3600:
                shutil.rmtree(q, ignore_errors=True)
        except OSError:
            pass

    def _term(*_):
        raise SystemExit(143)
    old_term = signal.signal(signal.SIGTERM, _term)
    d = tempfile.mkdtemp(prefix="rb_e2e_", dir=shm)
    try:
        paf, bed, out = (os.path.join(d, x) for x in ("w.paf", "w.bed", "out.paf"))
        t0 = time.perf_counter()
        with open(paf, "wb") as f:
            subprocess.check_call([rb, "synth-paf", "0x5EED0003", "0", str(n_rec)], stdout=f)
        with open(bed, "wb") as f:
            subprocess.check_call([rb, "synth-bed", str(n_win)], stdout=f)
        gen_s = time.perf_counter() - t0
        best = None
        pre = ["--gpus", str(gpus)] if gpus > 1 else []
        env = dict(os.environ)
        if os.environ.get("RB_BENCH_SAME_DEVICE") == "1":
            env["RB_GPUS_SAME_DEVICE"] = "1"
        for _ in range(2):  # (the first run also pages the binary and the HIP runtime in)
            if os.path.exists(out):
                os.unlink(out)  # (giving back the 19 GB of the run before is not part of this run: open(..., "wb") would do it inside the timed region)
            t0 = time.perf_counter()
            with open(out, "wb") as f:
                subprocess.check_call([rb, *pre, "liftover", "--bed", bed, paf], stdout=f, env=env)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        return {"e2e_paf_records_per_s": n_rec / best, "e2e": {"records": n_rec, "windows": n_win, "seconds": round(best, 3), "n_gpus": gpus,
                                                                "in_bytes": os.path.getsize(paf), "out_bytes": os.path.getsize(out),
                                                                "command": f"rb {' '.join(pre)} liftover --bed w.bed w.paf > out.paf (text in, text out; HIP start-up "
                                                                           "included; big plain files go through the pipelined route)",
                                                                "setup_s": round(gen_s, 1)}}
    finally:
        shutil.rmtree(d, ignore_errors=True)
        signal.signal(signal.SIGTERM, old_term)


def smi_facts():
    """Partition modes, power cap and clocks of GPU 0 as rocm-smi reports them -- run as a child process BEFORE this process touches the
    GPU (a process that has must not fork + exec).  None when rocm-smi is absent or says nothing parsable."""
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "--showmemorypartition", "--showcomputepartition", "--showmaxpower", "--showpower", "--showclocks", "--json"],
                             capture_output=True, text=True, timeout=60).stdout
        j = json.loads(out[out.index("{"):])
        c = j.get("card0") or next(iter(j.values()))
        pick = {}
        for k_, v in c.items():
            lk = k_.lower()
            if "partition" in lk or "power" in lk or lk.startswith("sclk") or lk.startswith("mclk") or lk.startswith("fclk"):
                pick[k_] = v
        return pick or None
    except Exception as e:  # (diagnostics only: never in the way of the bench)
        return {"error": f"{type(e).__name__}: {e}"}


class SmiSampler:
    """`rocm-smi` in a loop as a child process, started BEFORE this process touches the GPU (a process that has must not fork + exec) and
    stopped by its pid at the end: what the clocks (sclk / mclk / fclk), the temperatures and the package power read WHILE the clip
    kernel runs.  One JSON document per sample, each stamped with the wall clock."""

    def __init__(self):
        import shutil
        import subprocess
        import tempfile
        self.proc, self.path = None, None
        exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
        if not os.path.exists(exe):
            return
        fd, self.path = tempfile.mkstemp(prefix="rb_smi_", suffix=".txt")
        os.close(fd)
        # (it samples only once `<path>.go` exists: nothing polls the chip's management unit during the timed region)
        loop = (f'while [ ! -e {self.path}.go ]; do sleep 0.05; done; '
                f'while true; do echo "@@ $(date +%s.%N)"; {exe} --showclocks --showpower --showtemp --json 2>/dev/null; sleep 0.1; done')
        try:
            self.proc = subprocess.Popen(["bash", "-c", loop], stdout=open(self.path, "w"), stderr=subprocess.DEVNULL, start_new_session=True)
        except Exception:
            self.proc = None

    def go(self):
        if self.proc:
            open(self.path + ".go", "w").close()

    def stop(self, t_from, t_to):
        """-> {field: [min, max]} over the samples taken in [t_from, t_to] (time.time() stamps), or None"""
        if not self.proc:
            return None
        try:
            import signal
            os.killpg(self.proc.pid, signal.SIGTERM)  # (the exact process group this object started)
            self.proc.wait(timeout=5)
        except Exception:
            pass
        out = {}
        try:
            n = 0
            for chunk in open(self.path).read().split("@@ ")[1:]:
                head, _, body = chunk.partition("\n")
                ts = float(head.strip())
                if not (t_from <= ts <= t_to) or "{" not in body:
                    continue
                try:
                    j = json.loads(body[body.index("{"):body.rindex("}") + 1])
                except Exception:
                    continue
                c = j.get("card0") or next(iter(j.values()))
                n += 1
                for k_, v in c.items():
                    lk = k_.lower()
                    if not ("clock speed" in lk or "temperature" in lk or "power" in lk):
                        continue
                    try:
                        x = float(str(v).strip("()").lower().replace("mhz", "").replace("c", "").replace("w", ""))
                    except Exception:
                        continue
                    lo, hi = out.get(k_, (x, x))
                    out[k_] = (min(lo, x), max(hi, x))
            os.unlink(self.path)
            if os.path.exists(self.path + ".go"):
                os.unlink(self.path + ".go")
            return {"samples": n, **{k_: [v[0], v[1]] for k_, v in sorted(out.items())}} if n else {"samples": 0}
        except Exception as e:
            return {"error": f"{type(e).__name__}: {e}"}


def under_profiler():
    """True when a profiler's tool library rides in this process (rocprofv3 / rocprof preload theirs, and with --pmc it has initialised
    the GPU before python's first line runs).  Such a process must not start ANY child: no rocm-smi, no `rb`, no ranks."""
    e = os.environ
    if any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in e):
        return True
    pre = e.get("LD_PRELOAD", "")
    return any(t in pre for t in ("rocprof", "roctracer", "rocprofiler", "libkineto"))


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    profiled = under_profiler()
    if profiled:  # (no child process of any kind: see under_profiler)
        args.box_smi = False
        args.e2e_records = 0
        if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
            raise SystemExit("bench.py: --gpus N under a profiler would start the ranks from a process whose GPU is already initialised; "
                             "profile one rank (--gpus 1) or start the ranks with torch.distributed.run outside the profiler")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))  # (before torch / HIP are even imported)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.launch_dry_run:
        print(json.dumps({"rank": rank, "local_rank": local_rank, "world": world, "addr": os.environ.get("MASTER_ADDR"),
                          "port": os.environ.get("MASTER_PORT"), "ipc_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}), flush=True)
        return
    e2e = None
    want_smi = rank == 0 and args.box_smi and not args.no_box and not profiled
    smi = smi_facts() if want_smi else None
    sampler = SmiSampler() if want_smi else None  # (a child process, started before this one touches the GPU)
    if rank == 0 and args.e2e_records > 0 and args.workload == "config3" and args.op == "liftover" and not args.no_cpu_baseline and not profiled:
        # (child processes, before this one initialises the GPU; with N ranks: `rb --gpus N` on the same file while the other ranks
        #  wait for rank 0 at the rendezvous)
        e2e = e2e_leg(args.e2e_records, args.windows, world)
    import torch
    import torch.distributed as dist
    import rustybam_amd
    from rustybam_amd import shard
    from rustybam_amd import workload as wl

    use_dist = world > 1 or os.environ.get("RB_BENCH_FORCE_DIST") == "1"  # the latter: exercise the RCCL path on one GPU
    # RB_BENCH_SAME_DEVICE=1 (tests on a 1-GPU lease): every rank on GPU 0, so that the N > 1 path -- shard bounds, per-rank
    # generation, barrier, max-over-ranks time, gathered digest -- runs for real with N processes.  RCCL refuses two ranks on one
    # device ("duplicate GPU"), so the three control collectives then go over gloo with host tensors; the data path has no
    # collective either way.
    same_device = os.environ.get("RB_BENCH_SAME_DEVICE") == "1"
    dev_index = 0 if same_device else local_rank
    if use_dist:
        if same_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if dev_index >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {dev_index}, this node shows {torch.cuda.device_count()}")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    cdev = torch.device("cpu") if (use_dist and same_device) else dev  # where the control collectives' tensors live
    torch.cuda.set_stream(torch.cuda.Stream(dev))  # one real stream for torch's kernels AND the engine's (the default stream's
    stream = torch.cuda.current_stream().cuda_stream  # handle is NULL, which rb_ctx_create reads as "make a private stream")
    eng = rustybam_amd.Engine(dev_index, stream)

    if args.workload.startswith("config2") and "--records" not in sys.argv:
        args.records = 100_000  # BASELINE.json configs[1]: 1e5 records x one 1 Mbp window
    c4 = args.workload == "config4-shape"
    if c4 and "--records" not in sys.argv:
        args.records = 10_000_000  # BASELINE.json configs[3]: 1e7 records of 300 - 700 ops
    ops_lo, ops_hi = (args.ops_lo or 300, args.ops_hi or 700) if c4 else (1000, 9000)
    if args.scaling == "strong":  # one batch of --records, one op-balanced contiguous record range per rank
        seed_ = wl.SEED_CONFIG2 if args.workload.startswith("config2") else (wl.SEED_CONFIG4 if c4 else wl.SEED_CONFIG3)
        bounds = shard.shard_bounds(wl.op_offsets(wl.n_ops_lognormal(seed_, 0, args.records) if args.workload == "config2-lognormal"
                                                  else wl.n_ops(seed_, 0, args.records, ops_lo, ops_hi)), world)
        first, n_rec = int(bounds[rank]), int(bounds[rank + 1] - bounds[rank])
    else:
        n_rec = args.records
        first = rank * n_rec
    if args.workload in ("config3", "irregular", "config4-shape"):
        seed, placement = (wl.SEED_CONFIG4 if c4 else wl.SEED_CONFIG3), "uniform"
        w_c, w_st, w_en = wl.sliding_windows(args.windows)
    else:
        seed, placement = wl.SEED_CONFIG2, "overlap"
        w_c, w_st, w_en = np.zeros(1, np.uint32), np.array([12_000_000], np.uint64), np.array([13_000_000], np.uint64)
    if args.placement != "default":
        placement = args.placement
    lognormal = args.workload == "config2-lognormal"

    # ---- generate the shard in HBM ----
    t0 = time.time()
    nops = wl.n_ops_lognormal(seed, first, n_rec) if lognormal else wl.n_ops(seed, first, n_rec, ops_lo, ops_hi)
    op_off = wl.op_offsets(nops)
    total_ops = int(op_off[-1])

    def dev_u64(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)

    d_off = dev_u64(op_off)
    from rustybam_amd import capi as capi_mod
    lib_alloc = os.environ.get("RB_BENCH_TORCH_ALLOC") != "1"  # (diagnostics: the batch in torch's own allocations, as rounds 1 and 2)

    alloc_note = []

    def big(n, dtype, placed=1, score=None):  # -> (tensor, owner): the batch's large buffers (placed > 1: the output arena, rb_dev_alloc_placed[_by])
        if lib_alloc and not alloc_note:
            try:
                b_ = capi_mod.DevBuf(eng, torch, n, dtype, device=dev, placed_tries=placed, score=score)
                return b_.t, b_
            except Exception as e:  # (torch could not take the library's memory for a tensor of this device: its own allocator then)
                alloc_note.append(f"{type(e).__name__}: {e}")
        return torch.empty(n, dtype=dtype, device=dev), None

    d_ops, own_ops = big(total_ops + 64, torch.int32)
    eng.dev_synth_fill_ops(seed, first, n_rec, d_off.data_ptr(), d_ops.data_ptr())
    if args.workload == "irregular" or args.irregular_frac > 0:
        # op 1 (the first event) becomes an '=': three adjacent '=' ops that the reference's collapse merges (paf.rs:602-620); a third of
        # the records also end on an N, a third start on an H: no match op at that end.  None of them can take the streaming kernel.
        torch.cuda.synchronize()
        rr = torch.arange(n_rec, device=dev)
        every = 1 if args.workload == "irregular" else max(1, int(round(1.0 / args.irregular_frac)))
        sel = (rr + first) % every == 0
        i1 = (d_off[:-1] + 1)[sel]
        d_ops[i1] = (d_ops[i1] & ~15) | 7
        il = (d_off[1:] - 1)[sel & (rr % 3 == 1)]
        d_ops[il] = (d_ops[il] & ~15) | 3
        i0 = d_off[:-1][sel & (rr % 3 == 2)]
        d_ops[i0] = (d_ops[i0] & ~15) | 5
        torch.cuda.synchronize()
    zeros = torch.zeros(n_rec, dtype=torch.int64, device=dev)
    d_contig = torch.zeros(n_rec, dtype=torch.int32, device=dev)
    d_strand0 = torch.full((n_rec,), ord("+"), dtype=torch.uint8, device=dev)
    d_red = torch.empty(n_rec * 72, dtype=torch.uint8, device=dev)
    d_norm = torch.empty(n_rec * 64, dtype=torch.uint8, device=dev)
    v0 = eng.batch_view(n_rec, total_ops, d_ops.data_ptr(), d_off.data_ptr(), zeros.data_ptr(), zeros.data_ptr(),
                        zeros.data_ptr(), zeros.data_ptr(), d_strand0.data_ptr(), d_contig.data_ptr())
    torch.cuda.synchronize()  # (the engine has its own stream: torch's fills above must have landed before it reads them)
    eng.dev_scan_records(v0, d_red.data_ptr(), 0)
    torch.cuda.synchronize()
    red = d_red.cpu().numpy().view(rustybam_amd.REDUCE_DT)
    t_st, t_en, q_st, q_en, strand = wl.headers(seed, first, red["t_bases"], red["q_bases"], placement)
    d_tst, d_ten, d_qst, d_qen = dev_u64(t_st), dev_u64(t_en), dev_u64(q_st), dev_u64(q_en)
    d_strand = torch.from_numpy(strand).to(dev)
    view = eng.batch_view(n_rec, total_ops, d_ops.data_ptr(), d_off.data_ptr(), d_tst.data_ptr(), d_ten.data_ptr(),
                          d_qst.data_ptr(), d_qen.data_ptr(), d_strand.data_ptr(), d_contig.data_ptr())
    def view_now():  # (the batch view of the moment: the ops array may move once, when it is placed)
        return view
    # upload-time pass of the reference (Paf::from_file -> check_integrity) + remove_trailing_indels
    eng.dev_scan_records(view, d_red.data_ptr(), d_norm.data_ptr())
    torch.cuda.synchronize()
    red = d_red.cpu().numpy().view(rustybam_amd.REDUCE_DT)
    assert (red["status"] == 0).all(), "synthetic records must pass check_integrity"
    irregular = args.workload == "irregular"
    del d_red
    tp = time.perf_counter()
    plan = eng.plan_create(op_off, np.zeros(n_rec, np.uint32), w_c, w_st, w_en)
    plan_ms = (time.perf_counter() - tp) * 1e3  # host: canonical order, longest-first schedule, window grouping + their uploads
    gen_s = time.time() - t0

    # ---- size the outputs (first call tells what is needed) ----
    policy = (rustybam_amd.BSEARCH_LEGACY if args.legacy else rustybam_amd.BSEARCH_MODERN) | (rustybam_amd.LIFT_EARLY_EXIT if args.early_exit else 0) | (args.debug_skip << 8)
    if args.descriptors:
        policy |= rustybam_amd.LIFT_DESCRIPTORS
    if not args.unfused:
        policy |= rustybam_amd.LIFT_FUSED_SCAN
    d_cnt = torch.zeros(64, dtype=torch.uint8, device=dev)
    # out_ops: as many positional copies of the batch's op index space as the windows overlap deep (2 for the sliding windows; 1
    # for break-paf) + arena room; the clipped cigars land there while their record streams past
    rows_cap, out_cap = max(1024, 16 * n_rec), max(4096, eng.plan_out_capacity(plan, args.op == "break"))  # (a first guess; the counters of the first call say what is needed)

    # break-paf: the clip kernel finds the long indels itself (one walk of the ops); a batch it does not take says so in the
    # counters and is done with the collect pass in front (two walks) -- decided once, by the sizing call below
    brk_policy = [policy | (rustybam_amd.BREAK_ONE_WALK if (args.op == "break" and not args.descriptors and not args.two_walk) else 0)]

    nonlocal_view = [None]  # (set while a candidate of the ops array is being measured)
    ops_placement = None

    def run_op(ws, rows, out):
        out_ptr = out if isinstance(out, int) else out.data_ptr()  # (a raw address: a candidate of the arena's placement)
        view = nonlocal_view[0] or view_now()
        if args.op == "break":
            eng.dev_break(plan, view, d_norm.data_ptr(), 100, brk_policy[0], ws.data_ptr(), rows.data_ptr(), rows_cap, out_ptr,
                          out_cap, d_cnt.data_ptr())
        else:
            eng.dev_liftover(plan, view, d_norm.data_ptr(), policy, ws.data_ptr(), rows.data_ptr(), rows_cap, out_ptr,
                             out_cap, d_cnt.data_ptr())
    if args.descriptors:
        rows_cap = max(rows_cap, 16 * n_rec)
        out_cap = 4 * rows_cap + total_ops // 8 + 65536
    tz = time.perf_counter()
    for _ in range(6):
        d_ws, own_ws = big(eng.plan_workspace_bytes(plan, rows_cap), torch.uint8)
        d_rows, own_rows = big((rows_cap + 1) * 64, torch.uint8)
        d_out, own_out = big(out_cap + 64, torch.int32)
        run_op(d_ws, d_rows, d_out)
        torch.cuda.synchronize()
        cnt = d_cnt.cpu().numpy().view(rustybam_amd.COUNTERS_DT)[0]
        if cnt["redo_two_walk"] and (brk_policy[0] & rustybam_amd.BREAK_ONE_WALK):
            brk_policy[0] &= ~rustybam_amd.BREAK_ONE_WALK
            del d_ws, d_rows, d_out
            [o.free() for o in (own_ws, own_rows, own_out) if o]
            continue
        if not cnt["overflow"]:
            break
        rows_cap = max(rows_cap, int(cnt["n_hits"]) + 64)
        out_cap = max(out_cap * 2, int(int(cnt["out_ops_needed"]) * 1.25) + 4096)
        del d_ws, d_rows, d_out
        [o.free() for o in (own_ws, own_rows, own_out) if o]
    assert not cnt["overflow"], "could not size the output buffers"
    n_hits = int(cnt["n_hits"])
    sizing_ms = (time.perf_counter() - tz) * 1e3  # the calls that find rows_cap / out_cap (allocation included); once per batch shape
    # The output arena, placed by measurement (rb_dev_alloc_placed; set-up, like the sizing above): which physical pages the arena has
    # decides up to 20 % of the clip kernel's time on this part, reads do not care (profiles/r04_alloc_summary.md).  The library
    # allocates up to --placement-tries candidates, measures each, keeps the fastest and gives the others back.  The measure
    # (--placement-by): "launch" = this step on the candidate, three times under a timer (rb_dev_alloc_placed_by: what the arena is for);
    # "sweep" = the library's own store sweep (rb_dev_alloc_placed), which tells an 11 ms arena from a 9.2 ms one but not always
    # a 10.1 ms one from a 9.9 ms one.
    placement_ms = 0.0
    unplaced_ms = None  # the clip kernel (HIP events) on the buffers as rb_dev_alloc first returned them: what a caller that places nothing gets
    if args.placement_tries > 1 and own_out is not None:
        run_op(d_ws, d_rows, d_out)
        torch.cuda.synchronize()
        eng.set_timing(True)
        for _ in range(5):
            run_op(d_ws, d_rows, d_out)
        torch.cuda.synchronize()
        u_ = eng.get_timing()[-5:]
        eng.set_timing(False)
        unplaced_ms = float(np.mean(u_)) if len(u_) else None
        tp = time.perf_counter()
        del d_out
        own_out.free()

        def launch_score(ptr):
            # the clip kernels' own time on the candidate (HIP events inside the library, three launches), the measure the line reports --
            # not the wall time of three whole steps (round 4); the wall only where the library timed nothing
            run_op(d_ws, d_rows, ptr)  # (first touch of the candidate's pages)
            torch.cuda.synchronize()
            eng.set_timing(True)
            t_ = time.perf_counter()
            for _ in range(3):
                run_op(d_ws, d_rows, ptr)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t_) * 1e3 / 3
            ev_ = eng.get_timing()[-3:]
            eng.set_timing(False)
            return float(np.mean(ev_)) if len(ev_) == 3 else wall

        d_out, own_out = big(out_cap + 64, torch.int32, placed=args.placement_tries, score=launch_score if args.placement_by == "launch" else None)
        # ... and the INPUT: with the arena fixed, the same launch reading its ops from other pages differs by up to 8 % as well (9.23 /
        # 9.38 / 10.02 ms, profiles/r04_alloc_summary.md).  Candidates of the ops array, each filled with a copy of the ops and measured by
        # the step itself; rows and workspace do not care where they lie (9.42 - 9.48 ms).
        if args.placement_by == "launch" and own_ops is not None:
            class _Raw:  # (a candidate's address as a tensor, for the copy)
                def __init__(self, ptr, nbytes):
                    self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}
            ops_bytes = (total_ops + 64) * 4
            views = {}

            def ops_score(ptr):
                torch.as_tensor(_Raw(ptr, ops_bytes), device=dev).copy_(d_ops.view(torch.uint8)[:ops_bytes])
                views[ptr] = eng.batch_view(n_rec, total_ops, ptr, d_off.data_ptr(), d_tst.data_ptr(), d_ten.data_ptr(),
                                            d_qst.data_ptr(), d_qen.data_ptr(), d_strand.data_ptr(), d_contig.data_ptr())
                nonlocal_view[0] = views[ptr]
                try:
                    return launch_score(d_out)
                finally:
                    nonlocal_view[0] = None
            incumbent_ms = launch_score(d_out)  # (the ops array as it lies: a candidate like the others -- the pages the arena's losers gave back may be the slow ones)
            new_ops, new_own = big(total_ops + 64, torch.int32, placed=args.placement_tries, score=ops_score)
            if new_own is not None and new_own.placement:
                ops_placement = dict(new_own.placement, incumbent_ms=round(incumbent_ms, 4))
                if min(new_own.placement["launch_ms"]) < incumbent_ms:
                    old_t, old_own = d_ops, own_ops
                    d_ops, own_ops = new_ops, new_own
                    view = views[d_ops.data_ptr()]
                    del old_t
                    old_own.free()
                else:
                    ops_placement["kept"] = "incumbent"
                    del new_ops
                    new_own.free()
        run_op(d_ws, d_rows, d_out)
        torch.cuda.synchronize()
        placement_ms = (time.perf_counter() - tp) * 1e3

    def step():
        # the whole hot path from the packed ops.  liftover: one fused call -- remove_trailing_indels + check_integrity (which the
        # reference's aligned_pairs runs inside trim_paf_by_rgns, liftover.rs:119-121) are done by the clip kernel while it streams
        # each record (RB_LIFT_FUSED_SCAN: the normalised rows are an output of the step).
        if args.unfused:
            eng.dev_scan_records(view, 0, d_norm.data_ptr())
        run_op(d_ws, d_rows, d_out)

    def barrier():
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    eng.set_timing(True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = eng.get_timing()
    eng.set_timing(False)
    if os.environ.get("RB_BENCH_PLACEMENTS") and rank == 0 and args.op == "liftover":
        # experiment: the same launch with the output arena in OTHER physical pages of this process (earlier arenas stay allocated), and
        # the probe's write side alone on each: is the time of a launch a property of where its output lies?
        held = []
        src_b = (total_ops * 4) // 20480 * 20480
        for k_ in range(int(os.environ["RB_BENCH_PLACEMENTS"])):
            t_out, own_t = (d_out, None) if k_ == 0 else big(out_cap + 64, torch.int32)
            held.append((t_out, own_t))
            eng.set_timing(True)
            for _ in range(6):
                run_op(d_ws, d_rows, t_out)
            torch.cuda.synchronize()
            k_ms_ = float(np.mean(eng.get_timing()[-4:]))
            eng.set_timing(False)
            w_ms_ = eng.dev_box_probe(d_ops.data_ptr(), src_b, t_out.data_ptr(), t_out.data_ptr() + src_b, 5, scatter=1 | 8)[0] if out_cap * 4 >= 2 * src_b else float("nan")
            print(f"[placement {k_}] kernel {k_ms_:.3f} ms  probe writes alone {w_ms_:.3f} ms  arena at 0x{t_out.data_ptr():x}", file=sys.stderr)
        for t_out, own_t in held[1:]:
            del t_out
            if own_t:
                own_t.free()
        # ... and the other buffers, the arena fixed: the INPUT ops in other pages, then rows + workspace in other pages
        held = []
        for k_ in range(int(os.environ["RB_BENCH_PLACEMENTS"]) - 1):
            t_ops, own_t = big(total_ops + 64, torch.int32)
            t_ops.copy_(d_ops)
            held.append((t_ops, own_t))
            view_k = eng.batch_view(n_rec, total_ops, t_ops.data_ptr(), d_off.data_ptr(), d_tst.data_ptr(), d_ten.data_ptr(),
                                    d_qst.data_ptr(), d_qen.data_ptr(), d_strand.data_ptr(), d_contig.data_ptr())
            eng.set_timing(True)
            for _ in range(6):
                eng.dev_liftover(plan, view_k, d_norm.data_ptr(), policy, d_ws.data_ptr(), d_rows.data_ptr(), rows_cap, d_out.data_ptr(), out_cap, d_cnt.data_ptr())
            torch.cuda.synchronize()
            print(f"[ops placement {k_ + 1}] kernel {float(np.mean(eng.get_timing()[-4:])):.3f} ms  ops at 0x{t_ops.data_ptr():x}", file=sys.stderr)
            eng.set_timing(False)
        for t_ops, own_t in held:
            del t_ops
            if own_t:
                own_t.free()
        held = []
        for k_ in range(int(os.environ["RB_BENCH_PLACEMENTS"]) - 1):
            t_ws, own_a = big(eng.plan_workspace_bytes(plan, rows_cap), torch.uint8)
            t_rows, own_b = big((rows_cap + 1) * 64, torch.uint8)
            held.append((t_ws, own_a, t_rows, own_b))
            eng.set_timing(True)
            for _ in range(6):
                run_op(t_ws, t_rows, d_out)
            torch.cuda.synchronize()
            print(f"[rows + workspace placement {k_ + 1}] kernel {float(np.mean(eng.get_timing()[-4:])):.3f} ms", file=sys.stderr)
            eng.set_timing(False)
        for t_ws, own_a, t_rows, own_b in held:
            del t_ws, t_rows
            [o_.free() for o_ in (own_a, own_b) if o_]
        run_op(d_ws, d_rows, d_out)  # (the probe overwrote the arena: the checks below read this launch's output)
        torch.cuda.synchronize()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([float(total_ops), float(n_rec)], dtype=torch.float64, device=cdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        job_ops, job_recs = float(tot[0].item()), float(tot[1].item())
    else:
        job_ops, job_recs = float(total_ops), float(n_rec)

    # ---- post-run facts for the roofline (rank 0's shard) ----
    cnt = d_cnt.cpu().numpy().view(rustybam_amd.COUNTERS_DT)[0]
    # ---- digest of the whole job's output in gathered (canonical) order: every rank digests its rows with the number of rows /
    #      records of the ranks before it as bases; the sum is what one GPU holding all records would get (verification only,
    #      outside the timed region; the data path itself has no collective) ----
    mask64 = (1 << 64) - 1
    if use_dist:
        g = [torch.zeros(2, dtype=torch.int64, device=cdev) for _ in range(world)]
        dist.all_gather(g, torch.tensor([n_hits, n_rec], dtype=torch.int64, device=cdev))
        per_rank = [[int(x) for x in t.tolist()] for t in g]
    else:
        per_rank = [[n_hits, n_rec]]
    row_base = sum(h for h, _ in per_rank[:rank])
    d_dig = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    eng.dev_digest_rows(view, d_rows.data_ptr(), n_hits, d_out.data_ptr(), row_base, first, d_dig.data_ptr())
    torch.cuda.synchronize()
    if use_dist:
        g = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world)]
        dist.all_gather(g, d_dig.to(cdev))
        digest = sum(int(t.item()) & mask64 for t in g) & mask64
    else:
        digest = int(d_dig.item()) & mask64
    job_hits = sum(h for h, _ in per_rank)
    if args.debug_skip & 32:  # diagnostics: per-phase shader-clock sums of the clip kernel (last step, every 16th record)
        ph = [int(x) * 16 * 16 for x in cnt["phase"]]
        names = ["job+windows", "stream+resolve", "verdict+finalize", "reservation", "rows+end groups"]
        tot = sum(ph) or 1
        print("[phases] " + "  ".join(f"{n_}={v / tot:.3f}" for n_, v in zip(names, ph)) + f"  | mean cycles/record {tot / max(1, n_rec):.0f}",
              file=sys.stderr)
    rows_t = d_rows[: n_hits * 64].view(torch.int32).view(n_hits, 16)
    status = (rows_t[:, 2] & 0xFFFF)
    out_n = rows_t[:, 3].to(torch.int64)
    n_ok = int((status == 0).sum().item())
    n_out_ops = int((out_n * (status == 0)).sum().item())
    algo_bytes = wl.algorithmic_bytes(total_ops, n_rec, n_hits, n_out_ops)
    # (break-paf is priced with the same single-pass formula although this build walks the ops twice:
    #  collect pieces, clip; its rate is taken over the whole step, not one kernel)
    if args.descriptors:  # nothing is copied: 4 B/op + 48 B/record + (88 + 16) B per hit
        algo_bytes = wl.algorithmic_bytes(total_ops, n_rec, n_hits, 0) + 16 * n_hits
    k_ms = float(np.mean(kern_ms[-args.steps:])) if kern_ms else float("nan")
    if kern_ms and os.environ.get("RB_BENCH_VERBOSE"):  # experiments: the spread of the per-step clip-kernel times
        ks = np.sort(np.asarray(kern_ms[-args.steps:]))
        print(f"[kernel ms] min {ks[0]:.3f}  median {ks[len(ks) // 2]:.3f}  mean {ks.mean():.3f}  max {ks[-1]:.3f}"
              f"  | out_cap {out_cap} rows_cap {rows_cap} d_out 0x{d_out.data_ptr():x} d_ops 0x{d_ops.data_ptr():x}", file=sys.stderr)
    one_walk = args.op == "break" and bool(brk_policy[0] & rustybam_amd.BREAK_ONE_WALK)
    if (args.op == "break" and not one_walk) or irregular or not (k_ms == k_ms):  # (no single dominant kernel under HIP events: the rate is taken over the whole step)
        k_ms = elapsed / args.steps * 1e3
        if args.placement_tries > 1:
            unplaced_ms = None  # (measured on the clip kernels' events: not the basis of this line's rate)
    achieved = algo_bytes / (k_ms * 1e-3) / 1e9
    # HBM-side bytes per launch from the committed PMC run of this same workload (bench.py is not run under --pmc).  The file names the
    # hash of the kernel sources it was measured on: a figure measured on other sources is refused (null + a note), not reported
    traffic, traffic_note = None, None
    try:
        import glob
        for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r[0-9][0-9].json")), reverse=True):
            tj = json.load(open(tf))
            if tj["workload"] != {"records_per_gpu": n_rec, "windows": int(len(w_st)), "workload": args.workload} or args.descriptors or args.op != "liftover" \
                    or args.irregular_frac > 0:
                continue
            if tj.get("kernel_source_sha") == wl.kernel_source_sha():
                traffic, traffic_note = tj["traffic_bytes_per_launch"], f"{os.path.basename(tf)} (measured on these kernel sources, {tj.get('git_head', '?')})"
            else:
                traffic_note = f"{os.path.basename(tf)} is stale: measured on other kernel sources ({tj.get('kernel_source_sha')}), not reported"
            break
    except Exception:
        pass
    ks_ = np.sort(np.asarray(kern_ms[-args.steps:], dtype=np.float64)) if kern_ms else np.zeros(0)
    roofline = {"bound": "hbm", "kernel": ("rb_k_liftover_generic_wave (whole step: the streaming kernel only verifies and defers)" if irregular else "rb_k_liftover_stream") if args.op == "liftover"
                else ("rb_k_liftover_stream_brk" if one_walk else "rb_dev_break (rb_k_break_pieces + rb_k_liftover_stream)"), "achieved": round(achieved, 1), "peak": 8000.0,
                "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic, "traffic_source": traffic_note,
                "kernel_ms": round(k_ms, 4), "algorithmic_bytes": algo_bytes,
                # the clip kernel's own launches of the timed steps (HIP events on the launch stream): min / median / max
                "kernel_ms_steps": ({"min": round(float(ks_[0]), 4), "median": round(float(ks_[len(ks_) // 2]), 4), "max": round(float(ks_[-1]), 4),
                                     "n": int(len(ks_))} if len(ks_) else None),
                "frac_of_measured_copy_ceiling_6290": round(achieved / 6290.0, 4)}
    # ... and the same kernel on the buffers as first allocated (arena candidate 0, the ops array where it lay): five launches under HIP
    # events before any placement.  With --placement-tries 1 nothing is placed and the line's own figure is that number.
    if unplaced_ms is None and args.placement_tries <= 1 and k_ms == k_ms:
        unplaced_ms = k_ms
    roofline["unplaced"] = ({"kernel_ms": round(unplaced_ms, 4), "frac": round(algo_bytes / (unplaced_ms * 1e-3) / 8e12, 4),
                             "note": "arena and ops array as rb_dev_alloc first returned them (no rb_dev_alloc_placed)"} if unplaced_ms else None)

    result = {
        "metric": (("CIGAR-ops/s, liftover over 100 kb sliding windows (whole pass, inputs resident in HBM)" if args.workload == "config3" else
                    f"CIGAR-ops/s, liftover over 100 kb sliding windows, records of {ops_lo}-{ops_hi} ops (config 4's shape; whole pass, inputs resident in HBM)" if c4 else
                    f"CIGAR-ops/s, liftover over 100 kb sliding windows, irregular CIGARs{', legacy binary search' if args.legacy else ''} (generic kernel; whole pass, inputs resident in HBM)" if irregular else
                    "CIGAR-ops/s, liftover over one 1 Mbp window" + (", log-normal op counts" if lognormal else "") +
                    (", records placed uniformly on the target" if placement == "uniform" else "") + " (whole pass, inputs resident in HBM)") if args.op == "liftover"
                   else "CIGAR-ops/s, break-paf --max-size 100" + (f", records of {ops_lo}-{ops_hi} ops (config 4's shape)" if c4 else "") + " (whole pass, inputs resident in HBM)"),
        "value": job_ops * args.steps / elapsed,
        "unit": "CIGAR-ops/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": f"BASELINE.json {args.workload}: {n_rec} records/GPU ({'log-normal(ln 2000, 1.35) in [31, 80000]' if lognormal else f'uniform {ops_lo}-{ops_hi}'} ops, "
                               f"{total_ops} ops on rank 0) x {len(w_st)} windows, placement {placement}, seed {seed:#x}",
                   "records_per_gpu": n_rec, "windows": int(len(w_st)), "parallelism": f"record-range shard x{world} ({args.scaling}: "
                                   + (f"{args.records} records in all, cut on the op-count prefix" if args.scaling == "strong" else f"{args.records} records per GPU") + ")",
                   "full_walk": not args.early_exit, "clip_output": "descriptors" if args.descriptors else "copied ops",
                   "batch_buffers_chunked": {k_: (bool(o_.chunked) if o_ else None) for k_, o_ in (("ops", own_ops), ("workspace", own_ws), ("rows", own_rows), ("out_ops", own_out))},
                   "out_arena_placement": ({"tries_allowed": args.placement_tries, **own_out.placement, "seconds": round(placement_ms / 1e3, 2),
                                            "by": args.placement_by,
                                            "note": "rb_dev_alloc_placed[_by], part of the set-up: candidates of the output arena, each measured (ms: the clip kernels of the step "
                                                    "itself on the candidate under HIP events, or the library's store sweep), the fastest kept; which physical pages the arena has decides "
                                                    "up to 20 % of the clip kernel's time"}
                                           if (own_out is not None and own_out.placement) else None),
                   "ops_placement": ({"incumbent_ms": ops_placement["incumbent_ms"], "launch_ms": ops_placement["launch_ms"], "kept": ops_placement["kept"],
                                      "note": "the input ops array placed the same way, arena fixed: the step itself on the array as it lies (incumbent) and on "
                                              "every candidate (a copy of the ops); the array moves only to a faster candidate"} if ops_placement else None),
                   "batch_memory": ("rb_dev_alloc (2 MB physical chunks)" if lib_alloc and not alloc_note else
                                    "torch allocator (hipMalloc)" + (f"; rb_dev_alloc memory not usable as a tensor here: {alloc_note[0]}" if alloc_note else "")),
                   **({"break_walks": 1 if (brk_policy[0] & rustybam_amd.BREAK_ONE_WALK) else 2} if args.op == "break" else {}),
                   **({"irregular_frac": args.irregular_frac} if args.irregular_frac > 0 else {})},
        "paf_records_per_s": job_recs * args.steps / elapsed,
        "output_digest": f"{digest:#018x}", "job_records": int(job_recs), "job_hits": job_hits,
        "hits_per_gpu": n_hits, "ok_hits_per_gpu": n_ok, "out_ops_per_gpu": n_out_ops,
        "generic_hits_per_gpu": int(cnt["n_generic"]),
        # short records: tiles the step ran (0: none) and the records of tiles the tile kernel handed back to the per-record kernel (streamed
        # twice: a fallback storm would show here)
        **({"tiles_per_gpu": int(cnt["phase"][3]), "tile_records_handed_back_per_gpu": int(cnt["phase"][4]),
            "tile_records_handed_back_frac": round(int(cnt["phase"][4]) / max(1, n_rec), 5)} if not args.debug_skip else {}),
        "roofline": roofline,
        "setup_s": round(gen_s, 2),
        # once per (batch, windows), outside ms_per_step: the host-built plan and the output-sizing calls
        "plan_ms": round(plan_ms, 2), "sizing_ms": round(sizing_ms, 2),
    }
    if e2e:
        result.update(e2e)

    # ---- CPU baseline + sample parity (rank 0, N = 1 only) ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.descriptors and args.op == "liftover" and not irregular and args.irregular_frac == 0:
        from oracle import pyoracle  # checker / baseline only; never on the product path
        from rustybam_amd import capi
        threads = os.cpu_count() or 1

        def sample_idx(k):  # k records spread over the WHOLE batch (every n_rec // k-th one), not its first k
            stride = max(1, n_rec // max(1, k))
            return np.arange(0, n_rec, stride, dtype=np.int64)[:k]

        def sample(k):
            idx = sample_idx(k)
            so = wl.op_offsets(nops[idx])
            sops = np.empty(int(so[-1]), np.uint32)
            for j, r_ in enumerate(idx):  # (the generator is counter based: record r alone is record r of the batch)
                sops[int(so[j]):int(so[j + 1])] = capi.synth_fill_ops_host(seed, first + int(r_), np.array([0, nops[r_]], np.uint64))
            return pyoracle.Batch(sops, so, t_st[idx], t_en[idx], q_st[idx], q_en[idx], strand[idx], np.zeros(len(idx), np.uint32))

        def timed(sb_, nt):
            tb_ = time.perf_counter()
            res = pyoracle.liftover(sb_, w_c, w_st, w_en, n_threads=nt)
            return time.perf_counter() - tb_, res

        # calibrate on the reference's default pool (-t 8, cli.rs:18-19), then one sample for every thread count: the per-base
        # expansion (24 B per aligned base) is bound by the host's memory system, so more threads are not always faster
        t8 = min(8, threads)
        k0 = min(n_rec, 4 * t8)
        dt0, _ = timed(sample(k0), t8)
        k = int(max(k0, min(n_rec, 0.3 * args.cpu_seconds / max(dt0 / k0, 1e-6))))
        sb = sample(k)
        sample_ops = int(sb.op_off[-1])
        runs = {}
        for nt in sorted({t8, min(64, threads), threads}):
            dt, res = timed(sb, nt)
            runs[nt] = dt
            orows, oops = res
        best = min(runs, key=runs.get)
        cpu_s = runs[best]
        result["cpu_baseline"] = {"value": sample_ops / cpu_s, "unit": "CIGAR-ops/s", "cores": best, "kind": "port",
                                  "sample": f"{k} records of the same workload, every {max(1, n_rec // max(1, k))}th of the batch ({sample_ops} ops) x {len(w_st)} "
                                            f"windows, per-base oracle (aligned_pairs expansion) with OpenMP over "
                                            f"records, {cpu_s:.1f} s; the fastest of the thread counts tried",
                                  "records_per_s": k / cpu_s, "host_cores": threads,
                                  "by_threads": {str(nt): {"value": sample_ops / dt, "records_per_s": k / dt, "seconds": round(dt, 2)}
                                                 for nt, dt in sorted(runs.items())},
                                  "t8": {"value": sample_ops / runs[t8], "records_per_s": k / runs[t8], "cores": t8,
                                         "sample": "the same sample on the reference's default pool (-t 8, cli.rs:18-19)"}}
        # parity of the sample at full size: the GPU's rows of the sampled records (one contig: canonical order is record order, the rows
        # of record r are hit_off[r] .. hit_off[r + 1]) vs the oracle, bit for bit
        idx = sample_idx(k)
        hit_off_h = d_ws[: 8 * (n_rec + 1)].view(torch.int64).cpu().numpy()
        sel_rows = np.concatenate([np.arange(hit_off_h[r_], hit_off_h[r_ + 1]) for r_ in idx]) if len(idx) else np.zeros(0, np.int64)
        nrow = len(sel_rows)
        d_sel = torch.from_numpy(sel_rows).to(dev)
        grows = d_rows[: n_hits * 64].view(n_hits, 64)[d_sel].cpu().numpy().view(rustybam_amd.HIT_DT).reshape(-1)
        assert nrow == len(orows), f"sample parity: {nrow} GPU rows vs {len(orows)} oracle rows"
        assert np.array_equal(grows["rec"].astype(np.int64), idx[orows["rec"].astype(np.int64)]), "sample parity: rec"
        for key in ("win", "status", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len", "out_n"):
            ok = orows["status"] == 0 if key not in ("rec", "win", "status") else slice(None)
            assert np.array_equal(grows[key][ok].astype(np.uint64), orows[key][ok].astype(np.uint64)), f"sample parity: {key}"
        okrows = np.nonzero(orows["status"] == 0)[0]
        pick = okrows[:: max(1, len(okrows) // 200)]
        for i in pick:
            a = d_out[int(grows["out_off"][i]): int(grows["out_off"][i]) + int(grows["out_n"][i])].cpu().numpy().view(np.uint32)
            b = oops[int(orows["out_off"][i]): int(orows["out_off"][i]) + int(orows["out_n"][i])]
            assert np.array_equal(a, b), f"sample parity: cigar of row {i}"
        result["parity_sample"] = (f"ok: {nrow} rows of {k} records (every {max(1, n_rec // max(1, k))}th record of the batch) identical to the "
                                   f"per-base oracle, {len(pick)} cigars compared")
        # ---- "also time the op-space CPU path on the full input" -- oracle/rb_opspace.c, a CPU port of the op-space
        #      formulation (no per-base expansion), OpenMP over records, on ALL records of rank 0's batch when the host has the
        #      memory for them (20 GB of ops + 25 GB of clipped CIGARs), else on a fifth of them.  Its rows for the sample above
        #      must equal the per-base oracle's. ----
        try:
            avail_gb = int([ln for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")][0].split()[1]) / 1e6
        except Exception:
            avail_gb = 0.0
        k_os = n_rec if avail_gb > 16 * total_ops * 4 / 1e9 / 5 + 32 else max(k, n_rec // 5)
        if avail_gb > 8:
            ops_host = d_ops[: int(op_off[k_os])].cpu().numpy().view(np.uint32)
            ob = pyoracle.Batch(ops_host, op_off[: k_os + 1], t_st[:k_os], t_en[:k_os], q_st[:k_os], q_en[:k_os], strand[:k_os], np.zeros(k_os, np.uint32))
            os_runs = {}
            for nt in sorted({min(64, threads), threads}):
                tb_ = time.perf_counter()
                got = pyoracle.liftover_opspace(ob, w_c, w_st, w_en, n_threads=nt)
                os_runs[nt] = time.perf_counter() - tb_
                assert got is not None, "the op-space baseline refused the synthetic records"
                if nt == min(64, threads):
                    # the port's rows of the sampled records (record order, the same row ranges) against the per-base oracle's
                    in_os = sel_rows < (hit_off_h[k_os] if k_os < n_rec else len(got[0]))
                    srows = got[0][sel_rows[in_os]]
                    oro = orows[in_os]
                    assert np.array_equal(srows["rec"].astype(np.int64), idx[oro["rec"].astype(np.int64)]), "op-space baseline vs per-base oracle: rec"
                    for key in ("win", "status", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len", "out_n"):
                        okm = oro["status"] == 0 if key not in ("rec", "win", "status") else slice(None)
                        assert np.array_equal(srows[key][okm], oro[key][okm]), f"op-space baseline vs per-base oracle: {key}"
                    n_os_rows = len(got[0])
                    if k_os == n_rec:
                        # ---- parity on 100 % of the job: every GPU row against the op-space port's (which the sample above ties to the
                        #      per-base oracle), field by field, and the digest of every clipped CIGAR -- the device digest kernel over
                        #      the GPU's rows and clips against the same kernel over the port's rows and clips ----
                        tpf = time.perf_counter()
                        assert n_os_rows == n_hits, f"full parity: {n_hits} GPU rows vs {n_os_rows} rows of the op-space port"
                        grows_all = d_rows[: n_hits * 64].cpu().numpy().view(rustybam_amd.HIT_DT)
                        for key in ("rec", "win", "status", "out_n", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len"):
                            nbad = int((grows_all[key].astype(np.uint64) != got[0][key].astype(np.uint64)).sum())
                            assert nbad == 0, f"full parity: {key} differs in {nbad} of {n_hits} rows"
                        del grows_all
                        d_or = torch.from_numpy(capi.hit_rows_from(got[0]).view(np.uint8).reshape(-1)).to(dev)  # (the port's rows are 72 bytes, rb_hit_row 64)
                        d_oo = torch.from_numpy(got[1].view(np.int32)).to(dev)
                        d_dg = torch.zeros(1, dtype=torch.int64, device=dev)
                        torch.cuda.synchronize()
                        eng.dev_digest_rows(view, d_or.data_ptr(), n_hits, d_oo.data_ptr(), row_base, first, d_dg.data_ptr())
                        torch.cuda.synchronize()
                        odig = int(d_dg.item()) & mask64
                        del d_or, d_oo
                        assert world > 1 or odig == digest, f"full parity: digest of the port's clips {odig:#x} != the GPU's {digest:#x}"
                        result["parity_full"] = (f"ok: all {n_hits} rows of all {n_rec} records equal the op-space CPU port's, and the digest of "
                                                 f"all {n_out_ops} clipped ops equals the digest of the port's ({time.perf_counter() - tpf:.1f} s)")
                del got
            best_os = min(os_runs, key=os_runs.get)
            os_ops = int(op_off[k_os])
            result["cpu_baseline"]["opspace"] = {
                "value": os_ops / os_runs[best_os], "unit": "CIGAR-ops/s", "cores": best_os, "kind": "port",
                "records_per_s": k_os / os_runs[best_os], "rows": n_os_rows,
                "sample": (f"all {k_os} records of the batch" if k_os == n_rec else f"the first {k_os} records") +
                          f" ({os_ops} ops) x {len(w_st)} windows, op-space CPU port (oracle/rb_opspace.c: prefix sums + binary search per "
                          f"boundary, no per-base expansion), OpenMP over records, {os_runs[best_os]:.1f} s; its rows for the parity sample "
                          f"equal the per-base oracle's",
                "by_threads": {str(nt): {"value": os_ops / dt, "seconds": round(dt, 2)} for nt, dt in sorted(os_runs.items())}}
            del ops_host, ob

    # ---- the box: why this line reads what it reads on THIS machine (rank 0, outside the timed region, after every check: the probe
    #      overwrites the output arena).  (a) the clock the clip kernel holds: its diagnostics build (same code + s_memtime /
    #      s_memrealtime stamps around the streaming loop of every 16th record, written to a counter block nothing reads) after >= 3 s
    #      of back-to-back launches; the same run leaves the time each wave was done, hence how long the launch ran on after 95 % of
    #      them had retired.  (b) the library's memory-mix probe on these very buffers: the kernel's bytes in its access shape without
    #      its instructions -- a box that is slow at moving bytes shows here, a box that holds a lower clock shows in (a).
    if rank == 0 and not args.no_box and args.op == "liftover" and not args.descriptors and not irregular:
        box = {"smi_before_start": smi}
        try:
            dpol = policy | ((128 | 256) << 8)
            eng.set_timing(True)
            if sampler:
                sampler.go()
            tb0, n_diag, wall0 = time.perf_counter(), 0, time.time()
            while n_diag < 10 or time.perf_counter() - tb0 < 3.0:
                eng.dev_liftover(plan, view, d_norm.data_ptr(), dpol, d_ws.data_ptr(), d_rows.data_ptr(), rows_cap, d_out.data_ptr(), out_cap, d_cnt.data_ptr())
                n_diag += 1
                if n_diag % 10 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            wall1 = time.time()
            if sampler:  # rocm-smi's view of the chip while the clip kernel ran back to back (clocks incl. memory and fabric, temperatures, power)
                box["smi_under_load"] = sampler.stop(wall0 + 0.3, wall1)
                sampler = None
            dms = eng.get_timing()
            eng.set_timing(False)
            dc = d_cnt.cpu().numpy().view(rustybam_amd.COUNTERS_DT)[0]
            ph = [int(x) for x in dc["phase"]]
            if ph[1]:
                box["kernel_clock_mhz"] = round(ph[0] * 64.0 / ph[1] * 100.0, 1)
                box["kernel_clock_note"] = (f"diagnostics build of rb_k_liftover_stream, {n_diag} launches back to back ({time.perf_counter() - tb0:.1f} s), "
                                            f"stamps of {ph[2]} records of the last launch; that build's launches took {float(np.mean(dms[-10:])):.3f} ms")
            off = eng.plan_diag_stamps_offset(plan, rows_cap)
            d_ws[off: off + 4 * n_rec].zero_()
            torch.cuda.synchronize()
            eng.dev_liftover(plan, view, d_norm.data_ptr(), dpol, d_ws.data_ptr(), d_rows.data_ptr(), rows_cap, d_out.data_ptr(), out_cap, d_cnt.data_ptr())
            torch.cuda.synchronize()
            st_ = d_ws[off: off + 4 * n_rec].view(torch.int32).cpu().numpy().view(np.uint32)
            st_ = st_[st_ != 0]  # (a wave that left without a stamp -- a record the kernel handed to the general path -- has none)
            rel = np.sort((st_ - st_.min()).astype(np.uint32))  # (mod 2^32: a launch is far shorter than the counter's 43 s)
            box["launch_tail"] = {"ms_after_95pct_of_waves_done": round(float(rel[-1] - rel[int(0.95 * (len(rel) - 1))]) * 1e-5, 4),
                                  "ms_after_99pct": round(float(rel[-1] - rel[int(0.99 * (len(rel) - 1))]) * 1e-5, 4),
                                  "ms_first_to_last_wave_done": round(float(rel[-1]) * 1e-5, 4)}
            # where a record's cycles go on this box: one launch of the diagnostics build with its per-phase clock sums on (every 16th record)
            eng.dev_liftover(plan, view, d_norm.data_ptr(), policy | (32 << 8), d_ws.data_ptr(), d_rows.data_ptr(), rows_cap, d_out.data_ptr(), out_cap, d_cnt.data_ptr())
            torch.cuda.synchronize()
            pc = [int(x) * 256 for x in d_cnt.cpu().numpy().view(rustybam_amd.COUNTERS_DT)[0]["phase"]][:5]
            box["phase_cycles_per_record"] = dict(zip(["job_and_windows", "stream_and_resolve", "verdict_and_finalize", "reservation", "rows_and_end_groups"],
                                                      [round(v / max(1, n_rec)) for v in pc]))
            src_bytes = (total_ops * 4) // 20480 * 20480
            if out_cap * 4 >= 2 * src_bytes:
                pms, pmhz = eng.dev_box_probe(d_ops.data_ptr(), src_bytes, d_out.data_ptr(), d_out.data_ptr() + src_bytes, 10)
                box["probe_ms"] = round(pms, 4)
                box["probe_clock_mhz"] = round(pmhz, 1)
                box["probe_gb_s"] = round(2.2 * src_bytes / (pms * 1e-3) / 1e9, 1)
                sms, smhz = eng.dev_box_probe(d_ops.data_ptr(), src_bytes, d_out.data_ptr(), d_out.data_ptr() + src_bytes, 10, scatter=True)
                box["probe_scattered_ms"] = round(sms, 4)  # (the same bytes, the concurrently running waves spread over the whole arrays)
                box["probe_flat_ms"] = round(eng.dev_box_probe(d_ops.data_ptr(), src_bytes, d_out.data_ptr(), d_out.data_ptr() + src_bytes, 10, scatter=2)[0], 4)
                box["probe_flat_scattered_ms"] = round(eng.dev_box_probe(d_ops.data_ptr(), src_bytes, d_out.data_ptr(), d_out.data_ptr() + src_bytes, 10, scatter=3)[0], 4)
                # the two sides of the mix alone, and the clip kernel's diagnostics build without its speculative stores / without any
                # store of clipped ops: on this memory system the mix costs about what its reads and its writes cost one after the other
                box["probe_read_only_ms"] = round(eng.dev_box_probe(d_ops.data_ptr(), src_bytes, d_out.data_ptr(), d_out.data_ptr() + src_bytes, 10, scatter=1 | 4)[0], 4)
                box["probe_write_only_ms"] = round(eng.dev_box_probe(d_ops.data_ptr(), src_bytes, d_out.data_ptr(), d_out.data_ptr() + src_bytes, 10, scatter=1 | 8)[0], 4)
                for key, bits in (("kernel_without_speculative_stores_ms", 64), ("kernel_reads_only_ms", 64 | 512)):
                    eng.set_timing(True)
                    for _ in range(6):
                        eng.dev_liftover(plan, view, d_norm.data_ptr(), policy | (bits << 8), d_ws.data_ptr(), d_rows.data_ptr(), rows_cap, d_out.data_ptr(), out_cap, d_cnt.data_ptr())
                    torch.cuda.synchronize()
                    box[key] = round(float(np.mean(eng.get_timing()[-4:])), 4)
                    eng.set_timing(False)
                box["probe_note"] = (f"rb_dev_box_probe on this run's buffers: {src_bytes} B read from the ops array, 1.2 x that written to the output "
                                     "arena in the clip kernel's access shape (32 contiguous bytes per lane, two slots), no other instructions")
                if k_ms == k_ms:
                    box["kernel_over_probe"] = round(k_ms / (pms * algo_bytes / (2.2 * src_bytes)), 4)  # (probe scaled to the kernel's algorithmic bytes)
        except Exception as e:  # (diagnostics: never in the way of the line)
            box["error"] = f"{type(e).__name__}: {e}"
        result["box"] = box
    if sampler:
        sampler.stop(0, 0)
    if rank == 0:
        print(json.dumps(result))
    d_ops = d_ws = d_rows = d_out = rows_t = None
    [o.free() for o in (own_ops, own_ws, own_rows, own_out) if o]
    eng.plan_destroy(plan)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
