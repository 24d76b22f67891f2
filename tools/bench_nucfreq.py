"""per-position A/C/G/T counts, 30x coverage of one 250 Mbp contig by 15 kb reads.

Prints one JSON line shaped like bench.py's.  A step is one rb_dev_nucfreq call over the whole contig with everything
resident in HBM.  Parity of this workload is checked by tests/soak/soak_nucfreq.py; this script never touches the oracle.

  python tools/bench_nucfreq.py [--contig 250000000] [--coverage 30] [--read-len 15000] [--steps 5] [--warmup 1]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
SEED = 0x5EED0005
N_EVENTS = 15  # indel events per read: 16 M runs, 31 ops


def make_reads(contig, coverage, read_len, seed=SEED):
    """sorted starts; cigar = M (I|D) M ... M with query length read_len exactly"""
    rng = np.random.default_rng(seed)
    n = int(contig * coverage // read_len)
    if read_len < 2000:  # short reads: one M op each (--read-len 150: what the lane-per-read path of the tile kernel is for)
        pos = np.sort(rng.integers(0, contig - read_len, n)).astype(np.int64)
        return pos, np.full(n, (read_len << 4) | 0, np.uint32), np.arange(n + 1, dtype=np.uint64), n
    pos = np.sort(rng.integers(0, contig - 2 * read_len, n)).astype(np.int64)
    is_ins = rng.random((n, N_EVENTS)) < 0.5
    ev_len = np.where(rng.random((n, N_EVENTS)) < 0.7, 1, rng.integers(2, 30, (n, N_EVENTS))).astype(np.int64)
    ins_total = (ev_len * is_ins).sum(axis=1)
    m_total = read_len - ins_total                                     # query bases in M
    cuts = np.sort(rng.integers(1, (m_total - 1)[:, None], (n, N_EVENTS)), axis=1)
    cuts += np.arange(N_EVENTS)[None, :]                               # strictly increasing -> every M run >= 1
    edges = np.concatenate([np.zeros((n, 1), np.int64), cuts, (m_total + N_EVENTS)[:, None]], axis=1)
    m_len = np.diff(edges, axis=1) - np.concatenate([np.zeros((n, 1), np.int64), np.ones((n, N_EVENTS), np.int64)], axis=1)
    m_len[:, 0] = edges[:, 1]
    m_len = np.maximum(m_len, 1)
    m_len[:, -1] += m_total - m_len.sum(axis=1)                        # fix the total
    assert (m_len >= 1).all()
    ops = np.zeros((n, 2 * N_EVENTS + 1), np.uint32)
    ops[:, 0::2] = (m_len << 4).astype(np.uint32)                      # M = 0
    ops[:, 1::2] = ((ev_len << 4) | np.where(is_ins, 1, 2)).astype(np.uint32)
    op_off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(2 * N_EVENTS + 1))
    return pos, ops.reshape(-1), op_off, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contig", type=int, default=250_000_000)
    ap.add_argument("--coverage", type=int, default=30)
    ap.add_argument("--read-len", type=int, default=15000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--phases", action="store_true", help="a -DNF_DIAG variant of the library (RB_VARIANT): print the shader-clock length of a tile's phases, sampled on every 64th tile")
    ap.add_argument("--placement-tries", type=int, default=1, help="candidates rb_dev_alloc_placed may take for the counts array (store sweep)")
    a = ap.parse_args()
    import torch
    import rustybam_amd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream(dev))  # torch's kernels and the engine's on one real stream
    eng = rustybam_amd.Engine(0, torch.cuda.current_stream().cuda_stream)
    t0 = time.time()
    pos, ops, op_off, n = make_reads(a.contig, a.coverage, a.read_len)
    bytes_per_read = (a.read_len + 1) // 2
    g = torch.Generator(device=dev)
    g.manual_seed(SEED)
    lut = torch.tensor([(1 << (k >> 2)) << 4 | (1 << (k & 3)) for k in range(16)], dtype=torch.uint8, device=dev)
    d_seq = lut[torch.randint(0, 16, (n * bytes_per_read + 64,), dtype=torch.uint8, device=dev, generator=g).long()] if n * bytes_per_read < (1 << 28) else None
    from rustybam_amd import capi
    lib_alloc = os.environ.get("RB_BENCH_TORCH_ALLOC") != "1"  # the big arrays from the library's allocator (2 MB physical chunks)
    if d_seq is None:  # in pieces: the index tensor of a one-shot gather would be 8x the sequence
        own_seq = capi.DevBuf(eng, torch, n * bytes_per_read + 64, torch.uint8) if lib_alloc else None
        d_seq = own_seq.t if lib_alloc else torch.empty(n * bytes_per_read + 64, dtype=torch.uint8, device=dev)
        step = 1 << 27
        for o in range(0, d_seq.numel(), step):
            m = min(step, d_seq.numel() - o)
            d_seq[o:o + m] = lut[torch.randint(0, 16, (m,), dtype=torch.uint8, device=dev, generator=g).long()]
    i64 = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to(dev)
    d_pos, d_opoff = i64(pos), i64(op_off)
    d_ops = torch.from_numpy(np.concatenate([ops, np.zeros(8, np.uint32)]).view(np.int32)).to(dev)
    d_seqoff = i64(np.arange(n, dtype=np.uint64) * np.uint64(bytes_per_read))
    d_lseq = torch.full((n,), a.read_len, dtype=torch.int32, device=dev)
    d_tid = torch.zeros(n, dtype=torch.int32, device=dev)
    d_flag = torch.zeros(n, dtype=torch.int32, device=dev)
    d_rgtid = torch.zeros(1, dtype=torch.int32, device=dev)
    d_rgst = torch.zeros(1, dtype=torch.int64, device=dev)
    d_rgen = torch.full((1,), a.contig, dtype=torch.int64, device=dev)
    d_outoff = torch.tensor([0, a.contig], dtype=torch.int64, device=dev)
    own_counts = capi.DevBuf(eng, torch, a.contig * 4 + 16, torch.int32, placed_tries=a.placement_tries) if lib_alloc else None
    d_counts = own_counts.t if lib_alloc else torch.empty(a.contig * 4 + 16, dtype=torch.int32, device=dev)
    d_status = torch.empty(n + 1, dtype=torch.int32, device=dev)
    d_ctr = torch.zeros(6, dtype=torch.int64, device=dev)
    wsb = eng.nucfreq_workspace_bytes(n, 1, a.contig)
    d_ws = torch.empty(wsb + 256, dtype=torch.uint8, device=dev)
    ws_ptr = (d_ws.data_ptr() + 255) & ~255
    setup = time.time() - t0

    def step():
        eng.dev_nucfreq(n, d_ops.data_ptr(), d_opoff.data_ptr(), d_seq.data_ptr(), d_seqoff.data_ptr(), d_lseq.data_ptr(), d_tid.data_ptr(),
                        d_pos.data_ptr(), d_flag.data_ptr(), 1, d_rgtid.data_ptr(), d_rgst.data_ptr(), d_rgen.data_ptr(), d_outoff.data_ptr(),
                        a.contig, d_counts.data_ptr(), d_status.data_ptr(), d_ctr.data_ptr(), ws_ptr, wsb)
    torch.cuda.synchronize()  # (the engine has its own stream: torch's generation kernels must be done before it reads their output)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()          # device-wide synchronisation brackets the timed steps
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    ctr = d_ctr.cpu().numpy()
    if a.phases:  # (the workspace's layout up to the scans' partials: capi.hip nf_ws_layout; the diagnostics build adds its sums behind the eighth word)
        up = lambda x: (x + 255) & ~255
        off = up((n + 1) * 8)
        off = up(off + (n + 1) * 48)
        off = up(off + (1 + 2) * 8)
        base = ws_ptr - d_ws.data_ptr() + off + 64
        v = d_ws[base:base + 64].view(torch.int64).cpu().numpy()
        names = ["zeroing + first barrier", "first chunk scanned (records, ops: the dependent trips)", "the wave's reads", "waiting for the other waves",
                 "depth scan", "output", "whole tile"]
        waves = max(int(v[7]), 1)
        print({"sampled_waves": waves, **{nm: round(float(v[k]) / waves, 1) for k, nm in enumerate(names)}, "unit": "shader clocks per wave and tile (last call)"}, file=sys.stderr)
    assert int((d_status[:n] != 0).sum()) == 0 and ctr[3] == 0
    check = not os.environ.get("NF_NO_CHECK")
    c = d_counts[:a.contig * 4].view(-1, 4)
    tot = int((c[:, 0] & 0x7FFFFFFF).sum() + c[:, 1].sum() + c[:, 2].sum() + c[:, 3].sum())
    m_bases = int(((ops >> 4) * ((ops & 15) == 0)).sum())
    assert tot == m_bases or not check, (tot, m_bases)   # every M base lands inside the contig and is A/C/G/T: a checksum of the whole pile
    alg = n * bytes_per_read + 4 * len(ops) + 16 * a.contig + 44 * n
    traffic = None  # HBM-side bytes per launch from the committed PMC run of this same workload (this script is not run under --pmc)
    for name in ("traffic_nf_r06.json", "traffic_nf_r02.json", "traffic_nf_r01.json"):
        try:
            tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", name)))
            if tj["workload"] == {"contig": a.contig, "coverage": a.coverage, "read_len": a.read_len}:
                traffic = tj["traffic_bytes_per_launch"]
                break
        except Exception:
            pass
    print(json.dumps({
        "metric": "read bases piled up per second (A/C/G/T counts at every position, inputs resident in HBM)", "value": m_bases / (ms * 1e-3),
        "unit": "bases/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "dtype": "u8 / u16 in LDS, u32 out",
        "data": "synthetic", "config": {"workload": f"SURVEY 8d config5: {a.coverage}x of one {a.contig} bp contig, {n} reads of {a.read_len} bases, "
                                        f"{len(ops) // n} ops each, seed 0x5eed0005"},
        "positions_per_s": a.contig / (ms * 1e-3), "max_depth": int(ctr[0]), "covered": int(ctr[1]),
        "roofline": {"bound": "hbm", "kernel": "rb_k_nf_tiles (whole call)", "achieved": alg / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg / (ms * 1e-3) / 1e9 / 8000.0, "traffic": traffic, "algorithmic_bytes": alg,
                     "note": "bound by instruction issue, not by HBM (profiles/r06_nf_summary.md)"},
        "setup_s": round(setup, 2)}))


if __name__ == "__main__":
    main()
