"""BASELINE.json configs[2] at FULL size (1e6 records, 5e9 ops, 3000 sliding windows: what bench.py times) through properties that
need no oracle: every one of the ~12.6 M clipped records must pass the reference's own check_integrity (paf.rs:825-857: target span =
reference-consuming lengths, query span = query-consuming lengths, nmatch = M/=/X lengths, aln_len = all lengths), and the two
independent emission routes of the clip kernel -- copied ops and clip descriptors into the original CIGAR -- must describe the same
ops (a checksum per clip).  Segment sums come from prefix sums over the whole op arrays (torch, on the device)."""
import os

import numpy as np
import pytest

import rustybam_amd
from rustybam_amd import workload as wl

pytestmark = pytest.mark.gpu


def _seg(P, a, b):
    """sum over [a, b) from an inclusive prefix array"""
    import torch
    hi = P[(b - 1).clamp(min=0)]
    hi = torch.where(b > 0, hi, torch.zeros_like(hi))
    lo = P[(a - 1).clamp(min=0)]
    lo = torch.where(a > 0, lo, torch.zeros_like(lo))
    return hi - lo


def _config3(is_break=False, config2=False):
    import torch
    n_rec = int(os.environ.get("RB_FULLSIZE_RECORDS", "100000" if config2 else "1000000"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(dev))  # torch's kernels and the engine's on one real stream (a NULL handle would mean "private stream")
    eng = rustybam_amd.Engine(0, torch.cuda.current_stream().cuda_stream)
    seed = wl.SEED_CONFIG2 if config2 else wl.SEED_CONFIG3
    w_c, w_st, w_en = wl.sliding_windows(3000)
    if config2:  # BASELINE.json configs[1]: one 1 Mbp window, every record placed so that it overlaps it
        w_c, w_st, w_en = np.zeros(1, np.uint32), np.array([12_000_000], np.uint64), np.array([13_000_000], np.uint64)
    nops = wl.n_ops(seed, 0, n_rec)
    op_off = wl.op_offsets(nops)
    total_ops = int(op_off[-1])
    i64 = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
    d_off = i64(op_off)
    d_ops = torch.empty(total_ops + 64, dtype=torch.int32, device=dev)
    eng.dev_synth_fill_ops(seed, 0, n_rec, d_off.data_ptr(), d_ops.data_ptr())
    zeros = torch.zeros(n_rec, dtype=torch.int64, device=dev)
    d_contig = torch.zeros(n_rec, dtype=torch.int32, device=dev)
    d_strand0 = torch.full((n_rec,), ord("+"), dtype=torch.uint8, device=dev)
    d_red = torch.empty(n_rec * 72, dtype=torch.uint8, device=dev)
    d_norm = torch.empty(n_rec * 64, dtype=torch.uint8, device=dev)
    v0 = eng.batch_view(n_rec, total_ops, d_ops.data_ptr(), d_off.data_ptr(), zeros.data_ptr(), zeros.data_ptr(), zeros.data_ptr(),
                        zeros.data_ptr(), d_strand0.data_ptr(), d_contig.data_ptr())
    eng.dev_scan_records(v0, d_red.data_ptr(), 0)
    torch.cuda.synchronize()
    red = d_red.cpu().numpy().view(rustybam_amd.REDUCE_DT)
    t_st, t_en, q_st, q_en, strand = wl.headers(seed, 0, red["t_bases"], red["q_bases"], "overlap" if config2 else "uniform")
    d_c = [i64(x) for x in (t_st, t_en, q_st, q_en)]
    d_strand = torch.from_numpy(strand).to(dev)
    view = eng.batch_view(n_rec, total_ops, d_ops.data_ptr(), d_off.data_ptr(), *[x.data_ptr() for x in d_c], d_strand.data_ptr(), d_contig.data_ptr())
    plan = eng.plan_create(op_off, np.zeros(n_rec, np.uint32), w_c, w_st, w_en)
    d_cnt = torch.zeros(64, dtype=torch.uint8, device=dev)

    def run(policy, rows_cap, out_cap):
        if not (policy & rustybam_amd.LIFT_DESCRIPTORS):
            out_cap = max(out_cap, eng.plan_out_capacity(plan, is_break))
        for _ in range(6):
            ws = torch.empty(eng.plan_workspace_bytes(plan, rows_cap), dtype=torch.uint8, device=dev)
            rows = torch.full(((rows_cap + 1) * 64,), 0xEE, dtype=torch.uint8, device=dev)   # (a row nobody writes stays 0xEE..)
            out = torch.empty(out_cap + 64, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()   # the engine runs on its own stream: torch's fill must have landed before it starts
            if is_break:
                eng.dev_break(plan, view, d_norm.data_ptr(), 100, policy, ws.data_ptr(), rows.data_ptr(), rows_cap, out.data_ptr(), out_cap, d_cnt.data_ptr())
            else:
                eng.dev_liftover(plan, view, d_norm.data_ptr(), policy, ws.data_ptr(), rows.data_ptr(), rows_cap, out.data_ptr(), out_cap, d_cnt.data_ptr())
            torch.cuda.synchronize()
            cnt = d_cnt.cpu().numpy().view(rustybam_amd.COUNTERS_DT)[0]
            if not cnt["overflow"]:
                n = int(cnt["n_hits"])
                r = rows[:n * 64].view(torch.int32).view(n, 16).clone()
                torch.cuda.synchronize()   # ... and the copy must be done before the buffer is freed and handed to the next call
                return r, out
            rows_cap = max(rows_cap, int(cnt["n_hits"]) + 64)
            out_cap = max(out_cap * 2, int(int(cnt["out_ops_needed"]) * 1.25) + 4096)
            del ws, rows, out
        raise AssertionError("could not size the outputs")

    return dict(torch=torch, dev=dev, eng=eng, run=run, n_rec=n_rec, total_ops=total_ops, d_ops=d_ops, d_off=d_off, keep=(d_c, d_strand, d_contig, d_norm, zeros, plan, view),
                host=dict(seed=seed, nops=nops, op_off=op_off, t_st=t_st, t_en=t_en, q_st=q_st, q_en=q_en, strand=strand, windows=(w_c, w_st, w_en)), view=view)


def _u64(r, c):
    import torch
    return (r[:, c].to(torch.int64) & 0xFFFFFFFF) | (r[:, c + 1].to(torch.int64) << 32)


def _check_integrity(torch, dev, rows, out):
    """paf.rs:825-857 on every row: spans and counts against segment sums of the row's ops"""
    n = rows.shape[0]
    out_n = rows[:, 3].to(torch.int64)
    off = _u64(rows, 14)
    t_span, q_span = _u64(rows, 6) - _u64(rows, 4), _u64(rows, 10) - _u64(rows, 8)
    nmatch, aln_len = rows[:, 12].to(torch.int64) & 0xFFFFFFFF, rows[:, 13].to(torch.int64) & 0xFFFFFFFF
    cap = int((off + out_n).max().item())
    words = out[:cap]
    one = torch.ones((), dtype=torch.int32, device=dev)
    for name, class_mask, want in (("aln_len", None, aln_len), ("nmatch", 0x181, nmatch), ("target span", 0x18D, t_span), ("query span", 0x193, q_span)):
        x = words >> 4                                     # lengths (one 4 B/op temporary at a time: the arenas span 32 GB)
        if class_mask is not None:                         # ... of the ops whose code belongs to the class (paf.rs:946-996)
            m = words & 15
            m = torch.bitwise_right_shift(one * class_mask, m)
            m &= 1
            x *= m
            del m
        P = torch.cumsum(x, 0, dtype=torch.int64)
        got = _seg(P, off, off + out_n)
        bad = int((got != want).sum())
        assert bad == 0, f"{name}: {bad} of {n} clipped records fail check_integrity"
        del P, x, got
    return words, off, out_n


def test_full_size_liftover_integrity_and_descriptor_checksums():
    C = _config3()
    torch, dev, eng, run, n_rec, total_ops, d_ops, d_off = (C[k] for k in ("torch", "dev", "eng", "run", "n_rec", "total_ops", "d_ops", "d_off"))
    base = rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_FUSED_SCAN
    rows_c, out_c = run(base, 16 * n_rec, int(1.6 * total_ops))
    rows_d, desc = run(base | rustybam_amd.LIFT_DESCRIPTORS, rows_c.shape[0] + 64, 4 * (rows_c.shape[0] + 64) + 65536)
    n = rows_c.shape[0]
    assert n > 12 * n_rec and rows_d.shape[0] == n
    assert int((rows_c[:, 0] == -286331154).sum()) == 0 and int((rows_d[:, 0] == -286331154).sum()) == 0   # every row was written
    status = rows_c[:, 2] & 0xFFFF
    assert int((status != 0).sum()) == 0                       # every (record, window) hit of this workload clips
    # ---- the two routes agree on everything but where the ops are: same canonical order, same coordinates ----
    for c in (0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13):
        assert torch.equal(rows_c[:, c], rows_d[:, c]), f"column {c}"
    assert torch.equal(rows_c[:, 2] & 0xFFFF, rows_d[:, 2] & 0xFFFF)
    # ---- check_integrity on every clipped record (copied route) ----
    words, off_c, out_n = _check_integrity(torch, dev, rows_c, out_c)
    # ---- checksum per clip: copied ops vs the descriptor applied to the original cigar ----
    P_out = torch.cumsum(words, 0, dtype=torch.int64)
    sum_c = _seg(P_out, off_c, off_c + out_n)
    del P_out, words, out_c
    torch.cuda.empty_cache()
    d4 = desc[:4 * n].view(n, 4).to(torch.int64) & 0xFFFFFFFF   # first op (in the record's original cigar), count, first length, last length
    rec = rows_d[:, 0].to(torch.int64) & 0xFFFFFFFF
    first = d_off[rec] + d4[:, 0]
    cntd, fl, ll = d4[:, 1], d4[:, 2], d4[:, 3]
    assert torch.equal(cntd, out_n)
    P_in = torch.cumsum(d_ops[:total_ops], 0, dtype=torch.int64)
    sum_d = _seg(P_in, first, first + cntd)
    del P_in
    w_first = d_ops[first].to(torch.int64)
    w_last = d_ops[first + cntd - 1].to(torch.int64)
    one = cntd == 1
    new_first = torch.where(one & (fl > 0) & (ll > 0), fl + ll - (w_first >> 4), torch.where(fl > 0, fl, w_first >> 4))
    new_last = torch.where(ll > 0, ll, w_last >> 4)
    sum_d = sum_d + ((new_first - (w_first >> 4)) << 4) + torch.where(one, torch.zeros_like(sum_d), (new_last - (w_last >> 4)) << 4)
    bad = int((sum_d != sum_c).sum())
    assert bad == 0, f"{bad} of {n} clips differ between the copied and the descriptor route"
    eng.close()


def test_full_size_liftover_every_row_equals_the_opspace_oracle(oracle):
    """Parity on 100 % of the full-size job: all ~12.6 M hit rows of config 3 against the op-space CPU port (oracle/rb_opspace.c, itself
    held to the per-base oracle by tests/test_oracle_opspace.py) on all 1e6 records, field by field; and the order-sensitive digest
    of every clipped CIGAR: rb_dev_digest_rows (= its numpy twin, tests/test_gpu_digest.py) over the GPU's rows and clips against the
    same kernel over the oracle's rows and clips.  Needs the host memory for 20 GB of ops and 25 GB of clipped CIGARs."""
    try:
        avail_gb = int([ln for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")][0].split()[1]) / 1e6
    except Exception:
        avail_gb = 0.0
    n_full = int(os.environ.get("RB_FULLSIZE_RECORDS", "1000000"))
    if avail_gb < 70 * n_full / 1e6 + 8:
        pytest.skip(f"host has {avail_gb:.0f} GB available: not enough for the full-size op-space oracle")
    C = _config3()
    torch, dev, eng, run, n_rec, total_ops, d_ops = (C[k] for k in ("torch", "dev", "eng", "run", "n_rec", "total_ops", "d_ops"))
    H = C["host"]
    rows_g, out_g = run(rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_FUSED_SCAN, 16 * n_rec, int(1.6 * total_ops))
    n = rows_g.shape[0]
    ops_host = d_ops[:total_ops].cpu().numpy().view(np.uint32)
    ob = oracle.Batch(ops_host, H["op_off"], H["t_st"], H["t_en"], H["q_st"], H["q_en"], H["strand"], np.zeros(n_rec, np.uint32))
    got = oracle.liftover_opspace(ob, *H["windows"], n_threads=min(64, os.cpu_count() or 1))
    assert got is not None, "the op-space port refused the synthetic records"
    orows, oops = got
    assert len(orows) == n, f"{n} GPU rows, {len(orows)} oracle rows"
    grows = rows_g.cpu().numpy().view(np.uint8).reshape(n, 64).view(rustybam_amd.HIT_DT).reshape(n)
    for key in ("rec", "win", "status", "out_n", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len"):
        bad = int((grows[key].astype(np.uint64) != orows[key].astype(np.uint64)).sum())
        assert bad == 0, f"{key}: {bad} of {n} rows differ from the op-space oracle"
    assert int(((grows["flags"] & 1) != (orows["flags"] & 1)).sum()) == 0            # "strictly inside" clones (liftover.rs:23-25)
    # ---- every clipped CIGAR: one digest kernel, two inputs ----
    d_dig = torch.zeros(2, dtype=torch.int64, device=dev)
    view = C["view"]
    d_rows_g = rows_g.contiguous()
    torch.cuda.synchronize()
    eng.dev_digest_rows(view, d_rows_g.data_ptr(), n, out_g.data_ptr(), 0, 0, d_dig[0:].data_ptr())
    torch.cuda.synchronize()
    del out_g
    torch.cuda.empty_cache()
    from rustybam_amd import capi
    d_orows = torch.from_numpy(capi.hit_rows_from(orows).view(np.uint8).reshape(-1)).to(dev)   # (the oracle's rows are 72 bytes, rb_hit_row 64)
    d_oops = torch.from_numpy(oops.view(np.int32)).to(dev)
    torch.cuda.synchronize()
    eng.dev_digest_rows(view, d_orows.data_ptr(), n, d_oops.data_ptr(), 0, 0, d_dig[1:].data_ptr())
    torch.cuda.synchronize()
    dg = d_dig.cpu().numpy().view(np.uint64)
    assert dg[0] == dg[1], f"digest of the GPU's clips {int(dg[0]):#x} != digest of the oracle's {int(dg[1]):#x}"
    assert int(orows["out_n"].astype(np.int64).sum()) == len(oops) or len(oops) >= int(orows["out_n"].astype(np.int64).sum())
    eng.close()


def test_full_size_config2_one_window_every_record_overlaps(oracle):
    """BASELINE.json configs[1] at full size: 1e5 records of 1000-9000 ops, each placed so that it overlaps the one
    window chr1:12,000,000-13,000,000.  Properties that need no oracle: one hit per record, in record order; every clip passes the
    reference's check_integrity, lies inside the window and its record, and starts / ends within the window's first / last bases
    the record covers (a clip only shrinks to the next match op); plus the oracle on a sample, row by row and op by op, and the
    digest of the copied clips against the digest of their descriptors."""
    from rustybam_amd import capi
    C = _config3(config2=True)
    torch, dev, eng, run, n_rec, total_ops = (C[k] for k in ("torch", "dev", "eng", "run", "n_rec", "total_ops"))
    H = C["host"]
    base = rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_FUSED_SCAN
    rows, out = run(base, n_rec + 64, int(1.2 * total_ops))
    n = rows.shape[0]
    assert n == n_rec                                                        # every record overlaps the window exactly once
    assert torch.equal(rows[:, 0].to(torch.int64), torch.arange(n, device=dev)) and int(rows[:, 1].abs().sum()) == 0
    st = rows[:, 2] & 0xFFFF
    assert int((st >= 16).sum()) == 0 and int((st != 0).sum()) < n // 1000   # nothing panics; a window edge inside an indel gives None, rarely
    ok = st == 0
    r_ok = rows[ok]
    _check_integrity(torch, dev, r_ok, out)
    i64 = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
    t_st, t_en = i64(H["t_st"])[ok], i64(H["t_en"])[ok]
    c_st, c_en = _u64(r_ok, 4), _u64(r_ok, 6)
    lo, hi = torch.clamp(t_st, min=12_000_000), torch.clamp(t_en, max=13_000_000)
    inside = (r_ok[:, 2] >> 16) & 1
    free = inside == 0
    assert bool((c_st >= lo).all()) and bool((c_en <= hi).all()) and bool((c_st < c_en).all())
    # the reference walks from the window edge to the next match op: what it skips is one run of non-match ops (I / D of at most 5000 bases here)
    assert bool(((c_st - lo)[free] <= 5000).all()) and bool(((hi - c_en)[free] <= 5000).all())
    assert int(inside.sum()) == int(((i64(H["t_st"]) > 12_000_000) & (i64(H["t_en"]) < 13_000_000))[ok].sum())   # liftover.rs:23-25
    # ---- the oracle on the first records ----
    k = 300
    so = wl.op_offsets(H["nops"][:k])
    sops = capi.synth_fill_ops_host(H["seed"], 0, so)
    ob = oracle.Batch(sops, so, H["t_st"][:k], H["t_en"][:k], H["q_st"][:k], H["q_en"][:k], H["strand"][:k], np.zeros(k, np.uint32))
    orows, oops = oracle.liftover(ob, *H["windows"])
    g = rows[:k].contiguous().cpu().numpy().view(np.uint8).reshape(-1).view(rustybam_amd.HIT_DT)
    assert len(orows) == k
    for key in ("rec", "win", "status"):
        assert np.array_equal(g[key].astype(np.int64), orows[key].astype(np.int64)), key
    good = orows["status"] == 0
    for key in ("t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len", "out_n"):
        assert np.array_equal(g[key][good].astype(np.uint64), orows[key][good].astype(np.uint64)), key
    for i in np.nonzero(good)[0][::7]:
        a = out[int(g["out_off"][i]): int(g["out_off"][i]) + int(g["out_n"][i])].cpu().numpy().view(np.uint32)
        assert np.array_equal(a, oops[int(orows["out_off"][i]): int(orows["out_off"][i]) + int(orows["out_n"][i])]), i
    # ---- copied clips and descriptors describe the same records (rb_dev_digest_rows expands descriptors through the batch) ----
    dig = torch.zeros(2, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    eng.dev_digest_rows(C["view"], rows.data_ptr(), n, out.data_ptr(), 0, 0, dig[0:1].data_ptr())
    rows_d, desc = run(base | rustybam_amd.LIFT_DESCRIPTORS, n + 64, 4 * (n + 64) + 65536)
    eng.dev_digest_rows(C["view"], rows_d.data_ptr(), n, desc.data_ptr(), 0, 0, dig[1:2].data_ptr())
    torch.cuda.synchronize()
    assert int(dig[0]) == int(dig[1]) and int(dig[0]) != 0
    eng.close()


def test_full_size_nucfreq_checksums():
    """SURVEY 8d config 5 at full size (30x of a 250 Mbp contig, 5e5 reads of 15 kb): every M base lands in exactly one of the four
    counters of a position inside its read's span (sum of all counters = number of M bases), and coverage / maximum depth equal
    what a difference array over the reads' spans gives"""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from bench_nucfreq import make_reads, N_EVENTS
    contig = int(os.environ.get("RB_FULLSIZE_CONTIG", "250000000"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(dev))  # torch's kernels and the engine's on one real stream (a NULL handle would mean "private stream")
    eng = rustybam_amd.Engine(0, torch.cuda.current_stream().cuda_stream)
    pos, ops, op_off, n = make_reads(contig, 30, 15000)
    bpr = 7500
    lut = torch.tensor([(1 << (k >> 2)) << 4 | (1 << (k & 3)) for k in range(16)], dtype=torch.uint8, device=dev)
    d_seq = torch.empty(n * bpr + 64, dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    for o in range(0, d_seq.numel(), 1 << 27):
        m = min(1 << 27, d_seq.numel() - o)
        d_seq[o:o + m] = lut[torch.randint(0, 16, (m,), dtype=torch.uint8, device=dev, generator=g).long()]
    i64 = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to(dev)
    d_pos, d_opoff = i64(pos), i64(op_off)
    d_ops = torch.from_numpy(np.concatenate([ops, np.zeros(8, np.uint32)]).view(np.int32)).to(dev)
    d_seqoff = i64(np.arange(n, dtype=np.uint64) * np.uint64(bpr))
    d_lseq = torch.full((n,), 15000, dtype=torch.int32, device=dev)
    d_tid = torch.zeros(n, dtype=torch.int32, device=dev)
    d_flag = torch.zeros(n, dtype=torch.int32, device=dev)
    d_rgtid = torch.zeros(1, dtype=torch.int32, device=dev)
    d_rgst = torch.zeros(1, dtype=torch.int64, device=dev)
    d_rgen = torch.full((1,), contig, dtype=torch.int64, device=dev)
    d_outoff = torch.tensor([0, contig], dtype=torch.int64, device=dev)
    d_counts = torch.empty(contig * 4 + 16, dtype=torch.int32, device=dev)
    d_status = torch.empty(n + 1, dtype=torch.int32, device=dev)
    d_ctr = torch.zeros(6, dtype=torch.int64, device=dev)
    wsb = eng.nucfreq_workspace_bytes(n, 1, contig)
    d_ws = torch.empty(wsb + 256, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    eng.dev_nucfreq(n, d_ops.data_ptr(), d_opoff.data_ptr(), d_seq.data_ptr(), d_seqoff.data_ptr(), d_lseq.data_ptr(), d_tid.data_ptr(), d_pos.data_ptr(),
                    d_flag.data_ptr(), 1, d_rgtid.data_ptr(), d_rgst.data_ptr(), d_rgen.data_ptr(), d_outoff.data_ptr(), contig, d_counts.data_ptr(),
                    d_status.data_ptr(), d_ctr.data_ptr(), (d_ws.data_ptr() + 255) & ~255, wsb)
    torch.cuda.synchronize()
    ctr = d_ctr.cpu().numpy()
    assert int((d_status[:n] != 0).sum()) == 0 and ctr[3] == 0 and ctr[2] == 0
    c = d_counts[:contig * 4].view(-1, 4)
    covered = (c[:, 0] < 0)                                     # bit 31 of the A word
    total = int((c[:, 0] & 0x7FFFFFFF).sum() + c[:, 1].sum() + c[:, 2].sum() + c[:, 3].sum())
    o = ops.reshape(n, 2 * N_EVENTS + 1)
    assert total == int(((o >> 4) * ((o & 15) == 0)).sum())     # every M base counted once (all bases are A/C/G/T here)
    ref = ((o >> 4) * np.isin(o & 15, [0, 2])).sum(axis=1)
    d = torch.zeros(contig + 1, dtype=torch.int32, device=dev)
    d.index_add_(0, d_pos, torch.ones(n, dtype=torch.int32, device=dev))
    d.index_add_(0, i64(pos + ref.astype(np.int64)), -torch.ones(n, dtype=torch.int32, device=dev))
    depth = torch.cumsum(d[:contig], 0)
    assert int(depth.max()) == int(ctr[0]) and int((depth > 0).sum()) == int(ctr[1])
    assert torch.equal(depth > 0, covered)
    # a position's four counters never exceed its depth
    assert bool(((c[:, 0] & 0x7FFFFFFF) + c[:, 1] + c[:, 2] + c[:, 3] <= depth).all())
    eng.close()


def _check_break_rows(torch, dev, rows, out, n_rec):
    """every piece passes check_integrity, pieces of a record come in target order and do not overlap, and no piece contains an
    insertion or deletion longer than the limit (100)"""
    n = rows.shape[0]
    assert n >= n_rec and int((rows[:, 0] == -286331154).sum()) == 0
    st = rows[:, 2] & 0xFFFF
    ok = st == 0
    assert int((st >= 16).sum()) == 0 and int((~ok).sum()) < max(n // 10000, 4)   # a few pieces are None in the reference too (liftover.rs:52-102); nothing panics
    rows = rows[ok]
    words, off, out_n = _check_integrity(torch, dev, rows, out)
    rec = rows[:, 0].to(torch.int64) & 0xFFFFFFFF
    same = rec[1:] == rec[:-1]
    assert bool((rec[1:] >= rec[:-1]).all())                                        # record order (main.rs:274-280)
    assert bool((_u64(rows, 4)[1:][same] >= _u64(rows, 6)[:-1][same]).all())        # next piece starts at or after the previous one's end
    # long indels: prefix count of (I or D with len > 100) over the output; none may fall inside a piece
    big = (((words & 15) == 1) | ((words & 15) == 2)) & ((words >> 4) > 100)
    P = torch.cumsum(big, 0, dtype=torch.int64)
    assert int(_seg(P, off, off + out_n).sum()) == 0
    return rows


def test_full_size_break_paf_integrity(oracle):
    """break-paf --max-size 100 on the same 1e6 records: every piece passes check_integrity, pieces of a record come in target order
    and do not overlap, and no piece contains an insertion or deletion longer than the limit -- and (round 4) ALL pieces equal the
    op-space CPU port's (oracle/rb_opspace.c rbo_break_opspace_arrays, held to the per-base oracle by tests/test_oracle_opspace.py):
    every row field by field, every clipped CIGAR through the device digest over both outputs."""
    C = _config3(is_break=True)
    torch, dev, eng, run, n_rec, total_ops, d_ops = (C[k] for k in ("torch", "dev", "eng", "run", "n_rec", "total_ops", "d_ops"))
    rows, out = run(rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_FUSED_SCAN | rustybam_amd.BREAK_ONE_WALK, 8 * n_rec, int(1.4 * total_ops))
    assert rows.shape[0] > 2 * n_rec
    _check_break_rows(torch, dev, rows, out, n_rec)
    try:
        avail_gb = int([ln for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")][0].split()[1]) / 1e6
    except Exception:
        avail_gb = 0.0
    if avail_gb < 60 * n_rec / 1e6 + 8:  # (loudly: the property checks above have passed, the comparison of all pieces has NOT run)
        eng.close()
        pytest.skip(f"host has {avail_gb:.0f} GB available: the integrity properties of all {rows.shape[0]} pieces hold, but the comparison "
                    f"with the op-space port on all {n_rec} records needs {60 * n_rec / 1e6 + 8:.0f} GB and did not run")
    if True:
        from rustybam_amd import capi
        H = C["host"]
        n = rows.shape[0]
        ops_host = d_ops[:total_ops].cpu().numpy().view(np.uint32)
        ob = oracle.Batch(ops_host, H["op_off"], H["t_st"], H["t_en"], H["q_st"], H["q_en"], H["strand"], np.zeros(n_rec, np.uint32))
        got = oracle.break_opspace(ob, 100, n_threads=min(64, os.cpu_count() or 1))
        assert got is not None
        orows, oops = got
        assert len(orows) == n, f"{n} GPU pieces, {len(orows)} of the op-space port"
        grows = rows.cpu().numpy().view(np.uint8).reshape(n, 64).view(rustybam_amd.HIT_DT).reshape(n)
        for key in ("rec", "win", "status", "out_n", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len"):
            bad = int((grows[key].astype(np.uint64) != orows[key].astype(np.uint64)).sum())
            assert bad == 0, f"{key}: {bad} of {n} pieces differ from the op-space port"
        d_dig = torch.zeros(2, dtype=torch.int64, device=dev)
        d_rows_g = rows.contiguous()
        torch.cuda.synchronize()
        eng.dev_digest_rows(C["view"], d_rows_g.data_ptr(), n, out.data_ptr(), 0, 0, d_dig[0:].data_ptr())
        torch.cuda.synchronize()
        del out
        torch.cuda.empty_cache()
        d_orows = torch.from_numpy(capi.hit_rows_from(orows).view(np.uint8).reshape(-1)).to(dev)
        d_oops = torch.from_numpy(oops.view(np.int32)).to(dev)
        torch.cuda.synchronize()
        eng.dev_digest_rows(C["view"], d_orows.data_ptr(), n, d_oops.data_ptr(), 0, 0, d_dig[1:].data_ptr())
        torch.cuda.synchronize()
        dg = d_dig.cpu().numpy().view(np.uint64)
        assert dg[0] == dg[1], f"digest of the GPU's pieces {int(dg[0]):#x} != digest of the port's {int(dg[1]):#x}"
    eng.close()


def test_full_size_config4_trim_paf_then_break_paf(oracle, tmp_path):
    """BASELINE.json configs[3]: trim-paf (scores 1,1,1) -> break-paf --max-size 100 on 1e7 synthetic records
    (300-700 ops, 5e9 ops; 4 records per query whose consecutive query spans overlap by U[100, 10000] bases).  The batch stays in HBM
    from the first byte to the last: the passes of Paf::overlapping_paf_recs run over it in place (trim_driver.ResidentTrim:
    rb_dev_overlap_split + rb_dev_apply_pairs), then break-paf twice: straight off the batch as the passes left it (RB_LIFT_OP_STARTS, the
    route tools/bench_config4.py measures) and on the dense copy rb_dev_gather_records makes -- same pieces, same digest.  Checked:
    EVERY pair row of every pass (split, both cuts' coordinates, nmatch, aln_len, the kept range and its rewritten end words) against the
    op-space CPU port of the pair step; properties that need no oracle -- no two records of a query overlap any more, every record
    still passes check_integrity and lies inside the record it came from, every piece of break-paf passes the checks of the config-3
    test; and the per-base oracle CLI on 120 query groups spread over the whole batch, line by line, for both commands."""
    import torch
    from devutil import DevBatch
    from rbtest_util import unpack
    from rustybam_amd import capi, trim_driver
    n = int(os.environ.get("RB_FULLSIZE_C4_RECORDS", "10000000")) // 4 * 4
    seed = 0x5EED0004
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    eng = rustybam_amd.Engine(0, torch.cuda.current_stream().cuda_stream)
    nops = wl.n_ops(seed, 0, n, 300, 700)
    op_off = wl.op_offsets(nops)
    total_ops = int(op_off[-1])
    i64 = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
    d_off = i64(op_off)
    d_ops = torch.empty(total_ops + 64, dtype=torch.int32, device=dev)
    eng.dev_synth_fill_ops(seed, 0, n, d_off.data_ptr(), d_ops.data_ptr())
    zeros = torch.zeros(n, dtype=torch.int64, device=dev)
    d_strand0 = torch.full((n,), ord("+"), dtype=torch.uint8, device=dev)
    d_contig = torch.zeros(n, dtype=torch.int32, device=dev)
    d_red = torch.empty(n * 72, dtype=torch.uint8, device=dev)
    v0 = eng.batch_view(n, total_ops, d_ops.data_ptr(), d_off.data_ptr(), zeros.data_ptr(), zeros.data_ptr(), zeros.data_ptr(), zeros.data_ptr(),
                        d_strand0.data_ptr(), d_contig.data_ptr())
    torch.cuda.synchronize()
    eng.dev_scan_records(v0, d_red.data_ptr(), 0)
    torch.cuda.synchronize()
    red = d_red.cpu().numpy().view(rustybam_amd.REDUCE_DT)
    tb, qb = red["t_bases"].astype(np.uint64), red["q_bases"].astype(np.uint64)
    del d_red, red
    rng = np.random.default_rng(seed)
    q_st = np.zeros(n, np.uint64)
    ov = rng.integers(100, 10001, n).astype(np.uint64)
    for j in range(1, 4):       # each record starts `ov` bases before the previous one of its query ends (nothing contained)
        prev_en = q_st[j - 1::4] + qb[j - 1::4]
        q_st[j::4] = prev_en - np.minimum(ov[j::4], np.minimum(qb[j - 1::4], qb[j::4]) // np.uint64(2))
    q_en = q_st + qb
    t_st = rng.integers(0, 200_000_000, n).astype(np.uint64)
    t_en = t_st + tb
    strand = np.where(rng.integers(0, 2, n) == 0, ord("+"), ord("-")).astype(np.uint8)
    group = np.arange(n) // 4
    T = trim_driver.ResidentTrim(eng, torch, dev, d_ops, op_off, t_st, t_en, q_st, q_en, strand, group, room_factor=1.6)
    del d_ops
    # EVERY pair row of every pass against the op-space CPU port of the pair step (oracle/rb_opspace.c; tests/test_oracle_opspace.py
    # ties it to the per-base oracle): the port carries the passes over a host copy of the original ops as views
    from devutil import PairPortCheck
    ops_host = capi.synth_fill_ops_host(seed, 0, op_off)
    port = PairPortCheck(oracle, torch, T, ops_host, dict(op_off=op_off, t_st=t_st, t_en=t_en, q_st=q_st, q_en=q_en, strand=strand),
                         n_threads=min(64, os.cpu_count() or 1))
    T.run((1, 1, 1), rustybam_amd.BSEARCH_MODERN, on_pass=port)
    assert T.passes == 3 and T.pairs_done == 3 * (n // 4)         # (0,1) (1,2) (2,3) of every query, one per pass
    assert port.pairs == T.pairs_done and T.pairs_by_wave == T.pairs_done
    # ---- break-paf straight off the batch as the passes left it (RB_LIFT_OP_STARTS: no dense copy): its digest, for the route below ----
    S = DevBatch.from_trimmed(torch, eng, dev, T)
    srows, sout, scnt = S.run(None, max_size=100, rows_cap=4 * n, policy=rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_OP_STARTS | rustybam_amd.BREAK_ONE_WALK)
    assert not scnt["redo_two_walk"] and not scnt["overflow"]
    starts_pieces, starts_digest = int(srows.shape[0]), S.digest(srows, sout)
    starts_handed_back = int(scnt["phase"][4])
    del srows, sout, S
    torch.cuda.empty_cache()
    # ---- no overlap left inside a query ----
    qs, qe = T.q_st.reshape(-1, 4).astype(np.int64), T.q_en.reshape(-1, 4).astype(np.int64)
    assert (qe[:, :-1] <= qs[:, 1:]).all() and (qs < qe).all()
    assert (qs >= q_st.reshape(-1, 4).astype(np.int64)).all() and (qe <= q_en.reshape(-1, 4).astype(np.int64)).all()
    d_new, new_off, norm = T.gather()
    assert (norm["status"] == 0).all() and int(new_off[-1]) <= total_ops
    assert (norm["t_st"] >= t_st).all() and (norm["t_en"] <= t_en).all() and (norm["q_st"] == T.q_st).all() and (norm["q_en"] == T.q_en).all()
    # ---- check_integrity of every trimmed record, from segment sums over the dense ops ----
    fake = torch.zeros((n, 16), dtype=torch.int32, device=dev)
    dn = torch.from_numpy(norm.view(np.uint8).reshape(n, 64).view(np.int32).copy()).to(dev)     # norm rows as 16 int32 columns
    fake[:, 3] = dn[:, 9]                                   # out_n = n_ops
    fake[:, 4:12] = dn[:, 0:8]                              # t_st, t_en, q_st, q_en
    fake[:, 12], fake[:, 13] = dn[:, 12], dn[:, 13]         # nmatch, aln_len
    offs = torch.from_numpy(new_off[:-1].view(np.int64).copy()).to(dev)
    fake[:, 14], fake[:, 15] = (offs & 0xFFFFFFFF).to(torch.int32), (offs >> 32).to(torch.int32)
    _check_integrity(torch, dev, fake, d_new)
    del fake, dn
    # ---- break-paf --max-size 100 on the trimmed batch ----
    d_c = [torch.from_numpy(np.ascontiguousarray(norm[k]).view(np.int64)).to(dev) for k in ("t_st", "t_en", "q_st", "q_en")]
    B = DevBatch.from_device(torch, eng, dev, d_new, int(new_off[-1]), new_off, d_c, torch.from_numpy(strand).to(dev))
    T.release()
    torch.cuda.empty_cache()
    rows, out, cnt = B.run(None, max_size=100, rows_cap=4 * n)
    rows_ok = _check_break_rows(torch, dev, rows, out, n)
    # the route without the dense copy cut the same pieces: same count, same digest over every row and every clipped op
    assert starts_pieces == int(rows.shape[0]) and starts_digest == B.digest(rows, out) and starts_handed_back <= n // 1000
    # ---- the oracle CLI on 120 query groups spread over the whole batch (every (n / 480)-th group): trim-paf, then break-paf of its output ----
    step = max(1, (n // 4) // 120)
    recs = np.concatenate([np.arange(4 * g, 4 * g + 4) for g in range(0, n // 4, step)][:120])
    k = len(recs)
    paf = tmp_path / "c4.paf"
    with open(paf, "w") as f:
        for r in recs:
            cg = unpack(ops_host[int(op_off[r]):int(op_off[r + 1])])
            f.write(f"q{r // 4:07d}\t{int(q_en[r // 4 * 4 + 3]) + 1000}\t{int(q_st[r])}\t{int(q_en[r])}\t{chr(strand[r])}\tchr1\t250000000\t"
                    f"{int(t_st[r])}\t{int(t_en[r])}\t0\t0\t60\tcg:Z:{cg}\n")
    del ops_host
    rc, otrim = oracle.cli("trim-paf", paf)
    assert rc == 0
    mine = []
    for r in recs:                                                # (names sort like the record numbers: q0000000, q0000001, ...)
        cg = unpack(d_new[int(new_off[r]):int(new_off[r + 1])].cpu().numpy().view(np.uint32))
        mine.append(f"q{r // 4:07d}\t{int(q_en[r // 4 * 4 + 3]) + 1000}\t{int(norm['q_st'][r])}\t{int(norm['q_en'][r])}\t{chr(strand[r])}\tchr1\t250000000\t"
                    f"{int(norm['t_st'][r])}\t{int(norm['t_en'][r])}\t{int(norm['nmatch'][r])}\t{int(norm['aln_len'][r])}\t60\tid:Z:\tcg:Z:{cg}\n")
    assert "".join(mine).encode() == otrim
    trimmed = tmp_path / "c4_trim.paf"
    trimmed.write_bytes(otrim)
    rc, obreak = oracle.cli("break-paf", "--max-size", "100", trimmed)
    assert rc == 0
    in_sample = torch.zeros(n, dtype=torch.bool, device=dev)
    in_sample[torch.from_numpy(recs).to(dev)] = True
    hr = rows_ok[in_sample[rows_ok[:, 0].to(torch.int64) & 0xFFFFFFFF]].contiguous().cpu().numpy().view(np.uint8).reshape(-1).view(rustybam_amd.HIT_DT)
    allr = rows[in_sample[rows[:, 0].to(torch.int64) & 0xFFFFFFFF]].contiguous().cpu().numpy().view(np.uint8).reshape(-1).view(rustybam_amd.HIT_DT)
    assert len(allr) >= len(hr)
    mine = []
    for h in hr:
        r = int(h["rec"])
        cg = unpack(out[int(h["out_off"]):int(h["out_off"]) + int(h["out_n"])].cpu().numpy().view(np.uint32))
        mine.append(f"q{r // 4:07d}\t{int(q_en[r // 4 * 4 + 3]) + 1000}\t{int(h['q_st'])}\t{int(h['q_en'])}\t{chr(strand[r])}\tchr1\t250000000\t"
                    f"{int(h['t_st'])}\t{int(h['t_en'])}\t{int(h['nmatch'])}\t{int(h['aln_len'])}\t60\tid:Z:\tcg:Z:{cg}\n")
    assert "".join(mine).encode() == obreak
    eng.close()


def test_config4_25_contigs_deep_recursion(oracle, tmp_path):
    """SURVEY 8d config 4 as written: records over the 25 contigs of asm_small.bam's header (records per contig ~ length), and -- what
    the 1e7-record test above does not stress -- query groups of 2..9 records whose spans overlap their next TWO neighbours, so
    that a group needs many passes of Paf::overlapping_paf_recs (deep recursion, deferred pairs re-scanned after every cut).
    2e6 records through the device pass driver (rb_dev_trim_select + pair kernel, the numpy restatement of the selection checked
    on every pass), then break-paf; properties that need no oracle at full size, and the oracle CLI line by line on the first
    groups (trim-paf, then break-paf of its output, contigs included)."""
    import torch
    from devutil import DevBatch
    from rbtest_util import unpack
    from rustybam_amd import capi, trim_driver
    n_target = int(os.environ.get("RB_FULLSIZE_C4B_RECORDS", "2000000"))
    seed = 0x5EED0004
    contigs = [("chr1", 248387497), ("chr2", 242696747), ("chr3", 201106605), ("chr4", 193575430), ("chr5", 182045437), ("chr6", 172126870),
               ("chr7", 160567423), ("chr8", 146259322), ("chr9", 150617274), ("chr10", 134758122), ("chr11", 135127772), ("chr12", 133324781),
               ("chr13", 113566686), ("chr14", 101161492), ("chr15", 99753195), ("chr16", 96330493), ("chr17", 84276897), ("chr18", 80542536),
               ("chr19", 61707359), ("chr20", 66210247), ("chr21", 45090682), ("chr22", 51324926), ("chrX", 154259566), ("chrY", 62460029), ("chrM", 16569)]
    rng = np.random.default_rng(seed + 1)
    sizes = rng.integers(2, 10, n_target // 5)
    sizes = sizes[: np.searchsorted(np.cumsum(sizes), n_target)]
    n = int(sizes.sum())
    group = np.repeat(np.arange(len(sizes)), sizes)
    pos_in = np.arange(n) - np.repeat(np.cumsum(sizes) - sizes, sizes)
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    eng = rustybam_amd.Engine(0, torch.cuda.current_stream().cuda_stream)
    nops = wl.n_ops(seed, 0, n, 300, 700)
    op_off = wl.op_offsets(nops)
    total_ops = int(op_off[-1])
    i64 = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
    d_off = i64(op_off)
    d_ops = torch.empty(total_ops + 64, dtype=torch.int32, device=dev)
    eng.dev_synth_fill_ops(seed, 0, n, d_off.data_ptr(), d_ops.data_ptr())
    zeros = torch.zeros(n, dtype=torch.int64, device=dev)
    d_red = torch.empty(n * 72, dtype=torch.uint8, device=dev)
    v0 = eng.batch_view(n, total_ops, d_ops.data_ptr(), d_off.data_ptr(), zeros.data_ptr(), zeros.data_ptr(), zeros.data_ptr(), zeros.data_ptr(),
                        torch.full((n,), ord("+"), dtype=torch.uint8, device=dev).data_ptr(), torch.zeros(n, dtype=torch.int32, device=dev).data_ptr())
    torch.cuda.synchronize()
    eng.dev_scan_records(v0, d_red.data_ptr(), 0)
    torch.cuda.synchronize()
    red = d_red.cpu().numpy().view(rustybam_amd.REDUCE_DT)
    tb, qb = red["t_bases"].astype(np.uint64), red["q_bases"].astype(np.uint64)
    del d_red, red
    # record j of a group starts 40 % of the SHORTEST record length behind record j - 1: it overlaps j + 1 by ~60 % and j + 2 by ~20 %
    step = np.repeat(np.minimum.reduceat(qb, np.cumsum(sizes) - sizes) * 2 // 5, sizes)
    q_st = (pos_in.astype(np.uint64) * step + rng.integers(0, 50, n).astype(np.uint64))
    q_en = q_st + qb
    clen = np.array([c[1] for c in contigs], np.float64)
    contig = rng.choice(len(contigs), n, p=clen / clen.sum()).astype(np.uint32)
    t_st = (rng.random(n) * np.maximum(clen[contig] - tb.astype(np.float64) - 1, 1)).astype(np.uint64)
    t_en = t_st + tb
    strand = np.where(rng.integers(0, 2, n) == 0, ord("+"), ord("-")).astype(np.uint8)
    T = trim_driver.ResidentTrim(eng, torch, dev, d_ops, op_off, t_st, t_en, q_st, q_en, strand, group, room_factor=3.0)
    del d_ops
    from devutil import PairPortCheck                               # (every pair row of every pass against the op-space CPU port)
    ops_host = capi.synth_fill_ops_host(seed, 0, op_off)
    port = PairPortCheck(oracle, torch, T, ops_host, dict(op_off=op_off, t_st=t_st, t_en=t_en, q_st=q_st, q_en=q_en, strand=strand),
                         n_threads=min(64, os.cpu_count() or 1))
    T.run((1, 1, 1), rustybam_amd.BSEARCH_MODERN, check_host=True, on_pass=port)
    assert T.passes > 6, T.passes                                   # (a group of 9 with two overlaps per record: many rounds)
    assert port.pairs == T.pairs_done
    # ---- no overlap left inside a query (any two records of a group) ----
    order = np.lexsort((T.q_st.astype(np.int64), group))
    g_s, qs_s, qe_s = group[order], T.q_st[order].astype(np.int64), T.q_en[order].astype(np.int64)
    same = g_s[1:] == g_s[:-1]
    keep = ~T.contained[order]
    assert ((qe_s[:-1] <= qs_s[1:]) | ~same | ~keep[1:] | ~keep[:-1]).all()
    d_new, new_off, norm = T.gather()
    assert (norm["status"] == 0).all() and (norm["t_st"] >= t_st).all() and (norm["t_en"] <= t_en).all()
    fake = torch.zeros((n, 16), dtype=torch.int32, device=dev)
    dn = torch.from_numpy(norm.view(np.uint8).reshape(n, 64).view(np.int32).copy()).to(dev)
    fake[:, 3] = dn[:, 9]
    fake[:, 4:12] = dn[:, 0:8]
    fake[:, 12], fake[:, 13] = dn[:, 12], dn[:, 13]
    offs = torch.from_numpy(new_off[:-1].view(np.int64).copy()).to(dev)
    fake[:, 14], fake[:, 15] = (offs & 0xFFFFFFFF).to(torch.int32), (offs >> 32).to(torch.int32)
    _check_integrity(torch, dev, fake, d_new)
    del fake, dn
    d_c = [torch.from_numpy(np.ascontiguousarray(norm[k]).view(np.int64)).to(dev) for k in ("t_st", "t_en", "q_st", "q_en")]
    B = DevBatch.from_device(torch, eng, dev, d_new, int(new_off[-1]), new_off, d_c, torch.from_numpy(strand).to(dev))
    T.release()
    torch.cuda.empty_cache()
    rows, out, cnt = B.run(None, max_size=100, rows_cap=4 * n)
    rows_ok = _check_break_rows(torch, dev, rows, out, n)
    # ---- the oracle CLI on 151 query groups spread over the whole batch ----
    ends = np.cumsum(sizes)
    pick = np.arange(0, len(sizes), max(1, len(sizes) // 151))[:151]
    recs = np.concatenate([np.arange(ends[g] - sizes[g], ends[g]) for g in pick])
    q_len = {int(g): int(q_en[ends[g] - sizes[g]:ends[g]].max()) + 1000 for g in pick}
    line = lambda r, qs_, qe_, ts_, te_, nm, al, cg, with_id: (
        f"q{group[r]:07d}\t{q_len[int(group[r])]}\t{qs_}\t{qe_}\t{chr(strand[r])}\t{contigs[contig[r]][0]}\t{contigs[contig[r]][1]}\t{ts_}\t{te_}\t{nm}\t{al}\t60\t"
        + ("id:Z:\t" if with_id else "") + f"cg:Z:{cg}\n")
    paf = tmp_path / "c4b.paf"
    with open(paf, "w") as f:
        for r in recs:
            f.write(line(r, int(q_st[r]), int(q_en[r]), int(t_st[r]), int(t_en[r]), 0, 0, unpack(ops_host[int(op_off[r]):int(op_off[r + 1])]), False))
    del ops_host
    rc, otrim = oracle.cli("trim-paf", paf)
    assert rc == 0
    mine = [line(r, int(norm["q_st"][r]), int(norm["q_en"][r]), int(norm["t_st"][r]), int(norm["t_en"][r]), int(norm["nmatch"][r]), int(norm["aln_len"][r]),
                 unpack(d_new[int(new_off[r]):int(new_off[r + 1])].cpu().numpy().view(np.uint32)), True) for r in recs]
    assert "".join(mine).encode() == otrim                       # (names sort like the group numbers; file order inside a group)
    trimmed = tmp_path / "c4b_trim.paf"
    trimmed.write_bytes(otrim)
    rc, obreak = oracle.cli("break-paf", "--max-size", "100", trimmed)
    assert rc == 0
    in_sample = torch.zeros(n, dtype=torch.bool, device=dev)
    in_sample[torch.from_numpy(recs).to(dev)] = True
    hr = rows_ok[in_sample[rows_ok[:, 0].to(torch.int64) & 0xFFFFFFFF]].contiguous().cpu().numpy().view(np.uint8).reshape(-1).view(rustybam_amd.HIT_DT)
    mine = [line(int(h["rec"]), int(h["q_st"]), int(h["q_en"]), int(h["t_st"]), int(h["t_en"]), int(h["nmatch"]), int(h["aln_len"]),
                 unpack(out[int(h["out_off"]):int(h["out_off"]) + int(h["out_n"])].cpu().numpy().view(np.uint32)), True) for h in hr]
    assert "".join(mine).encode() == obreak
    eng.close()
