"""SURVEY 8d config 4 as ONE device pipeline: the CIGAR-walk stages of the README pipeline (`trim-paf | break-paf --max-size 100`) on
synthetic records of 300-700 ops, 4 records per query whose consecutive query spans overlap by U[100, 10000] bases (1e7 records = the
full size), the batch resident in HBM from the first stage to the last:

  scan     rb_dev_scan_records        remove_trailing_indels + check_integrity of every record (the row form: four records per wavefront)
  select   rb_dev_trim_select         per pass: the pair scan per query name and the choice of the pairs to cut
  pair     rb_dev_overlap_split       per pass: split + clip of the chosen pairs, IN PLACE (four pairs per wavefront, then the retries)
  apply    rb_dev_apply_pairs + rb_dev_trim_check
  break    rb_dev_break               on the batch as the passes left it (RB_LIFT_OP_STARTS: no rb_dev_gather_records in between)

Every stage is bracketed by HIP events on the engine's stream, with its outputs allocated and its plan built before the bracket (a
resident host has them with the batch).  One JSON line: per stage kernel_ms, algorithmic bytes, fraction of 8 TB/s; the pipeline's
records/s on the sum of the stages.  `--gather`: also the round-5 route (rb_dev_gather_records, then rb_dev_break with the fused scan on
the dense copy) for comparison.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split (tools/prof_c4.sh).  No oracle
here: parity of every stage is the business of tests/ (test_gpu_fullsize.py runs this workload; test_gpu_trim.py both break routes).

  python tools/bench_config4.py [--records 10000000] [--gather] [--host-buffers]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
SEED = 0x5EED0004
PEAK = 8e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=10_000_000)
    ap.add_argument("--gather", action="store_true", help="also time the route through rb_dev_gather_records (round 5)")
    ap.add_argument("--host-buffers", action="store_true", help="also time rb_host_overlap_split on host arrays (PCIe-inclusive, 2e6 records at most)")
    a = ap.parse_args()
    import torch
    import rustybam_amd
    from rustybam_amd import capi, trim_driver
    from devutil import DevBatch, config4_resident
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    eng = rustybam_amd.Engine(0, torch.cuda.current_stream().cuda_stream)
    t0 = time.time()
    T, h = config4_resident(torch, eng, dev, a.records, seed=SEED)
    n, total_ops, nops, off = h["n"], h["total_ops"], h["nops"], h["op_off"]
    t_st, t_en, q_st, q_en, strand = h["t_st"], h["t_en"], h["q_st"], h["q_en"], h["strand"]
    gen = time.time() - t0
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    MOD = rustybam_amd.BSEARCH_MODERN

    # ---- scan: the rows ResidentTrim made when it took the batch, timed here on the same view into a second array ----
    d_norm2 = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
    eng.dev_scan_records(T.view, 0, d_norm2.data_ptr())
    torch.cuda.synchronize()
    scan_ms = []
    for _ in range(3):
        e0, e1 = ev(), ev()
        e0.record()
        eng.dev_scan_records(T.view, 0, d_norm2.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        scan_ms.append(e0.elapsed_time(e1))
    assert torch.equal(d_norm2, T.d_norm[: n * 64])
    del d_norm2
    scan_bytes = 4 * total_ops + 48 * n + 64 * n

    # ---- trim-paf: the passes ----
    # (warm-up, as bench.py has one: the first trim call of a context allocates its scratch -- 40 MB for the pairs whose region does not fit
    #  LDS -- and the first launch of every kernel is not a launch like the others; eight records of the same batch, thrown away)
    group = np.arange(n) // 4
    W_ = trim_driver.ResidentTrim(eng, torch, dev, T.d_ops[: int(off[8])].clone(), off[:9], t_st[:8], t_en[:8], q_st[:8], q_en[:8], strand[:8], group[:8], room_factor=1.6)
    try:
        W_.run((1, 1, 1), MOD)
    except RuntimeError:
        if not os.environ.get("RB_C4_ONE_PASS"):  # (a diagnostic variant that ends its pairs early leaves rows nobody can use)
            raise
    W_.release()
    del W_
    T._pass_buffers()  # (the query groups and the passes' device buffers: part of having the batch resident, not of the passes)
    # (the list of pairs a first attempt declines grows with the largest pass seen: sized here, not inside the first pass's bracket)
    eng.trim_reserve(n // 4 + 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if os.environ.get("RB_C4_ONE_PASS"):  # (diagnostics, tools/prof_c4_decomp.sh: one pass of a library variant whose rows are wrong, then out)
        try:
            T.run((1, 1, 1), MOD, max_passes=1)
        except Exception as e:  # (more passes wanted, or rows a stopped variant left unfinished)
            print(f"one pass: {e}", file=sys.stderr)
        torch.cuda.synchronize()
        print(json.dumps({"one_pass": True}))
        return
    evs = []
    T.run((1, 1, 1), MOD, fetch=False, events=evs)  # (the batch goes on on the device: nothing of it is fetched inside the stage)
    torch.cuda.synchronize()
    t_trim = time.perf_counter() - t0
    k_sel, k_pair, k_apply = sum(e[1] for e in evs), sum(e[2] for e in evs), sum(e[3] for e in evs)
    pairs = T.pairs_done
    nops64 = nops.astype(np.int64)
    left = np.arange(n)[np.arange(n) % 4 != 3]
    pair_in = int((nops64[left] + nops64[left + 1]).sum())        # (upper bound: the middle records have been cut once already in later passes)
    # what a pair pass MOVES on a resident batch (RB_TRIM_IN_PLACE): both records' ops read (counted whole: the kernels read the ends that
    # overlap and nothing else, so this prices bytes they no longer touch -- kept as the figure the earlier rounds quote), a 128-byte row and
    # the two end words of each clip written
    pair_bytes = 4 * pair_in + 128 * pairs + 2 * 2 * 4 * pairs
    # ... and what the kernels touch: two regions of 64 ops (the first attempt's), two norm rows, the indices, the row, the end words
    pair_bytes_touched = pairs * (2 * 64 * 4 + 2 * 64 + 2 * 4 + 2 * 8 + 128 + 16)
    sel_bytes = len(evs) * (64 * n + 4 * n + 8 * (n // 4)) + 16 * pairs   # norm rows + order + group offsets read per pass, the chosen pairs written
    apply_bytes = pairs * (128 + 2 * (64 + 64 + 8))

    # ---- break-paf --max-size 100 on the batch as the passes left it ----
    in_place = T.pairs_by_wave == T.pairs_done  # (no pass moved a record: RB_LIFT_OP_STARTS applies)
    res_break = {}
    B = DevBatch.from_trimmed(torch, eng, dev, T)
    pol = MOD | rustybam_amd.LIFT_OP_STARTS | rustybam_amd.BREAK_ONE_WALK
    rows, out, cnt = B.run(None, max_size=100, rows_cap=4 * n, policy=pol)  # sizing (and the first launch of its kernels)
    n_pieces, one_walk = int(rows.shape[0]), not bool(cnt["redo_two_walk"])
    out_ops_emitted = int(rows[:, 3].to(torch.int64).sum().item())
    kept_ops = int(T.d_norm[: n * 64].view(torch.int32).view(n, 16)[:, 9].to(torch.int64).sum().item())
    del rows, out
    B.last = None
    plan = eng.plan_create(B.op_off_host, B.contig_host, None, None, None)
    rows_cap, out_cap = 4 * n, max(4096, eng.plan_out_capacity(plan, True))
    ws = torch.empty(eng.plan_workspace_bytes(plan, rows_cap), dtype=torch.uint8, device=dev)
    d_rows = torch.empty((rows_cap + 1) * 64, dtype=torch.uint8, device=dev)
    d_out = torch.empty(out_cap + 64, dtype=torch.int32, device=dev)
    brk_ms, brk_wall = [], []
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        tw = time.perf_counter()
        e0.record()
        eng.dev_break(plan, B.view, B.d_norm.data_ptr(), 100, pol, ws.data_ptr(), d_rows.data_ptr(), rows_cap, d_out.data_ptr(), out_cap, B.d_cnt.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        brk_wall.append(time.perf_counter() - tw)
        brk_ms.append(e0.elapsed_time(e1))
    c2 = B.d_cnt.cpu().numpy().view(rustybam_amd.COUNTERS_DT)[0]
    assert not c2["overflow"] and int(c2["n_hits"]) == n_pieces
    tiles, handed_back = int(c2["phase"][3]), int(c2["phase"][4])
    eng.plan_destroy(plan)
    del ws, d_rows, d_out
    break_bytes = 4 * kept_ops + 48 * n + 88 * n_pieces + 4 * out_ops_emitted

    def stage(ms, b, note=None):
        d = {"kernel_ms": round(ms, 3), "algorithmic_bytes": int(b), "achieved_GBps": round(b / (ms * 1e-3) / 1e9, 1), "frac": round(b / (ms * 1e-3) / PEAK, 4)}
        if note:
            d["note"] = note
        return d
    stages = {"scan": stage(min(scan_ms), scan_bytes, "rb_dev_scan_records, norm rows only: 4 B per op + 48 B header + 64 B row per record; best of 3"),
              "select": stage(k_sel, sel_bytes, "rb_dev_trim_select, all passes: norm rows + order + group offsets read per pass"),
              "pair": stage(k_pair, pair_bytes, "rb_dev_overlap_split, all passes, on the bytes the earlier rounds price (both records' ops whole + row + end words); "
                            f"on what the kernels touch ({pair_bytes_touched} B): {round(pair_bytes_touched / (k_pair * 1e-3) / PEAK, 4)}"),
              "apply": stage(k_apply, apply_bytes, "rb_dev_apply_pairs + rb_dev_trim_check, all passes"),
              "break": stage(min(brk_ms), break_bytes, "rb_dev_break with RB_LIFT_OP_STARTS | RB_BREAK_ONE_WALK on the batch as the passes left it: 4 B per kept op + 48 B per "
                             "record + 88 B per piece + 4 B per emitted op; best of 3")}
    total_ms = sum(s["kernel_ms"] for s in stages.values())
    res = {"workload": f"config 4: {n} records, {total_ops} ops, {n // 4} query groups of 4, seed 0x5eed0004; batch resident in HBM",
           "stages": stages, "pipeline_device_ms": round(total_ms, 3), "pipeline_records_per_s": n / (total_ms * 1e-3),
           "pipeline_cigar_ops_per_s": total_ops / (total_ms * 1e-3),
           "trim_passes": T.passes, "trim_pairs": pairs, "trim_wall_s": round(t_trim, 4), "trim_wall_over_kernels": round(t_trim * 1e3 / max(1e-9, k_sel + k_pair + k_apply), 2),
           "trim_per_pass_ms": [[int(e[0]), round(e[1], 3), round(e[2], 3), round(e[3], 3)] for e in evs],
           "pairs_by_wave_kernel": T.pairs_by_wave, "trim_in_place": bool(in_place),
           "break_pieces": n_pieces, "break_one_walk": one_walk, "break_wall_s": round(min(brk_wall), 4),
           "break_wall_over_kernels": round(min(brk_wall) * 1e3 / min(brk_ms), 3), "break_tiles": tiles, "break_records_handed_back": handed_back,
           "scan_ms_runs": [round(x, 3) for x in scan_ms], "break_ms_runs": [round(x, 3) for x in brk_ms], "setup_s": round(gen, 2)}
    if a.gather:  # the round-5 route: a dense copy, then break-paf with the fused scan on it
        T.fetch()
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record()
        d_new, new_off, norm = T.gather()
        e1.record()
        torch.cuda.synchronize()
        d_c = [torch.from_numpy(np.ascontiguousarray(norm[k]).view(np.int64)).to(dev) for k in ("t_st", "t_en", "q_st", "q_en")]
        G = DevBatch.from_device(torch, eng, dev, d_new, int(new_off[-1]), new_off, d_c, torch.from_numpy(strand).to(dev))
        polg = MOD | rustybam_amd.LIFT_FUSED_SCAN | rustybam_amd.BREAK_ONE_WALK
        G.run(None, max_size=100, rows_cap=4 * n, policy=polg)
        G.last = None
        plan = eng.plan_create(G.op_off_host, G.contig_host, None, None, None)
        out_cap = max(4096, eng.plan_out_capacity(plan, True))
        ws = torch.empty(eng.plan_workspace_bytes(plan, rows_cap), dtype=torch.uint8, device=dev)
        d_rows = torch.empty((rows_cap + 1) * 64, dtype=torch.uint8, device=dev)
        d_out = torch.empty(out_cap + 64, dtype=torch.int32, device=dev)
        gm = []
        for _ in range(3):
            torch.cuda.synchronize()
            g0, g1 = ev(), ev()
            g0.record()
            eng.dev_break(plan, G.view, G.d_norm.data_ptr(), 100, polg, ws.data_ptr(), d_rows.data_ptr(), rows_cap, d_out.data_ptr(), out_cap, G.d_cnt.data_ptr())
            g1.record()
            torch.cuda.synchronize()
            gm.append(g0.elapsed_time(g1))
        eng.plan_destroy(plan)
        res["gather_route"] = {"gather_ms_with_its_host_round_trips": round(e0.elapsed_time(e1), 3), "break_on_the_dense_copy_ms": round(min(gm), 3)}
    if a.host_buffers and n <= 2_000_000:
        ops_h = capi.synth_fill_ops_host(SEED, 0, off)
        t0 = time.time()
        eng.overlap_split(ops_h, off, t_st, t_en, q_st, q_en, strand, left.astype(np.uint32), (left + 1).astype(np.uint32))
        res["host_buffer_overlap_split_s"] = round(time.time() - t0, 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
