"""trim-paf on the device (pair kernel through the C ABI) vs the CPU oracle; bit-exact."""
import hashlib
import json
import os
import zlib

import numpy as np
import pytest

import rustybam_amd
from rustybam_amd import trim_driver
from rbtest_util import random_cigar, read_paf, recs_from_lines, sums, unpack

pytestmark = pytest.mark.gpu


def _pairs_batch(rng, n_pairs, mode, zero_bias=False, ops_range=(3, 60), max_overlap=None):
    cig, t_st, t_en, q_st, q_en, strand = [], [], [], [], [], []
    left, right = [], []
    for _ in range(n_pairs):
        ca = random_cigar(rng, int(rng.integers(*ops_range)), mode)
        cb = random_cigar(rng, int(rng.integers(*ops_range)), mode)
        (ra, qa), (rb, qb) = sums(ca), sums(cb)
        if min(qa, qb) < 2:
            continue
        a0 = 0 if (zero_bias and rng.random() < 0.5) else int(rng.integers(0, 1000))
        o = int(rng.integers(1, min(qa, qb) if max_overlap is None else min(qa, qb, max_overlap)))  # 1 <= overlap < both lengths: neither is contained
        b0 = a0 + qa - o
        for c, r, q, s0 in ((ca, ra, qa, a0), (cb, rb, qb, b0)):
            ts = int(rng.integers(0, 5000))
            cig.append(c); t_st.append(ts); t_en.append(ts + r); q_st.append(s0); q_en.append(s0 + q)
            strand.append(ord("+") if rng.random() < .5 else ord("-"))
        left.append(len(cig) - 2); right.append(len(cig) - 1)
    off = np.zeros(len(cig) + 1, np.uint64)
    off[1:] = np.cumsum([len(c) for c in cig])
    return dict(ops=np.concatenate(cig), op_off=off, t_st=np.array(t_st, np.uint64), t_en=np.array(t_en, np.uint64),
                q_st=np.array(q_st, np.uint64), q_en=np.array(q_en, np.uint64), strand=np.array(strand, np.uint8)), \
        np.array(left, np.uint32), np.array(right, np.uint32)


def _compare(rows, out, orows, oout, what):
    assert len(rows) == len(orows)
    bad = np.nonzero(rows["status"] != orows["status"])[0]
    assert len(bad) == 0, f"{what}: status differs at {bad[:5]}: gpu {rows['status'][bad[:5]]} oracle {orows['status'][bad[:5]]}"
    ok = orows["status"] == 0
    for k in ("split_idx", "split_score"):
        bad = np.nonzero(ok & (rows[k] != orows[k]))[0]
        assert len(bad) == 0, f"{what}: {k} differs at {bad[:5]}: gpu {rows[k][bad[:5]]} oracle {orows[k][bad[:5]]}"
    for k in ("t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len", "out_n"):
        bad = np.nonzero(ok[:, None] & (rows[k] != orows[k]))[0]
        assert len(bad) == 0, f"{what}: {k} differs at pairs {bad[:5]}: gpu {rows[k][bad[:3]]} oracle {orows[k][bad[:3]]}"
    for i in np.nonzero(ok)[0]:
        for s in (0, 1):
            a = out[int(rows["out_off"][i][s]):int(rows["out_off"][i][s]) + int(rows["out_n"][i][s])]
            b = oout[int(orows["out_off"][i][s]):int(orows["out_off"][i][s]) + int(orows["out_n"][i][s])]
            assert np.array_equal(a, b), f"{what}: cigar of pair {i} side {s}: gpu {unpack(a[:10])} oracle {unpack(b[:10])}"


def test_known_answer_ka3(engine, oracle, golden):
    k = json.load(open(os.path.join(golden, "known_answers.json")))["KA3_trim_pair"]
    r = recs_from_lines([k["left"], k["right"]])
    rows, out = engine.overlap_split(*r.arrays(), [0], [1], tuple(k["scores"]))
    assert rows["status"][0] == 0 and int(rows["split_idx"][0]) == 2 and int(rows["split_score"][0]) == 5
    got = [unpack(out[int(rows["out_off"][0][s]):int(rows["out_off"][0][s]) + int(rows["out_n"][0][s])]) for s in (0, 1)]
    assert got == [k["left_cigar"], k["right_cigar"]]


@pytest.mark.parametrize("mode", ["regular", "indel_ends", "wild"])
@pytest.mark.parametrize("policy", [rustybam_amd.BSEARCH_MODERN, rustybam_amd.BSEARCH_LEGACY])
@pytest.mark.parametrize("scores", [(1, 1, 1), (2, 3, 5)])
def test_pairs_random(engine, oracle, mode, policy, scores):
    rng = np.random.default_rng(zlib.crc32(f"{mode}{policy}{scores}".encode()))
    b, left, right = _pairs_batch(rng, 300, mode)
    rows, out = engine.overlap_split(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"],
                                     left, right, scores, policy)
    ob = oracle.Batch(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"])
    orows, oout = oracle.overlap_split(ob, left, right, scores, policy)
    _compare(rows, out, orows, oout, f"{mode} policy={policy}")


@pytest.mark.parametrize("ops_range", [(60, 200), (600, 900), (760, 776)])
def test_pairs_long_regular_cigars(engine, oracle, ops_range):
    """the wave-per-pair kernel stages a record 64 ops at a time and takes records of at most 768 ops; longer ones (and their
    partners) go to the serial kernel: both sides of that limit, several staging steps, both strands"""
    rng = np.random.default_rng(ops_range[0])
    b, left, right = _pairs_batch(rng, 40, "regular", ops_range=ops_range)
    for scores in ((1, 1, 1), (3, 1, 7)):
        rows, out = engine.overlap_split(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"], left, right, scores)
        ob = oracle.Batch(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"])
        orows, oout = oracle.overlap_split(ob, left, right, scores)
        _compare(rows, out, orows, oout, f"long regular {ops_range} {scores}")


@pytest.mark.parametrize("ops_range,max_overlap", [((2000, 9000), 4000), ((700, 1200), 150000), ((9000, 30000), 60)])
def test_pairs_records_longer_than_the_staged_region(engine, oracle, ops_range, max_overlap):
    """records of thousands of ops (whole-chromosome alignments) whose overlap is short: the wave kernel stages only the ops around
    the overlap, streams what lies in front of them for the totals and copies what the clip keeps; overlaps wider than the region
    still go to the serial kernel"""
    rng = np.random.default_rng(ops_range[0] + max_overlap)
    b, left, right = _pairs_batch(rng, 24, "regular", ops_range=ops_range, max_overlap=max_overlap)
    for scores in ((1, 1, 1), (3, 1, 7)):
        rows, out = engine.overlap_split(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"], left, right, scores)
        ob = oracle.Batch(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"])
        orows, oout = oracle.overlap_split(ob, left, right, scores)
        _compare(rows, out, orows, oout, f"long records {ops_range} overlap<{max_overlap} {scores}")
    by_wave = int((rows["_pad"] == 1).sum())
    if max_overlap <= 4000:
        assert by_wave >= len(rows) * 3 // 4, f"only {by_wave} of {len(rows)} pairs were done by the wave kernel"


def test_pairs_odd_geometries(engine, oracle):
    """regular records (the wave-per-pair kernel) whose query spans do not overlap, touch, coincide, or contain one another: the
    split degenerates and truncate_record_by_query runs into its assertions / not-found panics exactly as in the oracle"""
    rng = np.random.default_rng(99)
    cig, t_st, t_en, q_st, q_en, strand, left, right = [], [], [], [], [], [], [], []
    for rel in ("apart", "touch", "same", "contained", "contains", "one_base", "apart", "same", "contained"):
        for sa in "+-":
            for sb in "+-":
                ca = random_cigar(rng, int(rng.integers(5, 90)), "regular")
                cb = random_cigar(rng, int(rng.integers(5, 90)), "regular")
                (ra, qa), (rb, qb) = sums(ca), sums(cb)
                a0 = int(rng.integers(0, 3)) * 500
                if rel == "apart":
                    b0 = a0 + qa + 17
                elif rel == "touch":
                    b0 = a0 + qa
                elif rel == "one_base":
                    b0 = a0 + qa - 1
                elif rel == "same":
                    cb, rb, qb, b0 = ca, ra, qa, a0
                elif rel == "contained":       # right inside left
                    if qb >= qa:
                        ca, cb, ra, rb, qa, qb = cb, ca, rb, ra, qb, qa
                    b0 = a0 + (qa - qb) // 2
                else:                          # left starts first but right ends later and left ends inside right: plain overlap from the left end
                    b0 = a0 + max(qa // 3, 1)
                for c, r, q, s0, sd in ((ca, ra, qa, a0, sa), (cb, rb, qb, b0, sb)):
                    ts = int(rng.integers(0, 5000))
                    cig.append(c); t_st.append(ts); t_en.append(ts + r); q_st.append(s0); q_en.append(s0 + q); strand.append(ord(sd))
                left.append(len(cig) - 2); right.append(len(cig) - 1)
    off = np.zeros(len(cig) + 1, np.uint64)
    off[1:] = np.cumsum([len(c) for c in cig])
    b = dict(ops=np.concatenate(cig), op_off=off, t_st=np.array(t_st, np.uint64), t_en=np.array(t_en, np.uint64),
             q_st=np.array(q_st, np.uint64), q_en=np.array(q_en, np.uint64), strand=np.array(strand, np.uint8))
    left, right = np.array(left, np.uint32), np.array(right, np.uint32)
    for scores in ((1, 1, 1), (5, 1, 2)):
        rows, out = engine.overlap_split(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"], left, right, scores)
        ob = oracle.Batch(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"])
        orows, oout = oracle.overlap_split(ob, left, right, scores)
        _compare(rows, out, orows, oout, f"odd geometries {scores}")
    assert len(set(orows["status"].tolist())) >= 2          # some of these really are panics in the reference


@pytest.mark.parametrize("policy,key", [(rustybam_amd.BSEARCH_MODERN, "trim_paf_modern"),
                                        (rustybam_amd.BSEARCH_LEGACY, "trim_paf_legacy")])
def test_trim_paf_fixture_end_to_end(engine, golden, policy, key):
    """full `rb trim-paf` on the reference fixture through the device pair kernel: the printed PAF must
    have the digest the oracle CLI produces (tests/golden/digests.json)."""
    r = read_paf(os.path.join(golden, "asm_small.paf"))
    recs = [dict(q_name=r.q_name[i], q_len=r.q_len[i], t_name=r.t_name[i], t_len=r.t_len[i], mapq=r.mapq[i],
                 q_st=int(r.q_st[i]), q_en=int(r.q_en[i]), t_st=int(r.t_st[i]), t_en=int(r.t_en[i]),
                 strand=int(r.strand[i]), cigar=r.cigars[i], id="") for i in range(r.n)]
    out = trim_driver.overlapping_paf_recs(engine, recs, (1, 1, 1), False, policy)
    lines = []
    for x in out:
        lines.append("\t".join(map(str, [x["q_name"], x["q_len"], x["q_st"], x["q_en"], chr(x["strand"]), x["t_name"],
                                          x["t_len"], x["t_st"], x["t_en"], x["nmatch"], x["aln_len"], x["mapq"],
                                          "id:Z:" + x["id"], "cg:Z:" + unpack(x["cigar"])])) + "\n")
    dig = json.load(open(os.path.join(golden, "digests.json")))[key]["md5"]
    assert len(lines) == 249
    assert hashlib.md5("".join(lines).encode()).hexdigest() == dig


@pytest.mark.parametrize("policy,key", [(rustybam_amd.BSEARCH_MODERN, "trim_paf_modern"),
                                        (rustybam_amd.BSEARCH_LEGACY, "trim_paf_legacy")])
def test_trim_paf_fixture_with_the_batch_resident_on_the_device(golden, policy, key):
    """the same file through trim_driver.ResidentTrim: the batch is uploaded once, every pass cuts its pairs in place
    (rb_dev_overlap_split writing behind the ops, rb_dev_apply_pairs), only pair lists and a few row columns cross PCIe, and
    rb_dev_gather_records makes the result a dense batch again.  Same digest as the oracle CLI."""
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    eng = rustybam_amd.Engine(0, torch.cuda.current_stream().cuda_stream)
    r = read_paf(os.path.join(golden, "asm_small.paf"))
    names = {}
    group = np.array([names.setdefault(q, len(names)) for q in r.q_name])
    rank = {q: i for i, q in enumerate(sorted(names))}           # groups in the order of the names (the reference sorts by name)
    group = np.array([rank[q] for q in r.q_name])
    T = trim_driver.ResidentTrim(eng, torch, dev, r.ops, r.op_off, r.t_st, r.t_en, r.q_st, r.q_en, r.strand, group)
    norm0 = T.d_norm.cpu().numpy().view(rustybam_amd.NORM_DT)[:r.n].copy()
    T.run((1, 1, 1), policy)
    assert T.passes >= 2 and T.pairs_done > 100
    if policy == rustybam_amd.BSEARCH_MODERN:   # whole-chromosome alignments of up to 75 k ops: the wave kernel takes (nearly) all of their pairs
        assert T.pairs_by_wave >= 0.95 * T.pairs_done, f"{T.pairs_by_wave} of {T.pairs_done} pairs by the wave kernel"
    d_new, new_off, norm = T.gather()
    ops = d_new.cpu().numpy().view(np.uint32)
    lines = []
    for i in T.order:
        rid = ""
        if norm0[i]["lead_ops"] or norm0[i]["trail_ops"]:
            c = r.cigars[i]
            lead, trail = c[:norm0[i]["lead_ops"]], c[len(c) - norm0[i]["trail_ops"]:][::-1]
            rid = f"_TO.{unpack(lead)}.{unpack(trail)}"
        cg = ops[int(new_off[i]):int(new_off[i + 1])]
        lines.append("\t".join(map(str, [r.q_name[i], r.q_len[i], int(norm[i]["q_st"]), int(norm[i]["q_en"]), chr(r.strand[i]), r.t_name[i],
                                          r.t_len[i], int(norm[i]["t_st"]), int(norm[i]["t_en"]), int(norm[i]["nmatch"]), int(norm[i]["aln_len"]),
                                          r.mapq[i], "id:Z:" + rid, "cg:Z:" + unpack(cg)])) + "\n")
    dig = json.load(open(os.path.join(golden, "digests.json")))[key]["md5"]
    assert len(lines) == 249
    assert hashlib.md5("".join(lines).encode()).hexdigest() == dig
    eng.close()


@pytest.mark.parametrize("policy", [rustybam_amd.BSEARCH_MODERN, rustybam_amd.BSEARCH_LEGACY])
def test_pairs_unsorted_qpos_array(engine, oracle, policy):
    """q_st == 0 on '+' with a leading op that consumes no query: qpos_aln starts at q_pos = -1 (u64::MAX) and is not sorted;
    the pair kernel replays the binary search base by base (found by tests/soak/soak_trim.py)."""
    rng = np.random.default_rng(7002)
    b, left, right = _pairs_batch(rng, 200, "wild", zero_bias=True)
    rows, out = engine.overlap_split(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"], left, right, (1, 1, 1), policy)
    ob = oracle.Batch(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"], np.zeros(len(b["t_st"]), np.uint32))
    orows, oout = oracle.overlap_split(ob, left, right, (1, 1, 1), policy)
    assert (orows["status"] == 16).any() and (orows["status"] == 0).any()  # the corner is actually exercised
    _compare(rows, out, orows, oout, f"unsorted qpos policy {policy}")


def test_pass_selection_on_the_device_equals_the_reference_scan(engine):
    """rb_dev_trim_select against the pair scan of Paf::overlapping_paf_recs restated in numpy (paf.rs:231-284): per query group the
    pair of largest overlap, ties to the first in scan order; contained flags; the count of deferred pairs; groups of one record,
    of a few, and big ones (the whole wave works on those), equal overlaps, contained records, touching (zero-overlap) spans"""
    import torch
    from rustybam_amd import capi
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(99)
    sizes = [1, 2, 3, 4, 4, 5, 9, 47, 48, 49, 50, 130, 300] + [int(x) for x in rng.integers(1, 8, 400)]
    rng.shuffle(sizes)
    n = int(sum(sizes))
    grp_off = np.zeros(len(sizes) + 1, np.uint64)
    grp_off[1:] = np.cumsum(sizes)
    order = rng.permutation(n).astype(np.uint32)          # (the records of a group need not be neighbours in the batch)
    norm = np.zeros(n, capi.NORM_DT)
    q_st = rng.integers(0, 4000, n) // 50 * 50             # coarse grid: equal overlaps, exact containment and touching spans happen
    q_len = (rng.integers(1, 40, n)) * 50
    norm["q_st"], norm["q_en"] = q_st, q_st + q_len
    norm["n_ops"] = rng.integers(1, 900, n)

    def want():
        contained = np.zeros(n, np.uint8)
        pairs, deferred = [], 0
        for g in range(len(sizes)):
            recs = order[int(grp_off[g]):int(grp_off[g + 1])]
            best, cnt = None, 0
            for i in range(len(recs)):
                for j in range(i + 1, len(recs)):
                    a, b = int(recs[i]), int(recs[j])
                    ov = min(int(norm["q_en"][a]), int(norm["q_en"][b])) - max(int(norm["q_st"][a]), int(norm["q_st"][b]))
                    if ov < 1:
                        continue
                    if ov == int(norm["q_en"][b]) - int(norm["q_st"][b]):
                        contained[b] = 1
                    elif ov == int(norm["q_en"][a]) - int(norm["q_st"][a]):
                        contained[a] = 1
                    else:
                        cnt += 1
                        if best is None or ov > best[0]:
                            best = (ov, a, b) if norm["q_st"][a] <= norm["q_st"][b] else (ov, b, a)
            if best:
                pairs.append((best[1], best[2]))
                deferred += cnt - 1
        return contained, pairs, deferred

    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8)).to(dev)
    d_order, d_grp, d_norm = d(order), d(grp_off), d(norm)
    d_cont = torch.full((n + 1,), 7, dtype=torch.uint8, device=dev)
    G = len(sizes)
    d_left, d_right = torch.zeros(G + 1, dtype=torch.int32, device=dev), torch.zeros(G + 1, dtype=torch.int32, device=dev)
    d_poff = torch.zeros(G + 1, dtype=torch.int64, device=dev)
    d_pass = torch.zeros(64, dtype=torch.uint8, device=dev)
    d_scr = torch.zeros(engine.trim_select_scratch_bytes(G), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    engine.dev_trim_select(n, G, d_order.data_ptr(), d_grp.data_ptr(), d_norm.data_ptr(), 1000, d_cont.data_ptr(), d_left.data_ptr(),
                           d_right.data_ptr(), d_poff.data_ptr(), d_pass.data_ptr(), d_scr.data_ptr())
    engine.sync()
    ps = d_pass.cpu().numpy().view(capi.TRIM_PASS_DT)[0]
    contained, pairs, deferred = want()
    k = int(ps["n_pairs"])
    assert k == len(pairs) and int(ps["n_deferred"]) == deferred and len(pairs) > 100 and deferred > 100
    got = list(zip(d_left[:k].cpu().numpy().view(np.uint32).tolist(), d_right[:k].cpu().numpy().view(np.uint32).tolist()))
    assert got == pairs
    assert np.array_equal(d_cont[:n].cpu().numpy(), contained) and contained.sum() > 10
    need = np.array([int(norm["n_ops"][a]) + int(norm["n_ops"][b]) for a, b in pairs], np.int64)
    poff = d_poff[:k].cpu().numpy()
    assert np.array_equal(poff, 1000 + np.concatenate([[0], np.cumsum(need)[:-1]])) and int(ps["ops_end"]) == 1000 + int(need.sum())


@pytest.mark.parametrize("n,lo,hi", [(40_000, 300, 700), (60_000, 40, 120), (8_000, 1500, 2600)])
def test_break_paf_straight_off_the_trimmed_batch(n, lo, hi):
    """The README pipeline trim-paf | break-paf on a resident batch WITHOUT rb_dev_gather_records between the two (RB_LIFT_OP_STARTS): the
    passes cut the records in place, op_off becomes a table of starts, and rb_dev_break takes the batch as it lies -- the tile kernel
    streams over the gaps the cuts left between the records of a tile.  Same rows and the same clips (digest over every op) as the route
    through the dense copy, with the one-walk kernel and with the two-walk path; config 4's record shape, short records (several dozen
    records per tile, gaps as long as records) and records above the tile kernel's length."""
    import torch
    from devutil import DevBatch, config4_resident
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    eng = rustybam_amd.Engine(0, torch.cuda.current_stream().cuda_stream)
    T, h = config4_resident(torch, eng, dev, n, lo=lo, hi=hi)
    T.run((1, 1, 1), rustybam_amd.BSEARCH_MODERN)
    assert T.pairs_done > n // 4 and T.pairs_by_wave == T.pairs_done  # (every clip in place: nothing moved)
    res = {}
    for name, pol in (("starts one walk", rustybam_amd.LIFT_OP_STARTS | rustybam_amd.BREAK_ONE_WALK), ("starts two walks", rustybam_amd.LIFT_OP_STARTS)):
        B = DevBatch.from_trimmed(torch, eng, dev, T)
        rows, out, cnt = B.run(None, max_size=100, rows_cap=6 * n, policy=rustybam_amd.BSEARCH_MODERN | pol)
        assert not cnt["redo_two_walk"] and not cnt["overflow"], name
        hr, _ = B.host_rows(rows, out)
        res[name] = (hr.copy(), B.digest(rows, out), int(cnt["phase"][3]), int(cnt["phase"][4]))
    d_new, new_off, norm = T.gather()
    d_c = [torch.from_numpy(np.ascontiguousarray(norm[k]).view(np.int64)).to(dev) for k in ("t_st", "t_en", "q_st", "q_en")]
    G = DevBatch.from_device(torch, eng, dev, d_new, int(new_off[-1]), new_off, d_c, torch.from_numpy(h["strand"]).to(dev))
    rows, out, cnt = G.run(None, max_size=100, rows_cap=6 * n, policy=rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_FUSED_SCAN | rustybam_amd.BREAK_ONE_WALK)
    assert not cnt["redo_two_walk"] and not cnt["overflow"]
    want, _ = G.host_rows(rows, out)
    want_digest = G.digest(rows, out)
    for name, (got, dg, tiles, handed_back) in res.items():
        assert len(got) == len(want), name
        for k in ("rec", "win", "status", "out_n", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len"):
            assert np.array_equal(got[k], want[k]), f"{name}: {k}"
        assert dg == want_digest, name
        if hi <= 2048:  # (the short-record shape cuts some records below the 8 ops a tile's record needs: their tiles go back, same rows)
            assert tiles > 0 and (lo < 300 or handed_back <= n // 50), f"{name}: {tiles} tiles, {handed_back} records handed back to the per-record kernel"
    del B, G, rows, out
    T.release()
    torch.cuda.synchronize()
    eng.close()


@pytest.mark.parametrize("n,lo,hi", [(40_000, 300, 700), (12_000, 900, 2600)])
def test_liftover_straight_off_the_trimmed_batch(n, lo, hi):
    """rb_dev_liftover with RB_LIFT_OP_STARTS on the batch as trim-paf's passes left it: 600 sliding windows over the targets, monotone and
    (second shape) with records above the tile kernel's length; same rows and the same clip digest as liftover on the dense copy."""
    import torch
    from devutil import DevBatch, config4_resident
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    eng = rustybam_amd.Engine(0, torch.cuda.current_stream().cuda_stream)
    T, h = config4_resident(torch, eng, dev, n, lo=lo, hi=hi)
    T.run((1, 1, 1), rustybam_amd.BSEARCH_MODERN)
    assert T.pairs_by_wave == T.pairs_done
    w_st = np.arange(0, 200_200_000, 333_333, dtype=np.uint64)
    w = (np.zeros(len(w_st), np.uint32), w_st, w_st + np.uint64(400_000))
    B = DevBatch.from_trimmed(torch, eng, dev, T)
    rows, out, cnt = B.run(w, policy=rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_OP_STARTS, rows_cap=8 * n)
    assert not cnt["overflow"]
    got, got_digest, tiles = B.host_rows(rows, out)[0].copy(), B.digest(rows, out), int(cnt["phase"][3])
    d_new, new_off, norm = T.gather()
    d_c = [torch.from_numpy(np.ascontiguousarray(norm[k]).view(np.int64)).to(dev) for k in ("t_st", "t_en", "q_st", "q_en")]
    G = DevBatch.from_device(torch, eng, dev, d_new, int(new_off[-1]), new_off, d_c, torch.from_numpy(h["strand"]).to(dev))
    rows, out, cnt = G.run(w, rows_cap=8 * n)
    want, _ = G.host_rows(rows, out)
    assert len(got) == len(want) and len(want) > n // 2
    for k in ("rec", "win", "status", "flags", "out_n", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len"):
        sel = want["status"] == 0 if k not in ("rec", "win", "status") else slice(None)
        assert np.array_equal(got[k][sel], want[k][sel]), k
    assert got_digest == G.digest(rows, out) and tiles > 0
    del B, G, rows, out
    T.release()
    torch.cuda.synchronize()
    eng.close()
