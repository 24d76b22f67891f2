"""rb_dev_break with RB_BREAK_ONE_WALK: the clip kernel finds the long indels itself while it streams a record.  Same rows and clipped
CIGARs as the two-walk path and the oracle (liftover.rs:182-226).  Records it does not take (irregular CIGARs, records its fused
verification hands back, boundaries only the generic kernel resolves) are declined ONE BY ONE (round 3; round 2 redid the whole
batch): their pieces come from rb_k_break_pieces in list mode and the generic kernel, everybody else's rows stay."""
import numpy as np
import pytest

import rustybam_amd
from devutil import DevBatch
from rbtest_util import batch_args, digest_rows, random_batch, random_cigar, sums

pytestmark = pytest.mark.gpu
BASE = rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_FUSED_SCAN


@pytest.fixture(scope="module")
def ctx():
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    eng = rustybam_amd.Engine(0, torch.cuda.current_stream().cuda_stream)
    yield torch, eng, dev
    eng.close()


@pytest.mark.parametrize("max_size", [0, 2, 25, 100])
@pytest.mark.parametrize("seed", [1, 2])
def test_one_walk_equals_two_walks_and_the_oracle(ctx, oracle, max_size, seed):
    torch, eng, dev = ctx
    rng = np.random.default_rng(4000 + seed)
    # seed 2: leading / trailing indel runs, some of which make the reference panic -- with the fused scan such a record keeps its
    # rows (they carry the status) where the oracle has none, so that batch is compared between the two device paths only
    mode = "regular" if seed == 1 else "indel_ends"
    b = random_batch(rng, 400, mode, n_contig=1, long_frac=0.4)
    D = DevBatch(torch, eng, dev, b)
    rows2, out2, cnt2 = D.run(max_size=max_size, policy=BASE)
    assert cnt2["redo_two_walk"] == 0
    want = D.digest(rows2, out2)
    n_rows = rows2.shape[0]
    most = int(np.bincount(D.host_rows(rows2, out2)[0]["rec"].astype(np.int64)).max())
    if mode == "regular":
        orows, oops = oracle.break_paf(oracle.Batch(*batch_args(b), b["contig"]), max_size)
        assert n_rows == len(orows) and want == digest_rows(orows, oops)
    ok2 = _ok_records(D)
    want_ok = _rows_of(D, rows2, out2, ok2)
    rows1, out1, cnt1 = D.run(max_size=max_size, policy=BASE | rustybam_amd.BREAK_ONE_WALK)
    assert cnt1["redo_two_walk"] == 0  # nothing is handed back batch-wise any more
    assert np.array_equal(_ok_records(D), ok2)
    if mode == "regular":
        assert rows1.shape[0] == n_rows and D.digest(rows1, out1) == want
    else:  # (a record the reference panics on has rows that carry its status on one path, none on the other: compare the others)
        assert _rows_of(D, rows1, out1, ok2) == want_ok
    if max_size == 0:
        assert most > 32  # (records with more pieces than one pass of the kernel holds: several passes)


def _ok_records(D):
    """per record: did it pass remove_trailing_indels + check_integrity (the fused scan's final verdict in the norm rows)"""
    D.torch.cuda.synchronize()
    return D.d_norm.cpu().numpy().view(rustybam_amd.NORM_DT)["status"][:D.n_rec] == 0


def _rows_of(D, rows, out, ok):
    """rows (with their clipped CIGARs) of the records in `ok`, as comparable tuples in row order"""
    r, o = D.host_rows(rows, out)
    res = []
    for h in r:
        if not ok[int(h["rec"])]:
            continue
        cig = tuple(o[int(h["out_off"]):int(h["out_off"]) + int(h["out_n"])].tolist()) if h["status"] == 0 else ()
        res.append((int(h["rec"]), int(h["win"]), int(h["status"]), int(h["t_st"]), int(h["t_en"]), int(h["q_st"]), int(h["q_en"]),
                    int(h["nmatch"]), int(h["aln_len"]), cig))
    return res


def test_short_records_take_the_one_walk_path(ctx, oracle):
    """records of at most 30 ops cannot have more than 32 pieces: the one-walk path must take the batch, for every --max-size"""
    torch, eng, dev = ctx
    rng = np.random.default_rng(4100)
    b = random_batch(rng, 3000, "regular", n_contig=1, max_ops=30, long_frac=0.0)
    D = DevBatch(torch, eng, dev, b)
    for max_size in (0, 1, 7, 1000):
        orows, oops = oracle.break_paf(oracle.Batch(*batch_args(b), b["contig"]), max_size)
        rows, out, cnt = D.run(max_size=max_size, policy=BASE | rustybam_amd.BREAK_ONE_WALK)
        assert cnt["redo_two_walk"] == 0 and cnt["n_generic"] == 0
        assert rows.shape[0] == len(orows) and D.digest(rows, out) == digest_rows(orows, oops)
        r = D.host_rows(rows, out)[0]
        assert (np.diff(r["rec"].astype(np.int64)) >= 0).all()          # record order, then piece order
        same = np.flatnonzero(np.diff(r["rec"].astype(np.int64)) == 0)
        assert (r["win"][same + 1] == r["win"][same] + 1).all()


def test_declines_irregular_records_one_by_one(ctx, oracle):
    """N / S / H / P ops, zero lengths, adjacent ops of one type among regular records: the one-walk call completes, the irregular
    records' rows come from the generic kernel, and everything equals the two-walk path and the oracle"""
    torch, eng, dev = ctx
    for seed, mode, frac in ((4200, "mixed", 1.0), (4201, "spliced", 0.05), (4202, "wild", 0.01)):
        rng = np.random.default_rng(seed)
        b = random_batch(rng, 600, "regular", n_contig=1, long_frac=0.2)
        odd = random_batch(rng, 600, mode, n_contig=1, long_frac=0.2)
        pick = rng.random(600) < frac
        cigs = [(odd if pick[i] else b)["ops"][int((odd if pick[i] else b)["op_off"][i]):int((odd if pick[i] else b)["op_off"][i + 1])] for i in range(600)]
        m = {k: np.where(pick, odd[k], b[k]) for k in ("t_st", "t_en", "q_st", "q_en", "strand", "contig")}
        m["op_off"] = np.zeros(601, np.uint64)
        m["op_off"][1:] = np.cumsum([len(c) for c in cigs])
        m["ops"] = np.concatenate(cigs)
        D = DevBatch(torch, eng, dev, m)
        for max_size in (0, 10, 100):
            rows2, out2, cnt2 = D.run(max_size=max_size, policy=BASE)
            ok = _ok_records(D)
            want = _rows_of(D, rows2, out2, ok)
            rows1, out1, cnt1 = D.run(max_size=max_size, policy=BASE | rustybam_amd.BREAK_ONE_WALK)
            assert cnt1["redo_two_walk"] == 0 and cnt1["overflow"] == 0
            assert np.array_equal(_ok_records(D), ok)
            got = _rows_of(D, rows1, out1, ok)
            assert got == want, (mode, max_size, len(got), len(want))
            assert cnt1["n_generic"] > 0 or not pick.any()
            # the oracle on the records that pass: same pieces, same clipped CIGARs
            keep = np.flatnonzero(ok)
            sub = {k: m[k][keep] for k in ("t_st", "t_en", "q_st", "q_en", "strand", "contig")}
            sc = [cigs[i] for i in keep]
            sub["op_off"] = np.zeros(len(keep) + 1, np.uint64)
            sub["op_off"][1:] = np.cumsum([len(c) for c in sc])
            sub["ops"] = np.concatenate(sc) if sc else np.zeros(0, np.uint32)
            orows, oops = oracle.break_paf(oracle.Batch(*batch_args(sub), sub["contig"]), max_size)
            ogot = [(int(keep[int(h["rec"])]), int(h["status"]), int(h["t_st"]), int(h["t_en"]), int(h["q_st"]), int(h["q_en"]), int(h["nmatch"]), int(h["aln_len"]),
                     tuple(oops[int(h["out_off"]):int(h["out_off"]) + int(h["out_n"])].tolist()) if h["status"] == 0 else ()) for h in orows]
            assert [(g[0],) + g[2:] for g in got] == ogot, (mode, max_size)


def test_many_pieces_in_several_passes(ctx):
    torch, eng, dev = ctx
    # n long indels in one regular record, n + 1 pieces: exactly a pass, one more, several passes, pieces that span segments of the
    # stream (a record of 6000+ ops), long insertions next to long deletions (pieces without reference bases are no pieces)
    M, I_, D_ = 0, 1, 2
    for n_cut, m_len in ((31, 50), (32, 50), (40, 50), (200, 7), (1500, 3), (3000, 1)):
        cig = []
        for k in range(n_cut):
            cig += [(m_len << 4) | M, (500 << 4) | (D_ if k % 3 else I_)]
            if k % 7 == 0:
                cig += [(600 << 4) | (I_ if k % 3 else D_)]  # two long indels in a row
        cig.append((m_len << 4) | M)
        cig = np.array(cig, np.uint32)
        R, Q = sums(cig)
        one = dict(ops=cig, op_off=np.array([0, len(cig)], np.uint64), t_st=np.array([100], np.uint64), t_en=np.array([100 + R], np.uint64),
                   q_st=np.array([7], np.uint64), q_en=np.array([7 + Q], np.uint64), strand=np.array([ord("-")], np.uint8),
                   contig=np.zeros(1, np.uint32))
        D1 = DevBatch(torch, eng, dev, one)
        rows2, out2, cnt2 = D1.run(max_size=100, policy=BASE, rows_cap=8192)
        rows1, out1, cnt1 = D1.run(max_size=100, policy=BASE | rustybam_amd.BREAK_ONE_WALK, rows_cap=8192)
        assert cnt1["redo_two_walk"] == 0 and cnt2["redo_two_walk"] == 0
        assert rows1.shape[0] == rows2.shape[0] == n_cut + 1, (n_cut, rows1.shape, rows2.shape)
        assert D1.digest(rows1, out1) == D1.digest(rows2, out2), n_cut
    # long pieces (each keeps its place in an output slot) around a piece of one op at a pass boundary: pieces 30 and 32 then lie
    # in the same slot within one 16-byte group of each other, and belong to different passes
    for tiny_at in (31, 63, 33):
        cig = []
        for k in range(70):
            cig += [((1 if k == tiny_at else 300 + k) << 4) | M] + ([(2 << 4) | I_, (250 << 4) | M] if k != tiny_at else []) + [(500 << 4) | D_]
        cig.append((40 << 4) | M)
        cig = np.array(cig, np.uint32)
        R, Q = sums(cig)
        one = dict(ops=cig, op_off=np.array([0, len(cig)], np.uint64), t_st=np.array([0], np.uint64), t_en=np.array([R], np.uint64),
                   q_st=np.array([3], np.uint64), q_en=np.array([3 + Q], np.uint64), strand=np.array([ord("+")], np.uint8),
                   contig=np.zeros(1, np.uint32))
        D1 = DevBatch(torch, eng, dev, one)
        rows2, out2, cnt2 = D1.run(max_size=100, policy=BASE, rows_cap=8192)
        rows1, out1, cnt1 = D1.run(max_size=100, policy=BASE | rustybam_amd.BREAK_ONE_WALK, rows_cap=8192)
        assert cnt1["redo_two_walk"] == 0 and rows1.shape[0] == rows2.shape[0] == 71
        assert D1.digest(rows1, out1) == D1.digest(rows2, out2), tiny_at
