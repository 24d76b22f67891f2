"""rb_dev_digest_rows (the verification aid behind bench.py's output_digest): equals its numpy twin on the oracle's output, does not
depend on how a clip is stored (copied ops / descriptors), and the digests of record-range shards add up to the whole batch's -- the
1/2/4/8-GPU determinism check, run here as shards of one GPU."""
import numpy as np
import pytest

import rustybam_amd
from devutil import DevBatch
from rbtest_util import batch_args, digest_rows, random_batch, random_windows
from rustybam_amd import shard

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["regular", "mixed"])
def test_digest_matches_numpy_twin_on_oracle_output_and_shards_add_up(oracle, mode):
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    eng = rustybam_amd.Engine(0, torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(77 if mode == "regular" else 78)
    b = random_batch(rng, 500, mode, n_contig=1, long_frac=0.3)
    w = random_windows(rng, b, 80, True)
    orows, oops = oracle.liftover(oracle.Batch(*batch_args(b), b["contig"]), *w)
    want = digest_rows(orows, oops)
    D = DevBatch(torch, eng, dev, b)
    # (mixed: some records make the reference panic; with the fused scan their rows are kept and carry the status, the stand-alone
    #  scan drops them before the hit count as the oracle does)
    base = rustybam_amd.BSEARCH_MODERN | (rustybam_amd.LIFT_FUSED_SCAN if mode == "regular" else 0)
    rows, out, cnt = D.run(w, policy=base)
    assert rows.shape[0] == len(orows) and len(orows) > 100
    assert D.digest(rows, out) == want
    rows_d, out_d, _ = D.run(w, policy=base | rustybam_amd.LIFT_DESCRIPTORS)
    assert D.digest(rows_d, out_d) == want                      # descriptors into the original cigar: same records
    # a different order or a changed op changes it
    assert digest_rows(orows[::-1], oops) != want
    # ---- shards: 3 contiguous op-balanced record ranges, each digested with the rows / records before it as bases ----
    bounds = shard.shard_bounds(b["op_off"], 3)
    total, row_base = 0, 0
    for s in range(3):
        lo, hi = int(bounds[s]), int(bounds[s + 1])
        sb = shard.shard_slice(b, b["op_off"], lo, hi)
        S = DevBatch(torch, eng, dev, sb)
        r_s, o_s, _ = S.run(w, policy=base)
        total = (total + S.digest(r_s, o_s, row_base, lo)) & ((1 << 64) - 1)
        row_base += r_s.shape[0]
    assert row_base == len(orows) and total == want
    eng.close()
