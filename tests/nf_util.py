"""Helpers of the nucfreq tests: a BAM reader (gzip + struct), a random read generator, GPU-vs-oracle comparison."""
import gzip
import struct

import numpy as np

OPS = "MIDNSHP=X"
REF_OPS = {0, 2, 3, 7, 8}
QRY_OPS = {0, 1, 4, 7, 8}


class Reads:
    """SoA of BAM records in file order (the layout of rb_reads_view)."""

    def __init__(self, tid, pos, flag, cigars, seqs, l_seq=None):
        self.tid = np.asarray(tid, np.int32)
        self.pos = np.asarray(pos, np.int64)
        self.flag = np.asarray(flag, np.uint32)
        self.op_off = np.zeros(len(cigars) + 1, np.uint64)
        if len(cigars):
            self.op_off[1:] = np.cumsum([len(c) for c in cigars], dtype=np.uint64)
        self.ops = np.array([w for c in cigars for w in c], np.uint32)
        self.l_seq = np.asarray(l_seq if l_seq is not None else [len(s) for s in seqs], np.uint32)
        packed = []
        self.seq_off = np.zeros(len(seqs), np.uint64)
        o = 0
        for i, s in enumerate(seqs):  # s: sequence of nibbles
            self.seq_off[i] = o
            s = list(s) + ([0] if len(s) & 1 else [])
            b = bytes((s[k] << 4) | s[k + 1] for k in range(0, len(s), 2))
            packed.append(b)
            o += len(b)
        self.seq = np.frombuffer(b"".join(packed) + b"\0", np.uint8).copy()
        self.n = len(cigars)

    def args(self):
        return (self.tid, self.pos, self.flag, self.op_off, self.ops, self.l_seq, self.seq_off, self.seq)

    def slice(self, a, b):
        r = Reads.__new__(Reads)
        r.tid, r.pos, r.flag, r.l_seq, r.seq_off = self.tid[a:b], self.pos[a:b], self.flag[a:b], self.l_seq[a:b], self.seq_off[a:b]
        r.op_off, r.ops, r.seq, r.n = self.op_off[a:b + 1], self.ops, self.seq, b - a
        return r


def read_bam(path):
    """-> (ref names, ref lengths, Reads).  The CG:B,I long-cigar convention is not needed by the fixtures."""
    d = gzip.open(path, "rb").read()
    assert d[:4] == b"BAM\x01"
    p = 8 + struct.unpack_from("<i", d, 4)[0]
    n_ref = struct.unpack_from("<i", d, p)[0]
    p += 4
    names, lens = [], []
    for _ in range(n_ref):
        l = struct.unpack_from("<i", d, p)[0]
        names.append(d[p + 4:p + 4 + l - 1].decode())
        lens.append(struct.unpack_from("<i", d, p + 4 + l)[0])
        p += 8 + l
    tid, pos, flag, cigars, seqs = [], [], [], [], []
    while p < len(d):
        bs = struct.unpack_from("<i", d, p)[0]
        r = d[p + 4:p + 4 + bs]
        p += 4 + bs
        t, ps, l_rn, _mq, _bin, n_cig, fl, l_seq = struct.unpack_from("<iiBBHHHi", r, 0)
        c0 = 32 + l_rn
        cig = list(struct.unpack_from("<%dI" % n_cig, r, c0))
        sq = r[c0 + 4 * n_cig:c0 + 4 * n_cig + (l_seq + 1) // 2]
        nib = []
        for b in sq:
            nib += [b >> 4, b & 15]
        tid.append(t), pos.append(ps), flag.append(fl), cigars.append(cig), seqs.append(nib[:l_seq])
    return names, lens, Reads(tid, pos, flag, cigars, seqs)


def random_reads(rng, n, n_contig=2, span=20000, long_frac=0.1, odd_flags=True, max_ops=40):
    recs = []
    for _ in range(n):
        t = int(rng.integers(0, n_contig))
        ps = int(rng.integers(0, span))
        n_mid = int(rng.integers(1, max_ops)) if rng.random() > long_frac else int(rng.integers(max_ops, 12 * max_ops))
        cig = []
        if rng.random() < 0.15:
            cig.append((int(rng.integers(1, 50)), 5))
        if rng.random() < 0.3:
            cig.append((int(rng.integers(1, 200)), 4))
        mids = []
        for _k in range(n_mid):
            o = int(rng.choice([0, 0, 0, 7, 7, 8, 1, 2, 3, 6], p=None))
            l = int(rng.integers(1, 4)) if o in (8, 6) else (int(rng.integers(1, 30)) if o in (1, 2) else int(rng.integers(1, 400)))
            if o == 3:
                l = int(rng.integers(1, 3000))
            mids.append((l, o))
        if not any(o in REF_OPS for _l, o in mids):
            mids.append((int(rng.integers(1, 100)), 0))
        cig += mids
        if rng.random() < 0.3:
            cig.append((int(rng.integers(1, 200)), 4))
        if rng.random() < 0.15:
            cig.append((int(rng.integers(1, 50)), 5))
        if len(cig) == 1 and cig[0][1] not in (0, 7, 8):
            cig = [(cig[0][0], 0)]
        qlen = sum(l for l, o in cig if o in QRY_OPS)
        seq = rng.choice([1, 2, 4, 8, 1, 2, 4, 8, 1, 2, 4, 8, 15, 0, 3, 5, 10], size=qlen).tolist()
        fl = 0
        if odd_flags:
            fl = int(rng.choice([0, 0, 0, 16, 16, 2048, 256, 512, 1024, 4, 1 | 64]))
        recs.append((t, ps, fl, [(l << 4) | o for l, o in cig], seq))
    recs.sort(key=lambda r: (r[0], r[1]))
    return Reads([r[0] for r in recs], [r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs], [r[4] for r in recs])


def oracle_region(oracle, reads, tid, st, en, piece=10000):
    """the reference's per-10-kb fetch + pileup (main.rs:100-110) -> dense (covered mask, counts [en - st, 4])"""
    cov = np.zeros(en - st, bool)
    cnt = np.zeros((en - st, 4), np.uint64)
    for s0 in range(st, en, piece):
        s1 = min(s0 + piece, en)
        rc, p, c = oracle.nucfreq(*reads.args(), tid, s0, s1)
        assert rc == 0, rc
        cov[p.astype(np.int64) - st] = True
        cnt[p.astype(np.int64) - st] = c
    return cov, cnt


def check_regions(eng, oracle, reads, regions, nf_covered=0x80000000):
    rg = np.array(regions, np.int64).reshape(-1, 3)
    counts, status, ctr = eng.nucfreq(*reads.args(), rg[:, 0], rg[:, 1], rg[:, 2])
    o = 0
    total_cov = 0
    for t, st, en in rg.tolist():
        cov, cnt = oracle_region(oracle, reads, t, st, en)
        g = counts[o:o + (en - st)].astype(np.uint64)
        gcov = (g[:, 0] & nf_covered) != 0
        g[:, 0] &= nf_covered - 1
        assert np.array_equal(gcov, cov), (t, st, en, np.flatnonzero(gcov != cov)[:5])
        bad = np.flatnonzero((g != cnt).any(axis=1))
        assert bad.size == 0, (t, st, en, bad[:5] + st, g[bad[:3]], cnt[bad[:3]])
        o += en - st
        total_cov += int(cov.sum())
    assert ctr["n_covered"] == total_cov
    return counts, status, ctr
