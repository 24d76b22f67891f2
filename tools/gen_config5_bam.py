"""A BGZF-compressed BAM of the SURVEY 8d config-5 shape: `python tools/gen_config5_bam.py <contig_bp> <out.bam>` -- 30x coverage of one
contig `chrS` by 15 kb reads (the generator of tools/bench_nucfreq.py), bases A/C/G/T, no qualities.  Needs no GPU."""
import os
import struct
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_nucfreq import make_reads, SEED

contig, path = int(sys.argv[1]), sys.argv[2]
pos, ops, op_off, n = make_reads(contig, 30, 15000)
nop = int(op_off[1] - op_off[0])
rng = np.random.default_rng(SEED)
codes = np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88], np.uint8)
raw = bytearray(b"BAM\x01")
text = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chrS\tLN:%d\n" % contig
raw += struct.pack("<i", len(text)) + text + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chrS\0" + struct.pack("<i", contig)
qual = b"\xff" * 15000
rec_start, rec_end_pos = [], []
ref_len = ((ops.reshape(n, nop) >> 4) * np.isin(ops.reshape(n, nop) & 15, [0, 2])).sum(axis=1)
for r in range(n):
    rec_start.append(len(raw))
    name = b"r%d\0" % r
    cig = ops[r * nop:(r + 1) * nop].astype("<u4").tobytes()
    seq = codes[rng.integers(0, 16, 7500)].tobytes()
    body = struct.pack("<iiBBHHHiiii", 0, int(pos[r]), len(name), 60, 0, nop, 0, 15000, -1, -1, 0) + name + cig + seq + qual
    raw += struct.pack("<i", len(body)) + body
coff = []
with open(path, "wb") as f:                      # BGZF: gzip members of <= 64 KiB with the BC extra field, then the empty EOF member
    for o in list(range(0, len(raw), 0xFF00)) + [None]:
        coff.append(f.tell())
        chunk = bytes(raw[o:o + 0xFF00]) if o is not None else b""
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        bsize = len(comp) + 12 + 6 + 8
        f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize - 1) + comp + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
print(f"{n} reads, {len(raw)} BAM bytes", file=sys.stderr)

# ---- the .bai (SAM spec 5.2): bins of chunks of virtual offsets + the 16 kb linear index ----
def voff(u):
    return (coff[u // 0xFF00] << 16) | (u % 0xFF00)


def reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


bins, lin = {}, {}
starts = rec_start + [len(raw)]
for r in range(n):
    b, e = int(pos[r]), int(pos[r]) + int(ref_len[r])
    v0, v1 = voff(starts[r]), voff(starts[r + 1])
    ch = bins.setdefault(reg2bin(b, e), [])
    if ch and ch[-1][1] == v0:
        ch[-1][1] = v1
    else:
        ch.append([v0, v1])
    for w in range(b >> 14, ((e - 1) >> 14) + 1):
        lin[w] = min(lin.get(w, v0), v0)
n_intv = max(lin) + 1 if lin else 0
io = [lin.get(w, 0) for w in range(n_intv)]
for w in range(n_intv - 2, -1, -1):              # windows nobody overlaps take the next window's offset, as samtools writes them
    if io[w] == 0:
        io[w] = io[w + 1]
with open(path + ".bai", "wb") as f:
    f.write(b"BAI\x01" + struct.pack("<i", 1) + struct.pack("<i", len(bins)))
    for b, ch in bins.items():
        f.write(struct.pack("<Ii", b, len(ch)) + b"".join(struct.pack("<QQ", a, z) for a, z in ch))
    f.write(struct.pack("<i", n_intv) + b"".join(struct.pack("<Q", x) for x in io))
