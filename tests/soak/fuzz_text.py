"""Fuzz rb_k_parse_cigars against the oracle's parser: random byte strings over a CIGAR-like alphabet."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import rustybam_amd
from oracle import pyoracle as oracle
L = oracle.lib()
def oparse(b):
    ops, n = C.POINTER(C.c_uint32)(), C.c_size_t()
    rc = L.rbo_parse_cigar(b, C.c_size_t(len(b)), C.byref(ops), C.byref(n))
    if rc != 0:
        return None
    out = np.ctypeslib.as_array(ops, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint32)
    L.rbo_free(ops)
    return out
eng = rustybam_amd.Engine(0)
rng = np.random.default_rng(99)
alpha = np.frombuffer(b"0123456789MIDNSHP=X0123456789MIDNSHP=X0123456789MQ- +\n", dtype=np.uint8)
n_cases, bad, good = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 0, 0
for it in range(0, n_cases, 500):
    strs = []
    for _ in range(500):
        if rng.random() < 0.5:  # mostly valid, then damaged
            k = int(rng.integers(0, 400))
            pads = [0, 0, 0, 1, 3, 9] if rng.random() < 0.1 else [0, 0, 0, 0, 1]  # (ten digits or more: the device hands the string to the host)
            parts = [f"{'0' * int(rng.choice(pads))}{int(rng.integers(0, 10 ** int(rng.integers(1, 9))))}{'MIDNSHP=X'[int(rng.integers(0, 9))]}" for _ in range(k)]
            s = bytearray("".join(parts).encode())
            for _ in range(int(rng.integers(0, 2))):
                if s:
                    s[int(rng.integers(0, len(s)))] = int(alpha[int(rng.integers(0, len(alpha)))])
            strs.append(bytes(s))
        else:
            strs.append(bytes(alpha[rng.integers(0, len(alpha), int(rng.integers(0, 80)))]))
    op_off, ops, status = eng.parse_cigars(strs)
    for i, s in enumerate(strs):
        want = oparse(s)
        if status[i] == 3:  # handed to the host's parser
            continue
        if want is None:
            assert status[i] != 0, (s, "device accepted what the oracle rejects")
            bad += 1
        else:
            assert status[i] == 0, (s, int(status[i]))
            assert np.array_equal(ops[int(op_off[i]):int(op_off[i + 1])], want), s
            good += 1
print(f"fuzz ok: {good} accepted, {bad} rejected identically")
