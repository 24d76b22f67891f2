"""Guard for the streaming clip kernel's load ring: VGPRs v80..v95 are reserved outside the compiler's allocation
(k_liftover.hip: amdgpu_num_vgpr(80), the ring named literally in inline asm).  Nothing in the language guarantees that the
compiler stays out of them, so this script compiles k_liftover.hip to assembly (hipcc cross-compiles without a GPU) and fails
if any instruction outside the inline-asm blocks of rb_k_liftover_stream names a register in the ring -- a single register
v80..v95 or ANY tuple v[a:b] whose range intersects it (v[78:81] as well as v[80:83]).  Run by the Makefile on every build of
k_liftover.o and by tests/test_ring_registers.py.  `--tile`: the same for the tile kernel's ring (k_tile.hip, rb_k_liftover_tile*)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RING = (80, 95)
_SINGLE = re.compile(r"\bv(\d+)\b")
_TUPLE = re.compile(r"\bv\[(\d+):(\d+)\]")


def ring_uses(line, ring=RING):
    """Register operands of one assembly line that intersect the ring."""
    lo, hi = ring
    bad = [f"v[{a}:{b}]" for a, b in ((int(x), int(y)) for x, y in _TUPLE.findall(line)) if a <= hi and b >= lo]
    bad += [f"v{a}" for a in (int(x) for x in _SINGLE.findall(line)) if lo <= a <= hi]
    return bad


def check_assembly(text, expect_kernels=3, ring=RING, family="rb_k_liftover_stream"):
    """-> list of (kernel, line) offences; raises if the expected kernels are not in the assembly.
    k_liftover.hip: liftover, break-paf, diagnostics; k_liftover_list.hip: the two list forms (ring at v88..v103); k_tile.hip (family rb_k_liftover_tile): liftover, break-paf."""
    found, offences = 0, []
    for m in re.finditer(r"^(_Z\d+" + family + r"\w*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        found += 1
        in_asm = False
        for ln in m.group(2).splitlines():
            if "#ASMSTART" in ln:
                in_asm = True
            elif "#ASMEND" in ln:
                in_asm = False
            elif not in_asm and ring_uses(ln.split(";")[0], ring):
                offences.append((m.group(1), ln.strip()))
    if found != expect_kernels:
        raise RuntimeError(f"{found} builds of {family} in the assembly, {expect_kernels} expected")
    return offences


def spills(text, family="rb_k_liftover_stream"):
    """-> {kernel: (sgpr_spill_count, vgpr_spill_count, vgpr_count)} from the code-object metadata at the end of the assembly."""
    out = {}
    for m in re.finditer(r"\.name: +(_Z\d+" + family + r"\w*)\n(.*?)\.wavefront_size", text, re.S):
        f = dict(re.findall(r"\.(sgpr_spill_count|vgpr_spill_count|vgpr_count): +(\d+)", m.group(2)))
        out[m.group(1)] = (int(f.get("sgpr_spill_count", -1)), int(f.get("vgpr_spill_count", -1)), int(f.get("vgpr_count", -1)))
    return out


def compile_to_asm(hipcc, extra=(), source="k_liftover.hip"):
    src = os.path.join(ROOT, "rustybam_amd", "csrc", source)
    with tempfile.TemporaryDirectory(prefix="rb_ring_") as d:
        out = os.path.join(d, "k.s")
        flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", *extra]  # the Makefile's code-generation flags
        subprocess.check_call([hipcc, *flags, "-S", "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
        return open(out).read()


def ring_of(flags, macro="RB_RING_BASE"):
    """The ring a build uses: v80..v95 unless -DRB_RING_BASE=<n> (k_tile.hip: -DRBT_RING_BASE=<n>) moves it."""
    base, pf = RING[0], 2
    for f in flags:
        m = re.match(r"-D" + macro + r"=(\d+)$", f)
        if m:
            base = int(m.group(1))
        m = re.match(r"-DRBT?_PF=(\d+)$", f)
        if m:
            pf = int(m.group(1))
    return (base, base + 8 * pf - 1)


if __name__ == "__main__":
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    args = sys.argv[1:]
    tile = "--tile" in args  # k_tile.hip instead of k_liftover.hip
    args = [a for a in args if a != "--tile"]
    lst = "--list" in args  # k_liftover_list.hip: the list form of the per-record kernel, ring at v88..v103
    args = [a for a in args if a != "--list"]
    if lst:
        text = compile_to_asm(hipcc, args, "k_liftover_list.hip")
        bad = check_assembly(text, 2, (88, 103))
        sp = spills(text)
    elif tile:
        text = compile_to_asm(hipcc, args, "k_tile.hip")
        bad = check_assembly(text, 2, ring_of(args, "RBT_RING_BASE"), "rb_k_liftover_tile")
        sp = spills(text, "rb_k_liftover_tile")
    else:
        text = compile_to_asm(hipcc, args)
        bad = check_assembly(text, ring=ring_of(args))
        sp = spills(text)
    for k, ln in bad[:10]:
        print(f"ring register used by the compiler in {k}: {ln}", file=sys.stderr)
    print(f"check_ring: {'FAILED' if bad else 'ok'}; spills (sgpr, vgpr, vgprs) = {sp}")
    sys.exit(1 if bad else 0)
