"""The streaming clip kernel keeps its load ring in VGPRs v80..v95 that the compiler must not touch (k_liftover.hip: amdgpu_num_vgpr(80),
the ring named literally in inline asm).  Nothing in the language guarantees that: the registers the compiler spills scalar registers
into are placed behind its own allocation and have reached v80 in one build.  This test compiles the kernels to assembly (hipcc
cross-compiles without a GPU) and fails if any instruction outside the inline-asm blocks of rb_k_liftover_stream names v80..v95."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not present")
def test_compiler_stays_out_of_the_ring_registers():
    d = tempfile.mkdtemp(prefix="rb_ring_")
    try:
        out = os.path.join(d, "k.s")
        src = os.path.join(ROOT, "rustybam_amd", "csrc", "k_liftover.hip")
        flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off"]  # (the Makefile's code-generation flags)
        subprocess.check_call([HIPCC, *flags, "-S", "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
        text = open(out).read()
        ring = re.compile(r"\bv(8[0-9]|9[0-5])\b|v\[(8[0-9]|9[0-5]):")
        found = 0
        for m in re.finditer(r"^(_Z20rb_k_liftover_streamILb[01]E\w*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
            found += 1
            in_asm, bad = False, []
            for ln in m.group(2).splitlines():
                if "#ASMSTART" in ln:
                    in_asm = True
                elif "#ASMEND" in ln:
                    in_asm = False
                elif not in_asm and ring.search(ln):
                    bad.append(ln.strip())
            assert not bad, (m.group(1), bad[:5])
        assert found == 2, "both builds of rb_k_liftover_stream (liftover, break-paf in one walk) are expected in the assembly"
    finally:
        shutil.rmtree(d, ignore_errors=True)
