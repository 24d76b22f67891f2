"""The streaming clip kernel keeps its load ring in VGPRs v80..v95 that the compiler must not touch (k_liftover.hip: amdgpu_num_vgpr(80),
the ring named literally in inline asm).  tools/check_ring.py (also run by the Makefile on every build) compiles the kernels to assembly
and fails if anything outside the inline-asm blocks of rb_k_liftover_stream names a register of the ring, single or inside a tuple."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_ring  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def test_ring_pattern_sees_tuples_that_reach_into_the_ring():
    assert check_ring.ring_uses("v_mov_b32 v80, v1") == ["v80"]
    assert check_ring.ring_uses("global_load_dwordx4 v[78:81], v[2:3], off") == ["v[78:81]"]
    assert check_ring.ring_uses("v_lshlrev_b64 v[79:80], 2, v[4:5]") == ["v[79:80]"]
    assert check_ring.ring_uses("v_mov_b64 v[94:95], v[10:11]") == ["v[94:95]"]
    assert check_ring.ring_uses("v_mov_b64 v[95:96], v[10:11]") == ["v[95:96]"]
    assert check_ring.ring_uses("v_add_u32 v79, v96, v7") == []
    assert check_ring.ring_uses("v_mov_b64 v[76:79], v[96:99]") == []


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not present")
def test_compiler_stays_out_of_the_ring_registers():
    text = check_ring.compile_to_asm(HIPCC)
    bad = check_ring.check_assembly(text)  # raises unless both builds (liftover, break-paf in one walk) are in the assembly
    assert not bad, bad[:5]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not present")
def test_compiler_stays_out_of_the_tile_kernels_ring():
    """k_tile.hip (short records, one wave per tile of records): the same invisible ring, liftover and break-paf builds."""
    text = check_ring.compile_to_asm(HIPCC, source="k_tile.hip")
    bad = check_ring.check_assembly(text, 2, check_ring.RING, "rb_k_liftover_tile")
    assert not bad, bad[:5]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not present")
def test_compiler_stays_out_of_the_list_forms_ring():
    """k_liftover_list.hip: the per-record kernel over the tile kernel's hand-backs; its ring sits at v88..v103 because the loop's spilled
    scalar registers are parked at v80.. (this check found them there)."""
    text = check_ring.compile_to_asm(HIPCC, source="k_liftover_list.hip")
    bad = check_ring.check_assembly(text, 2, (88, 103))
    assert not bad, bad[:5]
