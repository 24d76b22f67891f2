"""rustybam_amd -- MI355X (gfx950) CIGAR-walk engine behind rustybam's liftover / break-paf / stats path.

The product is the C-ABI shared library ``librustybam_amd.so`` (include/rustybam_amd.h) built from
``rustybam_amd/csrc``.  This Python package is only the ctypes plumbing used by tests and bench.py;
it never falls back to a CPU implementation: if the library or a gfx950 device is missing, calls
raise.
"""
from .capi import (Engine, RbError, lib, lib_path, HIT_DT, NORM_DT, REDUCE_DT, COUNTERS_DT, PAIR_DT,  # noqa: F401
                   BSEARCH_MODERN, BSEARCH_LEGACY, LIFT_EARLY_EXIT, LIFT_DESCRIPTORS, LIFT_FUSED_SCAN, LIFT_OP_STARTS, BREAK_ONE_WALK, TRIM_IN_PLACE, HIT_INSIDE, HIT_GENERIC, HIT_DESCRIPTOR, NF_COVERED, RD_OK, RD_FILTERED, RD_BAD_CIGAR, RD_SEQ_SHORT, exported_symbols, declared_symbols)
