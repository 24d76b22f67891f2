// k_liftover_list.hip -- the per-record clip kernel (k_liftover.hip) over a LIST of records: the records of the tiles the tile kernel
// (k_tile.hip) handed back.  Workgroups that stay and take entry after entry.  A translation unit of its own because of its registers:
// the loop around the record's body makes the compiler park spilled scalar registers in two more vector registers than the plain
// kernels need, and it places them right behind its own allocation -- where the plain kernels keep their load ring (v80..v95,
// tools/check_ring.py found them at v80 v81).  Here the ring sits at v88..v103 and the compiler is held to 84 registers: four waves
// per SIMD instead of five, on a path that sees the odd record.
#define RB_LIST_TU 1
#define RB_RING_BASE 88
#define RB_SPILL_ROOM 4
#define RB_WPE 4, 5
#include "k_liftover.hip"
