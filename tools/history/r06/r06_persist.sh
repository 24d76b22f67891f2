#!/bin/bash
# round 6 (profiles/r06_stream_summary.md): the per-record clip kernel dispatched per record (default) against the same records through the
# persistent LIST form, interleaved on one box, plain allocations (--placement-tries 1).  The switch it set (RB_STREAM_PERSISTENT=1 in
# capi.hip's lift_common: the schedule copied into fb_list, fb_count = the number of records of the per-record launch,
# rb_launch_liftover_stream_list in place of rb_launch_liftover_stream) measured 5.4 % slower and is no longer in the tree.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in 0 1; do
    RB_STREAM_PERSISTENT=$v python3 bench.py --steps 10 --no-cpu-baseline --no-box --placement-tries 1 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('persistent=$v', 'step %.3f' % d['ms_per_step'], 'kernel %.3f' % d['roofline']['kernel_ms'], 'frac %.4f' % d['roofline']['frac'], d.get('output_digest'))"
  done
done
