#!/bin/bash
# same-box A/B of prebuilt library variants (rustybam_amd/variants/<name>.so): bench.py kernel time, two interleaved rounds
cd $GRAFT_REPO_ROOT
cp rustybam_amd/librustybam_amd.so /tmp/keep.so
for round in 1 2; do
  for n in "$@"; do
    cp rustybam_amd/variants/$n.so rustybam_amd/librustybam_amd.so
    python bench.py --steps 10 --no-cpu-baseline $AB_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n', 'step %.3f' % d['ms_per_step'], 'kernel %.3f' % d['roofline']['kernel_ms'], d.get('output_digest'))"
  done
done
cp /tmp/keep.so rustybam_amd/librustybam_amd.so
