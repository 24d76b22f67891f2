#!/bin/bash
# the record scan's variants (tools/mkvariant.sh <name> --src k_records.hip ...), timing + row digests (tools/scan_time.py), same box
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in "$@"; do cp rustybam_amd/variants/$v.so rustybam_amd/librustybam_amd.so; echo "== $v"; timeout -k 5 600 python3 tools/scan_time.py 2>&1 | tail -3; done
