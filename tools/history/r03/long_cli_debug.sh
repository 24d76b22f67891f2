#!/bin/bash
# diagnostics: rb vs the oracle CLI on the long-op cases of tests/test_long_ops.py, outputs kept under gpurun_out/
cd "$(dirname "$0")/.."
D=$(mktemp -d /tmp/rb_long_XXXX)
python3 - "$D" <<'PY'
import sys
sys.path.insert(0, "tests")
import test_long_ops as T
d = sys.argv[1]
open(d + "/long.paf", "w").write("\n".join(T.CASES[:-1]) + "\n")
open(d + "/w.bed", "w").write("".join(f"chrBig\t{a}\t{e}\n" for a, e in T.W))
PY
mkdir -p gpurun_out/long_cli
for cmd in "liftover --bed $D/w.bed $D/long.paf" "break-paf --max-size 100 $D/long.paf" "invert $D/long.paf" "stats --paf $D/long.paf"; do
  k=$(echo $cmd | cut -d' ' -f1)
  rustybam_amd/rb $cmd > gpurun_out/long_cli/$k.rb 2> gpurun_out/long_cli/$k.rb.err; echo "$k rb rc=$?"
  oracle/rb_oracle $cmd > gpurun_out/long_cli/$k.oracle 2> gpurun_out/long_cli/$k.oracle.err; echo "$k oracle rc=$?"
  cmp gpurun_out/long_cli/$k.rb gpurun_out/long_cli/$k.oracle && echo "$k: same"
done
rm -rf "$D"
