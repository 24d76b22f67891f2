#!/bin/bash
for D in "$@"; do
  make -s -C rustybam_amd/csrc clean >/dev/null
  make -s -j8 -C rustybam_amd/csrc all CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -ffp-contract=off $D" >/dev/null 2>&1
  echo "== $D"; python tools/dbg_tiled.py 2>&1 | grep -E "^rows|^q_st|^out_n"
done
