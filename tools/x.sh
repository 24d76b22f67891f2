make -s -C rustybam_amd/csrc clean >/dev/null
make -s -j8 -C rustybam_amd/csrc all CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -ffp-contract=off -DRB_LINE_ROUND" >/dev/null 2>&1
tools/prof_traffic.sh t2
