set -e
make -s -C rustybam_amd/csrc clean >/dev/null
make -s -j8 -C rustybam_amd/csrc all CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -ffp-contract=off -DRB_MS=2" >/dev/null 2>&1
python -m pytest tests/test_gpu_fullsize.py -x -q -k "liftover_integrity" 2>&1 | tail -15
