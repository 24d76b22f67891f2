#!/bin/bash
# tools/mkvariant.sh <name> [-D flags ...]: a variant of the library with k_liftover.hip rebuilt under extra flags, as
# rustybam_amd/variants/<name>.so (git-ignored, travels to the GPU box; tools/ab_so.sh times variants against each other on one box)
set -e
name=$1; shift
cd $(dirname $0)/../rustybam_amd/csrc
mkdir -p ../variants /tmp/rbvar_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-function -ffp-contract=off "$@" -c k_liftover.hip -o /tmp/rbvar_$name/k_liftover.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/$name.so capi.o k_records.o /tmp/rbvar_$name/k_liftover.o k_misc.o k_trim.o k_text.o k_nucfreq.o
echo "built variants/$name.so ($*)"
