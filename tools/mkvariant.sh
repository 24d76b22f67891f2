#!/bin/bash
# tools/mkvariant.sh <name> [--src k_xxx.hip] [-D flags ...]: a variant of the library with ONE kernel file (default k_liftover.hip)
# rebuilt under extra flags, as rustybam_amd/variants/<name>.so (git-ignored, travels to the GPU box; tools/ab_so.sh times variants
# against each other on one box; RB_VARIANT=<name> makes rustybam_amd.capi load it)
set -e
name=$1; shift
src=k_liftover.hip
if [ "$1" = "--src" ]; then src=$2; shift 2; fi
cd $(dirname $0)/../rustybam_amd/csrc
mkdir -p ../variants /tmp/rbvar_$name
obj=${src%.hip}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-function -ffp-contract=off "$@" -c $src -o /tmp/rbvar_$name/$obj
objs=""
for o in capi.o k_records.o k_liftover.o k_liftover_list.o k_tile.o k_misc.o k_trim.o k_trim4.o k_text.o k_nucfreq.o; do
  if [ "$o" = "$obj" ]; then objs="$objs /tmp/rbvar_$name/$obj"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/$name.so $objs
echo "built variants/$name.so ($src $*)"
