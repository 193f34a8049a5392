#!/bin/bash
# A variant build of the library for same-box A/B runs (tools/ab.sh): step.hip compiled with extra flags, the other
# objects taken from the installed build.   tools/build_variant.sh <name> [-DFLAG=1 ...]  ->  mpqe_amd/lib/alt/lib<name>.so
set -e
name=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
L=$root/mpqe_amd/lib
mkdir -p $L/alt /tmp/mpqe_variant
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -I$root/include -mllvm -amdgpu-mfma-vgpr-form=1 "$@" \
    -c $root/mpqe_amd/csrc/step.hip -o /tmp/mpqe_variant/step_$name.o 2>&1 | grep -E "error" || true
objs=$(ls $L/obj/*.o | grep -v "/step.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/mpqe_variant/step_$name.o -o $L/alt/lib$name.so
ls -la $L/alt/lib$name.so
