#!/bin/bash
# Same-box per-kernel durations of the general path for library variants: gpurun -- ./tools/ab_gen.sh <outdir> <name> ...
out=$1; shift
L=mpqe_amd/lib
cp $L/libmpqe_amd.so /tmp/lib_orig.so
for v in "$@"; do
  [ "$v" != cur ] && cp $L/alt/lib$v.so $L/libmpqe_amd.so
  echo "== $v"; ./tools/gen_prof.sh $out/$v | grep "gemm"
  cp /tmp/lib_orig.so $L/libmpqe_amd.so
done
echo "== cur, no mask"; ./tools/gen_prof.sh $out/nomask MPQE_DBG_GEN_NOMASK=1 | grep "gemm"
