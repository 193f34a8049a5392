#!/bin/bash
# Same-box bench of library variants:  gpurun -- ./tools/ab_bench.sh <outdir> "<bench flags>" <name> <name> ...
out=$1; shift
flags=$1; shift
mkdir -p $out
L=mpqe_amd/lib
cp $L/libmpqe_amd.so /tmp/lib_orig.so
for rep in ${REPS:-1 2}; do
for v in "$@"; do
  cp $L/alt/lib$v.so $L/libmpqe_amd.so
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-scatter $flags > $out/b_$v$rep.json 2> $out/b_$v$rep.err
  python - <<P
import json
try:
    d=json.loads(open('$out/b_$v$rep.json').read().strip().splitlines()[-1])
    print('$v $rep', round(d['value']/1e6,2), round(d['ms_per_step']*1e3,2), [ (k['kernel'][5:10], round(k['avg_launch_us'],1)) for k in d['kernels']])
except Exception as e:
    print('$v bench failed', e)
P
done
done
cp /tmp/lib_orig.so $L/libmpqe_amd.so
