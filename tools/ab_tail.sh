#!/bin/bash
# Same-box timing of library variants, weight-gradient launch in view:
#   variants at mpqe_amd/lib/alt/lib<name>.so;   gpurun -- ./tools/ab_tail.sh <outdir> <name> <name> ...
out=$1; shift
mkdir -p $out
L=mpqe_amd/lib
cp $L/libmpqe_amd.so /tmp/lib_orig.so
for rep in ${REPS:-1 2}; do
for v in "$@"; do
  cp $L/alt/lib$v.so $L/libmpqe_amd.so
  timeout -k 10 200 python tools/chain_timeline.py --tail > $out/tail_$v$rep.txt 2>&1
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-scatter > $out/b_$v$rep.json 2> $out/b_$v$rep.err
  echo "== $v $rep"; grep -A3 "weight-gradient launch" $out/tail_$v$rep.txt | grep -v "shader clock"; grep "BWD:" $out/tail_$v$rep.txt
  python - <<P
import json
try:
    d=json.loads(open('$out/b_$v$rep.json').read().strip().splitlines()[-1])
    print('bench', round(d['value']/1e6,2), round(d['ms_per_step']*1e3,2), [ (k['kernel'][5:10], round(k['avg_launch_us'],1)) for k in d['kernels']])
except Exception as e:
    print('bench failed', e)
P
done
done
cp /tmp/lib_orig.so $L/libmpqe_amd.so
