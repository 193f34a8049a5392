#!/bin/bash
# Same-box A/B of builds of the library (box-to-box differences are larger than most kernel changes):
#   put the builds at mpqe_amd/lib/alt/lib<name>.so, then   gpurun -- ./tools/ab_same_box.sh <name> <name> ...
set -e
L=mpqe_amd/lib
for v in "$@"; do
  cp $L/alt/lib$v.so $L/libmpqe_amd.so
  timeout -k 10 300 python -m pytest tests/test_step.py -m gpu -x -q > gpurun_out/ab_t_$v.log 2>&1 || { echo "TESTS FAILED for $v"; tail -5 gpurun_out/ab_t_$v.log; exit 1; }
done
for rep in 1 2; do
for v in "$@"; do
  cp $L/alt/lib$v.so $L/libmpqe_amd.so
  timeout -k 10 200 python tools/chain_timeline.py > gpurun_out/ab_tl_$v$rep.log 2>&1
  timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/ab_b_$v$rep.log 2>&1
  echo "== $v $rep"; grep -A1 "10 longest" gpurun_out/ab_tl_$v$rep.log | tail -1; grep "shader clock" gpurun_out/ab_tl_$v$rep.log
  python - <<P
import json
d=json.loads(open('gpurun_out/ab_b_$v$rep.log').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], [ (k['kernel'], round(k['avg_launch_us'],1)) for k in d['kernels']])
P
done
done
