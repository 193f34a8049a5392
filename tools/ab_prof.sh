#!/bin/bash
# Same-box per-kernel durations (rocprofv3 --kernel-trace --stats) of library variants, wrong-on-purpose builds allowed:
#   variants at mpqe_amd/lib/alt/lib<name>.so;   gpurun -- ./tools/ab_prof.sh <outdir> <name> <name> ...
out=$(realpath $1); shift
mkdir -p $out
root=$(pwd)
L=$root/mpqe_amd/lib
cp $L/libmpqe_amd.so /tmp/lib_orig.so
export TMPDIR=/tmp
for v in "$@"; do
  cp $L/alt/lib$v.so $L/libmpqe_amd.so
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$v -o p -- python3 $root/bench.py --no-cpu-baseline --no-self-check --steps 50 --repeats 3 > $out/prof_$v.log 2>&1)
  echo "== $v"
  python3 - <<P
import csv, json
try:
    d=json.loads([l for l in open('$out/prof_$v.log') if l.startswith('{')][-1])
    print('bench', round(d['value']/1e6,2), round(d['ms_per_step']*1e3,2))
except Exception as e:
    print('no bench line', e)
for r in csv.DictReader(open('$out/prof_$v/p_kernel_stats.csv')):
    if 'step_' in r['Name'] and 'upload' not in r['Name']: print('  ', r['Name'][:34], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
P
done
cp /tmp/lib_orig.so $L/libmpqe_amd.so
