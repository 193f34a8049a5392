#!/bin/bash
# per-readout kernel stats of bench.py (rocprofv3 --kernel-trace --stats)
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
for r in ${2:-mlp targetmlp concat}; do
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof_$r -o p -- python3 /root/repo/bench.py --readout $r --steps 100 --warmup 10 --no-cpu-baseline --no-scatter --no-pack-ms > /root/repo/$out/bench_$r.json 2> /root/repo/$out/bench_$r.err)
  echo "== $r"; python3 - <<P
import csv, json
for r in csv.DictReader(open('$out/prof_$r/p_kernel_stats.csv')):
    if 'step_' in r['Name'] and 'upload' not in r['Name']: print('  ', r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
try:
    d = json.loads([l for l in open('$out/bench_$r.json') if l.startswith('{')][-1])
    print('   us/step', round(d['ms_per_step']*1e3,1), 'roofline frac', round(d['roofline']['frac'],3), d['roofline']['kernel'])
except Exception as e: print('no line', e)
P
  head -8 $out/prof_$r/p_kernel_stats.csv > $out/kernel_stats_$r.csv
done
