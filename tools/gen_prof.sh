#!/bin/bash
# Per-kernel durations of the general-graph path at the stress shape (rocprofv3 --kernel-trace --stats):
#   gpurun -- ./tools/gen_prof.sh <outdir> [ENV=VALUE ...]
out=$(realpath -m $1); shift
mkdir -p $out
root=$(pwd)
export TMPDIR=/tmp
for kv in "$@"; do export $kv; done
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 $root/tools/scatter_bench.py > $out/work.json 2> $out/err.log
cd $root
python3 - <<P
import csv
for r in csv.DictReader(open('$out/prof/p_kernel_stats.csv')):
    if any(k in r['Name'] for k in ('rgcn_gen', 'segment_sum', 'bias_')): print('  ', r['Name'][:50], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
P
