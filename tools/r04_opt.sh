#!/bin/bash
# A/B of a debug option on one box:  tools/r04_opt.sh <tag> <OPTION> "<readouts>"
out=gpurun_out/$1; mkdir -p $out; opt=$2
export TMPDIR=/tmp; VALS=(${V0:-0} ${V1:-1})
for rep in ${REPS:-1 2}; do
for cfg in ${3:-mp mlp}; do
for v in 0 1; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-scatter --no-pack-ms --readout $cfg --debug-opt $opt=${VALS[$v]:-$v} > $out/b_${cfg}_$v$rep.json 2> $out/b_${cfg}_$v$rep.err
  python3 - $out/b_${cfg}_$v$rep.json "$cfg $opt=$v rep $rep" <<'P'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[2], 'us/step', round(d['ms_per_step'] * 1e3, 2), [(k['kernel'][5:10], round(k['avg_launch_us'], 1)) for k in d.get('kernels', [])])
except Exception as e:
    print(sys.argv[2], 'no bench line', e)
P
done; done; done
