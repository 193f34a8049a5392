#!/bin/bash
# A/B of the launch order (prologue in front of / behind the chain workgroups) on one box: debug option PROLOGUE_LAST
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_step.py -q -m gpu -k "mlp_readout_on_the_chain or chain_edge" > $out/gputest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 $out/gputest.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
for cfg in "mp" "mlp"; do
for pl in 0 1; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-scatter --no-pack-ms --readout $cfg --debug-opt PROLOGUE_LAST=$pl > $out/b_${cfg}_$pl$rep.json 2> $out/b_${cfg}_$pl$rep.err
  python3 - $out/b_${cfg}_$pl$rep.json "$cfg plast=$pl rep $rep" <<'P'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[2], 'us/step', round(d['ms_per_step'] * 1e3, 2), [(k['kernel'][5:10], round(k['avg_launch_us'], 1)) for k in d.get('kernels', [])])
except Exception as e:
    print(sys.argv[2], 'no bench line', e)
P
done; done; done
timeout -k 10 200 python tools/chain_timeline.py --readout mlp --debug-opt PROLOGUE_LAST=1 > $out/tl_mlp_pl1.txt 2>&1; grep -h "makespan\|distinct CUs" $out/tl_mlp_pl1.txt
