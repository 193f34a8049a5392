#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
for rep in 1 2; do
for cfg in "mp" "mlp"; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-scatter --no-pack-ms --readout $cfg > $out/b_${cfg}_$rep.json 2> $out/b_${cfg}_$rep.err
  python3 - $out/b_${cfg}_$rep.json "$cfg default rep $rep" <<'P'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[2], 'us/step', round(d['ms_per_step'] * 1e3, 2), [(k['kernel'][5:10], round(k['avg_launch_us'], 1)) for k in d.get('kernels', [])])
except Exception as e:
    print(sys.argv[2], 'no bench line', e)
P
done; done
timeout -k 10 900 python -m pytest tests/test_step.py tests/test_fused_gpu.py -q -m gpu > $out/gputest.log 2>&1; echo "pytest rc $?"; tail -4 $out/gputest.log
