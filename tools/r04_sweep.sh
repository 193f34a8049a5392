#!/bin/bash
# learned readouts on the chain form at other sizes: the fused step against the module path (max_rel_grad_diff) + timing
#   tools/r04_sweep.sh <tag> ["B:D B:D ..."]
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
for cfg in ${2:-64:128 128:128 2048:128 512:256 512:64 96:256}; do
  B=${cfg%%:*}; D=${cfg##*:}
  timeout -k 10 300 python tools/readout_step_bench.py --readouts mlp,targetmlp,concat --batch-size $B --embed-dim $D --steps 20 --warmup 5 > $out/s_${B}_$D.jsonl 2> $out/s_${B}_$D.err || { echo "FAILED B=$B D=$D"; tail -5 $out/s_${B}_$D.err; exit 1; }
  python3 - $out/s_${B}_$D.jsonl "B=$B D=$D" <<'P'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(sys.argv[2], d['readout'], 'fused ms', d['fused_fresh_ids_ms'], 'module ms', d['module_path_ms'], 'max rel grad diff %.2e' % d['max_rel_grad_diff'], 'loss', round(d['loss_fused'], 6), round(d['loss_module'], 6))
P
done
