#!/bin/bash
# round 4 GPU call: [tests] + A/B of debug options on one box + tail timelines
#   tools/r04_run.sh <tag> <tests: all|step|none> "<opt set 1>" "<opt set 2>" ...   (opt set: space-separated NAME=VALUE, or "-")
tag=$1; shift; tests=$1; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
case $tests in
  all) timeout -k 10 1100 python -m pytest tests -m gpu -q > $out/gputest.log 2>&1; echo "pytest rc $?"; tail -4 $out/gputest.log;;
  step) timeout -k 10 600 python -m pytest tests/test_step.py tests/test_fused_gpu.py tests/test_configs_gpu.py -m gpu -q -x > $out/gputest.log 2>&1; echo "pytest rc $?"; tail -4 $out/gputest.log;;
esac
i=0
for opts in "$@"; do
  i=$((i+1)); flags=""
  [ "$opts" != "-" ] && for o in $opts; do flags="$flags --debug-opt $o"; done
  for rep in 1 2; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-scatter --no-pack-ms $flags > $out/bench_$i$rep.json 2> $out/bench_$i$rep.err
  done
  timeout -k 10 200 python tools/chain_timeline.py --tail $flags > $out/tl_$i.txt 2>&1
done
python - "$out" "$@" <<'P'
import json, sys, glob
out = sys.argv[1]
for i, opts in enumerate(sys.argv[2:], 1):
    for rep in (1, 2):
        try:
            d = json.loads([l for l in open('%s/bench_%d%d.json' % (out, i, rep)) if l.startswith('{')][-1])
            print('[%s] rep %d: %.2f M q/s  %.2f us' % (opts, rep, d['value'] / 1e6, d['ms_per_step'] * 1e3),
                  [(k['kernel'][5:10], round(k['avg_launch_us'], 1)) for k in d.get('kernels', [])])
        except Exception as e:
            print('[%s] rep %d failed: %s' % (opts, rep, e))
P
