#!/bin/bash
# round 4, first GPU call: the whole -m gpu suite on the round's host-side changes, the bench line, TILE_N = 32 A/B + timelines
out=gpurun_out/r04a; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/gputest.log 2>&1; echo "pytest rc $?"; tail -3 $out/gputest.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-scatter > $out/bench_cur.json 2> $out/bench_cur.err; echo "bench rc $?"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-scatter --no-pack-ms --debug-opt TILE_N=32 > $out/bench_t32.json 2> $out/bench_t32.err
timeout -k 10 200 python tools/chain_timeline.py --tail > $out/tl_cur.txt 2>&1
timeout -k 10 200 python tools/chain_timeline.py --tail --debug-opt TILE_N=32 > $out/tl_t32.txt 2>&1
python - <<'P'
import json
for n in ('cur','t32'):
    try:
        d=json.loads([l for l in open('gpurun_out/r04a/bench_%s.json'%n) if l.startswith('{')][-1])
        print(n, round(d['value']/1e6,2), 'M q/s', round(d['ms_per_step']*1e3,2), 'us', [(k['kernel'][5:10], round(k['avg_launch_us'],1)) for k in d.get('kernels',[])])
    except Exception as e: print(n, 'failed', e)
P
