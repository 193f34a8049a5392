#!/bin/bash
out=gpurun_out/r04p; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 800 python -m pytest tests/test_parallel_gpu.py -m gpu -q -x > $out/gputest.log 2>&1; echo "pytest rc $?"; tail -15 $out/gputest.log
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --exchange p2p --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline --no-scatter > $out/bench_p2p.json 2> $out/bench_p2p.err; echo "bench p2p rc $?"
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline --no-scatter > $out/bench_gloo.json 2> $out/bench_gloo.err; echo "bench gloo rc $?"
python - <<'P'
import json
for n in ('p2p','gloo'):
    try:
        d=json.loads([l for l in open('gpurun_out/r04p/bench_%s.json'%n) if l.startswith('{')][-1])
        print(n, round(d['value']/1e6,2), 'M q/s', round(d['ms_per_step']*1e3,2), 'us', d.get('exchange'), d.get('exchange_note'))
    except Exception as e: print(n, 'failed', e); print(open('gpurun_out/r04p/bench_%s.err'%n).read()[-1500:])
P
