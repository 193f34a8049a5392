#!/bin/bash
# learned readouts: GPU tests, then the readout step bench and a kernel profile of bench.py --readout mlp
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_step.py tests/test_fused_gpu.py -q -m gpu -k "readout or learned" > $out/gputest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -15 $out/gputest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/readout_step_bench.py --readouts ${2:-mlp,targetmlp,concat} > $out/readout_bench.jsonl 2> $out/readout_bench.err; echo "bench rc $?"; cat $out/readout_bench.jsonl; tail -3 $out/readout_bench.err
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof -o ro -- python3 /root/repo/bench.py --readout mlp --steps 200 --warmup 20 --no-cpu-baseline > /root/repo/$out/bench_mlp.json 2> /root/repo/$out/bench_mlp.err; echo "prof rc $?"
cd /root/repo; tail -2 $out/bench_mlp.json | cut -c1-1500
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
