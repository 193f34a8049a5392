summ() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,2), round(d['replay']['ms_per_step']*1e3,2), [(k['kernel'],round(k['avg_launch_us'],1)) for k in d['kernels']], {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['pack_ms'].items() if v})"; }
timeout -k 10 900 python -m pytest tests/test_configs_gpu.py tests/test_parallel_gpu.py -m gpu -x -q --durations=8 2>&1 | tail -18
timeout -k 10 200 python bench.py --no-cpu-baseline --no-scatter --steps 20 --warmup 5 2>/dev/null | summ
