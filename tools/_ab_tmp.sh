timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
./tools/gen_prof.sh gpurun_out/r3_gen 2>&1 | tail -12
python - <<P
import json
d=json.load(open('gpurun_out/r3_gen/work.json'))
for k,v in d.items(): print(k, round(v['ms_fwd_bwd_median'],3), 'ms fwd+bwd')
P
