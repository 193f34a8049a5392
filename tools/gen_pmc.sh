#!/bin/bash
# PMC passes (separate runs, never with a trace domain) over the general-graph path at the stress shape:
#   gpurun -- ./tools/gen_pmc.sh <outdir>
out=$(realpath -m $1); mkdir -p $out
root=$(pwd)
export TMPDIR=/tmp
B="python3 $root/tools/scatter_bench.py"
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/pmc_sq -- $B > $out/pmc_sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B > $out/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_write -- $B > $out/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_tcc -- $B > $out/pmc_tcc.log 2>&1
cd $root
python3 tools/pmc_summary.py $out/pmc_sq $out/pmc_fetch $out/pmc_write $out/pmc_tcc > $out/pmc.json 2> $out/pmc.err
rm -rf $out/pmc_sq $out/pmc_fetch $out/pmc_write $out/pmc_tcc
python3 - <<P
import json
p=json.load(open('$out/pmc.json'))
for k,v in p.items():
    if 'rgcn_gen' in k or 'segment' in k: print(k[:46], {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ('hbm_bytes','l2_hit_rate','SQ_VALU_MFMA_BUSY_CYCLES','sq_wait_any_frac','GRBM_GUI_ACTIVE')})
P
