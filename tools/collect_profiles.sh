#!/bin/bash
# Collects the rocprofv3 summaries committed under profiles/ for one milestone (run on the MI355X box through gpurun):
#   gpurun -- ./tools/collect_profiles.sh gpurun_out/<tag>
# kernel-trace stats, the four --pmc passes of tools/pmc_summary.py (each in a run of its own, never combined with a
# trace domain), the chain / weight-gradient timelines, and the bench line (with the CPU baseline).
out=$(realpath -m $1); mkdir -p $out
root=$(pwd)
export TMPDIR=/tmp
B="python3 $root/bench.py --steps 20 --warmup 3 --repeats 10 --no-cpu-baseline --no-scatter --no-pack-ms --no-dropin-loop"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- $B > $out/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/pmc_sq -- $B > $out/pmc_sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B > $out/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_write -- $B > $out/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_tcc -- $B > $out/pmc_tcc.log 2>&1
cd $root
python3 tools/pmc_summary.py $out/pmc_sq $out/pmc_fetch $out/pmc_write $out/pmc_tcc > $out/pmc.json 2> $out/pmc.err
timeout -k 10 200 python3 tools/chain_timeline.py > $out/timeline.txt 2>&1
timeout -k 10 200 python3 tools/chain_timeline.py --tail >> $out/timeline.txt 2>&1
timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err
cp $out/stats/s_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
rm -rf $out/pmc_sq $out/pmc_fetch $out/pmc_write $out/pmc_tcc $out/stats/*trace* 
ls $out
