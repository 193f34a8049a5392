#!/bin/bash
# rocprofv3 summaries of the fused step for ONE bench configuration (run on the MI355X box through gpurun):
#   gpurun -- ./tools/profile_config.sh gpurun_out/<tag> [bench.py flags: --kg mutag --embed-dim 256 --readout sum ...]
# -> <tag>/kernel_stats.csv (--kernel-trace --stats), <tag>/pmc.json (four --pmc passes, each a run of its own, never combined
# with a trace domain: tools/pmc_summary.py applies the guide's unit and gfx950 corrections), <tag>/bench.json (the line,
# without the CPU baseline / scatter / drop-in records). The default workload's full set (timelines, CPU baseline):
# tools/collect_profiles.sh.
out=$(realpath -m $1); shift
mkdir -p $out
root=$(pwd)
export TMPDIR=/tmp
B="python3 $root/bench.py --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline --no-scatter --no-pack-ms --no-dropin-loop $@"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- $B > $out/stats.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/pmc_sq -- $B > $out/pmc_sq.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B > $out/pmc_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_write -- $B > $out/pmc_write.log 2>&1
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_tcc -- $B > $out/pmc_tcc.log 2>&1
cd $root
python3 tools/pmc_summary.py $out/pmc_sq $out/pmc_fetch $out/pmc_write $out/pmc_tcc > $out/pmc.json 2> $out/pmc.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-scatter --no-pack-ms --no-dropin-loop "$@" > $out/bench.json 2> $out/bench.err
cp $out/stats/s_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
rm -rf $out/pmc_sq $out/pmc_fetch $out/pmc_write $out/pmc_tcc $out/stats
ls $out
