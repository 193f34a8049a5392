#!/bin/bash
# Same-box A/B of builds of the library (box-to-box differences are larger than most kernel changes). Variants live at
# mpqe_amd/lib/alt/lib<name>.so and are selected through MPQE_AMD_LIB (mpqe_amd/_lib.py): the installed library is never
# overwritten, so an interrupted run leaves nothing behind.  `cur` = the installed library.
#   gpurun -- ./tools/ab.sh <outdir> <mode> "<bench flags>" <name> <name> ...
# modes: bench (bench.py line), prof (rocprofv3 --kernel-trace --stats of bench.py), timeline (chain + tail stamps),
#        tests (pytest tests/test_step.py -m gpu first; wrong-on-purpose builds must not use it)
out=$(realpath -m $1); shift
mode=$1; shift
flags=$1; shift
mkdir -p $out
root=$(pwd)
export TMPDIR=/tmp
lib_of() { if [ "$1" = cur ]; then echo ""; else echo "$root/mpqe_amd/lib/alt/lib$1.so"; fi; }
summ() { python3 - "$1" "$2" <<'P'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[2], 'M q/s', round(d['value'] / 1e6, 2), 'us/step', round(d['ms_per_step'] * 1e3, 2),
          'replay', round(d.get('replay', {}).get('ms_per_step', 0) * 1e3, 2),
          [(k['kernel'][5:10], round(k['avg_launch_us'], 1)) for k in d.get('kernels', [])])
except Exception as e:
    print(sys.argv[2], 'no bench line', e)
P
}
for rep in ${REPS:-1 2}; do
for v in "$@"; do
  export MPQE_AMD_LIB=$(lib_of $v)
  [ -z "$MPQE_AMD_LIB" ] && unset MPQE_AMD_LIB
  case $mode in
    tests)
      [ $rep = 1 ] && { timeout -k 10 400 python3 -m pytest tests/test_step.py -m gpu -x -q > $out/t_$v.log 2>&1 || { echo "TESTS FAILED for $v"; tail -5 $out/t_$v.log; exit 1; }; }
      timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-scatter $flags > $out/b_$v$rep.json 2> $out/b_$v$rep.err; summ $out/b_$v$rep.json "$v $rep";;
    bench)
      timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-scatter $flags > $out/b_$v$rep.json 2> $out/b_$v$rep.err; summ $out/b_$v$rep.json "$v $rep";;
    timeline)
      timeout -k 10 200 python3 tools/chain_timeline.py > $out/tl_$v$rep.txt 2>&1
      timeout -k 10 200 python3 tools/chain_timeline.py --tail >> $out/tl_$v$rep.txt 2>&1
      echo "== $v $rep"; grep -h "makespan\|BWD:\|shader clock" $out/tl_$v$rep.txt | head -8
      timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-scatter $flags > $out/b_$v$rep.json 2> $out/b_$v$rep.err; summ $out/b_$v$rep.json "$v $rep";;
    prof)
      [ $rep = 1 ] || continue
      (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$v -o p -- python3 $root/bench.py --no-cpu-baseline --no-scatter --steps 50 --repeats 3 $flags > $out/prof_$v.log 2>&1)
      summ $out/prof_$v.log "$v"
      python3 - <<P
import csv
for r in csv.DictReader(open('$out/prof_$v/p_kernel_stats.csv')):
    if 'step_' in r['Name'] and 'upload' not in r['Name']: print('  ', r['Name'][:34], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
P
      ;;
  esac
done
done
