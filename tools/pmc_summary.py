"""Summarise rocprofv3 --pmc passes into the JSON bench.py reads for `roofline.traffic`.

On the MI355X box (counters go in SEPARATE passes: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2;
never together with --sys-trace and friends):

    export TMPDIR=/tmp
    B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline"
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
        --output-format csv -d gpurun_out/pmc_sq -- $B
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- $B
    rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_write -- $B
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_tcc -- $B
    python tools/pmc_summary.py gpurun_out/pmc_sq gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_tcc \
        > gpurun_out/pmc.json

Per kernel (template arguments kept, parameter list dropped): average per launch of every counter;
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- on gfx950 FETCH_SIZE tallies 128-byte read requests at
64 bytes (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact for 16-byte-per-lane stores and float
atomics. Both sit on the L2's memory side: Infinity-Cache hits are included, so this is fabric traffic, an
upper bound on HBM traffic."""
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.strip('"')
    depth, out = 0, []
    for ch in name:                      # cut the parameter list: first '(' outside template brackets
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            break
        out.append(ch)
    s = ''.join(out).strip()
    return s[5:] if s.startswith('void ') else s


def main(dirs):
    acc = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r['Kernel_Name'])
                key = (k, r['Counter_Name'])
                disp = (f, r.get('Dispatch_Id', ''))
                a = acc.setdefault(key, {})
                a[disp] = a.get(disp, 0.0) + float(r['Counter_Value'])      # rows per XCD / instance add up
    out = {}
    for (k, c), per in sorted(acc.items()):
        out.setdefault(k, {})[c] = sum(per.values()) / max(len(per), 1)
        out[k]['launches_' + c] = len(per)
    for k, rec in out.items():
        if 'FETCH_SIZE' in rec and 'WRITE_SIZE' in rec:
            rec['hbm_bytes'] = int((2 * rec['FETCH_SIZE'] + rec['WRITE_SIZE']) * 1024)
        w = rec.get('SQ_WAVE_CYCLES')
        if w:
            for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
                if c in rec:
                    rec[c.lower() + '_frac'] = rec[c] / w
        if 'TCC_HIT_sum' in rec and 'TCC_MISS_sum' in rec and rec['TCC_HIT_sum'] + rec['TCC_MISS_sum'] > 0:
            rec['l2_hit_rate'] = rec['TCC_HIT_sum'] / (rec['TCC_HIT_sum'] + rec['TCC_MISS_sum'])
    out['_note'] = ('rocprofv3 --pmc, separate passes (SQ; FETCH_SIZE; WRITE_SIZE + GRBM; TCC hit/miss) of `bench.py '
                    '--steps 20 --warmup 3 --no-cpu-baseline`, averages per launch; FETCH_SIZE / WRITE_SIZE in KiB as '
                    'reported; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 rule, MI355X_MICROARCH.md HBM '
                    'section); L2 memory-side counters: Infinity-Cache hits included (fabric traffic, an upper bound '
                    'on HBM traffic).')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1:])
