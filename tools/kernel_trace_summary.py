"""Per-kernel totals, or a timeline window, of a `rocprofv3 --kernel-trace` run (its results .db), per iteration of a loop.

    cd /tmp && rocprofv3 --kernel-trace -d OUT -o x -- python3 <repo>/tools/dropin_loop_bench.py --readout mlp --iters 300 --module-iters 0
    python tools/kernel_trace_summary.py OUT 640                 # launches, average us and us per iteration, per kernel
    python tools/kernel_trace_summary.py OUT --window 0.3 75     # 75 launches from 30 % into the trace: start, end, queue, grid

The window shows which launches of different streams (hardware queues) overlap: how the drop-in's side streams were checked
(DESIGN.md 1a)."""
import glob
import sqlite3
import sys


def tables(c):
    names = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    return ([t for t in names if 'kernel_dispatch' in t][0], [t for t in names if 'kernel_symbol' in t][0])


def main():
    db = glob.glob(sys.argv[1] + '/**/*.db', recursive=True) or glob.glob(sys.argv[1] + '/*.db')
    c = sqlite3.connect(db[0])
    kd, ks = tables(c)
    if len(sys.argv) > 2 and sys.argv[2] == '--window':
        frac, count = float(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 75
        rows = c.execute("select s.kernel_name, d.start, d.end, d.queue_id, d.grid_size_x from %s d join %s s on d.kernel_id = s.id "
                         "order by d.start" % (kd, ks)).fetchall()
        win = rows[int(len(rows) * frac): int(len(rows) * frac) + count]
        t0 = win[0][1]
        for name, st, en, q, g in win:
            print('%9.1f %9.1f  q%-3d grid %-8d %s' % ((st - t0) / 1e3, (en - t0) / 1e3, q, g, name.split('(')[0][:60]))
        return
    iters = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rows = c.execute("select s.kernel_name, count(*), avg(d.end - d.start), sum(d.end - d.start) from %s d join %s s on "
                     "d.kernel_id = s.id group by s.kernel_name order by 4 desc" % (kd, ks)).fetchall()
    total = sum(r[3] for r in rows)
    for name, n, avg, tot in rows[:20]:
        print('%-72s %8.1f per iteration %9.2f us  %8.1f us per iteration' % (name[:72], n / iters, avg / 1e3, tot / 1e3 / iters))
    print('GPU busy per iteration: %.1f us' % (total / 1e3 / iters))


if __name__ == '__main__':
    main()
