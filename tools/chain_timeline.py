"""Timeline of the chain kernel's workgroups for one bench step (diagnostics).

    python tools/chain_timeline.py [--batch-size 512] [--embed-dim 128] [--readout mp] [--kg aifb]

Uses mpqe_debug_chain_stamps: every workgroup records the device wall clock (100 MHz) at its phase
boundaries and where it ran (XCC, SE, CU). Prints, per query type, the mean duration of each phase, and
the placement (workgroups per CU, busiest CUs)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument('--kg', default='aifb')
    ap.add_argument('--embed-dim', type=int, default=128)
    ap.add_argument('--batch-size', type=int, default=512)
    ap.add_argument('--readout', default='mp')
    ap.add_argument('--no-prune', action='store_true')
    ap.add_argument('--out', default='')
    ap.add_argument('--tail', action='store_true', help='time stamps of the weight-gradient launch instead')
    ap.add_argument('--merged', action='store_true', help='merged launch: chain workgroups and tiles on one clock')
    ap.add_argument('--merge-tail', action='store_true', help='force the merged launch')
    ap.add_argument('--trace', action='store_true', help='CHAIN_DBG=6 builds: per-item cycle stamps of block 0')
    ap.add_argument('--touch', default='step', choices=['step', 'pack'])
    ap.add_argument('--reps', type=int, default=1)
    ap.add_argument('--eight-waves', action='store_true')
    ap.add_argument('--no-ksplit', action='store_true')
    ap.add_argument('--debug-opt', action='append', default=[], metavar='NAME=VALUE')
    ap.add_argument("--no-self-check", action="store_true")
    args = ap.parse_args()
    from mpqe_amd import ops, synthetic
    from mpqe_amd.data_utils import make_feature_modules
    from mpqe_amd.encoders import DirectEncoder
    from mpqe_amd.fused import FusedTrainStep
    from mpqe_amd.model import RGCNEncoderDecoder
    for kv in args.debug_opt:
        name, _, val = kv.partition('=')
        ops.lib().mpqe_debug_option(name.encode(), int(val or 1), 1)
    torch.manual_seed(0)
    dev = torch.device('cuda:0')
    D = args.embed_dim
    schema = synthetic.make_schema(*synthetic.KG_SHAPES[args.kg], seed=0)
    graph = synthetic.SchemaGraph(schema, D)
    fm, node_maps = make_feature_modules(schema.ids, D, schema.num_entities)
    adaptive = args.readout == 'mp'
    model = RGCNEncoderDecoder(graph, DirectEncoder(None, fm, node_maps), readout=args.readout, num_layers=3,
                               shared_layers=False, adaptive=adaptive, weight_decay=0).to(dev)
    model.validate = False
    data = bench.StepData(schema, model, args.batch_size, np.random.RandomState(1000), dev)
    step = FusedTrainStep(model, prune=not args.no_prune, merge_tail=True if (args.merge_tail or args.merged) else None,
                          touch=args.touch, eight_waves=args.eight_waves, ksplit=not args.no_ksplit)
    packed = bench.pack_for_fused(step, data, resident=True)       # (ids in HBM, as in the timed loop of bench.py)
    assert step.uses_chain(packed), 'this step does not take the chain kernel'
    for _ in range(5):
        step.run(packed)
    torch.cuda.synchronize()
    if args.merged:
        # merged launch: chain workgroups and weight-gradient tiles on ONE clock (both stamp sets, one run)
        cap = 16 * sum((b + 15) // 16 for b in packed.sizes)
        stamps = torch.zeros(cap * 8, dtype=torch.int64, device=dev)
        tcap = 4096
        tst = torch.zeros(tcap * 8, dtype=torch.int64, device=dev)
        ops.lib().mpqe_debug_chain_stamps(stamps.data_ptr(), cap)
        ops.lib().mpqe_debug_tail_stamps(tst.data_ptr(), tcap)
        step.run(packed)
        torch.cuda.synchronize()
        ops.lib().mpqe_debug_chain_stamps(None, 0)
        ops.lib().mpqe_debug_tail_stamps(None, 0)
        st = stamps.cpu().numpy().reshape(cap, 8)
        tt = tst.cpu().numpy().reshape(tcap, 8)
        used = st[:, 6] != 0
        used[len(used) // 2:] = False                 # (second half of the buffer: shader-clock ticks)
        st = st[used]
        t0 = st[:, 0].min()
        batch = (st[:, 7] >> 40) & 0xff
        print('chain workgroups: %d, rows published (stamp 5) %.1f..%.1f us, end %.1f..%.1f us'
              % (len(st), (st[:, 5].min() - t0) * 0.01, (st[:, 5].max() - t0) * 0.01, (st[:, 6].min() - t0) * 0.01,
                 (st[:, 6].max() - t0) * 0.01))
        for b in np.unique(batch):
            m = batch == b
            print('   batch %2d: %3d workgroups, start %5.1f..%5.1f, rows out %5.1f..%5.1f, end %5.1f'
                  % (b, m.sum(), (st[m, 0].min() - t0) * 0.01, (st[m, 0].max() - t0) * 0.01, (st[m, 5].min() - t0) * 0.01,
                     (st[m, 5].max() - t0) * 0.01, (st[m, 6].max() - t0) * 0.01))
        tt = tt[tt[:, 0] != 0]
        if len(tt):
            us = lambda c: (tt[:, c] - t0) * 0.01
            print('tiles: %d; dispatched %.1f..%.1f us, rows ready (waited) %.1f..%.1f, end %.1f..%.1f'
                  % (len(tt), us(0).min(), us(0).max(), us(4).min(), us(4).max(), us(1).min(), us(1).max()))
            order = np.argsort(us(1))
            print('   [tile, dispatched, ready, first rows, K loop done, end]')
            for k in list(order[:6]) + list(order[len(order) // 2 - 3: len(order) // 2 + 3]) + list(order[-12:]):
                print('   %4d  %6.1f %6.1f %6.1f %6.1f %6.1f' % (k, us(0)[k], us(4)[k], us(5)[k], us(6)[k], us(1)[k]))
            print('   run time after ready: mean %.1f min %.1f max %.1f us' % ((us(1) - us(4)).mean(), (us(1) - us(4)).min(),
                                                                               (us(1) - us(4)).max()))
        return
    if args.tail:
        tcap = 4096
        tst = torch.zeros(tcap * 8, dtype=torch.int64, device=dev)
        ops.lib().mpqe_debug_tail_stamps(tst.data_ptr(), tcap)
        step.run(packed)
        torch.cuda.synchronize()
        ops.lib().mpqe_debug_tail_stamps(None, 0)
        tt = tst.cpu().numpy().reshape(tcap, 8)
        tt = tt[tt[:, 0] != 0]
        t0 = tt[:, 0].min()
        start, end = (tt[:, 0] - t0) * 0.01, (tt[:, 1] - t0) * 0.01
        gw = tt[:, 1] != 0
        hw, xcc = tt[:, 3] & 0xffffffff, (tt[:, 3] >> 32) & 0xf
        place = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xf)
        dur = end[gw] - start[gw]
        print('weight-gradient launch: %d workgroups (%d tiles, %d zero-fill), tiles: start %.1f..%.1f us, duration mean '
              '%.1f min %.1f max %.1f, last end %.1f us' % (len(tt), gw.sum(), (~gw).sum(), start[gw].min(),
                                                            start[gw].max(), dur.mean(), dur.min(), dur.max(), end[gw].max()))
        mhz = tt[gw, 2] / np.maximum(dur, 1e-3)
        print('   shader clock while the tiles ran: mean %.0f MHz (min %.0f, max %.0f)' % (mhz.mean(), mhz.min(), mhz.max()))
        ph = (tt[gw][:, [4, 5, 6, 1]] - tt[gw][:, [0, 4, 5, 6]]) * 0.01
        print('   tile phases, mean us: record %.1f, first K-step landed %.1f, K loop %.1f, stores %.1f'
              % tuple(ph.mean(axis=0)))
        order = np.argsort(-end[gw])[:8]
        for k in order:
            i = np.nonzero(gw)[0][k]
            print('   block %4d  start %5.1f  dur %5.1f  end %5.1f  on-CU mates %d' % (i, start[i], end[i] - start[i], end[i],
                                                                                     int((place[gw] == place[i]).sum())))
        uo = (~gw) & (tt[:, 6] >= 1) & (tt[:, 6] <= 7) & (tt[:, 5] != 0)      # vector ops of the backward post-pass
        if uo.any():
            uend = (tt[uo, 5] - t0) * 0.01
            kinds = tt[uo, 6] - 1
            print('   post-pass vector ops: %d workgroups, start %.1f..%.1f us, end %.1f..%.1f us' %
                  (uo.sum(), start[uo].min(), start[uo].max(), uend.min(), uend.max()))
            for kd, nm in ((2, "RED"), (1, "BWD"), (4, "R1"), (5, "CLOSURE"), (6, "ROWS")):
                m = kinds == kd
                if m.any():
                    print('      %s: %d, ends %s' % (nm, m.sum(), np.round(np.sort(uend[m])[::max(1, m.sum() // 12)], 1).tolist()))
                if kd == 6 and m.any():
                    rows = tt[uo][m]
                    st_, en_ = (rows[:, 0] - t0) * 0.01, (rows[:, 5] - t0) * 0.01
                    print('         table rows / loss: start %.1f .. %.1f, duration mean %.1f max %.1f' % (st_.min(), st_.max(), (en_ - st_).mean(), (en_ - st_).max()))
                if kd == 5 and m.any():      # closures: start | programme in LDS | pre phase done | item stream done | end
                    rows = tt[uo][m]
                    for rr in rows[np.argsort(rows[:, 5])]:
                        print('         closure: start %.1f, block %.1f, pre %.1f, items %.1f, end %.1f' %
                              tuple((rr[k] - t0) * 0.01 for k in (0, 2, 3, 4, 5)))
        hist = np.histogram(dur, bins=[0, 4, 6, 8, 10, 12, 14, 16, 20, 30])
        print('   duration histogram (us):', list(zip(hist[1][:-1].tolist(), hist[0].tolist())))
        return
    cap = 16 * sum((b + 15) // 16 for b in packed.sizes)     # the launch grid has holes (placement by XCD)
    stamps = torch.zeros(cap * 8, dtype=torch.int64, device=dev)
    ops.lib().mpqe_debug_chain_stamps(stamps.data_ptr(), cap)
    import ctypes
    worst = None
    for rep in range(args.reps):        # several stamped runs: [launch duration by events | chain makespan | sort end] per run
        stamps.zero_()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for e in evs:
            e.record()
        step.run(packed, events=(ctypes.c_void_p * 4)(*[e.cuda_event for e in evs]))
        torch.cuda.synchronize()
        r = stamps.cpu().numpy()
        s8 = r.reshape(cap, 8)
        nre = sum((b + 15) // 16 for b in packed.sizes)
        g = (nre + 7) // 8 * 8
        while int((s8[:g, 6] != 0).sum()) < nre or bool((s8[g:2 * g, 6] != 0).any()):
            g += 8
        m = s8[:g][s8[:g, 6] != 0]
        ns = (packed.touch_entries + 1023) // 1024 if packed.step_flags else 0
        so = r[g * 16: g * 16 + 8 * ns].reshape(ns, 8)
        print('run %d: chain launch %.1f us (events), tail %.1f; chain workgroups %.1f .. %.1f us; sort %s'
              % (rep, evs[0].elapsed_time(evs[1]) * 1e3, evs[2].elapsed_time(evs[3]) * 1e3, 0.0, (m[:, 6].max() - m[:, 0].min()) * 0.01,
                 ('%.1f .. %.1f' % ((so[:, 0].min() - m[:, 0].min()) * 0.01, (so[:, 7].max() - m[:, 0].min()) * 0.01)) if ns else '-'))
        span = m[:, 6].max() - m[:, 0].min()
        if worst is None or span > worst[0]:
            worst = (span, r.copy())
    ops.lib().mpqe_debug_chain_stamps(None, 0)
    raw = worst[1]                                  # the details below: the slowest of the runs
    st = raw.reshape(cap, 8)
    nreal = sum((b + 15) // 16 for b in packed.sizes)
    grid = (nreal + 7) // 8 * 8                     # the launch grid: 8 x (longest XCD list), holes included
    while int((st[:grid, 6] != 0).sum()) < nreal or bool((st[grid:2 * grid, 6] != 0).any()):
        grid += 8
    trace_words = raw[grid * 16: grid * 16 + 8192].copy()
    nsort = (packed.touch_entries + 1023) // 1024 if packed.step_flags else 0
    sort_st = raw[grid * 16: grid * 16 + 8 * nsort].reshape(nsort, 8).copy()     # (the in-step sort's workgroups: behind the chain entries)
    st[2 * grid:] = 0                               # (a CHAIN_DBG=6 build keeps its trace behind the stamps)
    used = np.nonzero(st[:, 6] != 0)[0]
    if args.trace:
        tr = trace_words.reshape(2, 2, 2048)                                 # [tracing wave][fwd / bwd][entry]
        names = {0: 'item start', 1: 'mfma done', 2: 'handoff done', 3: 'epilogue done', 4: 'partials added',
                 5: 'transformed', 6: 'tile stored', 7: 'rows stored'}
        for w in range(2):
            for d, dn in enumerate(('fwd', 'bwd')):
                v = tr[w, d]
                v = v[v != 0]
                if not len(v):
                    continue
                t, tag = v >> 3, v & 7
                print('wave %d (%s K part) %s: %d stamps, %d cycles total' % (2 * w, 'first' if w == 0 else 'last', dn,
                                                                              len(v), t[-1] - t[0]))
                print('    sequence (tag:cycles since the previous stamp): '
                      + ' '.join('%d:%d' % (int(tag[k]), int(t[k] - t[k - 1])) for k in range(1, len(v))))
                seg = {}
                for k in range(1, len(v)):
                    key = '%s -> %s' % (names[int(tag[k - 1])], names[int(tag[k])])
                    seg.setdefault(key, []).append(int(t[k] - t[k - 1]))
                for key, vals in seg.items():
                    print('    %-32s n=%3d  mean %6.0f  min %5d  max %5d  sum %7d' % (key, len(vals), np.mean(vals),
                                                                                  min(vals), max(vals), sum(vals)))
    ticks = st[grid:2 * grid]
    sel = used[used < grid]
    mhz = (ticks[sel, 1] - ticks[sel, 0]) / ((st[sel, 6] - st[sel, 0]) * 0.01)
    print('shader clock while the workgroups ran: mean %.0f MHz (min %.0f, max %.0f)' % (mhz.mean(), mhz.min(), mhz.max()))
    st = st[sel]
    nblk = st.shape[0]
    t = (st[:, :7] - st[:, 0].min()) * 0.01              # us
    hw = st[:, 7] & 0xffffffff
    xcc = (st[:, 7] >> 32) & 0xf
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    place = xcc * 1000 + se * 100 + sh * 16 + cu
    names = ['A1 ids', 'A2 gather', 'forward', 'score', 'backward', 'anchors']
    print('workgroups %d, makespan %.1f us (first start %.1f, last start %.1f)'
          % (nblk, t[:, 6].max(), t[:, 0].min(), t[:, 0].max()))
    if nsort and sort_st[:, 7].all():
        ts = (sort_st - st[:, 0].min()) * 0.01
        n_st = int((sort_st[0, :7] != 0).sum())
        print('touch-plan sort: %d workgroups; start %.1f..%.1f us, keys done %.1f..%.1f, after grid barriers %s, end %.1f..%.1f'
              % (nsort, ts[:, 0].min(), ts[:, 0].max(), ts[:, 1].min(), ts[:, 1].max(),
                 ' | '.join('%.1f..%.1f' % (ts[:, k].min(), ts[:, k].max()) for k in range(2, n_st)), ts[:, 7].min(), ts[:, 7].max()))
    uniq, cnt = np.unique(place, return_counts=True)
    print('distinct CUs used %d; workgroups per CU: %s' % (len(uniq), dict(zip(*np.unique(cnt, return_counts=True)))))
    # which batch a block belongs to cannot be read back from the library; durations by total time instead
    dur = t[:, 6] - t[:, 0]
    order = np.argsort(-dur)
    print('phase means over all workgroups (us): ' + ', '.join(
        '%s %.1f' % (n, (t[:, k + 1] - t[:, k]).mean()) for k, n in enumerate(names)))
    print('10 longest workgroups: [block, start, A1, A2, fwd, score, bwd, anchors, total, cu-mates]')
    for bidx in order[:10]:
        mates = int((place == place[bidx]).sum())
        print('  %4d  start %5.1f  ' % (bidx, t[bidx, 0]) + ' '.join('%5.1f' % (t[bidx, k + 1] - t[bidx, k]) for k in range(6))
              + '  total %5.1f  mates %d' % (dur[bidx], mates))
    print('10 shortest:')
    for bidx in order[-10:]:
        mates = int((place == place[bidx]).sum())
        print('  %4d  start %5.1f  ' % (bidx, t[bidx, 0]) + ' '.join('%5.1f' % (t[bidx, k + 1] - t[bidx, k]) for k in range(6))
              + '  total %5.1f  mates %d' % (dur[bidx], mates))
    print('workgroups per XCD: %s' % dict(zip(*np.unique(xcc, return_counts=True))))
    batch, fops = (st[:, 7] >> 40) & 0xff, (st[:, 7] >> 48) & 0xff
    print('per batch (library order): [batch, forward K-blocks, blocks, XCDs, mean of A1, A2, fwd, score, bwd, anchors, total; max end]')
    for bi in np.unique(batch):
        m = batch == bi
        print('  %3d  kblocks %2d  blocks %3d  xcc %s  ' % (bi, int(fops[m][0]), int(m.sum()), sorted(set(xcc[m].tolist())))
              + ' '.join('%5.1f' % (t[m, k + 1] - t[m, k]).mean() for k in range(6))
              + '  total %5.1f  max end %5.1f  mates %.1f' % (dur[m].mean(), t[m, 6].max(),
                                                              np.mean([(place == pl).sum() for pl in place[m]])))
    if args.out:
        json.dump(dict(t=t.tolist(), place=place.tolist()), open(args.out, 'w'))


if __name__ == '__main__':
    main()
