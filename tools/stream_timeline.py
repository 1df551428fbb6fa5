"""dev tool: how the two streams of a training step share the chip.  From a rocprofv3 --kernel-trace rocpd database: per
queue / stream busy time, the union, the time both are busy, and per kernel family the duration inside vs outside overlap.
usage: python tools/stream_timeline.py results.db"""
import bisect
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info('kernels')")]
    key = 'stream_id' if 'stream_id' in cols else ('queue_id' if 'queue_id' in cols else None)
    print('columns:', cols)
    rows = db.execute(f'select start, end, name, {key} from kernels order by start').fetchall()
    # whole steps only: the window between the optimizer launches that end the 6th-last and the last step
    ends = [e for s, e, n, q in rows if 'sgd_multi_kernel' in n]
    nsteps = min(5, len(ends) - 1)
    lo, hi = ends[-1 - nsteps], ends[-1]
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    print(f'{nsteps} steps, {(hi - lo) / 1e6 / nsteps:.2f} ms per step')
    span = rows[-1][1] - rows[0][0]
    by = {}
    for s, e, n, q in rows:
        by.setdefault(q, []).append((s, e, n))
    print(f'window {span / 1e6:.1f} ms, {len(rows)} kernels')
    for q, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
        print(f'  {key} {q}: {len(v)} kernels, busy {sum(e - s for s, e, _ in v) / 1e6:.1f} ms')
    ev = []
    for s, e, n, q in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, hist = 0, ev[0][0], {}
    for t, d in ev:
        hist[depth] = hist.get(depth, 0) + (t - last)
        depth += d
        last = t
    for k in sorted(hist):
        print(f'  {k} kernel(s) in flight: {hist[k] / 1e6:.1f} ms ({100.0 * hist[k] / span:.1f} %)')
    side = max(by, key=lambda q: (sum('wgrad' in n for _, _, n in by[q]) / (1.0 + len(by[q]))) if len(by[q]) > 50 else 0)
    iv = sorted((s, e) for s, e, _ in by[side])
    starts = [a for a, _ in iv]

    def overlapped(s, e):
        i = max(0, bisect.bisect_left(starts, s) - 1)
        tot = 0
        while i < len(iv) and iv[i][0] < e:
            tot += max(0, min(e, iv[i][1]) - max(s, iv[i][0]))
            i += 1
        return tot
    fam = {}
    for q, v in by.items():
        if q == side:
            continue
        for s, e, n in v:
            f = n.split('<')[0].split('(')[0].replace('void ', '').replace('evk::', '')
            o = overlapped(s, e)
            a = fam.setdefault(f, [0, 0, 0])
            a[0] += e - s
            a[1] += o
            a[2] += 1
    print(f'side stream = {key} {side}; main-stream families: launches, total ms, of which beside a side-stream kernel')
    for f, (tot, o, c) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f'  {f:44s} {c:6d} {tot / 1e6:8.2f} {o / 1e6:8.2f} ({100.0 * o / tot:.0f} %)')


if __name__ == '__main__':
    main(*sys.argv[1:])
