"""dev tool: forward / backward split of a training step from a rocprofv3 --kernel-trace rocpd database — wall time of each
phase, busy time per stream inside it, time with 0 / 1 / 2 / 3 kernels in flight, idle time of the first stream.
usage: python tools/phase_timeline.py results.db"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute('select start, end, name, stream_id from kernels order by start').fetchall()
    ends = [e for s, e, n, q in rows if 'sgd_multi_kernel' in n]
    nsteps = min(5, len(ends) - 1)
    tot = {'fwd': [0.0, {}, {}], 'bwd': [0.0, {}, {}]}
    for k in range(nsteps):
        lo, hi = ends[-2 - k], ends[-1 - k]
        step = [r for r in rows if r[0] >= lo and r[1] <= hi]
        fb = next(r[0] for r in step if 'dice_bwd' in r[2] or 'bce_bwd' in r[2] or 'ce_bwd' in r[2])
        for name, a, b in (('fwd', lo, fb), ('bwd', fb, hi)):
            t = tot[name]
            t[0] += (b - a) / 1e6
            ev = []
            for s, e, n, q in step:
                s2, e2 = max(s, a), min(e, b)
                if e2 > s2:
                    t[1][q] = t[1].get(q, 0.0) + (e2 - s2) / 1e6
                    ev.append((s2, 1))
                    ev.append((e2, -1))
            ev.sort()
            depth, last = 0, a
            for tt, d in ev:
                t[2][depth] = t[2].get(depth, 0.0) + (tt - last) / 1e6
                depth += d
                last = tt
            t[2][0] = t[2].get(0, 0.0) + (b - last) / 1e6
    for name in ('fwd', 'bwd'):
        t = tot[name]
        print(f'{name}: {t[0] / nsteps:.3f} ms per step; busy per stream ' +
              ', '.join(f'{q}: {v / nsteps:.3f}' for q, v in sorted(t[1].items())) +
              '; in flight ' + ', '.join(f'{d}: {v / nsteps:.3f}' for d, v in sorted(t[2].items())))


if __name__ == '__main__':
    main(sys.argv[1])
