"""dev tool: HBM bytes per kernel CLASS (name + grid size) from the two PMC passes of tools/profile_round.sh — where the family
totals of profiles/rNN_traffic.json come from, launch shape by launch shape.  Same unit handling as tools/traffic_from_pmc.py
(KB per dispatch summed over the XCDs, FETCH_SIZE doubled).  usage: python tools/traffic_by_kernel.py fetch.db write.db [family]"""
import os
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import families  # noqa: E402


def load(db_path, counter):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info('pmc_events')")]
    grid = 'grid_size_x' if 'grid_size_x' in cols else ('grid_x' if 'grid_x' in cols else None)
    q = f'select name, dispatch_id, counter_value{", " + grid if grid else ""} from pmc_events where counter_name = ?'
    acc, meta = defaultdict(float), {}
    for row in db.execute(q, (counter,)):
        acc[row[1]] += row[2]
        meta[row[1]] = (row[0], row[3] if grid else 0)
    return acc, meta


def main(fetch_db, write_db, only=None):
    f_acc, f_meta = load(fetch_db, 'FETCH_SIZE')
    w_acc, w_meta = load(write_db, 'WRITE_SIZE')
    cls = defaultdict(lambda: [0, 0.0, 0, 0.0])
    for d, v in f_acc.items():
        name, g = f_meta[d]
        c = cls[(families.bare(name) + name[name.find('<'):name.find('>') + 1] if '<' in name else families.bare(name), g)]
        c[0] += 1
        c[1] += 2.0 * 1024.0 * v
    for d, v in w_acc.items():
        name, g = w_meta[d]
        c = cls[(families.bare(name) + name[name.find('<'):name.find('>') + 1] if '<' in name else families.bare(name), g)]
        c[2] += 1
        c[3] += 1024.0 * v
    rows = []
    for (name, g), (nf, fb, nw, wb) in cls.items():
        fam = families.family_of('evk::' + name.split('<')[0] + '(')
        if only and fam != only:
            continue
        rows.append((fb + wb, name, g, nf, fb / max(nf, 1), wb / max(nw, 1), fam))
    tot = sum(r[0] for r in rows)
    print(f'{"kernel class":70s} {"grid":>9s} {"n":>5s} {"fetch MB":>9s} {"write MB":>9s} {"share":>6s}')
    for t, name, g, n, fb, wb, fam in sorted(rows, reverse=True)[:60]:
        print(f'{name[:70]:70s} {g:9d} {n:5d} {fb / 1e6:9.1f} {wb / 1e6:9.1f} {100.0 * t / tot:5.1f}%')


if __name__ == '__main__':
    main(*sys.argv[1:4])
