"""dev tool: per-kernel PMC counter means from a rocprofv3 rocpd sqlite db (one row per kernel name)."""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else 'evk'
rows = db.execute("select name, dispatch_id, duration, counter_name, counter_value from pmc_events").fetchall()
agg = defaultdict(lambda: defaultdict(list)); dur = defaultdict(dict)
for name, disp, d, cn, cv in rows:
    if pat in name:
        agg[name][cn].append(cv); dur[name][disp] = d
for name, cs in agg.items():
    ds = list(dur[name].values())
    print(f'\n{name[:100]}  dispatches={len(ds)} avg_dur_us={sum(ds)/len(ds)/1e3:.1f}')
    for cn, v in sorted(cs.items()):
        print(f'   {cn:28s} mean {sum(v)/len(v):.4g}  (n={len(v)})')
