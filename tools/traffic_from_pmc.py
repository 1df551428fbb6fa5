"""Build profiles/rNN_traffic.json from two rocprofv3 PMC passes of bench.py (MI355X_MICROARCH.md §HBM):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir_fetch> -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timer
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d <dir_write> -o r -- python bench.py ... (same command)
    python tools/traffic_from_pmc.py <fetch.db> <write.db> profiles/r01_traffic.json
Counters are KB per dispatch summed over the XCDs; on gfx950 FETCH_SIZE reports half of a wide coalesced
stream (x2 correction); WRITE_SIZE is calibrated against bn_apply, which writes exactly one tensor."""
import json
import sqlite3
import sys
from collections import defaultdict

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import families  # noqa: E402

FAMILY_KEYS = ('conv_igemm', 'conv_wgrad', 'bn', 'resample_loss')   # tools/families.py: one table for every profile tool


def per_dispatch(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute('select name, dispatch_id, counter_value from pmc_events where counter_name = ?', (counter,)).fetchall()
    acc = defaultdict(float)
    name_of = {}
    for name, disp, v in rows:
        acc[disp] += v
        name_of[disp] = name
    return acc, name_of


def main(fetch_db, write_db, out, steps=7):
    steps = int(steps)   # steps the profiled command ran: 1 warm-up + 3 unblocked-host probes + 3 timed (tools/profile_round.sh)
    res = {}
    f_acc, f_name = per_dispatch(fetch_db, 'FETCH_SIZE')
    w_acc, w_name = per_dispatch(write_db, 'WRITE_SIZE')
    for fam in FAMILY_KEYS:
        fsel = [v for d, v in f_acc.items() if families.family_of(f_name[d]) == fam]
        wsel = [v for d, v in w_acc.items() if families.family_of(w_name[d]) == fam]
        if not fsel or not wsel:
            continue
        fb = 2.0 * 1024.0 * sum(fsel) / len(fsel)
        wb = 1024.0 * sum(wsel) / len(wsel)
        res[fam] = {'launches': len(fsel), 'kernel_launches_per_step': round(len(fsel) / steps, 1),
                    'fetch_bytes_per_launch': round(fb), 'write_bytes_per_launch': round(wb),
                    'hbm_bytes_per_launch': round(fb + wb),
                    'kernels': sorted({families.bare(f_name[d]) for d in f_acc if families.family_of(f_name[d]) == fam})}
    res['method'] = __doc__.split('\n\n')[0] if False else (
        'rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE on `python bench.py --steps 3 '
        '--warmup 1 --no-cpu-baseline --no-kernel-timer`; counters are KB summed over the 8 XCDs per dispatch; FETCH_SIZE '
        'doubled (gfx950 reports half of a wide coalesced stream, guide §HBM); families as tools/families.py defines them (conv_igemm = every forward / data-gradient '
        'convolution kernel incl. conv1x1_ps; conv_wgrad; bn_*; resample_loss = bilinear_* / nearest2x_* / bce_* / dice_* / ce_*), the kernels '
        'counted are listed per family; tools/traffic_from_pmc.py')
    with open(out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1)[:1200])


if __name__ == '__main__':
    main(*sys.argv[1:5])
