"""Per-kernel pipe utilisation from rocprofv3 --pmc passes over bench.py (tools/pmc_bench.sh): for the kernels that take
the most time, the mean counters per dispatch and the derived ratios

  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES)      matrix-pipe occupancy over the 1024 SIMDs (SQ_BUSY_CYCLES
                                                                      is summed over the 32 shader engines)
  lds_busy    = SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles)
  lds_conflict= SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_any / wait_inst / active = the three disjoint shares of SQ_WAVE_CYCLES (MI355X_MICROARCH.md, PMC slots)
  clock GHz   = GRBM_GUI_ACTIVE / 8 XCDs / duration (duration of the profiled dispatch: profiled passes clock lower)

usage: python tools/pmc_bench_summary.py pass1.db pass2.db ... [top=14]"""
import os
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import families  # noqa: E402


def load(paths):
    agg = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for p in paths:
        db = sqlite3.connect(p)
        for name, disp, d, cn, cv in db.execute('select name, dispatch_id, duration, counter_name, counter_value from pmc_events'):
            agg[name][(p, disp, cn)].append(cv)
            dur[name][(p, disp)] = d
    out = {}
    for name, cs in agg.items():
        per = defaultdict(list)
        for (p, disp, cn), v in cs.items():
            per[cn].append(sum(v))          # a dispatch's counter summed over its instances (XCDs / SEs)
        ds = list(dur[name].values())
        out[name] = dict(n=len(ds), dur_us=sum(ds) / len(ds) / 1e3, total_us=sum(ds) / 1e3 / max(1, len(paths)),
                         c={k: sum(v) / len(v) for k, v in per.items()})
    return out


def main():
    paths = [a for a in sys.argv[1:] if a.endswith('.db')]
    jpath = [a for a in sys.argv[1:] if a.endswith('.json')]
    top = 14
    k = load(paths)
    if jpath:
        # machine-readable: per kernel family the time-weighted matrix-pipe occupancy (bench.py puts it into `roofline`)
        import json
        out = {}
        for fam in ('conv_igemm', 'conv_wgrad', 'bn'):   # tools/families.py: one table for every profile tool
            num = den = 0.0
            per = {}
            for n, r in k.items():
                if families.family_of(n) != fam:
                    continue
                c = r['c']
                if 'SQ_BUSY_CYCLES' not in c or 'SQ_VALU_MFMA_BUSY_CYCLES' not in c:
                    continue
                busy = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (32.0 * c['SQ_BUSY_CYCLES'])
                w = r['total_us']
                num, den = num + busy * w, den + w
                per[n[:100]] = dict(mfma_busy=round(busy, 4), avg_us=round(r['dur_us'], 1), dispatches_per_pass=r['n'] // max(1, len(paths)))
            if den > 0:
                out[fam] = dict(mfma_busy=round(num / den, 4), kernels=per)
        out['method'] = ('rocprofv3 --kernel-trace --pmc <4 counters per pass> on `python bench.py --steps 2 --warmup 1 '
                         '--no-cpu-baseline --no-kernel-timer` (tools/pmc_bench.sh); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / '
                         '(32 x SQ_BUSY_CYCLES) per dispatch, averaged per kernel, weighted by kernel time over the family')
        with open(jpath[0], 'w') as f:
            json.dump(out, f, indent=1)
    names = sorted(k, key=lambda n: -k[n]['total_us'])[:top]
    print(__doc__.split('usage')[0])
    for n in names:
        r = k[n]
        c = r['c']
        g = lambda x: c.get(x, float('nan'))
        cyc = g('SQ_BUSY_CYCLES') / 32.0
        line = [f"{n[:110]}", f"  dispatches/pass {r['n'] // max(1, len(paths))}  avg {r['dur_us']:.1f} us"]
        line.append(f"  mfma_busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / (1024.0 * cyc):.3f}   lds_busy {g('SQ_LDS_IDX_ACTIVE') / (256.0 * cyc):.3f}"
                    f"   lds_conflict/active {g('SQ_LDS_BANK_CONFLICT') / max(1.0, g('SQ_LDS_IDX_ACTIVE')):.3f}"
                    f"   clock {g('GRBM_GUI_ACTIVE') / 8.0 / (r['dur_us'] * 1e3):.2f} GHz")
        wc = g('SQ_WAVE_CYCLES')
        line.append(f"  of wave cycles: wait_any {g('SQ_WAIT_ANY') / wc:.3f}  wait_inst_any {g('SQ_WAIT_INST_ANY') / wc:.3f}"
                    f"  (lds {g('SQ_WAIT_INST_LDS') / wc:.3f})  active_any {g('SQ_ACTIVE_INST_ANY') / wc:.3f}"
                    f"  active_valu {g('SQ_ACTIVE_INST_VALU') / wc:.3f}  active_lds {g('SQ_ACTIVE_INST_LDS') / wc:.3f}"
                    f"  vmem_cycles {g('SQ_INST_CYCLES_VMEM') / wc:.3f}")
        line.append(f"  per dispatch: waves {g('SQ_WAVES'):.0f}  valu insts {g('SQ_INSTS_VALU'):.3g}  lds insts {g('SQ_INSTS_LDS'):.3g}"
                    f"  mfma busy cycles {g('SQ_VALU_MFMA_BUSY_CYCLES'):.3g}  sq busy cycles {g('SQ_BUSY_CYCLES'):.3g}")
        print('\n'.join(line) + '\n')


if __name__ == '__main__':
    main()
