"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) --kernel-trace --stats capture into a committed,
human-readable per-kernel table: calls, total / average / min / max duration, share of GPU time.

    python tools/rocpd_summary.py gpurun_out/prof_r01/farseg_results.db profiles/r01_kernel_stats
writes <out>.md and <out>.csv
"""
import csv
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import families  # noqa: E402


def main(db_path, out, steps=28, mode='two'):
    steps = int(steps)   # bench.py --steps 20 --warmup 5 runs 28 steps since round 3 (+ 3 empty-queue host probes)
    db = sqlite3.connect(db_path)
    rows = db.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                      'from kernels group by name order by sum(duration) desc').fetchall()
    total = sum(r[2] for r in rows)
    span = db.execute('select min(start), max(end) from kernels').fetchone()
    with open(out + '.csv', 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct'])
        for n, c, s, a, mn, mx in rows:
            w.writerow([n, c, round(s / 1e3, 1), round(a / 1e3, 2), round(mn / 1e3, 2), round(mx / 1e3, 2),
                        round(100.0 * s / total, 2)])
    # one family table for every profile tool (tools/families.py); the partition is asserted exact
    fam, rest = families.split(rows)
    label_of = {k: lab for k, lab, _ in families.FAMILIES}
    fam_rows = []
    for key in ('conv_igemm', 'conv_wgrad'):
        sel = fam[key]
        if sel:
            tot = sum(r[2] for r in sel)
            fam_rows.append((label_of[key], sum(r[1] for r in sel), tot))
    if fam['conv_wgrad']:
        sel = fam['conv_wgrad'] + fam['conv_wgrad_aux']
        fam_rows.append(('conv_wgrad as bench.py brackets it (one span per C-ABI call: + splitk_reduce, colsum_*, pack_f16x2 / '
                         'pack_planar kernels; "launches" = weight-gradient kernels)', sum(r[1] for r in fam['conv_wgrad']),
                         sum(r[2] for r in sel)))
    for key in ('conv_wgrad_aux', 'bn', 'resample_loss', 'pointwise', 'operand_prep', 'optimizer'):
        sel = fam[key]
        if sel:
            fam_rows.append((label_of[key], sum(r[1] for r in sel), sum(r[2] for r in sel)))
    rest_tot = sum(r[2] for r in rest)
    fam_rows.append(('every other kernel (ATen element-wise kernels, runtime copies / fills, any evk:: kernel no family claims)',
                     sum(r[1] for r in rest), rest_tot))
    claimed = sum(sum(r[2] for r in v) for v in fam.values()) + rest_tot
    assert abs(claimed - total) <= 1e-9 * total, (claimed, total)
    # the partition the table prints: every row except the repeated "as bench.py brackets it" one
    assert abs(sum(t for lab, c, t in fam_rows if not lab.startswith('conv_wgrad as bench.py')) - total) <= 1e-9 * total
    fam_rows = [(lab, c, t, t / max(1, c)) for lab, c, t in fam_rows]
    unclaimed = [r for r in rest if 'evk::' in r[0] and r[2] >= 0.003 * total]
    with open(out + '.md', 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats summary\n\n' + (
                '**Single-stream run (`EVK_WGRAD_STREAM=0`): every kernel alone on the chip — the durations behind the `achieved` / '
                '`frac` / `avg_launch_us` fields of the bench line\'s roofline objects (its two single-stream sampled steps).**\n\n'
                if mode == 'single' else
                '**Default run: weight gradients on a second stream (DESIGN 2.8).  Durations of kernels that ran beside a kernel of '
                'the other stream are longer than alone, and the families below add up to MORE than the step — compare with the '
                '`*_overlapped` fields of the bench line; the kernels alone are in `*_kernel_stats_single_stream.md`.**\n\n') +
                'source: `' + ('EVK_WGRAD_STREAM=0 ' if mode == 'single' else '') +
                'rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 '
                f'--no-cpu-baseline --no-graph-line` (tools/profile_round.sh; rocpd db summarised by tools/rocpd_summary.py; {steps} steps: 5 warm-up + 3 '
                f'empty-queue host probes + 20 timed); {sum(r[1] for r in rows)} '
                f'dispatches, {total / 1e6:.1f} ms of kernel time over a {(span[1] - span[0]) / 1e6:.1f} ms window\n\n')
        f.write('Kernel families as bench.py times them with HIP events.  Its `avg_launch_us` is per C-ABI CALL (a strided data '
                'gradient is up to four kernels, a weight gradient carries its split-K reduce / column sums / pack passes), so compare '
                f'`avg_launch_us x launches_per_step` with the ms-per-step column ({steps} steps in this run):\n\n'
                '| family | launches | total ms | avg us per launch | ms per step | % |\n|---|---:|---:|---:|---:|---:|\n')
        for label, calls, tot, avg in fam_rows:
            f.write(f'| {label} | {calls} | {tot / 1e6:.2f} | {avg / 1e3:.1f} | {tot / 1e6 / steps:.2f} | {100.0 * tot / total:.2f} |\n')
        f.write('\nThe family rows partition the capture (asserted: families + rest = total; the weight-gradient row "as bench.py '
                'brackets it" repeats the conv_wgrad row plus its auxiliaries).  `evk::` kernels above 0.3 % of the kernel time that '
                'no family claims: ' + (', '.join(f'`{families.bare(r[0])}` {100.0 * r[2] / total:.2f} %' for r in unclaimed) or 'none') + '.\n')
        f.write('\n| kernel | family | calls | total ms | avg us | min us | max us | % |\n|---|---|---:|---:|---:|---:|---:|---:|\n')
        for n, c, s, a, mn, mx in rows:
            f.write(f'| `{n[:110]}` | {families.family_of(n) or "-"} | {c} | {s / 1e6:.2f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | '
                    f'{100.0 * s / total:.2f} |\n')
    print(f'wrote {out}.md and {out}.csv')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 28, sys.argv[4] if len(sys.argv) > 4 else 'two')
