"""dev tool: side-by-side diff of two kernel-stats CSVs written by tools/rocpd_summary.py (template arguments kept, call
arguments dropped).  usage: python tools/ab_diff.py A.csv B.csv [rows]"""
import collections, csv, re, sys


def load(p):
    fam = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(p)):
        k = re.sub(r'\(.*', '', r['kernel']).replace('void evk::', '').replace('evk::', '')[:64]
        fam[k][0] += float(r['total_us']) / 1e3
        fam[k][1] += int(r['calls'])
    return fam


a, b = load(sys.argv[1]), load(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
keys = sorted(set(a) | set(b), key=lambda k: -max(a[k][0], b[k][0]))
print('total ms  A %.1f  B %.1f' % (sum(v[0] for v in a.values()), sum(v[0] for v in b.values())))
for k in keys[:n]:
    print('%-66s %8.2f (%5d) %8.2f (%5d)  %+7.2f' % (k, a[k][0], a[k][1], b[k][0], b[k][1], a[k][0] - b[k][0]))
