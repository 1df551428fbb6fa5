# dev tool (GPU box): HBM fetch / write bytes and L2 hits / misses of the 3x3 kernels on the 256 -> 256 layer shapes of tools/wino_probe.py,
# one PMC pass per counter set (no trace domain beside --kernel-trace).  Round 6: Winograd kernel, 3x3x256 @128^2 forward: WRITE_SIZE
# 268 MB (exactly the output), L2 misses 5.9 M x 64 B = 376 MB against 268 MB of input x 1.27 of halo overlap = 340 MB, L2 hit rate 0.87.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_one; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  EVK_WINO=1 ONLY=256-256@ rocprofv3 --kernel-trace --pmc $c -d $O/$n -o r -- python $R/tools/wino_probe.py > $O/$n.log 2>&1
done
cd $R
python - <<'PY'
import sqlite3, glob, collections
for n in ('FETCH_SIZE','WRITE_SIZE','TCC_HIT_sum'):
    dbs = glob.glob(f'gpurun_out/pmc_one/{n}/*.db')
    if not dbs: print(n, 'no db'); continue
    db = sqlite3.connect(dbs[0])
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); name = {}
    for nm, d, cn, v in db.execute('select name, dispatch_id, counter_name, counter_value from pmc_events'):
        acc[d][cn] += v; name[d] = nm
    per = collections.defaultdict(list)
    for d, cs in acc.items():
        per[name[d][:60]].append(cs)
    for k, v in per.items():
        if 'conv3x3' not in k: continue
        keys = sorted(v[0].keys())
        print(n, k, len(v), {kk: [round(x[kk]) for x in v[:14]] for kk in keys})
PY
rm -rf $O
