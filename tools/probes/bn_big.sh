cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bn_big; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/t -o bn -- python $R/tools/probes/bn_big.py > $O/log.txt 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob('$O/t/**/*.db', recursive=True)[0])
rows = db.execute("select name, grid_x, workgroup_x, duration, start from kernels where name like '%bn_%' order by start").fetchall()
seen = {}
for n, g, w, d, s in rows:
    key = (n.split('(')[0][-40:], g, w)
    seen.setdefault(key, []).append(d / 1e3)
with open('$R/gpurun_out/bn_big.txt', 'w') as f:
    for k, v in seen.items():
        v = sorted(v)
        f.write(f'{k[0]:42s} grid {k[1]:9d} x{k[2]:4d}  n={len(v):3d}  min {v[0]:7.1f}  med {v[len(v)//2]:7.1f} us\n')
PY
rm -rf $O
