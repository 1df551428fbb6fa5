cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tbk; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph-line --no-kernel-timer > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph-line --no-kernel-timer > $O/write.log 2>&1
cd $R
python tools/traffic_by_kernel.py $(ls $O/fetch/*.db | head -1) $(ls $O/write/*.db | head -1) > gpurun_out/traffic_by_kernel.txt 2>&1
python - <<'PY' >> gpurun_out/traffic_by_kernel.txt 2>&1
import sqlite3, glob
db = sqlite3.connect(glob.glob('gpurun_out/tbk/fetch/*.db')[0])
print([r[1] for r in db.execute("pragma table_info('pmc_events')")])
PY
rm -rf $O/fetch $O/write
head -70 gpurun_out/traffic_by_kernel.txt
