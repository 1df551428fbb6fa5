import os, sys, traceback, runpy
orig = os.sched_setaffinity
def spy(pid, mask):
    print('SETAFFINITY', pid, len(mask), file=sys.stderr); traceback.print_stack(file=sys.stderr); return orig(pid, mask)
os.sched_setaffinity = spy
import threading
def aff(tag): print('AFF', tag, len(os.sched_getaffinity(0)), [l.strip() for l in open('/proc/self/status') if 'Cpus_allowed_list' in l], file=sys.stderr, flush=True)
aff('start')
import torch; aff('torch')
torch.cuda.init(); aff('cuda init')
x = torch.zeros(4, device='cuda'); torch.cuda.synchronize(); aff('first kernel')
sys.path.insert(0, os.getcwd()); import ever_amd; aff('ever_amd')
sys.argv = ['bench.py', '--steps', '4', '--warmup', '2', '--no-cpu-baseline', '--no-graph-line'] + sys.argv[1:]
import atexit; atexit.register(lambda: aff('exit'))
runpy.run_path('bench.py', run_name='__main__')
