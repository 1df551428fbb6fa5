"""dev tool (GPU box): the software-pipelined one-tap kernel (csrc/conv1x1_sp.hip: s128 / s64 = ring of four, t128 / t64 =
ring of three) against the dispatch default and the other one-tap forms on the FarSeg 1x1 shapes."""
import os, sys
os.environ['EVK_TUNE'] = '1'
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, 'ab_c1dma.py')).read().replace('\nmain()\n', '\n')
ns = {'__file__': os.path.join(here, 'ab_c1dma.py'), '__name__': 'ab'}
exec(compile(src, ns['__file__'], 'exec'), ns)
problem, timeit, B = ns['problem'], ns['timeit'], ns['B']
forms = sys.argv[1].split(',') if len(sys.argv) > 1 else ['e128', 'd128', 'q128', 's128', 't128', 's64']
quick = len(sys.argv) > 2 and sys.argv[2] == 'quick'
tot = {}
for (h, ci, co) in ns['SHAPES']:
    for packed in ((1,) if quick else (0, 1)):
        for stats in ((0,) if quick else (0, 1)):
            fn, out, keep = problem(h, ci, co, packed, stats)
            os.environ['EVK_X3_FORCE'] = ''
            timeit(fn, 5)
            base = timeit(fn)
            ref = out.clone()
            res = {}
            for cfg in forms:
                os.environ['EVK_X3_FORCE'] = cfg
                out.zero_()
                t = timeit(fn)
                err = float((out - ref).abs().max() / ref.abs().max())
                res[cfg] = (t, err)
            os.environ['EVK_X3_FORCE'] = ''
            mb = B * h * h * (ci + co) * 4 / 1e6
            for k, v in res.items():
                a = tot.get(k, 0.0)
                tot[k] = a + v[0]
            tot['default'] = tot.get('default', 0.0) + base
            print(f'{ci:4d}->{co:4d} @{h:3d}^2 pk={packed} st={stats} {mb:6.1f} MB  default {base:6.1f} | '
                  + ' '.join(f'{k}={v[0]:.1f}' + ('' if v[1] < 1e-5 else f'(ERR {v[1]:.1e})') for k, v in res.items()), flush=True)
print('sums:', {k: round(v) for k, v in tot.items()})
