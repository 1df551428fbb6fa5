"""dev tool (GPU box): ablations of the one-tap DMA kernel on the small-map shapes (EVK_C1_DMA_DBG: 1 no activation DMA,
2 no weight DMA, 4 no compute, 8 no stores) — is the K loop bound by the memory path or by the CU?"""
import os, sys
sys.argv = [sys.argv[0], 'none']
os.environ['EVK_TUNE'] = '1'
import importlib.util
spec = importlib.util.spec_from_file_location('ab', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ab_c1dma.py'))
src = open(spec.origin).read().replace('\nmain()\n', '\n')
ns = {'__file__': spec.origin, '__name__': 'ab'}
exec(compile(src, spec.origin, 'exec'), ns)
for (h, ci, co) in [(32, 1024, 256), (32, 256, 1024), (16, 2048, 512), (16, 512, 2048), (64, 512, 128), (64, 128, 512)]:
    for packed in (0, 1):
        fn, out, keep = ns['problem'](h, ci, co, packed, 0)
        for force in ('e128', 'd128'):
            os.environ['EVK_X3_FORCE'] = force
            row = []
            for dbg in (0, 8, 4, 12, 3, 11, 1, 2, 5, 6):
                os.environ['EVK_C1_DMA_DBG'] = str(dbg)
                row.append(f'dbg{dbg}={ns["timeit"](fn):.1f}')
            os.environ['EVK_C1_DMA_DBG'] = '0'
            print(f'{ci:4d}->{co:4d} @{h:3d}^2 pk={packed} {force}: ' + ' '.join(row), flush=True)
