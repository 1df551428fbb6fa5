"""probe (GPU box): evk_bn_finalize_parts back to back on record counts of the 128^2 / 256^2 maps.  Round 6 used it to compare the
one-channel-per-workgroup finalisation (bn_parts_final_kernel<1, 256>) with a four-channel form reading 16-byte pieces
(EVK_BN_PARTS_ONE switched between them; the quad form is not in the tree): 2048 records x 256 channels 15.2 vs 18.3 us,
1024 x 256 6.9 vs 6.7, 2048 x 64 7.1 vs 7.9, -0.25 % on the step (profiles/r06_experiments/bench_parts_final.txt,
ab_bn_parts_quad.txt).  Both forms issue 2048 x 3 single-line requests per workgroup (a record row is 3 KB from the next): the
launch is bound by the rate at which ONE CU resolves divergent lines, not by L2 traffic, so fewer, wider workgroups lose."""
import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import os, torch
from ever_amd import _C
dev = torch.device('cuda:0')
for nparts, C in [(2048, 256), (1024, 256), (2048, 64), (4096, 64), (1024, 128)]:
    parts = torch.rand(nparts, 3, C, device=dev) + 1.0
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    sm, si, ss = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(2 * C, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for one in ('1', '0'):
        os.environ['EVK_BN_PARTS_ONE'] = one
        def run():
            _C.call('evk_bn_finalize_parts', parts.data_ptr(), nparts, C, 1000000, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(),
                    rv.data_ptr(), 0.1, 1e-5, sm.data_ptr(), si.data_ptr(), ss.data_ptr(), st)
        for _ in range(20): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(500): run()
        e1.record(); torch.cuda.synchronize()
        print(f'nparts {nparts} C {C} one-channel={one}: {e0.elapsed_time(e1) / 500 * 1e3:.2f} us per call (back to back)')
