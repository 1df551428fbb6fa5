"""probe (GPU box): what the vendor library's dense fp16 / bf16 GEMM sustains on this part — a yardstick for the split-MFMA
convolution kernels (three fp16 MFMA products per fp32 product: algorithmic TF = raw MFMA TF / 3).  usage: python tools/probes/gemm_yardstick.py"""
import torch
dev = torch.device('cuda:0')
def run(m, n, k, dt, it=30):
    a = torch.randn(m, k, device=dev, dtype=dt); b = torch.randn(k, n, device=dev, dtype=dt)
    for _ in range(5): a @ b
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): a @ b
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / it * 1e-3
    return 2.0 * m * n * k / t / 1e12, t * 1e6
for dt in (torch.float16, torch.bfloat16):
    for (m, n, k) in [(8192, 8192, 8192), (262144, 256, 2304), (262144, 256, 256), (65536, 512, 128), (16384, 256, 1024), (16384, 256, 2304)]:
        tf, us = run(m, n, k, dt)
        print(f'{str(dt):16s} M={m:7d} N={n:5d} K={k:5d}: {tf:7.1f} TFLOP/s  {us:8.1f} us', flush=True)
# weight-gradient shapes: out[Cout][K] = dy^T [Cout x M] . im2col [M x K], reduction over M pixels
print('weight-gradient shapes (A^T B, reduction over the pixels):')
for dt in (torch.float16,):
    for (m, co, k) in [(262144, 256, 2304), (262144, 256, 256), (65536, 128, 1152), (16384, 256, 2304), (16384, 256, 1024), (4096, 512, 4608)]:
        a = torch.randn(m, co, device=dev, dtype=dt); b = torch.randn(m, k, device=dev, dtype=dt)
        f = lambda: a.t() @ b
        for _ in range(5): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30): f()
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) / 30 * 1e-3
        print(f'{str(dt):16s} pixels={m:7d} Cout={co:5d} K={k:5d}: {2.0 * m * co * k / t / 1e12:7.1f} TFLOP/s  {t * 1e6:8.1f} us', flush=True)
