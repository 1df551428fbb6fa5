"""dev tool (GPU box): per-layer "floor vs passes" table of the convolution launches of one FarSeg-R50
training step.  Every conv launch of configuration c2 is timed (HIP events, single stream: each kernel alone) under the
three split arithmetics, which differ ONLY in the number of MFMA partial products per product —
    bf16 1 (one plane per operand)      f16x2 3 (two planes)      bf16x3 6 (three planes)
and, per layer shape (family, GFLOP, algorithmic MB):
    per-pass = (t3 - t1) / 2            what one more MFMA pass over the layer costs
    floor    = t1 - per-pass            the part of the launch that does not depend on the MFMA count
    bytes    = algorithmic bytes / 6.3 TB/s   (the guide's measured copy rate)
    ideal3   = max(3 x per-pass, bytes)  and  t3 / ideal3
usage: python tools/floor_table.py [config]      (worker: python tools/floor_table.py --worker <math> <config> <out.json>)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(math, cfg, out):
    sys.path.insert(0, ROOT)
    import torch
    import ever_amd as er
    from ever_amd.hip import timing
    from ever_amd.hip import functional as HF
    import bench
    HF.set_conv_math(math)
    HF.set_wgrad_stream(False)
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    m, inputs = bench.make_workload(er, cfg, dev, bench.BATCH, 0)[:2]
    m = m.to(dev).train()
    recs = []
    for it in range(5):
        t = timing.KernelTimer() if it >= 2 else None
        if t: t.__enter__()
        loss = sum(v for k, v in m(*inputs).items() if k.endswith('loss')); loss.backward()
        if t: t.__exit__()
        m.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        if t: recs.append([(f, fl, nb, s.elapsed_time(e) * 1e3) for f, fl, nb, s, e, _sc in t.records if fl > 0])
    # the i-th conv launch of a step is the same layer in every step and under every arithmetic: keep the order
    n = len(recs[0])
    assert all(len(r) == n for r in recs)
    rows = [dict(family=recs[0][i][0], gflop=recs[0][i][1] / 1e9, mb=recs[0][i][2] / 1e6,
                 us=min(r[i][3] for r in recs)) for i in range(n)]
    json.dump(rows, open(out, 'w'))


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
    data = {}
    for math in ('bf16', 'f16x2', 'bf16x3'):
        out = f'/tmp/floor_{math}.json'
        subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', math, cfg, out], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        data[math] = json.load(open(out))
    n = len(data['f16x2'])
    assert len(data['bf16']) == n and len(data['bf16x3']) == n
    agg = {}
    for i in range(n):
        r = data['f16x2'][i]
        k = (r['family'], round(r['gflop'], 3), round(r['mb'], 1))
        a = agg.setdefault(k, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += data['bf16'][i]['us']; a[2] += r['us']; a[3] += data['bf16x3'][i]['us']
    tot = {m: sum(r['us'] for r in data[m]) for m in data}
    print(f'config {cfg}: {n} convolution launches per step (forward + data gradient = conv_igemm, weight gradient = conv_wgrad), '
          f'single stream, best of 3 timed steps per launch')
    print(f'conv time per step: 1 product {tot["bf16"] / 1e3:.2f} ms, 3 products {tot["f16x2"] / 1e3:.2f} ms, 6 products '
          f'{tot["bf16x3"] / 1e3:.2f} ms  ->  fit conv_ms = a + b x products: b = {(tot["f16x2"] - tot["bf16"]) / 2e3:.2f} (1->3), '
          f'{(tot["bf16x3"] - tot["f16x2"]) / 3e3:.2f} (3->6) ms per product, a = {(tot["bf16"] - (tot["f16x2"] - tot["bf16"]) / 2) / 1e3:.2f} ms')
    print('family        GFLOP      MB   n |     t1      t3      t6 us | per-pass   floor   bytes@6.3 | floor/bytes  t3/ideal3 | excess us/step')
    rows = []
    for (f, gf, mb), (c, t1, t3, t6) in agg.items():
        t1, t3, t6 = t1 / c, t3 / c, t6 / c
        pp = max(0.0, (t3 - t1) / 2)
        floor = t1 - pp
        bt = mb / 6.3
        ideal = max(3 * pp, bt, 1e-3)
        rows.append((c * (t3 - ideal), f, gf, mb, c, t1, t3, t6, pp, floor, bt, ideal))
    sum_ex = 0.0
    for ex, f, gf, mb, c, t1, t3, t6, pp, floor, bt, ideal in sorted(rows, reverse=True):
        sum_ex += ex
        print(f'{f:11s} {gf:8.3f} {mb:7.1f} {c:3d} | {t1:6.1f} {t3:7.1f} {t6:7.1f}    | {pp:7.1f} {floor:8.1f} {bt:9.1f}   | '
              f'{floor / max(bt, 1e-3):8.2f} {t3 / ideal:10.2f}   | {ex:8.1f}')
    print(f'sum over layers of n x (t3 - max(3 x per-pass, bytes-time)) = {sum_ex / 1e3:.2f} ms per step of {tot["f16x2"] / 1e3:.2f}')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--worker':
        worker(*sys.argv[2:5])
    else:
        main()
