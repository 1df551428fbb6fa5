"""bench.py — FarSeg-R50 training-step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic tiles already resident in HBM:
forward (ResNet-50 encoder -> FPN -> FS-Relation -> decoder -> head -> BCE+dice) + backward (+ the
RCCL gradient all-reduce under DDP, overlapped with backward) + fused SGD update.  Workload =
BASELINE.json configs[1]: FarSeg ResNet-50 FPN, 3-band 512x512, batch 16 per GPU (weak scaling).

Rank 0 prints ONE JSON line: the contract fields plus
  roofline     — achieved MFMA rate of the dominant kernel family (conv forward + data gradient:
                 conv3x3_halo_x3 / conv1x1_dma / conv_igemm_x3ws / conv_igemm_x3 kernels) = algorithmic FLOPs / HIP-event
                 time of its launches over the timed region, against the peak of the instruction the
                 arithmetic issues: dense 16-bit MFMA 2500 TF / partial products per fp32 product — 3 for the default
                 f16x2 arithmetic (2-term scaled fp16 split) = 833.3 TF, 6 under --conv-math bf16x3 = 416.7 TF
                 (157.3 TF, v_mfma_f32_32x32x2_f32, under --conv-math f32);
  roofline_wgrad / roofline_encoder — the same for the weight-gradient family and for the ResNet-50
                 encoder's convolutions alone (forward + data gradient + weight gradient of `en.*`:
                 the stack BASELINE.json's 0.6 target is stated on);
  roofline_hbm_* — achieved algorithmic GB/s of the BatchNorm and resample/loss families vs 8 TB/s;
                 (weight gradients run on a second stream, DESIGN 2.8: the sampled steps of the timed region alternate —
                 step 0 single-stream = each kernel alone = the achieved / frac fields; step K/2 as every other step =
                 the *_overlapped fields)
  cpu_baseline — the CPU oracle (stock PyTorch port of the reference path) timed on this box's host
                 cores on a bounded sample of the same workload (N=1 only).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')   # (ever_amd/__init__.py; here before torch can touch the HIP runtime)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-in MFMA peak (= fp32 vector peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak (v_mfma_f32_32x32x16_{bf16,f16})
X3_PASSES = 6                   # bf16 MFMA partial products per fp32 product in the split kernels
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable on a float4 copy)
GF_FWD_BWD_PER_TILE = 342.7     # conv GFLOP fwd+bwd per 512x512x3 tile, default head
TILE, BANDS, BATCH = 512, 3, 16


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1,
                   help='ranks = GPUs of this node.  Under torchrun (WORLD_SIZE set) it must equal the world size; without it '
                        'and N > 1 this script starts the N ranks itself (torch.distributed.run, rendezvous on 127.0.0.1)')
    p.add_argument('--dry-launch', action='store_true',
                   help='launch check without a GPU: start the ranks exactly as a real run would, join a gloo group, have rank 0 '
                        'print what every rank saw (tests/test_bench_launch_cpu.py), and exit')
    p.add_argument('--steps', type=int, default=40)
    p.add_argument('--warmup', type=int, default=5)
    p.add_argument('--batch', type=int, default=BATCH, help='tiles per GPU (default: the BASELINE config)')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-baseline-only', action='store_true', help='(internal) time the CPU oracle and print its JSON object')
    p.add_argument('--no-graph-line', action='store_true',
                   help='skip the `hip_graph_replay` object (the same step captured as one hipGraph and replayed, measured in a '
                        'child process after the timed region; never the headline value)')
    p.add_argument('--ddp', choices=['flat', 'torch'], default='flat',
                   help='gradient exchange under --gpus N>1: ever_amd FlatGradDDP (default) or torch DistributedDataParallel')
    p.add_argument('--conv-math', choices=['f16x2', 'bf16x3', 'f32', 'bf16'], default=None,
                   help='convolution arithmetic (default: ever_amd default = f16x2, 2-term fp16 split with per-tensor scale, 3 '
                        'MFMA products, fp32 grade; bf16x3 = 3-term bf16 split, 6 products, fp32 grade; f32 = exact fp32 '
                        'MFMA; bf16 = plain bf16 operands, the --mixed_precision bf16 mode: NOT the headline configuration)')
    p.add_argument('--no-kernel-timer', action='store_true')
    p.add_argument('--host-cores', type=int, default=None,
                   help='CPUs this rank\'s enqueuing threads are kept on (ever_amd.core.device.pin_host_threads): default '
                        'EVK_HOST_CORES, else 4 (the inputs are resident in HBM, there are no loader workers to squeeze); '
                        '0 = only restore the launch mask if the HIP runtime widened it')
    p.add_argument('--graph', action='store_true',
                   help='run the step as one captured hipGraph (ever_amd/core/graph.py; N = 1, no per-kernel event timer): '
                        'what the host costs then is in host_*_ms_per_step')
    p.add_argument('--config', choices=['c2', 'c3', 'c4', 'c5'], default='c2',
                   help='BASELINE.json configs[1..4] at N = 1 (per-GPU size): c2 = the headline (default, the only one the '
                        'driver times); c3 = FarSeg++ R50 4-band 1024x1024 batch 8; c4 = ChangeStar (FarSeg-R50 + ChangeMixin) '
                        '2 x (3x512x512) batch 8; c5 = FreeNet 200-band 610x340 scene (padded to 616x344), batch 1')
    return p.parse_args()


def make_workload(er, cfg, dev, batch, rank):
    """(model, args of model(*args), unit name, units per step, metric, workload text, GFLOP fwd+bwd per unit or None)"""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(2333 + rank)
    if cfg == 'c2':
        x, y = make_batch(dev, batch, rank)
        return (er.module.FarSeg(dict()), (x, y), 'tiles/s', batch, '512x512 tiles/sec fwd+bwd, FarSeg-R50',
                f'FarSeg ResNet-50 FPN (FarSegHead defaults, BCE+dice), 3-band 512x512, batch {batch}/GPU, fwd+bwd+SGD step, '
                'inputs resident in HBM', GF_FWD_BWD_PER_TILE, [BANDS, TILE, TILE])
    if cfg == 'c3':
        b = 8 if batch == BATCH else batch
        x = torch.randn(b, 4, 1024, 1024, device=dev, generator=g)
        y = (torch.rand(b, 1024, 1024, device=dev, generator=g) < 0.3).long()
        y[:, :8, :8] = 255
        return (er.module.FarSegPP(dict(encoder=dict(in_channels=4))), (x, y), 'tiles/s', b,
                '1024x1024 tiles/sec fwd+bwd, FarSeg++-R50',
                f'FarSeg++ ResNet-50 (FSRelationV2 head, BCE+dice), 4-band 1024x1024, batch {b}/GPU, fwd+bwd+SGD step, inputs '
                'resident in HBM (BASELINE.json configs[2] at its per-GPU size)', None, [4, 1024, 1024])
    if cfg == 'c4':
        b = 8 if batch == BATCH else batch
        x = torch.randn(b, 6, TILE, TILE, device=dev, generator=g)
        y = dict(cls=(torch.rand(b, TILE, TILE, device=dev, generator=g) < 0.3).long(),
                 cls2=(torch.rand(b, TILE, TILE, device=dev, generator=g) < 0.3).long())
        y['change'] = (y['cls'] != y['cls2']).long()
        y['change'][:, :8, :8] = 255
        return (er.module.ChangeStarFarSeg(dict()), (x, y), 'pairs/s', b,
                'bitemporal 512x512 tile pairs/sec fwd+bwd, ChangeStar(FarSeg-R50)',
                f'ChangeStar (FarSeg-R50 + ChangeMixin; semantic BCE+dice on both dates, change BCE on both orders), bitemporal '
                f'2 x (3x512x512), batch {b} pairs/GPU, fwd+bwd+SGD step, inputs resident in HBM (BASELINE.json configs[3] at its '
                'per-GPU size)', None, [6, TILE, TILE])
    from ever_amd.module.freenet import divisible_pad
    x = divisible_pad(torch.randn(1, 200, 610, 340, device=dev, generator=g), 8)
    y = divisible_pad(torch.randint(0, 17, (1, 610, 340), device=dev, generator=g).float(), 8).long()
    return (er.module.FreeNet(dict()), (x, y), 'scenes/s', 1, '200-band 610x340 scenes/sec fwd+bwd, FreeNet',
            'FreeNet patch-free hyperspectral model (GroupNorm conv blocks, nearest top-down, CE on labelled pixels), one '
            '200-band 610x340 scene padded to 616x344, fwd+bwd+SGD step, input resident in HBM (BASELINE.json configs[4])',
            None, [200, 616, 344])


def make_batch(dev, batch, rank):
    """N(0,1) image, P(fg)=0.3 labels with an 8x8 ignore(255) corner, seed 2333+rank."""
    g = torch.Generator(device=dev)
    g.manual_seed(2333 + rank)
    x = torch.randn(batch, BANDS, TILE, TILE, device=dev, generator=g)
    y = (torch.rand(batch, TILE, TILE, device=dev, generator=g) < 0.3).long()
    y[:, :8, :8] = 255
    return x, y


def pmc_kernel_launches_per_step(family):
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')), reverse=True):
        try:
            with open(path) as f:
                return json.load(f)[family]['kernel_launches_per_step']
        except (OSError, KeyError, ValueError):
            continue
    return None


def pmc_traffic(family):
    """(HBM bytes per launch, file) of a kernel family from the newest committed rocprofv3 PMC passes (FETCH_SIZE x2 +
    WRITE_SIZE, MI355X_MICROARCH.md §HBM; profiles/rNN_traffic.json states the method); (None, None) if absent."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')), reverse=True):
        try:
            with open(path) as f:
                return json.load(f)[family]['hbm_bytes_per_launch'], os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def pmc_busy(family):
    """matrix-pipe occupancy of a kernel family from the newest committed PMC summary (profiles/rNN_pmc_conv_kernels.json,
    tools/pmc_bench.sh), or None"""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_conv_kernels.json')), reverse=True):
        try:
            with open(path) as f:
                return json.load(f)[family]['mfma_busy'], os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def host_cores():
    """Cores this process may actually use: the cgroup CPU quota if one is set (the GPU box exposes 256
    logical CPUs under a 16-core quota; oversubscribing it makes oneDNN 100x slower), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def library_gemm_yardstick(dev):
    """What the vendor library's dense fp16 GEMM (torch.matmul = hipBLASLt) sustains on THIS box, after the timed region: the
    practical ceiling of the matrix pipe under this part's power limit, beside the nominal 2500 TF the roofline divides by.
    The f16x2 arithmetic needs three such products per fp32 product (tools/probes/gemm_yardstick.py has the per-layer shapes)."""
    out = {}
    try:
        for name, (m, n, k) in (('8192x8192x8192', (8192, 8192, 8192)), ('262144x256x2304 (3x3x256 @128^2 as a plain GEMM)', (262144, 256, 2304))):
            a = torch.randn(m, k, device=dev, dtype=torch.float16)
            b = torch.randn(k, n, device=dev, dtype=torch.float16)
            for _ in range(3):
                a @ b
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                a @ b
            e.record()
            torch.cuda.synchronize()
            tf = 2.0 * m * n * k / (s.elapsed_time(e) / 10 * 1e-3) / 1e12
            out[name] = {'tflops': round(tf, 1), 'frac_of_nominal_2500': round(tf / PEAK_BF16_MFMA_TFLOPS, 3),
                         'algorithmic_tflops_at_3_products': round(tf / 3, 1)}
            del a, b
    except Exception as ex:       # noqa: BLE001 (a yardstick, never a reason to lose the line)
        out['error'] = str(ex)[:200]
    return out


def board_power_line(step, seconds=2.5, device_index=0):
    """Board power while the SAME training step keeps running after the timed region (rocm-smi sampled every ~0.4 s from a
    thread; DESIGN 2.10: every phase of this step runs at 73-100 % of the part's power cap, so the step is bounded by its
    energy, not by a kernel schedule).  None when rocm-smi is not on the box.  Never part of the timed region."""
    import shutil
    import subprocess
    import threading
    if shutil.which('rocm-smi') is None:
        return None
    samples, stop = [], [False]

    def sampler():
        time.sleep(0.5)
        while not stop[0]:
            try:
                out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True,
                                     timeout=10).stdout
                d = json.loads(out)
                card = d.get(f'card{device_index}') or next(iter(d.values()))
                w = next((float(v) for k, v in card.items() if 'Power (W)' in k), None)
                clk = next((int(''.join(c for c in v if c.isdigit())) for k, v in card.items() if k.startswith('sclk clock speed')), None)
                if w is not None:
                    samples.append((w, clk))
            except Exception:      # noqa: BLE001 (a probe, never a reason to lose the line)
                pass
            time.sleep(0.3)

    cap = 1400.0   # (the part's default; replaced by what the board reports)
    try:
        d = json.loads(subprocess.run(['rocm-smi', '--showmaxpower', '--json'], capture_output=True, text=True, timeout=10).stdout)
        card = d.get(f'card{device_index}') or next(iter(d.values()))
        cap = next((float(v) for k, v in card.items() if 'Max Graphics Package Power' in k or 'Max' in k), cap) or cap
    except Exception:      # noqa: BLE001
        pass
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        n += 10
    elapsed = time.perf_counter() - t0
    stop[0] = True
    th.join(timeout=15)
    if not samples:
        return None
    w = sum(s[0] for s in samples) / len(samples)
    clks = [s[1] for s in samples if s[1]]
    return {'avg_board_power_w': round(w, 1), 'power_cap_w': round(cap, 1), 'frac_of_cap': round(w / cap, 3), 'card': device_index,
            'avg_sclk_mhz': round(sum(clks) / len(clks)) if clks else None, 'samples': len(samples),
            'joules_per_step': round(w * elapsed / n, 2), 'ms_per_step_during_probe': round(elapsed / n * 1e3, 3),
            'note': 'rocm-smi during extra steps after the timed region; by phase (profiles/r05_experiments/power_phase.txt): 3x3 halo '
                    'convolution 1380 W, one-tap 256->256 @128^2 1400 W, BatchNorm backward 1217 W, device copy 1018 W, idle 283 W'}


def graph_replay_line(args):
    """The same workload with the step captured once as a hipGraph and replayed (`bench.py --graph`, ever_amd/core/graph.py;
    bit-identical to the eager step: tests/test_graph_gpu.py), measured in a CHILD process after the timed region — a capture
    problem can then not take the headline line with it.  Reported beside `value`, never instead of it: a replay exposes no
    per-launch HIP events, so it carries no roofline."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--graph', '--steps', str(args.steps), '--warmup', str(max(args.warmup, 4)),
           '--config', args.config, '--no-cpu-baseline', '--no-graph-line']
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        rows = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
        d = json.loads(rows[-1])
        return {'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'],
                'host_enqueue_ms_per_step': d['host_enqueue_ms_per_step'],
                'how': 'python bench.py --graph (child process, same box, right after the timed region): forward + losses + '
                       'backward + fused SGD captured once by torch.cuda.graph and replayed; EVK_GRAPH=1 in the Launcher.  Behind the eager step by '
                       '3-4 %: a replayed graph serialises its two branches at every fork, so the capture forks the weight-gradient '
                       'branch once per 32 layers instead of once per layer (DESIGN 3; 1 / 2 / 4 / 8 / 32 launches per fork: 513.7 / '
                       '520.8 / 529.5 / 528.7 / 550.6 tiles/s where eager is 572.4, profiles/r05_experiments/ab_graph_batch.txt)'}
    except Exception as e:   # never at the expense of the headline line
        return {'value': None, 'error': f'{type(e).__name__}: {str(e)[:160]}'}


def cpu_baseline(seconds_budget=20.0):
    """The oracle (CPU port of the reference path, stock torch.nn) on the host cores: FarSeg-R50 fwd+bwd+SGD
    on 3x512x512 tiles at batch 2 (bounded sample: one warm-up + as many steps as fit the budget)."""
    from oracle import farseg_ref
    torch.manual_seed(0)
    cores = host_cores()
    torch.set_num_threads(cores)
    net = farseg_ref.FarSegRef('resnet50', BANDS, 1).train()
    opt = torch.optim.SGD(net.parameters(), lr=0.007, momentum=0.9, weight_decay=1e-4)
    b = 2
    x = torch.randn(b, BANDS, TILE, TILE)
    y = (torch.rand(b, TILE, TILE) < 0.3).long()
    y[:, :8, :8] = 255

    def step():
        out = net(x, y)
        sum(out.values()).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)

    step()
    t0, n = time.time(), 0
    while True:
        step()
        n += 1
        if time.time() - t0 > seconds_budget or n >= 8:
            break
    dt = (time.time() - t0) / n
    return dict(value=round(b / dt, 3), unit='tiles/s', cores=cores, kind='port',
                sample=f'oracle FarSegRef-R50 fwd+bwd+SGD, batch {b} of 3x512x512, {n} timed steps after 1 warm-up, '
                       f'torch {torch.__version__} CPU, {cores} threads')


def cpu_baseline_child():
    """cpu_baseline() in a CHILD process started on the CPU set this process was launched with: the parent has narrowed its own
    mask to a few CPUs (--host-cores) and its OpenMP workers inherited that; the child initialises no GPU runtime, so its mask
    stays what it is given."""
    import subprocess
    from ever_amd.core.device import launch_affinity
    mask = launch_affinity()
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-only'], capture_output=True, text=True,
                             timeout=900, preexec_fn=(lambda: os.sched_setaffinity(0, mask)) if mask else None)
        rows = [ln for ln in out.stdout.splitlines() if ln.startswith('{"value"')]
        return json.loads(rows[-1])
    except Exception as e:   # never at the expense of the headline line
        return {'value': None, 'error': f'{type(e).__name__}: {str(e)[:160]}'}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` with no torchrun around it: become the launcher.  The reference's contract is an `env://`
    rendezvous prepared by torchrun (`ever/trainer/th_ddp_trainer.py:13-17`, README "torchrun --nproc_per_node"); the same
    ranks, environment and argv are produced here, so a rank cannot tell the two ways of starting apart."""
    import subprocess
    if args.gpus < 1:
        raise SystemExit(f'bench.py: --gpus {args.gpus}: need at least one rank')
    if not args.dry_launch:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f'bench.py: --gpus {args.gpus} asked for, but this node shows {have} GPU(s) '
                             f'(torch.cuda.device_count()); refusing to report an n_gpus={args.gpus} line from fewer devices')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('OMP_NUM_THREADS', '1' if args.dry_launch else str(max(1, host_cores() // args.gpus)))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def dry_launch(args, world, rank, local_rank):
    """Every rank joins a gloo group over the rendezvous it was handed and reports; rank 0 prints one JSON line."""
    import torch.distributed as dist
    dist.init_process_group(backend='gloo', init_method='env://', rank=rank, world_size=world)
    mine = torch.tensor([rank, local_rank, int(os.environ.get('WORLD_SIZE', -1)), os.getpid()], dtype=torch.int64)
    seen = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(seen, mine)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({'dry_launch': True, 'n_gpus': world, 'gpus_arg': args.gpus,
                          'ranks': [{'rank': int(t[0]), 'local_rank': int(t[1]), 'world_size_env': int(t[2]), 'pid': int(t[3])}
                                    for t in seen]}), flush=True)


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()), flush=True)
        return
    if 'WORLD_SIZE' not in os.environ and (args.gpus > 1 or args.dry_launch):
        self_launch(args)    # does not return
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree '
                         '(n_gpus in the JSON line is the number of ranks that actually ran)')
    if args.dry_launch:
        return dry_launch(args, world, rank, local_rank)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (the HIP path has no CPU fallback)')
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f'bench.py: rank {rank} (local rank {local_rank}) has no GPU of its own: this node shows '
                         f'{torch.cuda.device_count()} device(s) for {world} ranks')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_ddp = world > 1 or os.environ.get('EVK_BENCH_FORCE_DDP') == '1'   # the env knob runs the DDP path on 1 GPU (tests)
    if use_ddp:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        # the image exports NCCL_DEBUG=VERSION: RCCL's banner would follow the JSON line on stdout
        if 'EVK_NCCL_DEBUG' in os.environ:
            os.environ['NCCL_DEBUG'] = os.environ['EVK_NCCL_DEBUG']
        else:
            os.environ.pop('NCCL_DEBUG', None)
        dist.init_process_group(backend='nccl', init_method='env://', rank=rank, world_size=world,
                                device_id=dev)   # RCCL over xGMI, communicator bound to this rank's GPU
    import ever_amd as er
    from ever_amd import _C
    from ever_amd.core.device import pin_host_threads
    torch.cuda.init()
    if args.host_cores is None:
        args.host_cores = int(os.environ.get('EVK_HOST_CORES', '4') or 0)
    pin_host_threads(local_rank, args.host_cores)   # (after the runtime is up: its initialisation may reset the mask)
    from ever_amd.hip import timing
    from ever_amd.hip import functional as HF
    if args.conv_math:
        HF.set_conv_math(args.conv_math)
    conv_math = HF.get_conv_math()
    _C.load()

    torch.manual_seed(2333)
    model, inputs, unit, units_per_step, metric, workload, gf_per_unit, tile_shape = make_workload(er, args.config, dev,
                                                                                                  args.batch, rank)
    model = model.to(dev).train()      # c2: R50 encoder + FarSegHead reference defaults
    ddp = model
    if use_ddp:
        if args.ddp == 'flat':   # the trainer's default exchange: one pack launch + one RCCL all-reduce per 64 MB bucket
            from ever_amd.trainer.grad_reducer import FlatGradDDP
            ddp = FlatGradDDP(model, bucket_cap_mb=64)
        else:
            ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank,
                                                            bucket_cap_mb=64, gradient_as_bucket_view=True)
    opt = er.opt.FusedSGD(model.parameters(), lr=0.007, momentum=0.9, weight_decay=1e-4)

    def eager_step(*data):
        out = ddp(*data)
        sum(v for k, v in out.items() if k.endswith('loss')).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return out

    if args.graph:
        if use_ddp:
            raise SystemExit('--graph: single-process only (no collective inside the captured step)')
        from ever_amd.core.graph import GraphedTrainStep
        graphed = GraphedTrainStep(eager_step, opt, modules=(model,))
        args.no_kernel_timer = True

        def step():
            graphed(*inputs)
    else:
        def step():
            eager_step(*inputs)

    def fence():
        torch.cuda.synchronize()
        if use_ddp:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # what enqueuing ONE step costs the host when nothing holds it back: the launch queue is empty (fence above), no event
    # timer, no synchronisation inside; median of three (`host_enqueue_ms_per_step` of the timed loop
    # reads ~ the GPU step time because the full queue back-pressures the host)
    unblocked = []
    for _ in range(3):
        t_h = time.perf_counter()
        step()
        unblocked.append(time.perf_counter() - t_h)
        fence()
    host_unblocked_ms = sorted(unblocked)[1] * 1e3
    # HIP-event timing of the launches costs 2-3 ms per step (an event pair around each of ~250 C-ABI calls), so it
    # samples two steps of the timed region rather than all of them.
    if use_ddp and hasattr(ddp, 'measure_exposed'):
        ddp.measure_exposed = True
    timer = None if args.no_kernel_timer else timing.KernelTimer()
    # With the weight gradients on their own stream a launch bracketed by events shares the chip with the other
    # stream's kernels and its duration says how the two split it, not how good the kernel is.  So the sampled steps of the
    # timed region are two: step 0 runs single-stream (the figures the roofline objects quote: each kernel alone), step K/2 as
    # every other step runs (the *_overlapped fields).  `value` includes all of them: an event pair around every launch costs
    # 2-3 ms of pipeline bubbles per sampled step, a single-stream step another 1.4 ms, so two sampled steps whatever K is.
    two_streams = timer is not None and HF.wgrad_stream_enabled() and not args.graph
    timer_ov = timing.KernelTimer() if two_streams else None
    # one step of each kind: every launch of a step is in it (159 + 84 convolution calls, ~110 BatchNorm calls), and a
    # sampled step costs 2-4 ms
    alone_at, ov_at = {0}, ({args.steps // 2} if (two_streams and args.steps >= 2) else set())
    sampled = sampled_ov = 0
    t0 = time.perf_counter()
    marks = []
    for i in range(args.steps):
        marks.append(time.perf_counter())
        if timer is not None and i in ov_at:
            with timer_ov:
                step()
            sampled_ov += 1
        elif timer is not None and i in alone_at:
            if two_streams:
                HF.set_wgrad_stream(False)
            with timer:
                step()
            if two_streams:
                HF.set_wgrad_stream(True)
            sampled += 1
        else:
            step()
    enqueued = time.perf_counter() - t0      # host time to enqueue the K steps (no synchronisation inside the loop)
    if os.environ.get('EVK_BENCH_MARKS'):
        marks.append(time.perf_counter())
        print('host ms per step:', [round((b - a) * 1e3, 1) for a, b in zip(marks, marks[1:])], file=sys.stderr)
    fence()
    elapsed = time.perf_counter() - t0
    overlapped = timer_ov.summary() if (timer_ov is not None and sampled_ov) else None
    if os.environ.get('EVK_BENCH_MARKS'):
        print('weight gradients side/main:', HF.wgrad_stream_stats, file=sys.stderr)
    per_rank = None
    if use_ddp:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
        # per-rank host cost and exposed all-reduce time: what a rank's Python thread needs to enqueue a
        # step (timed loop / empty queue) and how long its compute stream waited for the last gradient bucket
        mine = torch.tensor([enqueued / args.steps * 1e3, host_unblocked_ms,
                             ddp.exposed_ms() if hasattr(ddp, 'exposed_ms') else 0.0], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        per_rank = [[round(float(v), 3) for v in r.tolist()] for r in allr]

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        tiles_s = world * units_per_step * args.steps / elapsed
        line = {
            'metric': metric, 'value': round(tiles_s, 3 if tiles_s < 100 else 2), 'unit': unit,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
            'host_enqueue_ms_per_step': round(enqueued / args.steps * 1e3, 3),
            'host_unblocked_ms_per_step': round(host_unblocked_ms, 3),
            'host_cores': host_cores(), 'host_affinity': len(os.sched_getaffinity(0)),
            'host_pinned_cores': args.host_cores,   # (bench.py pins its enqueuing threads by default; the Launcher only under EVK_HOST_CORES,
                                                   #  because DataLoader workers inherit the mask: tiles/s is the same either way — the step is
                                                   #  GPU-bound — only host_*_ms_per_step move, DESIGN 2.10 "Host threads")
            'hip_graph': bool(args.graph),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if conv_math == 'bf16' else 'f32', 'data': 'synthetic',
            'config': {'workload': workload,
                       'global_batch': world * units_per_step, 'tile': tile_shape,
                       'parallelism': f'dp{world}' if world > 1 else 'single',
                       'arithmetic': ('fp32 in / fp32 accumulate / fp32 out; conv operands divided by a per-tensor power of two '
                                      'and split into 2 fp16 terms (22-bit operands), 3 fp16-MFMA partial products each '
                                      '(error < 2^-21 per product; as accurate vs fp64 as the 6-product bf16 split on every '
                                      'layer shape, tools/check_f16x2.py)') if conv_math == 'f16x2'
                       else ('fp32 in / fp32 accumulate / fp32 out; conv products from an exact 3-term bf16 split, '
                             '6 bf16-MFMA partial products each (error < 2^-24 per product)') if conv_math == 'bf16x3'
                       else ('fp32 tensors; conv operands rounded to bf16 once, 1 bf16-MFMA product, fp32 accumulate '
                             '(--mixed_precision bf16; BatchNorm, resampling, losses fp32)') if conv_math == 'bf16'
                       else 'fp32 MFMA (exact fmaf chain)'},
        }
        if gf_per_unit is not None:
            line['config']['whole_model_tflops'] = round(tiles_s * gf_per_unit / 1e3 / world, 2)
        if per_rank is not None:
            line['per_rank'] = {'host_enqueue_ms_per_step': [r[0] for r in per_rank],
                                'host_unblocked_ms_per_step': [r[1] for r in per_rank],
                                'exposed_allreduce_ms_per_step': [r[2] for r in per_rank]}
        if use_ddp:
            line['config']['gradient_exchange'] = ('FlatGradDDP: 64 MB buckets, one pack launch + one RCCL all-reduce per bucket'
                                                   if args.ddp == 'flat' else 'torch DistributedDataParallel') + \
                (' (world 1, forced by EVK_BENCH_FORCE_DDP)' if world == 1 else '')
        if timer is not None:
            x3 = conv_math in ('f16x2', 'bf16x3', 'bf16')
            passes = {'f16x2': 3, 'bf16x3': X3_PASSES, 'bf16': 1}.get(conv_math, 1)
            # split kernels: every algorithmic fp32 FLOP costs X3_PASSES bf16 MFMA FLOPs, so the roofline for
            # algorithmic FLOP/s is the dense bf16 MFMA peak / X3_PASSES
            peak = PEAK_BF16_MFMA_TFLOPS / passes if x3 else PEAK_FP32_MFMA_TFLOPS
            fam = timer.summary(peak_flops=peak * 1e12, peak_bytes=8.0e12)
            own = ('sum over the launches of each one\'s own roofline time, max(FLOP / MFMA peak of the arithmetic, algorithmic '
                   'bytes / 8 TB/s), over their measured time: the one-tap layers of stages 1-2 are bound by their bytes')
            peak_note = (f'dense bf16/fp16 MFMA peak {PEAK_BF16_MFMA_TFLOPS:.0f} TF / {passes} partial product(s) per product'
                         if x3 else 'dense f32-input MFMA peak (v_mfma_f32_32x32x2_f32)')
            convs = [fam[k] for k in ('conv_igemm', 'conv_wgrad', 'conv_igemm_f32', 'conv_wgrad_f32') if k in fam]
            if convs:   # algorithmic convolution work of one step as the spans saw it (forward + both gradients)
                line['config']['conv_gflop_per_step'] = round(sum(c['flops'] for c in convs) / max(1, sampled) / 1e9, 1)
                line['config']['conv_ms_per_step'] = round(sum(c['seconds'] for c in convs) / max(1, sampled) * 1e3, 3)
            if args.config == 'c5':
                # BASELINE.json configs[4] "channel-heavy implicit-GEMM stress": the Cin = 200 first convolution alone
                # (forward + weight gradient; the image needs no data gradient)
                f200 = 2.0 * 616 * 344 * 96 * 200 * 9
                recs = [r for r in timer.records if abs(r[1] - f200) < 1.0]
                if recs:
                    sec = sum(r[3].elapsed_time(r[4]) for r in recs) * 1e-3
                    ach = f200 * len(recs) / sec / 1e12
                    line['roofline_cin200'] = {'bound': 'mfma', 'kernel': 'first convolution, 3x3, Cin 200 -> 96 @616x344 '
                                               '(forward + weight gradient launches)', 'achieved': round(ach, 2),
                                               'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                                               'launches_per_step': len(recs) // max(1, sampled), 'sampled_steps': sampled,
                                               'avg_launch_us': round(sec / len(recs) * 1e6, 2)}
            ig = fam.get('conv_igemm' if x3 else 'conv_igemm_f32')
            if ig:
                ach = ig['flops'] / ig['seconds'] / 1e12
                # the committed PMC passes were collected under the default arithmetic
                traffic, traffic_file = pmc_traffic('conv_igemm') if (conv_math == 'f16x2' and args.config == 'c2') else (None, None)
                line['roofline'] = {
                    'bound': 'mfma',
                    'kernel': ('evk::conv3x3_wino_x3_kernel / conv3x3_halo_x3_kernel / conv1x1_ps2_kernel / conv1x1_dma_kernel / conv1x1_sp_kernel / conv_igemm_x3ws_kernel / conv_igemm_x3_kernel' if x3
                               else 'evk::conv_igemm_kernel') + ' (conv forward + data-gradient launches)',
                    'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'peak_note': peak_note,
                    'frac': round(ach / peak, 4), 'traffic': traffic,
                    'traffic_unit': f'HBM bytes per launch (PMC, {traffic_file})',
                    'algorithmic_bytes_per_launch': round(ig['bytes'] / ig['launches']),
                    'launches_per_step': ig['launches'] // max(1, sampled), 'sampled_steps': sampled,
                    'avg_launch_us': round(ig['seconds'] / ig['launches'] * 1e6, 2),
                    'algorithmic_gflop_per_launch': round(ig['flops'] / ig['launches'] / 1e9, 3),
                    'frac_of_launch_bounds': round(ig['bound_seconds'] / ig['seconds'], 4), 'frac_of_launch_bounds_note': own}
                if conv_math == 'f16x2' and args.config == 'c2':
                    busy, busy_file = pmc_busy('conv_igemm')
                    line['roofline']['mfma_busy'] = busy
                    line['roofline']['mfma_busy_note'] = f'matrix-pipe occupancy, time-weighted over the family (PMC, {busy_file})'
            wg = fam.get('conv_wgrad' if x3 else 'conv_wgrad_f32')
            if wg:
                ach = wg['flops'] / wg['seconds'] / 1e12
                wtraffic, _ = pmc_traffic('conv_wgrad') if (conv_math == 'f16x2' and args.config == 'c2') else (None, None)
                line['roofline_wgrad'] = {'bound': 'mfma',
                                          'kernel': ('evk::conv_wgrad_x3ws_kernel / conv_wgrad_x3_kernel' if x3 else
                                                     'evk::conv_wgrad_kernel') + ' (+split-K reduce)',
                                          'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                                          'frac': round(ach / peak, 4), 'traffic': wtraffic,
                                          'algorithmic_bytes_per_launch': round(wg['bytes'] / wg['launches']),
                                          'launches_per_step': wg['launches'] // max(1, sampled), 'sampled_steps': sampled,
                                          'avg_launch_us': round(wg['seconds'] / wg['launches'] * 1e6, 2),
                                          'algorithmic_gflop_per_launch': round(wg['flops'] / wg['launches'] / 1e9, 3),
                                          'frac_of_launch_bounds': round(wg['bound_seconds'] / wg['seconds'], 4)}
                if conv_math == 'f16x2' and args.config == 'c2':
                    line['roofline_wgrad']['mfma_busy'] = pmc_busy('conv_wgrad')[0]
            # traffic and algorithmic bytes on ONE denominator — per step (`traffic` is per KERNEL launch,
            # `algorithmic_bytes_per_launch` per C-ABI call, and a strided data gradient or a split-K weight gradient is
            # several kernels per call)
            for key, fname, e in (('roofline', 'conv_igemm', ig), ('roofline_wgrad', 'conv_wgrad', wg)):
                if key in line and e:
                    per_step = e['bytes'] / max(1, sampled)
                    line[key]['algorithmic_bytes_per_step'] = round(per_step)
                    kl = pmc_kernel_launches_per_step(fname)
                    if line[key].get('traffic') and kl:
                        line[key]['traffic_per_step'] = round(line[key]['traffic'] * kl)
                        line[key]['traffic_over_algorithmic'] = round(line[key]['traffic'] * kl / per_step, 3)
                        line[key]['traffic_per_step_note'] = (f'PMC bytes per kernel launch x {kl} kernel launches per step of the family '
                                                              '(profiles/r*_traffic.json); the two *_per_step fields are comparable')
            # every convolution launch of the step as one object, and the whole step against the same peak: `roofline` alone
            # (forward + data gradient) reads better than either
            if ig and wg:
                fl, sec = ig['flops'] + wg['flops'], ig['seconds'] + wg['seconds']
                line['roofline_conv_all'] = {
                    'bound': 'mfma', 'kernel': 'every convolution launch of the step (forward + data gradient + weight gradient, '
                                               'incl. split-K reduce, bias column sums and pack passes)',
                    'achieved': round(fl / sec / 1e12, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                    'frac': round(fl / sec / 1e12 / peak, 4), 'gflop_per_step': round(fl / max(1, sampled) / 1e9, 1),
                    'ms_per_step': round(sec / max(1, sampled) * 1e3, 3), 'sampled_steps': sampled,
                    'frac_of_launch_bounds': round((ig['bound_seconds'] + wg['bound_seconds']) / sec, 4)}
                step_tf = fl / max(1, sampled) / (ms * 1e-3) / 1e12
                line['roofline_step'] = {
                    'bound': 'mfma', 'kernel': 'the whole training step (algorithmic convolution FLOP of a step over ms_per_step: '
                                               'BatchNorm, resampling, losses, optimizer and every gap included)',
                    'achieved': round(step_tf, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(step_tf / peak, 4)}
            # the ResNet-50 encoder's convolutions alone (BASELINE.json north_star: >= 0.6 x MFMA roofline on this
            # stack): forward + data gradient + weight gradient of every `en.*` convolution, the 7x7 stem on the
            # exact-fp32 kernel included, all priced against the default arithmetic's peak
            enc = [fam.get('encoder/' + k) for k in ('conv_igemm', 'conv_wgrad', 'conv_igemm_f32', 'conv_wgrad_f32')]
            enc = [e for e in enc if e]
            if enc:
                fl, sec = sum(e['flops'] for e in enc), sum(e['seconds'] for e in enc)
                bsec = sum(e['bound_seconds'] for e in enc)
                parts = {}
                for k in ('conv_igemm', 'conv_wgrad', 'conv_igemm_f32', 'conv_wgrad_f32'):
                    e = fam.get('encoder/' + k)
                    if e:
                        parts[k] = {'tflops': round(e['flops'] / e['seconds'] / 1e12, 2),
                                    'ms_per_step': round(e['seconds'] / max(1, sampled) * 1e3, 3),
                                    'gflop_per_step': round(e['flops'] / max(1, sampled) / 1e9, 1)}
                line['roofline_encoder'] = {
                    'bound': 'mfma', 'kernel': 'all convolution launches of the ResNet-50 encoder (forward + data gradient '
                                               '+ weight gradient of en.*)',
                    'achieved': round(fl / sec / 1e12, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                    'frac': round(fl / sec / 1e12 / peak, 4),
                    'frac_of_launch_bounds': round(bsec / sec, 4),
                    'gflop_per_step': round(fl / max(1, sampled) / 1e9, 1), 'sampled_steps': sampled,
                    'ms_per_step': round(sec / max(1, sampled) * 1e3, 3), 'families': parts}
            if overlapped is not None:
                for key, fname in (('roofline', 'conv_igemm' if x3 else 'conv_igemm_f32'),
                                   ('roofline_wgrad', 'conv_wgrad' if x3 else 'conv_wgrad_f32')):
                    a = overlapped.get(fname)
                    if a and key in line:
                        ach = a['flops'] / a['seconds'] / 1e12
                        line[key].update({'achieved_overlapped': round(ach, 2), 'frac_overlapped': round(ach / peak, 4),
                                          'avg_launch_us_overlapped': round(a['seconds'] / a['launches'] * 1e6, 2)})
                line['roofline']['two_streams'] = (
                    f'weight gradients run on a second stream beside the main stream\'s kernels (DESIGN 2.8).  achieved / frac / '
                    f'avg_launch_us: each kernel alone, from the {sampled} sampled steps of the timed region that run single-stream; '
                    f'*_overlapped: the same launches in the {sampled_ov} sampled steps that run as every other step does (a launch '
                    'then shares the chip with the other stream\'s kernels).  rocprofv3 summaries of both: profiles/')
            for fam_name, label in (('bn', 'evk::bn_* (BatchNorm+residual+ReLU forward/backward passes)'),
                                    ('resample_loss', 'evk::bilinear_fwd/bwd + bce/dice kernels (upsample x2/x4, pixel losses)')):
                hb = fam.get(fam_name)
                if hb and hb['seconds'] > 0:
                    gbs = hb['bytes'] / hb['seconds'] / 1e9
                    line['roofline_hbm_' + fam_name] = {
                        'bound': 'hbm', 'kernel': label, 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                        'frac': round(gbs / PEAK_HBM_GBS, 4),
                        # PMC bytes are per KERNEL launch (a call is 2-3 kernels): compare with algorithmic_bytes_per_kernel
                        'traffic': (pmc_traffic(fam_name)[0] if (conv_math == 'f16x2' and args.config == 'c2') else None),
                        'traffic_unit': f'HBM bytes per KERNEL launch (PMC, {pmc_traffic(fam_name)[1]}; kernels listed there)',
                        'algorithmic_bytes_per_call': round(hb['bytes'] / hb['launches']),
                        'algorithmic_bytes_per_kernel': (round(hb['bytes'] / max(1, sampled) / pmc_kernel_launches_per_step(fam_name))
                                                         if pmc_kernel_launches_per_step(fam_name) else None),
                        'calls_per_step': hb['launches'] // max(1, sampled), 'sampled_steps': sampled,
                        'avg_call_us': round(hb['seconds'] / hb['launches'] * 1e6, 2),
                        'algorithmic_bytes_per_step': round(hb['bytes'] / max(1, sampled))}
                    o = line['roofline_hbm_' + fam_name]
                    if o['traffic'] and pmc_kernel_launches_per_step(fam_name):
                        o['traffic_per_step'] = round(o['traffic'] * pmc_kernel_launches_per_step(fam_name))
                        o['traffic_over_algorithmic'] = round(o['traffic_per_step'] / o['algorithmic_bytes_per_step'], 3)
                    a = overlapped.get(fam_name) if overlapped is not None else None
                    if a and a['seconds'] > 0:
                        g1 = a['bytes'] / a['seconds'] / 1e9
                        line['roofline_hbm_' + fam_name].update({'achieved_overlapped': round(g1, 1),
                                                                 'frac_overlapped': round(g1 / PEAK_HBM_GBS, 4)})
        if world == 1 and not args.graph and 'roofline' in line:
            line['roofline']['library_fp16_gemm'] = library_gemm_yardstick(dev)
        if world == 1 and not args.graph and not args.no_graph_line and not use_ddp:
            line['board_power'] = board_power_line(step, device_index=local_rank)
        if world == 1 and not use_ddp and not args.graph and not args.no_graph_line and conv_math == 'f16x2':
            line['hip_graph_replay'] = graph_replay_line(args)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline_child()
        out_line = json.dumps(line)
    if use_ddp:
        torch.distributed.destroy_process_group()   # RCCL prints its version banner here: keep the JSON line last
    if rank == 0:
        sys.stdout.flush()
        sys.stderr.flush()
        print(out_line, flush=True)


if __name__ == '__main__':
    main()
