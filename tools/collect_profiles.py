"""Copy the round's evidence from gpurun_out/ (tools/profile_all.sh) into profiles/<tag>_*: python tools/collect_profiles.py r05"""
import glob, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r05'
G, P = os.path.join(R, 'gpurun_out'), os.path.join(R, 'profiles')
pairs = {
    'prof_round/bench.json': 'bench.json', 'prof_round/bench_under_rocprof.json': 'bench_under_rocprof.json',
    'prof_round/bench_under_rocprof_single_stream.json': 'bench_under_rocprof_single_stream.json',
    'prof_round/kernel_stats.md': 'kernel_stats.md', 'prof_round/kernel_stats.csv': 'kernel_stats.csv',
    'prof_round/kernel_stats_single_stream.md': 'kernel_stats_single_stream.md',
    'prof_round/kernel_stats_single_stream.csv': 'kernel_stats_single_stream.csv',
    'prof_round/traffic.json': 'traffic.json', 'prof_round/stream_timeline.txt': 'stream_timeline.txt',
    f'pmc_bench_{tag}.txt': 'pmc_conv_kernels.txt', f'pmc_bench_{tag}.json': 'pmc_conv_kernels.json',
    'final/gputest_tail.txt': 'gputest_tail.txt', 'final/floor_table_c2.txt': 'floor_table_c2.txt',
    'final/bench_single_stream.json': 'bench_single_stream.json', 'final/bench_ddp_flat_world1.json': 'bench_ddp_flat_world1.json',
    'final/bench_conv_math_bf16.json': 'bench_conv_math_bf16.json', 'final/bench_conv_math_bf16x3.json': 'bench_conv_math_bf16x3.json',
    'final/bench_host_1core.json': 'bench_host_1core.json', 'final/bench_host_2cores.json': 'bench_host_2cores.json',
    'final/bench_host_full.json': 'bench_host_full.json', 'final/g_graph.json': 'bench_graph.json',
}
for c in ('c2', 'c3', 'c4', 'c5'):
    pairs[f'final/layer_table_{c}.txt'] = f'layer_table_{c}.txt'
    if c != 'c2':
        pairs[f'final/bench_{c}.json'] = f'bench_{c}.json'
n = 0
for src, dst in pairs.items():
    s = os.path.join(G, src)
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copyfile(s, os.path.join(P, f'{tag}_{dst}'))
        n += 1
    else:
        print('missing', src)
X = os.path.join(P, f'{tag}_experiments')
os.makedirs(X, exist_ok=True)
for name in ('final/power_probe.txt', 'final/dma_patterns.txt', 'final/dma_issue.txt', 'final/ab_c1sp.txt', 'sp_abl.txt', 'ab_c1_small.txt',
             'at_halo_pipe.txt', 'at_halo_old.txt', 'at_k1.txt', 'ab_bn_elems.txt', 'bn_big.txt', 'reduce_patterns.txt', 'kstats_ps2.md',
             'kstats_ps2off.md', 'power_phase.txt', 'power_step.txt', 'host_profile.txt', 'train_sanity.txt'):
    s = os.path.join(G, name)
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copyfile(s, os.path.join(X, os.path.basename(name)))
        n += 1
print(f'copied {n} files into profiles/{tag}_*')
