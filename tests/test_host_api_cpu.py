"""CPU: host-side mirror of the reference API — config, registry, builder, schedules, samplers,
module construction / state-dict compatibility, C-ABI symbol table (no compute calls)."""
import ctypes
import json
import os
import re
import subprocess

import numpy as np
import pytest
import torch

import ever_amd as er
from ever_amd import _C
from oracle import farseg_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


# ------------------------------------------------------------------ config / registry / builder
def test_attrdict_recursive_update_and_cli_overrides(tmp_path):
    cfg = er.AttrDict.from_dict(dict(model=dict(type='X', params=dict(a=1, b=dict(c=2))), train=dict(num_iters=10)))
    cfg.update(dict(model=dict(params=dict(b=dict(d=3)))))
    assert cfg.model.params.a == 1 and cfg.model.params.b.c == 2 and cfg['model']['params']['b']['d'] == 3
    cfg.update_from_list(['train.num_iters', '20', 'model.params.b.c', '(1, 2)', 'model.name', 'farseg'])
    assert cfg.train.num_iters == 20 and cfg.model.params.b.c == (1, 2) and cfg.model.name == 'farseg'
    cfg.stages = [dict(k=1), dict(k=2)]
    cfg.update(dict(stages=[dict(k=5)]))
    assert isinstance(cfg.stages[0], er.AttrDict) and cfg.stages[0].k == 5
    p = tmp_path / 'c.pkl'
    cfg.to_pickle(str(p))
    back = er.config.import_config(str(p))
    assert back.train.num_iters == 20 and back.to_dict()['model']['params']['a'] == 1
    (tmp_path / 'cfg.py').write_text('config = dict(model=dict(type="M", params=dict(w=3)))\n')
    assert er.config.import_config(str(tmp_path / 'cfg.py')).model.params.w == 3


def test_registry_decorator_and_builder():
    reg = er.registry.Registry()

    @reg.register()
    def foo():
        return 1

    @reg.register('alias')
    @reg.register('alias2')
    def bar():
        return 2

    reg.register('direct', 3)
    assert reg['foo']() == 1 and reg['alias']() == 2 and reg['alias2']() == 2 and reg['direct'] == 3
    with pytest.raises(ValueError):
        er.builder.make_model(dict(type='__nope__', params={}))
    m = er.builder.make_model(dict(type='FarSegHead', params=dict()))
    assert isinstance(m, er.ERModule) and m.config.fpn.out_channels == 256
    opt = er.builder.make_optimizer(er.AttrDict.from_dict(dict(type='sgd', params=dict(lr=0.1, momentum=0.9),
                                                               grad_clip=dict(max_norm=35, norm_type=2))),
                                    params=m.parameters())
    assert opt.er_config.grad_clip.max_norm == 35


def test_lr_schedules_match_reference_tables():
    with open(os.path.join(GOLD, 'op_kats.json')) as f:
        k = json.load(f)

    class FakeOpt:
        def __init__(self):
            self.param_groups = [dict(lr=None)]

    def lr_at(s, step):
        o = FakeOpt()
        s.step(step, o)
        return o.param_groups[0]['lr']

    poly = er.builder.make_learningrate(dict(type='poly', params=dict(
        base_lr=0.007, power=0.9, max_iters=30000, warmup=dict(type='linear', step=100, ratio=0.1))))
    for s, v in k['poly'].items():
        assert lr_at(poly, int(s)) == pytest.approx(v, rel=1e-12)
    cos = er.builder.make_learningrate(dict(type='cosine', params=dict(base_lr=0.007, max_iters=30000, eta_min=1e-6)))
    for s, v in k['cosine'].items():
        assert lr_at(cos, int(s)) == pytest.approx(v, rel=1e-12)
    ms = er.builder.make_learningrate(dict(type='multistep', params=dict(steps=(60000, 80000), base_lr=0.02, gamma=0.1)))
    for s, v in k['multistep'].items():
        assert lr_at(ms, int(s)) == pytest.approx(v, rel=1e-12)
    # SURVEY §8 c3 literal pins
    assert lr_at(poly, 0) == pytest.approx(7e-4) and lr_at(poly, 15000) == pytest.approx(3.762496e-3, rel=1e-6)
    assert lr_at(cos, 7500) == pytest.approx(5.975020e-3, rel=1e-6)
    const = er.builder.make_learningrate(dict(type='constant', params=dict(base_lr=0.1)))
    o = FakeOpt()
    const.step(5, o)
    assert o.param_groups[0]['lr'] is None  # ConstantLearningRate never touches the optimizer


def test_step_distributed_sampler_shards_are_disjoint_and_step_seeded():
    from ever_amd.data import StepDistributedSampler
    s = StepDistributedSampler(range(10))
    s.set_step(3)
    assert list(s) == [6, 0, 3, 7, 8, 5, 1, 9, 2, 4]  # SURVEY §8 c3 (torch.randperm, world 1, step 3)
    shards = []
    for rank in range(4):
        s = StepDistributedSampler(range(10))
        s.num_replicas, s.rank = 4, rank
        s.num_samples, s.total_size = 3, 12
        s.set_step(7)
        shards.append(list(s))
    flat = sum(shards, [])
    assert len(flat) == 12 and set(flat) == set(range(10))  # wrap-padded to a multiple of world


# ------------------------------------------------------------------ modules: keys, counts, loud failure
def test_state_dict_keys_equal_the_reference_layout():
    hip = er.module.FarSeg(dict())
    ref = farseg_ref.FarSegRef('resnet50', 3, 1)  # pinned key-for-key to the reference in gen_golden.py
    assert list(hip.state_dict().keys()) == list(ref.state_dict().keys())
    for (k, a), (_, b) in zip(hip.state_dict().items(), ref.state_dict().items()):
        assert tuple(a.shape) == tuple(b.shape), k
    n_params = sum(p.numel() for p in hip.parameters())
    assert len(list(hip.parameters())) == 238 and abs(n_params / 1e6 - 33.875) < 0.01
    assert sum(p.numel() for p in hip.en.parameters()) == sum(p.numel() for p in ref.en.parameters())
    n_bn = sum(isinstance(m, torch.nn.BatchNorm2d) for m in hip.modules())
    n_conv = sum(isinstance(m, torch.nn.Conv2d) for m in hip.modules())
    assert (n_bn, n_conv) == (68, 85)
    for k in ('en.resnet.conv1.weight', 'en.resnet.layer1.0.bn1.running_mean', 'head.fpn.fpn_inner1.0.weight',
              'head.fs_relation.scene_encoder.0.0.weight', 'head.fpn_decoder.blocks.3.2.0.weight',
              'head.fpn_decoder.classifier.0.bias'):
        assert k in hip.state_dict()
    # reference weights load into the HIP model and back, conv weights stay OHWI in memory
    sd = {k: torch.randn_like(v) if v.is_floating_point() else v for k, v in ref.state_dict().items()}
    hip.load_state_dict(sd, strict=True)
    w = hip.en.resnet.layer1[0].conv2.weight
    assert w.permute(0, 2, 3, 1).is_contiguous() and torch.equal(w, sd['en.resnet.layer1.0.conv2.weight'])


def test_encoder_options_follow_the_reference():
    enc = er.module.ResNetEncoder(dict(resnet_type='resnet18', in_channels=4, output_stride=16, freeze_at=2,
                                       batchnorm_trainable=False))
    assert enc.resnet.conv1.weight.shape == (64, 4, 7, 7) and 'fc' not in dict(enc.resnet.named_children())
    l4 = enc.resnet.layer4[0]
    assert l4.conv1.stride == (1, 1) and l4.conv2.dilation == (2, 2) and l4.downsample[0].stride == (1, 1)
    assert not any(p.requires_grad for p in enc.resnet.layer1.parameters())
    assert not any(p.requires_grad for m in enc.modules() if isinstance(m, torch.nn.BatchNorm2d) for p in m.parameters())
    enc.train()
    assert all(not m.training for m in enc.modules() if isinstance(m, torch.nn.BatchNorm2d))
    with pytest.raises(ValueError):
        er.module.ResNetEncoder(dict(output_stride=4))


def test_hip_modules_refuse_cpu_tensors():
    from ever_amd.hip.functional import HipPathError
    m = er.module.FarSeg(dict(encoder=dict(resnet_type='resnet18'),
                              head=dict(fpn=dict(in_channels_list=(64, 128, 256, 512), out_channels=256),
                                        fs_relation=dict(scene_embedding_channels=512))))
    with pytest.raises(HipPathError):
        m(torch.randn(1, 3, 64, 64), torch.zeros(1, 64, 64, dtype=torch.long))
    with pytest.raises(HipPathError):
        er.module.loss.binary_cross_entropy_with_logits(torch.randn(1, 1, 4, 4), torch.zeros(1, 4, 4).long())


def test_functional_facade_reads_and_forwards_assignments():
    """hip/functional.py is a facade over the family modules (_base, streams, conv, norm, pointwise, losses): every public
    name and every `HF._X` switch callers use is visible on it, and ASSIGNING through it rebinds the name in each family
    module that holds it — a flag defined in _base.py and imported by value into norm.py must change in both, or the tests
    and tools that flip `HF._LAZY_RES` / `HF._rank_sum_hook` / `HF._cuda_set_stream` would flip nothing."""
    from ever_amd.hip import functional as HF
    from ever_amd.hip import _base, conv, losses, norm, pointwise, streams
    assert all(hasattr(HF, n) for n in HF.__all__)
    assert HF.conv2d is conv.conv2d and HF.relu is pointwise.relu and HF.batch_norm_act is norm.batch_norm_act
    assert HF.bce_with_logits is losses.bce_with_logits and HF.wait_wgrad_stream is streams.wait_wgrad_stream
    assert HF._WGRAD_STREAM is streams._WGRAD_STREAM and HF._ZERO_POOL is _base._ZERO_POOL        # containers: shared objects
    prev = HF._LAZY_RES
    try:
        HF._LAZY_RES = not prev
        assert _base._LAZY_RES is (not prev) and norm._LAZY_RES is (not prev) and HF._LAZY_RES is (not prev)
    finally:
        HF._LAZY_RES = prev
    assert norm._LAZY_RES is prev
    marker = object()
    try:
        HF._rank_sum_hook = marker
        assert losses._rank_sum_hook is marker
    finally:
        HF._rank_sum_hook = None
    saved = streams._cuda_set_stream
    try:
        HF._cuda_set_stream = None
        assert streams._cuda_set_stream is None
    finally:
        HF._cuda_set_stream = saved
    prev_math = HF.get_conv_math()
    try:
        HF.set_conv_math('f32')
        assert _base.get_conv_math() == 'f32' and not conv._f16x2()
    finally:
        HF.set_conv_math(prev_math)


def test_to_hip_retargets_a_stock_model():
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(True),
                              torch.nn.MaxPool2d(3, 2, 1))
    keys = list(net.state_dict())
    er.module.to_hip(net)
    assert isinstance(net, er.module.HipSequential) and isinstance(net[0], er.module.Conv2d)
    assert isinstance(net[1], er.module.BatchNorm2d) and list(net.state_dict()) == keys
    grouped = er.module.to_hip(torch.nn.Sequential(torch.nn.Conv2d(4, 4, 3, groups=2)))   # runs dense (ResNeXt)
    assert isinstance(grouped[0], er.module.Conv2d) and grouped[0].weight.shape == (4, 2, 3, 3)
    with pytest.raises(NotImplementedError):
        er.module.to_hip(torch.nn.Sequential(torch.nn.Conv2d(4, 4, 3, padding=1, padding_mode='reflect')))
    with pytest.raises(NotImplementedError):
        er.module.to_hip(torch.nn.Sequential(torch.nn.ConvTranspose2d(4, 4, 2, 2, groups=2)))


# ------------------------------------------------------------------ the C-ABI
def _header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'ever_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(evk_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    lib_path = _C.lib_path()
    assert os.path.exists(lib_path), 'libever_hip.so missing: run __graft_entry__.build()'
    declared = _header_symbols()
    assert len(declared) >= 40
    out = subprocess.run(['nm', '-D', '--defined-only', lib_path], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r'\bT (evk_[a-z0-9_]+)', out))
    missing = [s for s in declared if s not in exported]
    assert not missing, f'declared in include/ever_hip.h but not exported: {missing}'
    assert sorted(_C.SIGNATURES) == declared, 'ctypes signature table out of sync with the header'
    lib = _C.load()
    assert lib.evk_abi_version() == 21 and lib.evk_build_arch() == b'gfx950'
    # argument validation happens before any launch: safe without a GPU
    d = _C.ConvDesc(1, 8, 8, 3, 8, 8, 4, 1, 1, 1, 1, 0, 0, 1, 1)
    rc = lib.evk_conv2d_fwd(ctypes.byref(d), 1, 1, None, 1, 0, None)
    assert rc == -2 and b'multiple of 4' in lib.evk_last_error()   # EVK_E_UNSUPPORTED: Cin % 4
    assert lib.evk_bn_fwd_train(None, None, None, None, None, None, 0.1, 1e-5, None, None, None, 4, 4, 0, None, 0, None, None) == -1
    assert lib.evk_conv2d_wgrad_workspace_bytes(ctypes.byref(_C.ConvDesc(2, 8, 8, 64, 8, 8, 64, 3, 3, 1, 1, 1, 1, 1, 1))) > 0


def test_split_job_tables_are_host_side_and_match_the_plane_layouts():
    """evk_conv2d_split_jobs / _job_count / evk_split_job_pairs run on the host (no launch): job counts, plane offsets
    and sizes follow the layouts evk_conv2d_split_weight_bytes promises, and the layout signature the plane cache keys on
    separates geometries whose kernels read different planes (LDS-halo 3x3 vs generic) while merging those that do not."""
    from ever_amd.hip import weight_planes as wp
    lib = _C.load()

    def jobs_of(d, for_dgrad):
        n = lib.evk_conv2d_split_job_count(ctypes.byref(d), for_dgrad)
        arr = (_C.SplitJob * max(n, 1))()
        got = lib.evk_conv2d_split_jobs(ctypes.byref(d), 4096, for_dgrad, 1 << 20, arr, n)
        assert 0 <= got <= n
        return [arr[i] for i in range(got)], arr

    # 1x1 stride 2 (a ResNet down-sampling shortcut): forward = one job; data gradient = only the residue class that
    # has a tap produces planes, although the layout reserves room for all four
    d = _C.ConvDesc(16, 128, 128, 256, 64, 64, 512, 1, 1, 2, 2, 0, 0, 1, 1)
    f, _keep_f = jobs_of(d, 0)
    assert len(f) == 1 and f[0].kind == 0 and lib.evk_split_job_pairs(ctypes.byref(f[0])) == 512 * 256 // 2
    g, _keep_g = jobs_of(d, 1)
    assert lib.evk_conv2d_split_job_count(ctypes.byref(d), 1) == 4 and len(g) == 1 and g[0].kind == 1
    assert lib.evk_split_job_pairs(ctypes.byref(g[0])) == 256 * 512 // 2
    # 3x3 stride 2: four residue classes with 4 / 2 / 2 / 1 taps, laid out back to back inside the plane buffer
    d = _C.ConvDesc(2, 32, 32, 64, 16, 16, 128, 3, 3, 2, 2, 1, 1, 1, 1)
    g, _keep = jobs_of(d, 1)
    assert len(g) == 4
    offs = [j.out - (1 << 20) for j in g]
    assert offs == sorted(offs) and offs[0] == 0
    total = lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 1)
    last_bytes = 3 * 64 * g[-1].arg[10] * 2
    assert offs[-1] + last_bytes == total
    assert sum(lib.evk_split_job_pairs(ctypes.byref(j)) for j in g) == 64 * 128 * 9 // 2   # K = taps*Cout is a multiple of 32 here
    # 3x3 'same': big maps take the LDS-halo layout, small grids the generic one -> different cache signatures;
    # two batch sizes that both take the halo kernel share one
    big = _C.ConvDesc(16, 128, 128, 256, 128, 128, 256, 3, 3, 1, 1, 1, 1, 1, 1)
    big2 = _C.ConvDesc(8, 128, 128, 256, 128, 128, 256, 3, 3, 1, 1, 1, 1, 1, 1)
    tiny = _C.ConvDesc(1, 8, 16, 256, 8, 16, 256, 3, 3, 1, 1, 1, 1, 1, 1)
    assert jobs_of(big, 0)[0][0].kind == 2 and jobs_of(tiny, 0)[0][0].kind == 0
    assert wp._layout(big, 0)[0] == wp._layout(big2, 0)[0] != wp._layout(tiny, 0)[0]
    assert wp._layout(big, 0)[1] == lib.evk_conv2d_split_weight_bytes(ctypes.byref(big), 0)
    # argument checking
    arr = (_C.SplitJob * 1)()
    assert lib.evk_conv2d_split_jobs(ctypes.byref(d), 4096, 1, 1 << 20, arr, 1) == -1   # needs room for 4 jobs
    assert lib.evk_conv2d_split_multi(None, None, 0, None) == 0
    assert lib.evk_conv2d_split_multi(None, None, 5, None) == -1


def test_pending_log_total_is_the_sum_of_the_losses_only():
    """reference launcher.py:203-222: total_loss sums the `*loss` entries; averaged tensors (grad_norm from grad_clip,
    tensor metrics) and python extras are merged afterwards and never enter the total."""
    import torch
    from ever_amd.core.launcher import Launcher
    lz = Launcher.__new__(Launcher)          # only _start_log / log_info_dict are exercised (no model, no files)
    out = Launcher.log_info_dict(lz, {'cls_loss': torch.tensor(1.0), 'dice_loss': torch.tensor(0.25),
                                      'grad_norm': torch.tensor(5.0), 'acc': torch.tensor([0.25, 0.75]), 'n': 3})
    assert out['total_loss'] == 1.25
    assert out['cls_loss'] == 1.0 and out['dice_loss'] == 0.25 and out['grad_norm'] == 5.0 and out['acc'] == 0.5
    assert out['n'] == 3


def test_launcher_amp_flags_for_hip_models(tmp_path):
    """a9: `--mixed_precision` on the HIP path: bf16 selects the plain-bf16 convolution kernels (no autocast, no scaler);
    fp16 keeps the default arithmetic — already fp16 MFMAs on scaled, split operands — and adds the reference's GradScaler
    protocol (launcher.py:46-80)."""
    import torch
    import pytest
    import ever_amd as er
    from ever_amd.core.launcher import Launcher
    from ever_amd.hip import functional as HF
    hip = torch.nn.Sequential(er.module.Conv2d(8, 8, 1))
    prev = HF.get_conv_math()
    try:
        lz = Launcher(str(tmp_path), hip, torch.optim.SGD(hip.parameters(), lr=0.1), None, mixed_precision='fp16')
        assert HF.get_conv_math() == 'f16x2' and lz._amp and lz._amp is not True and lz.scaler is not None
    finally:
        HF.set_conv_math(prev)
    try:   # bf16 = the plain-bf16 convolution arithmetic, no autocast region, no GradScaler
        lz = Launcher(str(tmp_path), hip, torch.optim.SGD(hip.parameters(), lr=0.1), None, mixed_precision='bf16')
        assert HF.get_conv_math() == 'bf16' and lz._amp is False and lz.scaler is None
    finally:
        HF.set_conv_math(prev)
    stock = torch.nn.Sequential(torch.nn.Conv2d(8, 8, 1))       # stock torch models keep the reference's autocast
    Launcher(str(tmp_path), stock, torch.optim.SGD(stock.parameters(), lr=0.1), None, mixed_precision='bf16')


def test_fused_sgd_relays_momentum_buffers_of_a_torch_sgd_checkpoint():
    """Resuming from a reference / torch.optim.SGD checkpoint: its momentum buffers are NCHW-dense while the HIP
    convolution weights are channels_last (OHWI memory).  The fused kernel pairs elements by memory offset, so the
    buffers must be re-laid to the parameter's strides (same logical values)."""
    import torch
    import ever_amd as er
    conv = er.module.Conv2d(3, 5, 3)
    assert conv.weight.stride() == (27, 1, 9, 3)
    stock = torch.nn.Conv2d(3, 5, 3)
    stock.load_state_dict(conv.state_dict())
    ref_opt = torch.optim.SGD(stock.parameters(), lr=0.1, momentum=0.9)
    stock(torch.randn(2, 3, 8, 8)).sum().backward()
    ref_opt.step()
    state = ref_opt.state_dict()
    assert state['state'][0]['momentum_buffer'].stride() == (27, 9, 3, 1)
    opt = er.opt.FusedSGD(conv.parameters(), lr=0.1, momentum=0.9)
    opt.load_state_dict(state)
    buf = opt.state[conv.weight]['momentum_buffer']
    assert buf.stride() == conv.weight.stride()
    assert torch.equal(buf, state['state'][0]['momentum_buffer'])           # logical values untouched
    assert torch.equal(opt.state[conv.bias]['momentum_buffer'], state['state'][1]['momentum_buffer'])


def test_farsegpp_head_builds_fsrelation_v2_with_reference_keys():
    import ever_amd as er
    from ever_amd.module.fs_relation import FSRelation, FSRelationV2
    assert isinstance(er.module.FarSegHead(dict()).fs_relation, FSRelation)
    h = er.module.FarSegPPHead(dict())
    assert isinstance(h.fs_relation, FSRelationV2)
    assert isinstance(er.module.FarSegHead(dict(relation_version='v2')).fs_relation, FSRelationV2)
    keys = set(h.state_dict())
    assert {'fs_relation.project.0.0.weight', 'fs_relation.project.3.1.running_var',
            'fs_relation.scene_encoder.2.1.weight', 'fs_relation.scene_encoder.0.4.bias'} <= keys
    from oracle import farseg_ref
    ora = farseg_ref.FarSegHeadRef(relation_version='v2')
    assert list(ora.state_dict().keys()) == list(h.state_dict().keys())
    assert er.registry.MODEL['FarSegPP'] is er.module.FarSegPP


def test_weight_scale_slots_are_reused_when_weights_die():
    """hip/weight_planes.py: one word of the absmax array per live weight tensor; a long session (many models built and
    dropped) must not run out of words."""
    import gc
    import torch
    from ever_amd.hip import weight_planes as wp
    saved = (wp._SLOTS, wp._next_slot[0], list(wp._free_slots))
    wp.clear()
    wp._SLOTS = 4
    try:
        ts = [torch.zeros(1) for _ in range(4)]
        assert sorted(wp._slot(t) for t in ts) == [0, 1, 2, 3]
        assert wp._slot(ts[2]) == wp._slot(ts[2])
        dead = wp._slot(ts[1])
        del ts[1]
        gc.collect()
        t = torch.zeros(1)
        assert wp._slot(t) == dead
        import pytest
        with pytest.raises(RuntimeError, match='live weight tensors'):
            wp._slot(torch.zeros(1))
    finally:
        wp.clear()
        wp._SLOTS, wp._next_slot[0] = saved[0], saved[1]
        wp._free_slots.extend(saved[2])


def test_pin_host_threads_groups_by_rank_and_restores_launch_mask():
    """core/device.py: rank r takes the r-th group of the launch CPU set; cores=0 restores the launch mask (the HIP runtime may
    widen it when it initialises); threads started afterwards inherit it."""
    import os
    import threading
    from ever_amd.core.device import launch_affinity, pin_host_threads
    if not hasattr(os, 'sched_setaffinity'):
        pytest.skip('no affinity call on this OS')
    allowed = sorted(launch_affinity())
    try:
        if len(allowed) >= 4:
            a, b = pin_host_threads(0, 2), pin_host_threads(1, 2)
            assert sorted(a) == allowed[0:2] and sorted(b) == allowed[2:4]
            seen = []
            t = threading.Thread(target=lambda: seen.append(frozenset(os.sched_getaffinity(0))))
            t.start(); t.join()
            assert seen[0] == b
        assert sorted(pin_host_threads(0, len(allowed) + 3)) == allowed      # more than there are: the whole launch set
        os.environ['LOCAL_WORLD_SIZE'] = str(len(allowed))                   # no room for a group of two per local rank: nobody narrowed
        try:
            assert len(allowed) < 2 or sorted(pin_host_threads(0, 2)) == allowed
        finally:
            del os.environ['LOCAL_WORLD_SIZE']
    finally:
        assert sorted(pin_host_threads(0, 0)) == allowed
