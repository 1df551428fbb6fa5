"""TEST INFRASTRUCTURE ONLY — pins the oracle against the REAL reference and writes tests/golden/*.

Runs in the build container only (needs /root/reference; never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

1. imports Z-Zheng/ever from /root/reference with three in-memory stub modules for its missing,
   path-irrelevant dependencies (wandb, prettytable, albumentations);
2. builds the reference ResNetEncoder + FarSegHead and the oracle restatement (oracle/farseg_ref.py),
   loads the SAME portable weights (oracle/portable.py) into both, runs forward + backward on CPU and
   requires bit-equality of logits, losses and every parameter gradient  -> the oracle is pinned;
3. writes the golden vectors (outputs only; weights/inputs are regenerated from the portable hash):
   per-op known answers, end-to-end logits / losses / gradient digests, LR-schedule and sampler tables.
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')


def _install_stubs():
    wandb = types.ModuleType('wandb')
    wandb.run = None
    for fn in ('log', 'init', 'login', 'finish'):
        setattr(wandb, fn, lambda *a, **k: None)
    sys.modules['wandb'] = wandb
    pt = types.ModuleType('prettytable')

    class PrettyTable:
        def __init__(self, field_names=None, **k):
            self.field_names, self._rows = list(field_names or []), []

        def add_row(self, r):
            self._rows.append(r)

        def get_string(self):
            return ''

    pt.PrettyTable = PrettyTable
    sys.modules['prettytable'] = pt
    alb = types.ModuleType('albumentations')
    alb.RandomScale = type('RandomScale', (), {})
    alb.DualTransform = type('DualTransform', (), {})
    albp = types.ModuleType('albumentations.pytorch')
    albp.ToTensorV2 = type('ToTensorV2', (), {})
    alb.pytorch = albp
    sys.modules['albumentations'] = alb
    sys.modules['albumentations.pytorch'] = albp


def import_reference():
    _install_stubs()
    sys.path.insert(0, REF)
    import ever  # noqa
    return ever


def grad_digest(named_params):
    """Compact, order-stable summary of a gradient set: L2 norm, sum, 4 strided samples and the projection on a
    hash-generated +-1 direction (oracle/portable.py: `sign_vector`) per tensor."""
    from oracle import portable
    d = {}
    for k, p in named_params:
        g = p.grad.detach().double().reshape(-1)
        idx = np.linspace(0, g.numel() - 1, 4).astype(np.int64)
        proj = float((g.numpy() * portable.sign_vector(k, g.numel())).sum())
        d[k] = [float(g.norm()), float(g.sum())] + [float(g[i]) for i in idx] + [proj]
    return d


BIAS_KEY = 'head.fpn_decoder.classifier.0.bias'


def build_pair(er, resnet_type, in_channels, num_classes=1, decoder_channels=256, classifier_kernel=1,
               relation_version='v1', classifier_bias=None):
    from ever.module.resnet import ResNetEncoder
    from ever.module.fs_relation import FarSegHead, FSRelationV2
    from oracle import farseg_ref, portable
    widths = (64, 128, 256, 512) if resnet_type in ('resnet18', 'resnet34') else (256, 512, 1024, 2048)
    en = ResNetEncoder(dict(resnet_type=resnet_type, in_channels=in_channels))
    head = FarSegHead(dict(
        fpn=dict(in_channels_list=widths, out_channels=256),
        fs_relation=dict(scene_embedding_channels=widths[-1], in_channels_list=(256,) * 4, out_channels=256,
                         scale_aware_proj=True),
        fpn_decoder=dict(in_channels=256, out_channels=decoder_channels, in_feat_output_strides=(4, 8, 16, 32),
                         out_feat_output_stride=4,
                         classifier_config=dict(scale_factor=4.0, num_classes=num_classes, kernel_size=classifier_kernel))))
    if relation_version == 'v2':
        # FarSeg++: the reference ships FSRelationV2 (fs_relation.py:76-163) but no head that composes it; the
        # composition is FarSegHead.forward (:175-181) with the relation module swapped.  Dropout2d p=0 makes the
        # training-mode run deterministic (the mask path is tested against its formula in test_next_rows_gpu.py).
        head.fs_relation = FSRelationV2(widths[-1], (256,) * 4, 256, scale_aware_proj=True)
        for mod in head.fs_relation.modules():
            if isinstance(mod, torch.nn.Dropout2d):
                mod.p = 0.0

    class RefModel(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.en, self.head = en, head

        def forward(self, x):
            return self.head(self.en(x))

    ref = RefModel()
    ora = farseg_ref.FarSegRef(resnet_type, in_channels, num_classes, decoder_channels, classifier_kernel,
                               relation_version=relation_version, dropout=0.0)
    assert list(ref.state_dict().keys()) == list(ora.state_dict().keys()), 'state-dict keys differ'
    filled = portable.fill_state_dict(ora.state_dict())
    if classifier_bias is not None:
        filled[BIAS_KEY] = np.asarray(classifier_bias, dtype=np.float32)
    farseg_ref.load_portable_weights(ref, filled)
    farseg_ref.load_portable_weights(ora, filled)
    return ref, ora


def pick_classifier_bias(er, name, x, cfg):
    """The fixture's classifier bias, moved so that NO pixel of the training-mode prediction sits near its decision
    boundary (SURVEY §7 'argmax bit-exactness near ties'): the threshold in the widest empty interval of the logits
    (one class), or the best of 400 hashed per-class bias vectors (several classes).  Returns the bias values."""
    from oracle import portable
    ref, _ = build_pair(er, *cfg)
    ref.train()
    with torch.no_grad():
        lg = ref(x).numpy()
    b0 = ref.state_dict()[BIAS_KEY].numpy().astype(np.float64)
    if lg.shape[1] == 1:
        s, half = portable.widest_gap_shift(lg)
        bias = (b0 + s).astype(np.float32)
    else:
        db, gap = portable.widest_gap_bias(lg, name)
        bias = (b0 + db.astype(np.float64)).astype(np.float32)
    return bias


def e2e_case(er, name, resnet_type, in_channels, n, hw, num_classes=1, decoder_channels=256, classifier_kernel=1,
             relation_version='v1'):
    import ever.module.loss as rloss
    import torch.nn.functional as F
    from oracle import portable
    torch.manual_seed(0)
    cfg = (resnet_type, in_channels, num_classes, decoder_channels, classifier_kernel, relation_version)
    x_np, y_np = portable.synthetic_batch(name, n, in_channels, hw, hw, num_classes)
    x, y = torch.from_numpy(x_np), torch.from_numpy(y_np)
    bias = pick_classifier_bias(er, name, x, cfg)
    ref, ora = build_pair(er, *cfg, classifier_bias=bias)
    ref.train()
    ora.train()
    lg_ref = ref(x)
    if num_classes == 1:
        losses_ref = dict(bce_loss=rloss.binary_cross_entropy_with_logits(lg_ref, y, ignore_index=255),
                          dice_loss=rloss.dice_loss_with_logits(lg_ref, y, ignore_index=255))
    else:
        losses_ref = dict(cls_loss=F.cross_entropy(lg_ref, y, ignore_index=255))
    sum(losses_ref.values()).backward()
    lg_ora = ora.logits(x)
    losses_ora = ora(x, y)
    sum(losses_ora.values()).backward()
    # ---- pin: restatement == reference, bit for bit
    lg_ora2 = ora.logits(x)  # BN running stats moved; logits in train mode do not depend on them
    assert torch.equal(lg_ref, lg_ora) and torch.equal(lg_ora, lg_ora2), f'{name}: logits differ from the reference'
    for k in losses_ref:
        assert torch.equal(losses_ref[k], losses_ora[k]), f'{name}: {k} differs from the reference'
    for (ka, pa), (kb, pb) in zip(ref.named_parameters(), ora.named_parameters()):
        assert ka == kb and torch.equal(pa.grad, pb.grad), f'{name}: grad of {ka} differs from the reference'
    print(f'[pin] {name}: oracle == reference (logits, losses, {len(list(ref.parameters()))} grads) bit-exact')
    # fp64 run of the SAME reference modules: measures how far fp32 rounding alone moves each
    # gradient on this input (tiny tiles give 8-sample BatchNorm statistics in layer4, which amplify
    # rounding by ~1e4); the GPU parity test sizes its gradient tolerance from this.
    ref64, _ = build_pair(er, *cfg, classifier_bias=bias)
    ref64 = ref64.double().train()
    lg64 = ref64(x.double())
    if num_classes == 1:
        l64 = (rloss.binary_cross_entropy_with_logits(lg64, y, ignore_index=255).double() +
               rloss.dice_loss_with_logits(lg64, y, ignore_index=255).double())
    else:
        l64 = F.cross_entropy(lg64, y, ignore_index=255)
    l64.backward()
    gnorm64 = {k: float(p.grad.norm()) for k, p in ref64.named_parameters()}
    # the fp64 run's digest as well: how far fp32 rounding alone moves the samples / projection of each tensor
    digest64 = grad_digest(ref64.named_parameters())
    logits_noise = float((lg64.detach() - lg_ref.detach().double()).abs().max() / lg64.detach().abs().max())
    rng = float(lg_ref.detach().abs().max())
    margin = portable.mask_margin(lg_ref.detach().numpy())
    m64 = portable.mask_margin(lg64.detach().numpy())
    flips64 = int(((lg_ref.detach().numpy() > 0) != (lg64.detach().numpy() > 0)).sum()) if num_classes == 1 else \
        int((lg_ref.detach().numpy().argmax(1) != lg64.detach().numpy().argmax(1)).sum())
    print(f'[margin] {name}: smallest decision margin {margin.min() / rng:.2e} of the logit range '
          f'({int((margin < 1e-3 * rng).sum())} of {margin.size} pixels inside 1e-3; fp64 reference: '
          f'{m64.min() / rng:.2e}, {flips64} fp32-vs-fp64 mask flips)')
    # eval-mode logits (running statistics after one training step)
    ref.eval()
    with torch.no_grad():
        lg_eval = ref(x)
    buffers = {k: v for k, v in ref.state_dict().items() if 'running_' in k}
    bdig = {k: [float(v.double().sum()), float(v.double().norm())] for k, v in buffers.items()}
    np.savez_compressed(os.path.join(OUT, f'e2e_{name}.npz'), logits=lg_ref.detach().numpy(),
                        logits_eval=lg_eval.numpy(),
                        **{k: np.float64(v.item()) for k, v in losses_ref.items()})
    meta = dict(resnet_type=resnet_type, in_channels=in_channels, n=n, hw=hw, num_classes=num_classes,
                decoder_channels=decoder_channels, classifier_kernel=classifier_kernel,
                relation_version=relation_version, classifier_bias=[float(b) for b in bias],
                min_margin_rel=float(margin.min() / rng), pixels_inside_1e3=int((margin < 1e-3 * rng).sum()),
                losses={k: float(v.item()) for k, v in losses_ref.items()},
                grads=grad_digest(ref.named_parameters()), grad_norm_fp64=gnorm64, grads_fp64=digest64, logits_fp32_vs_fp64=logits_noise,
                running=bdig,
                argmax_margin=float(margin.min()))
    with open(os.path.join(OUT, f'e2e_{name}.json'), 'w') as f:
        json.dump(meta, f)


def bf16_autocast_case(er, name):
    """the reference's OWN bf16 mode (core/launcher.py:40-80 wraps model + losses in
    torch.autocast(bfloat16); module/ops.py:152-166 and fpn.py:96-102 keep the resampling in fp32) on an existing
    fixture's weights, classifier bias and input, run on the CPU backend's autocast: logits and losses for
    tests/test_bf16_mode_gpu.py to hold `--mixed_precision bf16` against, plus how far this mode sits from the same
    reference in fp32 (the yardstick of the tolerance)."""
    import ever.module.loss as rloss
    from oracle import portable
    with open(os.path.join(OUT, f'e2e_{name}.json')) as f:
        meta = json.load(f)
    cfg = (meta['resnet_type'], meta['in_channels'], meta['num_classes'], meta['decoder_channels'],
           meta['classifier_kernel'], meta['relation_version'])
    x_np, y_np = portable.synthetic_batch(name, meta['n'], meta['in_channels'], meta['hw'], meta['hw'], meta['num_classes'])
    x, y = torch.from_numpy(x_np), torch.from_numpy(y_np)
    ref, _ = build_pair(er, *cfg, classifier_bias=np.asarray(meta['classifier_bias'], dtype=np.float32))
    ref.train()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        lg = ref(x)
        losses = dict(bce_loss=rloss.binary_cross_entropy_with_logits(lg, y, ignore_index=255),
                      dice_loss=rloss.dice_loss_with_logits(lg, y, ignore_index=255))
    lg32 = np.load(os.path.join(OUT, f'e2e_{name}.npz'))['logits']
    lg = lg.detach().float().numpy()
    dist = float(np.abs(lg - lg32).max() / np.abs(lg32).max())
    flips = int(((lg > 0) != (lg32 > 0)).sum())
    print(f'[bf16] {name}: reference under CPU autocast(bfloat16) vs its fp32 run: logits {dist:.2e} of the range, '
          f'{flips} of {lg.size} mask pixels differ; losses {[round(float(v), 5) for v in losses.values()]} '
          f'(fp32: {[round(v, 5) for v in meta["losses"].values()]})')
    np.savez_compressed(os.path.join(OUT, f'e2e_{name}_bf16autocast.npz'), logits=lg,
                        **{k: np.float64(float(v)) for k, v in losses.items()})
    with open(os.path.join(OUT, f'e2e_{name}_bf16autocast.json'), 'w') as f:
        json.dump(dict(source='reference ResNetEncoder + FarSegHead + loss under torch.autocast("cpu", torch.bfloat16), '
                              'weights / bias / input of the fp32 fixture', logits_vs_fp32=dist, mask_flips_vs_fp32=flips,
                       losses={k: float(v) for k, v in losses.items()}), f)


def op_kats(er):
    """Known answers of the in-tree loss / resampling ops on the inputs of SURVEY §8 (a10-a12, c3)."""
    import ever.module.loss as rloss
    import torch.nn.functional as F
    k = {}
    lg = torch.tensor([[.5, -1.], [2., 0.]]).reshape(1, 1, 2, 2)
    yb = torch.tensor([[1, 0], [1, 255]]).reshape(1, 2, 2)
    k['bce'] = rloss.binary_cross_entropy_with_logits(lg, yb).item()
    k['ls_bce'] = rloss.label_smoothing_binary_cross_entropy(lg, yb).item()
    k['dice_binary'] = rloss.dice_loss_with_logits(lg, yb).item()
    l3 = torch.tensor([[[1., 0.], [0., 2.]], [[0., 1.], [0., 0.]], [[-1., 0.], [3., 0.]]]).reshape(1, 3, 2, 2)
    y3 = torch.tensor([[0, 1], [2, 255]]).reshape(1, 2, 2)
    k['dice_3class'] = rloss.dice_loss_with_logits(l3, y3).item()
    k['dice_3class_ignore_ch0'] = rloss.dice_loss_with_logits(l3, y3, ignore_channel=0).item()
    k['ce_3class'] = F.cross_entropy(l3, y3, ignore_index=255).item()
    k['ls_ce_3class'] = rloss.label_smoothing_cross_entropy(l3, y3, ignore_index=255).item()
    k['tversky_a03'] = rloss.tversky_loss_with_logits(l3, y3, alpha=0.3).item()
    up = torch.nn.UpsamplingBilinear2d(scale_factor=2)(torch.tensor([[0., 1.], [2., 3.]]).reshape(1, 1, 2, 2))
    k['bilinear2x_of_0123'] = up.reshape(-1).tolist()
    # schedules (ever/opt/learning_rate.py)
    from ever.opt.learning_rate import PolyLearningRate, CosineAnnealingLearningRate, MultiStepLearningRate

    class FakeOpt:
        def __init__(self):
            self.param_groups = [dict(lr=None)]

    def lr_at(sched, step):
        o = FakeOpt()
        sched.step(step, o)
        return o.param_groups[0]['lr']

    poly = PolyLearningRate(0.007, 0.9, 30000, warmup=dict(type='linear', step=100, ratio=0.1))
    k['poly'] = {str(s): lr_at(poly, s) for s in (0, 50, 100, 101, 15000, 29999)}
    k['cosine'] = {str(s): lr_at(CosineAnnealingLearningRate(0.007, 30000, 1e-6), s) for s in (0, 7500, 30000)}
    ms = MultiStepLearningRate((60000, 80000), 0.02, 0.1)
    k['multistep'] = {str(s): lr_at(ms, s) for s in (0, 60000, 60001, 80001)}
    from ever.magic.bigimage.sliding_window import sliding_window
    k['sliding_window_1000x700_512_256'] = np.asarray(sliding_window((1000, 700), 512, 256)).tolist()
    with open(os.path.join(OUT, 'op_kats.json'), 'w') as f:
        json.dump(k, f, indent=1)
    print('[kat] wrote op_kats.json:', {a: b for a, b in k.items() if isinstance(b, float)})


def block_vectors(er):
    """Per-block golden vectors at toy shapes: FPN, FSRelation, AssymetricDecoder outputs + input grads,
    taken from the reference modules with portable weights."""
    from ever.module.fpn import FPN, AssymetricDecoder
    from ever.module.fs_relation import FSRelation
    from oracle import portable
    out = {}
    feats = [torch.from_numpy(portable.normalish(f'blk/f{i}', (2, c, s, s))).requires_grad_()
             for i, (c, s) in enumerate([(64, 16), (128, 8), (256, 4), (512, 2)])]
    fpn = FPN((64, 128, 256, 512), 64)
    fpn.load_state_dict({k: torch.from_numpy(v) for k, v in portable.fill_state_dict(fpn.state_dict()).items()})
    rel = FSRelation(512, (64,) * 4, 64, True)
    rel.load_state_dict({k: torch.from_numpy(v) for k, v in portable.fill_state_dict(rel.state_dict()).items()})
    dec = AssymetricDecoder(64, 32, classifier_config=dict(scale_factor=4.0, num_classes=3, kernel_size=3))
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in portable.fill_state_dict(dec.state_dict()).items()})
    for m in (fpn, rel, dec):
        m.train()
    p = fpn(feats)
    scene = torch.nn.functional.adaptive_avg_pool2d(feats[-1], 1)
    r = rel(scene, p)
    o = dec(r)
    w = torch.from_numpy(portable.normalish('blk/w', tuple(o.shape)))
    (o * w).sum().backward()
    for i, t in enumerate(p):
        out[f'fpn{i}'] = t.detach().numpy()
    for i, t in enumerate(r):
        out[f'rel{i}'] = t.detach().numpy()
    out['dec'] = o.detach().numpy()
    for i, f in enumerate(feats):
        out[f'dfeat{i}'] = f.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'blocks.npz'), **out)
    print('[blk] wrote blocks.npz')


def toy_dataset(n=4, c=4, hw=64):
    """n hash-generated (image, label) pairs: the dataset of the Launcher pin (both sides build it)."""
    from oracle import portable
    xs, ys = [], []
    for i in range(n):
        x, y = portable.synthetic_batch(f'toy/{i}', 1, c, hw, hw, 1)
        xs.append(torch.from_numpy(x[0]))
        ys.append(torch.from_numpy(y[0]))
    return xs, ys


def launcher_case(er):
    """Pin of SURVEY §8 c5: the REFERENCE Launcher drives the reference blocks (R18, 4-band, batch 2,
    SGD m=.9 wd=1e-4, poly lr .01/.9/3 iters, BCE+dice) for 3 iterations on CPU; the unsmoothed
    per-step losses, the lr sequence and the final weights' digest are the fixture that the build's
    own Launcher + Trainer must reproduce (tests/test_plumbing_cpu.py)."""
    import tempfile
    tb = types.ModuleType('torch.utils.tensorboard')
    tb.SummaryWriter = type('SummaryWriter', (), {'__init__': lambda s, *a, **k: None,
                                                  'add_scalar': lambda s, *a, **k: None,
                                                  'add_histogram': lambda s, *a, **k: None})
    sys.modules['torch.utils.tensorboard'] = tb
    import ever.module.loss as rloss
    from ever.core.launcher import Launcher
    from ever.opt.learning_rate import PolyLearningRate
    from ever.core.builder import make_optimizer
    from oracle import portable
    ref, _ = build_pair(er, 'resnet18', 4)

    class Model(er.ERModule):
        def __init__(self, config):
            super().__init__(config)
            self.en, self.head = ref.en, ref.head

        def forward(self, x, y=None):
            lg = self.head(self.en(x))
            if self.training:
                return dict(bce_loss=rloss.binary_cross_entropy_with_logits(lg, y, ignore_index=255),
                            dice_loss=rloss.dice_loss_with_logits(lg, y, ignore_index=255))
            return lg

        def set_default_config(self):
            self.config.update(dict())

    model = Model(dict())
    xs, ys = toy_dataset()
    loader = torch.utils.data.DataLoader(list(zip(xs, ys)), batch_size=2, shuffle=False)
    sched = PolyLearningRate(0.01, 0.9, 3)
    opt_cfg = er.config.AttrDict.from_dict(dict(type='sgd', params=dict(momentum=0.9, weight_decay=1e-4, lr=sched.base_lr)))
    opt = make_optimizer(opt_cfg, params=model.custom_param_groups())
    records = []
    with tempfile.TemporaryDirectory() as d:
        tl = Launcher(d, model, opt, sched, mixed_precision='fp32')
        orig = tl._logger.train_log

        def spy(**kw):
            records.append(dict(step=int(kw['step']), lr=float(kw['lr']), **{k: float(v) for k, v in kw['loss_dict'].items()}))
            return orig(**kw)

        tl._logger.train_log = spy
        tl.train_by_config(loader, config=er.config.AttrDict.from_dict(dict(num_iters=3, save_ckpt_interval_epoch=1000)))
        files = sorted(os.listdir(d))
        with open(os.path.join(d, 'checkpoint_info.json')) as f:
            index = json.load(f)
    digest = {k: [float(v.double().sum()), float(v.double().norm())] for k, v in model.state_dict().items()}
    with open(os.path.join(OUT, 'launcher_r18.json'), 'w') as f:
        json.dump(dict(records=records, files=[x for x in files if not x.endswith('.log')], index=index,
                       final_state=digest), f)
    print('[launcher] reference Launcher steps:', records)


def next_rows_kats(er):
    """losses (tversky / focal / sigmoid-focal / OHEM), PixelMetric formulas on a
    confusion matrix, sliding-window boxes.  Inputs from oracle/portable.py, outputs of the reference."""
    from oracle import portable
    from ever.module import loss as L
    from ever.metric.pixel import PixelMetric
    from ever.magic.bigimage.sliding_window import sliding_window
    out, arrays = {}, {}
    # --- losses on [2, C, 12, 10] logits
    for c in (1, 4):
        key = f'next_loss_c{c}'
        z = torch.from_numpy(portable.uniform(key, (2, c, 12, 10), -3.0, 3.0)).requires_grad_()
        y = torch.from_numpy(portable.integers(key + '_y', (2, 12, 10), max(c, 2)).astype(np.int64))
        y[0, :2, :3] = 255
        for alpha, beta, gamma in ((0.3, None, 1.0), (0.7, 0.5, 2.0)):
            z.grad = None
            v = L.tversky_loss_with_logits(z, y, alpha, beta, gamma, smooth_value=1.0, ignore_index=255, sync_statistics=False)
            v.backward()
            tag = f'tversky_c{c}_a{alpha}_g{gamma}'
            out[tag] = float(v)
            arrays[tag + '_grad'] = z.grad.numpy().copy()
    zf = torch.from_numpy(portable.uniform('next_focal', (2, 3, 9, 7), -4.0, 4.0)).requires_grad_()
    yf = torch.from_numpy((portable.uniform01('next_focal_y', 2 * 3 * 9 * 7) > 0.6).astype(np.float32).reshape(2, 3, 9, 7))
    for normalize in (False, True):
        zf.grad = None
        # the reference's normalize branch only broadcasts for already-flat inputs (loss.py:166-174)
        v = L.focal_loss(zf.reshape(-1), yf.reshape(-1), gamma=2.0, normalize=True) if normalize else L.focal_loss(zf, yf, gamma=2.0)
        v.backward()
        out[f'focal_norm{int(normalize)}'] = float(v)
        arrays[f'focal_norm{int(normalize)}_grad'] = zf.grad.numpy().copy()
    for alpha, gamma, red in ((-1.0, 2.0, 'mean'), (0.25, 1.5, 'sum')):
        zf.grad = None
        v = L.sigmoid_focal_loss(zf, yf, alpha, gamma, red)
        v.backward()
        tag = f'sigmoid_focal_a{alpha}_g{gamma}_{red}'
        out[tag] = float(v)
        arrays[tag + '_grad'] = zf.grad.numpy().copy()
    lo = torch.from_numpy(portable.uniform('next_ohem', (3, 50), 0.0, 2.0)).clone()
    lo[0, :7] = 0.0
    lo.requires_grad_()
    v = L.online_hard_example_mining(lo, 0.4)
    v.backward()
    out['ohem_0.4'] = float(v)
    arrays['ohem_0.4_grad'] = lo.grad.numpy().copy()
    # --- pixel metric
    for c in (2, 7):
        yt = portable.integers(f'next_cm_t{c}', (3, 40, 33), c).astype(np.int64)
        yp = portable.integers(f'next_cm_p{c}', (3, 40, 33), c).astype(np.int64)
        yp = np.where(portable.uniform01(f'next_cm_m{c}', yt.size).reshape(yt.shape) < 0.6, yt, yp)
        pm = PixelMetric(c)
        pm.forward(yt[:2], yp[:2])
        pm.forward(torch.from_numpy(yt[2:]), torch.from_numpy(yp[2:]))
        cm = pm.dense_cm
        tb = pm.summary_all(dense_cm=cm)
        out[f'metric_c{c}'] = dict(
            cm=cm.tolist(), iou=PixelMetric.compute_iou_per_class(cm).tolist(),
            f1=PixelMetric.compute_F_measure_per_class(cm).tolist(),
            precision=PixelMetric.compute_precision_per_class(cm).tolist(),
            recall=PixelMetric.compute_recall_per_class(cm).tolist(),
            oa=float(PixelMetric.compute_overall_accuracy(cm)), kappa=float(PixelMetric.cohen_kappa_score(cm)),
            table_fields=list(tb.field_names), table_rows=[[x if isinstance(x, str) else float(x) for x in r] for r in tb._rows])
    # --- sliding window
    sw = {}
    for size, k, st in (((1024, 1024), 512, 256), ((610, 340), (256, 128), (200, 100)), ((100, 90), 512, 256),
                        ((513, 700), 512, 512), ((64, 64), 64, 32)):
        sw[f'{size}|{k}|{st}'] = sliding_window(size, k, st).tolist()
    out['sliding_window'] = sw
    with open(os.path.join(OUT, 'next_kats.json'), 'w') as f:
        json.dump(out, f)
    np.savez_compressed(os.path.join(OUT, 'next_kats.npz'), **arrays)
    print('next-row KATs:', len(out), 'values,', len(arrays), 'arrays')


def api_holes_kats(er):
    """Round 6 (VERDICT r5 "missing" 3): the FPN's top blocks (reference fpn.py:109-141), the 'sum' / 'none' reductions of
    label_smoothing_cross_entropy and binary_cross_entropy_with_logits (loss.py:207-235), ResNetEncoder with a GroupNorm
    norm_layer (resnet.py:213-225).  Inputs / weights from oracle/portable.py, outputs of the reference."""
    import functools
    from oracle import portable
    from ever.module import fpn as RF
    from ever.module import loss as L
    from ever.module.resnet import ResNetEncoder
    arrays, out = {}, {}
    chans, sizes = (16, 32, 64, 128), (32, 16, 8, 4)
    for tag, top in (('maxpool', lambda: RF.LastLevelMaxPool()), ('p6p7_c5', lambda: RF.LastLevelP6P7(128, 32)),
                     ('p6p7_p5', lambda: RF.LastLevelP6P7(32, 32))):
        torch.manual_seed(0)
        m = RF.FPN(chans, 32, top_blocks=top())
        m.load_state_dict({k: torch.from_numpy(v) for k, v in portable.fill_state_dict(m.state_dict()).items()})
        xs = [torch.from_numpy(portable.normalish(f'fpn_top/x{i}', (2, c, s, s))).requires_grad_() for i, (c, s) in enumerate(zip(chans, sizes))]
        outs = m(xs)
        gouts = [torch.from_numpy(portable.normalish(f'fpn_top/{tag}/g{i}', tuple(o.shape))) for i, o in enumerate(outs)]
        torch.autograd.backward(outs, gouts)
        for i, o in enumerate(outs):
            arrays[f'fpn_{tag}/out{i}'] = o.detach().numpy()
        for i, x in enumerate(xs):
            arrays[f'fpn_{tag}/dx{i}'] = x.grad.numpy()
        for k, p_ in m.named_parameters():
            arrays[f'fpn_{tag}/grad/{k}'] = p_.grad.numpy()
    # --- label smoothing cross entropy
    z = torch.from_numpy(portable.uniform('ls_ce', (2, 5, 12, 10), -3.0, 3.0)).requires_grad_()
    y = torch.from_numpy(portable.integers('ls_ce_y', (2, 12, 10), 5).astype(np.int64))
    y[1, 3:6, 2:9] = 255
    for red in ('mean', 'sum'):
        z.grad = None
        v = L.label_smoothing_cross_entropy(z, y, eps=0.1, reduction=red, ignore_index=255)
        v.backward()
        out[f'ls_ce_{red}'] = float(v)
        arrays[f'ls_ce_{red}_grad'] = z.grad.numpy().copy()
    zf = torch.from_numpy(portable.uniform('ls_ce_flat', (40, 6), -3.0, 3.0)).requires_grad_()
    yf = torch.from_numpy(portable.integers('ls_ce_flat_y', (40,), 6).astype(np.int64))
    v = L.label_smoothing_cross_entropy(zf, yf, eps=0.2, reduction='none', ignore_index=255)   # (defined: nothing ignored)
    gv = torch.from_numpy(portable.normalish('ls_ce_flat_g', tuple(v.shape)))
    v.backward(gv)
    arrays['ls_ce_none'] = v.detach().numpy()
    arrays['ls_ce_none_grad'] = zf.grad.numpy().copy()
    # --- binary cross entropy
    zb = torch.from_numpy(portable.uniform('bce_none', (2, 1, 9, 11), -4.0, 4.0)).requires_grad_()
    yb = torch.from_numpy((portable.uniform01('bce_none_y', 2 * 9 * 11) > 0.6).astype(np.int64).reshape(2, 9, 11))
    yb[0, :3, 4:] = 255
    for tag, fn in (('bce_none', lambda: L.binary_cross_entropy_with_logits(zb, yb.reshape(2, 1, 9, 11).float(), 'none', 255)),
                    ('bce_none_pw', lambda: L.binary_cross_entropy_with_logits(zb, yb.reshape(2, 1, 9, 11).float(), 'none', 255,
                                                                               pos_weight=torch.tensor(2.5))),
                    ('lsbce_none', lambda: L.label_smoothing_binary_cross_entropy(zb, yb.reshape(2, 1, 9, 11).float(), 0.1, 'none', 255))):
        zb.grad = None
        v = fn()
        gv = torch.from_numpy(portable.normalish(tag + '_g', tuple(v.shape)))
        v.backward(gv)
        arrays[tag] = v.detach().numpy()
        arrays[tag + '_grad'] = zb.grad.numpy().copy()
    # --- ResNet-18 encoder with GroupNorm(8, C)
    torch.manual_seed(0)
    enc = ResNetEncoder(dict(resnet_type='resnet18', in_channels=4, pretrained=False,
                             norm_layer=functools.partial(torch.nn.GroupNorm, 8)))
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in portable.fill_state_dict(enc.state_dict()).items()})
    enc.train()
    x = torch.from_numpy(portable.normalish('gn_enc/x', (2, 4, 64, 64)))
    feats = enc(x)
    gouts = [torch.from_numpy(portable.normalish(f'gn_enc/g{i}', tuple(o.shape))) for i, o in enumerate(feats)]
    torch.autograd.backward(feats, gouts)
    for i, o in enumerate(feats):
        arrays[f'gn_enc/out{i}'] = o.detach().numpy()
    # (11 M parameters: digests — norm, sum, four samples, a +-1 projection per tensor — instead of the gradients themselves)
    out['gn_enc_grad_digest'] = grad_digest([(k, p_) for k, p_ in enc.named_parameters() if p_.grad is not None])
    with open(os.path.join(OUT, 'r6_api.json'), 'w') as f:
        json.dump(out, f)
    np.savez_compressed(os.path.join(OUT, 'r6_api.npz'), **arrays)
    print('round-6 API KATs:', len(out), 'values,', len(arrays), 'arrays')


def fsrel_v2_case(er):
    """FSRelationV2 (reference fs_relation.py:76-163) forward + backward with portable weights; Dropout2d is set to
    p = 0 so that the training-mode run is deterministic (the mask path is tested against its formula)."""
    from oracle import portable
    from ever.module.fs_relation import FSRelationV2
    for sap in (True, False):
        torch.manual_seed(0)
        m = FSRelationV2(128, (64, 64, 64, 64), 64, scale_aware_proj=sap)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in portable.fill_state_dict(m.state_dict()).items()})
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout2d):
                mod.p = 0.0
        m.train()
        scene = torch.from_numpy(portable.normalish('fsv2/scene', (2, 128, 1, 1))).requires_grad_()
        feats = [torch.from_numpy(portable.normalish(f'fsv2/f{i}', (2, 64, s, s))).requires_grad_()
                 for i, s in enumerate((16, 8, 4, 2))]
        outs = m(scene, feats)
        gouts = [torch.from_numpy(portable.normalish(f'fsv2/g{i}', tuple(o.shape))) for i, o in enumerate(outs)]
        torch.autograd.backward(outs, gouts)
        arrays = {f'out{i}': o.detach().numpy() for i, o in enumerate(outs)}
        arrays['dscene'] = scene.grad.numpy()
        for i, f in enumerate(feats):
            arrays[f'dfeat{i}'] = f.grad.numpy()
        for k, p_ in m.named_parameters():
            arrays['grad/' + k] = p_.grad.numpy()
        for k, b in m.named_buffers():
            arrays['buf/' + k] = b.numpy()
        m.eval()
        with torch.no_grad():
            for i, o in enumerate(m(scene, feats)):
                arrays[f'eval_out{i}'] = o.numpy()
        np.savez_compressed(os.path.join(OUT, f'fsrel_v2_sap{int(sap)}.npz'), **arrays)
        print('FSRelationV2 scale_aware_proj =', sap, ':', len(arrays), 'arrays')


E2E_CASES = [
    ('r18_4band_64', 'resnet18', 4, 2, 64, {}),
    ('r50_3band_64', 'resnet50', 3, 2, 64, {}),
    ('r50_3band_128', 'resnet50', 3, 2, 128, {}),
    ('r50_3band_64_c16', 'resnet50', 3, 2, 64, dict(num_classes=16, decoder_channels=128, classifier_kernel=3)),
    ('r50_3band_256', 'resnet50', 3, 2, 256, {}),                       # well-conditioned gradients: tight bound
    ('pp_r50_4band_64', 'resnet50', 4, 2, 64, dict(relation_version='v2')),   # FarSeg++ (config C3's model)
]


BF16_CASES = ('r18_4band_64', 'r50_3band_64')


def main():
    os.makedirs(OUT, exist_ok=True)
    er = import_reference()
    print('reference ever', er.__version__, 'torch', torch.__version__)
    torch.set_num_threads(8)
    only = sys.argv[1] if len(sys.argv) > 1 else None
    if only == 'launcher':
        launcher_case(er)
        return
    if only == 'e2e':
        for a in E2E_CASES:
            if len(sys.argv) < 3 or a[0] in sys.argv[2:]:
                e2e_case(er, *a[:5], **a[5])
        return
    if only == 'next':
        next_rows_kats(er)
        fsrel_v2_case(er)
        return
    if only == 'holes':
        api_holes_kats(er)
        return
    if only == 'bf16':
        for name in BF16_CASES:
            bf16_autocast_case(er, name)
        return
    op_kats(er)
    block_vectors(er)
    next_rows_kats(er)
    api_holes_kats(er)
    fsrel_v2_case(er)
    launcher_case(er)
    for a in E2E_CASES:
        e2e_case(er, *a[:5], **a[5])
    for name in BF16_CASES:
        bf16_autocast_case(er, name)
    with open(os.path.join(OUT, 'PROVENANCE.json'), 'w') as f:
        json.dump(dict(reference='Z-Zheng/ever', version=er.__version__, torch=torch.__version__,
                       generated_by='oracle/gen_golden.py',
                       note='outputs only; inputs and weights come from oracle/portable.py'), f)


if __name__ == '__main__':
    main()
