"""ReLU bits (include/ever_hip.h: evk_bn_fwd_train_parts_bits / evk_bn_bwd_bits / evk_conv2d_dgrad_f16x2_masked): the
BatchNorm + add + ReLU that ends a residual block (reference ever/module/_resnets.py:95-112) keeps one bit per output
element; its backward reads the bits instead of the output tensor, and hands the identity branch's gradient on unmasked
(a view of the incoming gradient + the bits) to be masked by its consumer.  The masked values are the same numbers either
way, so every gradient of a network must be BIT-identical with the mechanism on and off."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _unpack_bits(words, n4):
    """host model of csrc/common.hpp relu_bits_*: bool [n4, 4]"""
    w = words.cpu().numpy().astype(np.uint32)
    j = np.arange(n4)
    base = (j >> 6) * 8 + ((j >> 5) & 1) * 4
    out = np.zeros((n4, 4), dtype=bool)
    for e in range(4):
        out[:, e] = (w[base + e] >> (j & 31).astype(np.uint32)) & 1
    return out


@pytest.mark.parametrize('rows,c', [(4096, 256), (1000, 64), (77, 12)])
def test_forward_bits_are_the_sign_of_the_output_and_backward_takes_them(cuda, rows, c):
    from ever_amd import _C
    lib = _C.load()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(rows + c)
    x = torch.randn(rows, c, generator=g).to(cuda)
    res = torch.randn(rows, c, generator=g).to(cuda)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(cuda), (0.1 * torch.randn(c, generator=g)).to(cuda)
    # one record per channel over all rows: (count, mean, M2)
    parts = torch.stack([torch.full((c,), float(rows)), x.mean(0).cpu(), ((x - x.mean(0)) ** 2).sum(0).cpu()]).to(cuda).contiguous()
    wsb = lib.evk_bn_workspace_bytes(rows, c)
    ws = torch.empty(wsb, dtype=torch.uint8, device=cuda)
    y = torch.empty_like(x)
    mean, invstd = torch.empty(c, device=cuda), torch.empty(c, device=cuda)
    nwords = lib.evk_relu_bits_bytes(rows * c) // 4
    bits = torch.full((nwords,), -1, dtype=torch.int32, device=cuda)
    _C.call('evk_bn_fwd_train_parts_bits', x.data_ptr(), res.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, None, 0.1, 1e-5,
            y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), rows, c, 1, parts.data_ptr(), 1, ws.data_ptr(), wsb, None,
            bits.data_ptr(), st)
    torch.cuda.synchronize()
    n4 = rows * c // 4
    got = _unpack_bits(bits, n4)
    assert np.array_equal(got, (y > 0).cpu().numpy().reshape(n4, 4))
    # backward: bits instead of y, with and without the masked residual gradient written
    dy = torch.randn(rows, c, generator=g).to(cuda)
    outs = []
    for mode in ('y', 'bits', 'lazy'):
        dx, dres = torch.empty_like(x), torch.empty_like(x)
        dg, db = torch.empty(c, device=cuda), torch.empty(c, device=cuda)
        _C.call('evk_bn_bwd_bits', dy.data_ptr(), x.data_ptr(), y.data_ptr() if mode == 'y' else None, gamma.data_ptr(),
                beta.data_ptr(), mean.data_ptr(), invstd.data_ptr(), dx.data_ptr(), None if mode == 'lazy' else dres.data_ptr(),
                dg.data_ptr(), db.data_ptr(), rows, c, 1, 1, ws.data_ptr(), wsb, None,
                None if mode == 'y' else bits.data_ptr(), st)
        if mode == 'lazy':
            _C.call('evk_relu_bits_apply', dy.data_ptr(), bits.data_ptr(), dres.data_ptr(), dy.numel(), st)
        torch.cuda.synchronize()
        outs.append((dx, dres, dg, db))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)
    assert torch.equal(outs[0][1], dy * (y > 0))


def _grads(cuda, meta, bits, lazy):
    from ever_amd.hip import functional as F
    from oracle import portable
    from tests.test_e2e_gpu import _hip_model
    prev = F._RELU_BITS, F._LAZY_RES
    F._RELU_BITS, F._LAZY_RES = bits, lazy
    try:
        m = _hip_model(meta, cuda).train()
        x, y = portable.synthetic_batch('relubits', 2, meta['in_channels'], 64, 64, 1)
        x, y = torch.from_numpy(x).to(cuda), torch.from_numpy(y).to(cuda)
        losses = m.loss(m.head(m.en(x)), y)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    finally:
        F._RELU_BITS, F._LAZY_RES = prev


@pytest.mark.parametrize('resnet_type', ['resnet18', 'resnet50'])
def test_network_gradients_are_bit_identical_with_and_without_relu_bits(cuda, resnet_type):
    meta = dict(resnet_type=resnet_type, in_channels=3, num_classes=1, decoder_channels=256, classifier_kernel=1)
    from ever_amd.hip import functional as F
    ref = _grads(cuda, meta, False, False)
    # (on these 64 x 64 tiles the last stages' maps have so few rows that their convolutions run the small-M kernel, which
    # hands no statistics records over: those blocks keep the y-reading path — 4 of 8 / 13 of 16 blocks use the bits)
    for bits, lazy in ((True, False), (True, True)):
        F.relu_bits_stats.update({k: 0 for k in F.relu_bits_stats})
        got = _grads(cuda, meta, bits, lazy)
        bad = [k for k in ref if not torch.equal(ref[k], got[k])]
        assert not bad, (bits, lazy, bad[:5])
        st = dict(F.relu_bits_stats)
        assert st['forward'] >= 4, st
        if lazy:   # every such block hands its shortcut gradient on unmasked: to the fork node's data gradient, or — where
            # the shortcut is a convolution — to that convolution's BatchNorm; nothing had to be materialised
            assert st['lazy'] == st['forward'] and st['masked_dgrad'] + st['masked_bn'] == st['lazy'], st
            assert st['masked_dgrad'] >= 2 and st['masked_bn'] >= 1 and st['materialized'] == 0, st
        else:
            assert st['lazy'] == st['masked_dgrad'] == st['masked_bn'] == 0, st


def test_masked_accumulate_in_the_data_gradient(cuda):
    """dx = dgrad(dy) + (accum where its bit is set) equals dgrad with the pre-masked accum, bit for bit"""
    from ever_amd.hip import functional as F
    prev = F.set_conv_math('f16x2')
    try:
        g = torch.Generator().manual_seed(4)
        x = torch.randn(2, 64, 24, 24, generator=g).to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_()
        w = (torch.randn(128, 64, 1, 1, generator=g) / 8).to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_()

        class Conv:
            weight, bias, stride, padding, dilation = w, None, (1, 1), (0, 0), (1, 1)
        out = []
        for lazy in (False, True):
            prev_l = F._LAZY_RES
            F._LAZY_RES = lazy
            try:
                x.grad = w.grad = None
                h, s = F.conv2d_fork(x, Conv)
                # a stand-in for the block: BatchNorm(h) reduced to 64 channels is not needed — use s as the residual of a
                # BatchNorm over a second convolution's output with the same shape as x
                w2 = (torch.randn(64, 128, 1, 1, generator=torch.Generator().manual_seed(9)) / 11).to(cuda)
                w2 = w2.contiguous(memory_format=torch.channels_last)
                z = F.conv2d(h, w2, None, bn_stats=True)
                gm, bt = torch.ones(64, device=cuda), torch.zeros(64, device=cuda)
                yb = F.batch_norm_act(z, gm, bt, None, None, True, 0.1, 1e-5, residual=s, relu=True, lazy_res=True)
                (yb * yb).sum().backward()
                torch.cuda.synchronize()
                out.append((x.grad.clone(), w.grad.clone()))
            finally:
                F._LAZY_RES = prev_l
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    finally:
        F.set_conv_math(prev)
