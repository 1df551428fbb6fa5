"""Pixel losses with ignore_index (API of reference ever/module/loss.py), HIP-backed masked reductions."""
from ..hip import functional as HF

__all__ = ['binary_cross_entropy_with_logits', 'dice_loss_with_logits', 'cross_entropy',
           'label_smoothing_cross_entropy', 'label_smoothing_binary_cross_entropy', 'soft_cross_entropy',
           'tversky_loss_with_logits', 'focal_loss', 'sigmoid_focal_loss', 'online_hard_example_mining',
           'cross_entropy_per_pixel']


def binary_cross_entropy_with_logits(output, target, reduction='mean', ignore_index=255, pos_weight=None):
    """reference loss.py:229-235 (reduction 'mean' | 'sum' | 'none' — 'none': one loss per NON-ignored pixel, a 1-D tensor
    in pixel order, as the reference's masked_select leaves it)"""
    return HF.bce_with_logits(output, target, ignore_index=ignore_index, pos_weight=pos_weight, reduction=reduction)


def dice_loss_with_logits(y_pred, y_true, smooth_value=1.0, ignore_index=255, ignore_channel=-1, *,
                          sync_statistics=True):
    """reference loss.py:54-75 ; sufficient statistics are summed across ranks before the ratio."""
    return HF.dice_loss_with_logits(y_pred, y_true, smooth_value, ignore_index, ignore_channel, sync_statistics)


def cross_entropy(output, target, ignore_index=255):
    """F.cross_entropy(output, target, ignore_index=ignore_index) as used by EVer model code."""
    return HF.cross_entropy(output, target, ignore_index=ignore_index)


def label_smoothing_cross_entropy(output, target, eps=0.1, reduction='mean', ignore_index=-1):
    """reference loss.py:207-219 (reduction 'mean' | 'sum' | 'none')"""
    return HF.cross_entropy(output, target, ignore_index=ignore_index, label_smoothing=eps, reduction=reduction)


def label_smoothing_binary_cross_entropy(output, target, eps=0.1, reduction='mean', ignore_index=255):
    """reference loss.py:222-226"""
    return HF.bce_with_logits(output, target, ignore_index=ignore_index, label_smoothing=eps, reduction=reduction)


def soft_cross_entropy(input, target):
    """reference loss.py:238-242"""
    return HF.soft_cross_entropy(input, target)


def tversky_loss_with_logits(y_pred, y_true, alpha, beta=None, gamma=1.0, smooth_value=1.0, ignore_index=255,
                             reduction='mean', *, sync_statistics=True):
    """reference loss.py:78-143"""
    return HF.tversky_loss_with_logits(y_pred, y_true, alpha, beta, gamma, smooth_value, ignore_index, reduction,
                                       sync_statistics)


def focal_loss(y_pred, y_true, gamma=2.0, normalize=False):
    """reference loss.py:158-176"""
    return HF.focal_loss(y_pred, y_true, gamma, normalize)


def sigmoid_focal_loss(y_pred, y_true, alpha=-1, gamma=2, reduction='mean'):
    """reference loss.py:179-201"""
    return HF.sigmoid_focal_loss(y_pred, y_true, alpha, gamma, reduction)


def online_hard_example_mining(losses, keep_ratio):
    """reference loss.py:146-155"""
    from ..hip import functional_next as HN
    return HN.online_hard_example_mining(losses, keep_ratio)


def cross_entropy_per_pixel(output, target, ignore_index=255):
    """F.cross_entropy(..., reduction='none'): the per-pixel losses OHEM selects from"""
    from ..hip import functional_next as HN
    return HN.cross_entropy_per_pixel(output, target, ignore_index)
