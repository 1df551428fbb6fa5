"""tools/families.py is the ONE kernel-family table of the profile tools (a new kernel was dispatched but
matched none of the three tools' private pattern lists).  Every `__global__` kernel of ever_amd/csrc must be claimed by a
family, the convolution kernels by the convolution families, and the partition helper must be exact."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import families  # noqa: E402


def _kernels():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, 'ever_amd', 'csrc')):
        if fn.endswith('.hip'):
            src = open(os.path.join(ROOT, 'ever_amd', 'csrc', fn)).read()
            for m in re.finditer(r'__global__(?:\s+__launch_bounds__\([^;{]*?\))?\s+void\s+(\w+)\s*\(', src):
                names.add(m.group(1))
    return names


def test_every_kernel_of_the_library_has_a_family():
    ks = _kernels()
    assert len(ks) > 80
    # launch-probe / runtime helpers that are not part of a training step
    helpers = {k for k in ks if k.startswith(('spin_', 'evk_probe', 'overlap_', 'census_'))}
    unclaimed = sorted(k for k in ks - helpers if families.family_of('evk::' + k + '(args)') is None)
    assert not unclaimed, unclaimed


def test_convolution_kernels_are_in_the_convolution_families():
    for k in _kernels():
        fam = families.family_of('void evk::' + k + '<128, true>(evk::IGemmArgs)')
        if k.startswith(('conv1x1_', 'conv3x3_halo', 'conv_igemm')):
            assert fam == 'conv_igemm', (k, fam)
        if k.startswith('conv_wgrad'):
            assert fam == 'conv_wgrad', (k, fam)
        if k.startswith('bn_'):
            assert fam == 'bn', (k, fam)


def test_split_is_an_exact_partition():
    rows = [('void evk::conv1x1_ps2_kernel<true, false>(evk::IGemmArgs)', 3, 10.0), ('evk::splitk_reduce_kernel(float*)', 5, 2.0),
            ('void at::native::vectorized_elementwise_kernel<4>', 7, 1.5), ('evk::bn_apply_kernel<true>(float*)', 2, 4.0)]
    fam, rest = families.split(rows)
    assert [r[0] for r in fam['conv_igemm']] == [rows[0][0]] and [r[0] for r in fam['conv_wgrad_aux']] == [rows[1][0]]
    assert [r[0] for r in fam['bn']] == [rows[3][0]] and [r[0] for r in rest] == [rows[2][0]]
