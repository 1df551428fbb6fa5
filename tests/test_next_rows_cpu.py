"""SURVEY §8 f3 host pieces against vectors taken from the imported reference (tests/golden/next_kats.json,
made by oracle/gen_golden.py next): sliding-window enumeration and the PixelMetric formulas / table."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
with open(os.path.join(GOLD, 'next_kats.json')) as f:
    KATS = json.load(f)


def test_sliding_window_boxes_match_reference():
    from ever_amd.magic.bigimage import sliding_window
    for key, boxes in KATS['sliding_window'].items():
        size, k, st = (eval(s) for s in key.split('|'))
        got = sliding_window(size, k, st)
        assert got.tolist() == boxes, key


def test_pixel_metric_formulas_and_table_match_reference():
    from ever_amd.metric import PixelMetric
    for c in (2, 7):
        ref = KATS[f'metric_c{c}']
        cm = np.asarray(ref['cm'], dtype=np.float32)
        np.testing.assert_allclose(PixelMetric.compute_iou_per_class(cm), ref['iou'], rtol=1e-6)
        np.testing.assert_allclose(PixelMetric.compute_F_measure_per_class(cm), ref['f1'], rtol=1e-6)
        np.testing.assert_allclose(PixelMetric.compute_precision_per_class(cm), ref['precision'], rtol=1e-6)
        np.testing.assert_allclose(PixelMetric.compute_recall_per_class(cm), ref['recall'], rtol=1e-6)
        assert abs(PixelMetric.compute_overall_accuracy(cm) - ref['oa']) < 1e-6
        assert abs(PixelMetric.cohen_kappa_score(cm) - ref['kappa']) < 1e-6
        tb = PixelMetric(c).summary_all(dense_cm=cm)
        assert list(tb.field_names) == ref['table_fields']
        for got, want in zip(tb._rows, ref['table_rows']):
            for g, w in zip(got, want):
                if isinstance(w, str):
                    assert g == w
                else:
                    assert abs(float(g) - w) < 1e-6


def test_confusion_matrix_host_path_counts_like_reference():
    """CPU inputs are counted on the host (as the reference does); same inputs as the golden generator."""
    from oracle import portable
    from ever_amd.metric import PixelMetric
    for c in (2, 7):
        yt = portable.integers(f'next_cm_t{c}', (3, 40, 33), c).astype(np.int64)
        yp = portable.integers(f'next_cm_p{c}', (3, 40, 33), c).astype(np.int64)
        yp = np.where(portable.uniform01(f'next_cm_m{c}', yt.size).reshape(yt.shape) < 0.6, yt, yp)
        pm = PixelMetric(c)
        pm.forward(yt[:2], yp[:2])
        pm.forward(torch.from_numpy(yt[2:]), torch.from_numpy(yp[2:]))
        assert pm.dense_cm.tolist() == KATS[f'metric_c{c}']['cm']
        pm.reset()
        assert pm.dense_cm.sum() == 0
