"""PixelMetric: IoU / F1 / precision / recall / OA / kappa from the confusion matrix (API and formulas of
reference ever/metric/pixel.py:14-206; EPS = 1e-7 as there).  Counting runs on the GPU (confusion_matrix.py);
the C x C summary arithmetic is host numpy, identical to the reference's."""
import logging
import os
import time

import numpy as np

from .confusion_matrix import ConfusionMatrix
from ..core.dist import is_main_process, all_gather

EPS = 1e-7

__all__ = ['PixelMetric', 'AccTable']


class AccTable(object):
    """Minimal stand-in for the prettytable-based table of the reference (same accessors)."""

    def __init__(self, field_names=None):
        self.field_names = list(field_names or [])
        self._rows = []

    def add_row(self, row):
        self._rows.append(list(row))

    @staticmethod
    def _get_data(data, class_index=None):
        if isinstance(class_index, int):
            return data[class_index]
        if isinstance(class_index, (list, tuple)):
            return [data[c] for c in class_index]
        return data

    def get(self, col_name, row_index=None):
        idx = self.field_names.index(col_name)
        return self._get_data([r[idx] for r in self._rows], row_index)

    def f1(self, class_index=None):
        return self.get('f1', class_index)

    def iou(self, class_index=None):
        return self.get('iou', class_index)

    def precision(self, class_index=None):
        return self.get('precision', class_index)

    def recall(self, class_index=None):
        return self.get('recall', class_index)

    def get_string(self):
        cols = [self.field_names] + [[str(x) for x in r] for r in self._rows]
        widths = [max(len(str(row[i])) for row in cols) for i in range(len(self.field_names))]
        line = '+' + '+'.join('-' * (w + 2) for w in widths) + '+'
        fmt = lambda row: '| ' + ' | '.join(str(x).center(w) for x, w in zip(row, widths)) + ' |'
        return '\n'.join([line, fmt(self.field_names), line] + [fmt(r) for r in cols[1:]] + [line])

    __str__ = get_string

    def to_dataframe(self):
        import pandas as pd
        return pd.DataFrame(self._rows, columns=self.field_names)

    def to_csv(self, csv_file):
        self.to_dataframe().to_csv(csv_file, index=False)


class PixelMetric(ConfusionMatrix):
    def __init__(self, num_classes, logdir=None, logger=None, class_names=None):
        super(PixelMetric, self).__init__(num_classes)
        if logdir is not None:
            os.makedirs(logdir, exist_ok=True)
        self.logdir = logdir
        if logdir is not None and logger is None:
            from ..core.logger import get_console_file_logger
            self._logger = get_console_file_logger('PixelMetric', logging.INFO, self.logdir)
        else:
            self._logger = logger
        self._class_names = class_names
        if class_names:
            assert num_classes == len(class_names)

    @property
    def logger(self):
        return self._logger

    # ---- formulas (row = ground truth, column = prediction)
    @staticmethod
    def compute_iou_per_class(confusion_matrix):
        over_pred = np.sum(confusion_matrix, axis=0)
        over_true = np.sum(confusion_matrix, axis=1)
        diag = np.diag(confusion_matrix)
        return diag / (over_pred + over_true - diag + EPS)

    @staticmethod
    def compute_recall_per_class(confusion_matrix):
        return np.diag(confusion_matrix) / (np.sum(confusion_matrix, axis=1) + EPS)

    @staticmethod
    def compute_precision_per_class(confusion_matrix):
        return np.diag(confusion_matrix) / (np.sum(confusion_matrix, axis=0) + EPS)

    @staticmethod
    def compute_overall_accuracy(confusion_matrix):
        return np.sum(np.diag(confusion_matrix)) / (np.sum(confusion_matrix) + EPS)

    @staticmethod
    def compute_F_measure_per_class(confusion_matrix, beta=1.0):
        p = PixelMetric.compute_precision_per_class(confusion_matrix)
        r = PixelMetric.compute_recall_per_class(confusion_matrix)
        return (1 + beta ** 2) * p * r / ((beta ** 2) * p + r + EPS)

    @staticmethod
    def cohen_kappa_score(cm_th):
        cm_th = cm_th.astype(np.float32)
        n = cm_th.shape[0]
        sum0, sum1 = cm_th.sum(axis=0), cm_th.sum(axis=1)
        expected = np.outer(sum0, sum1) / (np.sum(sum0) + EPS)
        w = np.ones([n, n])
        w.flat[::n + 1] = 0
        return 1. - np.sum(w * cm_th) / (np.sum(w * expected) + EPS)

    def _gathered(self):
        return sum(all_gather(self.dense_cm))  # multi-GPU evaluation: every rank holds a disjoint shard

    def _log_summary(self, table, dense_cm):
        if self.logger is not None:
            self.logger.info('\n' + table.get_string())
            if self.logdir is not None:
                cm_dir = os.path.join(self.logdir, 'cm')
                os.makedirs(cm_dir, exist_ok=True)
                stamp = time.strftime('%Y-%m-%d-%H:%M:%S', time.localtime())
                np.save(os.path.join(cm_dir, f'confusion_matrix-{stamp}-{time.time()}.npy'), dense_cm)
        else:
            print(table)

    def summary_iou(self):
        dense_cm = self._gathered()
        iou = PixelMetric.compute_iou_per_class(dense_cm)
        tb = AccTable(['class', 'iou'])
        for idx, v in enumerate(iou):
            tb.add_row([idx, v])
        tb.add_row(['mIoU', iou.mean()])
        if is_main_process():
            self._log_summary(tb, dense_cm)
        return tb

    def summary_all(self, dense_cm=None, dec=5):
        if dense_cm is None:
            dense_cm = self._gathered()
        iou = np.round(PixelMetric.compute_iou_per_class(dense_cm), dec)
        f1 = np.round(PixelMetric.compute_F_measure_per_class(dense_cm, beta=1.0), dec)
        prec = np.round(PixelMetric.compute_precision_per_class(dense_cm), dec)
        rec = np.round(PixelMetric.compute_recall_per_class(dense_cm), dec)
        miou, mf1, mprec, mrec = (np.round(v.mean(), dec) for v in (iou, f1, prec, rec))
        oa = np.round(PixelMetric.compute_overall_accuracy(dense_cm), dec)
        kappa = np.round(PixelMetric.cohen_kappa_score(dense_cm), dec)
        named = bool(self._class_names)
        tb = AccTable((['name'] if named else []) + ['class', 'iou', 'f1', 'precision', 'recall'])
        lead = (lambda i: [self._class_names[i]]) if named else (lambda i: [])
        pad = [''] if named else []
        for idx in range(len(iou)):
            tb.add_row(lead(idx) + [idx, iou[idx], f1[idx], prec[idx], rec[idx]])
        tb.add_row(pad + ['mean', miou, mf1, mprec, mrec])
        tb.add_row(pad + ['OA', oa, '-', '-', '-'])
        tb.add_row(pad + ['Kappa', kappa, '-', '-', '-'])
        if is_main_process():
            self._log_summary(tb, dense_cm)
        return tb
