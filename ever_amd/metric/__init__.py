from .confusion_matrix import ConfusionMatrix
from .pixel import PixelMetric, AccTable

__all__ = ['ConfusionMatrix', 'PixelMetric', 'AccTable']
