from .callback import Callback, EvaluationCallback, SaveCheckpointCallback
from .configurable import ConfigurableMixin
from .learning_rate import LearningRateBase
from .module import ERModule
from .dataloader import ERDataLoader, ERDataset

__all__ = ['ERModule', 'ConfigurableMixin', 'ERDataLoader', 'ERDataset', 'Callback', 'SaveCheckpointCallback',
           'EvaluationCallback', 'LearningRateBase']
