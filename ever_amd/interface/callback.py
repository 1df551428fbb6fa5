"""Epoch-boundary / end-of-training hooks (API of reference ever/interface/callback.py:1-113)."""


class Callback:
    def __init__(self, epoch_interval, only_master, prior=100, before_train=False, after_train=False):
        self._epoch_interval = epoch_interval
        self._only_master = only_master
        self._prior = prior
        self._launcher = None
        self.before_train = before_train
        self.after_train = after_train

    def name(self):
        return ''

    def func(self):
        return NotImplemented

    interval = property(lambda self: self._epoch_interval)
    only_master = property(lambda self: self._only_master)
    prior = property(lambda self: self._prior)
    launcher = property(lambda self: self._launcher)
    model = property(lambda self: self._launcher.model)
    model_without_ddp = property(lambda self: self._launcher.model_without_ddp)
    unwrapped_model = property(lambda self: self._launcher.unwrapped_model)
    model_dir = property(lambda self: self._launcher.model_dir)
    global_step = property(lambda self: self._launcher.global_step)
    learning_rate = property(lambda self: self._launcher.lr)
    logger = property(lambda self: self._launcher.logger)

    def set_launcher(self, launcher):
        self._launcher = launcher

    def info(self, msg):
        self._launcher.info(msg)

    def save_model(self, filename=None):
        self._launcher.save_model(filename)


class SaveCheckpointCallback(Callback):
    def __init__(self, epoch_interval):
        super().__init__(epoch_interval=epoch_interval, only_master=True, prior=0, before_train=False,
                         after_train=True)

    def func(self):
        self.launcher.checkpoint.save()

    def name(self):
        return 'SaveCheckpoint'


class EvaluationCallback(Callback):
    def __init__(self, dataloader, epoch_interval, only_master, after_train=True, config=None):
        super().__init__(epoch_interval=epoch_interval, only_master=only_master, before_train=False,
                         after_train=after_train)
        self._dataloader = dataloader
        self._config = config

    def func(self):
        self.launcher.evaluate(self._dataloader, config=self._config)

    def name(self):
        return 'Evaluation'
