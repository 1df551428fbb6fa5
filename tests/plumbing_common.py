"""Shared fixtures of the CPU plumbing tests: an ERModule that wraps the ORACLE model (stock torch,
CPU) so that ever_amd's Launcher / Trainer / DDP loop can be exercised without a GPU
(BASELINE config 1: "FarSeg ResNet-18, 4-band 256x256 tiles, batch 2, CPU-only ... plumbing").
The HIP modules themselves have no CPU path, by design."""
import torch

import ever_amd as er
from oracle import farseg_ref, portable


@er.registry.MODEL.register('OracleFarSeg', override=True, verbose=False)
class OracleFarSeg(er.ERModule):
    def __init__(self, config):
        super().__init__(config)
        net = farseg_ref.FarSegRef(self.config.resnet_type, self.config.in_channels, 1)
        farseg_ref.load_portable_weights(net, portable.fill_state_dict(net.state_dict()))
        self.en, self.head = net.en, net.head

    def forward(self, x, y=None):
        lg = self.head(self.en(x))
        if self.training:
            return dict(bce_loss=farseg_ref.bce_ref(lg, y), dice_loss=farseg_ref.dice_ref(lg, y))
        return lg

    def set_default_config(self):
        self.config.update(dict(resnet_type='resnet18', in_channels=4))


class ToyTiles(torch.utils.data.Dataset):
    """hash-generated (image, label) pairs, identical to oracle/gen_golden.py:toy_dataset"""

    def __init__(self, n=4, c=4, hw=64):
        self.items = []
        for i in range(n):
            x, y = portable.synthetic_batch(f'toy/{i}', 1, c, hw, hw, 1)
            self.items.append((torch.from_numpy(x[0]), torch.from_numpy(y[0])))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


@er.registry.DATALOADER.register('ToyTilesLoader', override=True, verbose=False)
class ToyTilesLoader(er.ERDataLoader):
    def __init__(self, config):
        super().__init__(config)

    @property
    def dataloader_params(self):
        ds = ToyTiles(self.config.n, self.config.c, self.config.hw)
        if self.config.distributed:
            sampler = er.data.StepDistributedSampler(ds)
        else:
            sampler = torch.utils.data.SequentialSampler(ds)
        return dict(dataset=ds, batch_size=self.config.batch_size, sampler=sampler, num_workers=0)

    def set_default_config(self):
        self.config.update(dict(n=4, c=4, hw=64, batch_size=2, distributed=False))


CONFIG_TEMPLATE = '''
config = dict(
    model=dict(type='OracleFarSeg', params=dict(resnet_type='resnet18', in_channels=4)),
    data=dict(train=dict(type='ToyTilesLoader', params=dict(n={n}, c=4, hw={hw}, batch_size=2, distributed={dist}))),
    optimizer=dict(type='sgd', params=dict(momentum=0.9, weight_decay=1e-4), grad_clip=dict(max_norm=35, norm_type=2)),
    learning_rate=dict(type='poly', params=dict(base_lr=0.01, power=0.9, max_iters={iters})),
    train=dict(forward_times=1, num_iters={iters}, distributed={dist}, log_interval_step=1, save_ckpt_interval_epoch=1000),
    test=dict(),
)
'''
