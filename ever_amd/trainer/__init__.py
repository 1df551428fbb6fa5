"""Command-line entry of EVer training (API of reference ever/trainer/__init__.py:14-67)."""
import argparse
import os

from .th_ddp_trainer import THDDPTrainer
from .trainer import Trainer

TRAINER = dict(th_ddp=THDDPTrainer, base=Trainer)


def get_default_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--config_path', default=None, type=str, help='path to config file')
    p.add_argument('--model_dir', default=None, type=str, help='path to model directory')
    p.add_argument('--local_rank', type=int, default=None)
    p.add_argument('--trainer', default='th_ddp', type=str, help='type of trainer')
    p.add_argument('--find_unused_parameters', action='store_true', help='whether to find unused parameters')
    p.add_argument('--mixed_precision', default='fp32', type=str, help='datatype', choices=['fp32', 'fp16', 'bf16'])
    p.add_argument('--use_wandb', action='store_true', help='whether to use wandb for logging')
    p.add_argument('--project', default=None, type=str, help='Project name for init wandb')
    p.add_argument('--entity', default=None, type=str, help='Entity for init wandb')
    p.add_argument('opts', help='Modify config options using the command-line', default=None, nargs=argparse.REMAINDER)
    return p


def get_trainer(trainer_name=None, parser=None, return_args=False, argv=None):
    parser = parser or get_default_parser()
    args = parser.parse_args(argv)
    assert args.config_path is not None, 'The `config_path` is needed'
    assert args.model_dir is not None, 'The `model_dir` is needed'
    if args.use_wandb:
        assert args.project is not None, '`project` is needed if you use wandb'
    if args.local_rank is None:  # torchrun exports LOCAL_RANK; plain `python train.py` is rank 0
        args.local_rank = int(os.environ.get('LOCAL_RANK', 0))
    name = trainer_name or args.trainer
    if name not in TRAINER:
        raise KeyError(f'unknown trainer {name}; available: {sorted(TRAINER)}')
    t = TRAINER[name](args)
    return (t, args) if return_args else t
