"""Build a model from a config and load a training checkpoint for inference (API of reference
ever/api/infer_tool.py:16-74; checkpoint layout of ever/core/checkpoint.py).  `export_model` (TorchScript
tracing) is not offered: the HIP layers are ctypes calls, not traceable ATen ops."""
import os
from pathlib import Path

import torch

from ..core import checkpoint, config
from ..core.builder import make_model
from ..core.logger import info

__all__ = ['build_from_file', 'build_and_load_from_file', 'build_from_model_dir']


def build_from_file(config_path):
    cfg = config.import_config(config_path)
    return make_model(cfg['model'])


def _load_any(path):
    return torch.load(path, map_location='cpu', weights_only=False)


def build_and_load_from_file(config_path, checkpoint_path):
    model = build_from_file(config_path)
    blob = _load_any(checkpoint_path)
    if isinstance(blob, dict) and checkpoint.CheckPoint.MODEL in blob:
        state = checkpoint.remove_module_prefix(blob[checkpoint.CheckPoint.MODEL])
        global_step = blob[checkpoint.CheckPoint.GLOBALSTEP]
    else:  # a bare state dict named checkpoint-<step>.pth
        state = checkpoint.remove_module_prefix(blob)
        global_step = int(Path(checkpoint_path).name.split('.')[0].split('-')[1])
    model.eval()
    model.load_state_dict(state)
    info('[Load params] from {}'.format(checkpoint_path))
    return model, global_step


def build_from_model_dir(model_dir, checkpoint_name=None):
    pkl_cfg, py_cfg = os.path.join(model_dir, 'config.pkl'), os.path.join(model_dir, 'config.py')
    if os.path.exists(pkl_cfg):
        cfg_path = pkl_cfg
    elif os.path.exists(py_cfg):
        cfg_path = py_cfg
    else:
        raise FileNotFoundError('The config file is not found in model_dir.')
    if checkpoint_name is None:  # the best model if there is one, else the last checkpoint
        best = os.path.join(model_dir, 'model-best.pth')
        if os.path.exists(best):
            model = build_from_file(cfg_path)
            model.eval()
            model.load_state_dict(checkpoint.remove_module_prefix(_load_any(best)))
            info('[Load params] from {}'.format(best))
            return model, 'best'
        fps = sorted(Path(model_dir).glob('checkpoint-*.pth'),
                     key=lambda e: int(e.name.replace('checkpoint-', '').replace('.pth', '')))
        if not fps:
            raise FileNotFoundError(f'no checkpoint-*.pth in {model_dir}')
        checkpoint_name = fps[-1].name
    return build_and_load_from_file(cfg_path, os.path.join(model_dir, checkpoint_name))
