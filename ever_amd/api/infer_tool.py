"""Build a model from a config and load a training checkpoint for inference (API of reference
ever/api/infer_tool.py:16-74; checkpoint layout of ever/core/checkpoint.py).  `export_model` traces the eval-mode forward with
`torch.jit.trace` as the reference does: the no-grad forward of every HIP layer kind is a registered `ever_amd::` operator
(hip/oplib.py), so the TorchScript file holds those operator calls and the weights; loading it needs `import ever_amd`
(operator registration) and the GPU — there is no CPU path to fall back to."""
import os
from pathlib import Path

import torch

from ..core import checkpoint, config
from ..core.builder import make_model
from ..core.logger import info

__all__ = ['build_from_file', 'build_and_load_from_file', 'build_from_model_dir', 'export_model', 'trace_model']


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


def trace_model(model, example, fold=True):
    """`torch.jit.trace` of an eval-mode HIP model (reference infer_tool.py:72).  fold: BatchNorm folded into the preceding
    convolutions first (module/fold.py) — the traced graph then holds `ever_amd::conv2d_folded` nodes with the folded weights
    as constants; the model's own parameters are left as they are."""
    from ..module.fold import fold_batchnorm
    model = model.eval()
    if fold:
        fold_batchnorm(model)
    with torch.no_grad():
        return torch.jit.trace(model, example, check_trace=False)


def export_model(config_path, checkpoint_path, input_shape, output_path, device='cuda'):
    """reference infer_tool.py:70-74: build, load, trace on an all-ones input of `input_shape`, save as TorchScript."""
    model, gs = build_and_load_from_file(config_path, checkpoint_path)
    model = model.to(device)
    traced = trace_model(model, torch.ones(input_shape, device=device))
    torch.jit.save(traced, output_path)
    info('[export model] to {}'.format(output_path))
    return traced
