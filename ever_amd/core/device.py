import torch

__all__ = ['auto_device']


def auto_device():
    return torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')
