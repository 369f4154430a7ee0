"""Utility functions (host-side mirror of pyroved.utils for the SVI hot path)."""
from .coord import generate_grid, transform_coordinates
from .data import init_dataloader
from .nn import (get_activation, get_bnorm, get_conv, get_maxpool,
                 set_deterministic_mode, to_onehot, Concat, _to_device, activation_name)
from .prob import get_sampler

__all__ = ['generate_grid', 'transform_coordinates', 'get_sampler', 'init_dataloader',
           'get_activation', 'get_bnorm', 'get_conv', 'get_maxpool',
           'to_onehot', 'set_deterministic_mode', 'Concat']
