"""Utility functions (host-side mirror of pyroved.utils for the SVI hot path)."""
from .coord import generate_grid, transform_coordinates, generate_latent_grid, generate_latent_grid_traversal
from .data import init_dataloader, init_ssvae_dataloaders, iter_batches
from .nn import (get_activation, get_bnorm, get_conv, get_maxpool,
                 set_deterministic_mode, to_onehot, average_weights, Concat, _to_device, activation_name)
from .prob import get_sampler
from .viz import plot_grid_traversal, plot_img_grid, plot_spect_grid
from .gp import gp_model

__all__ = ['generate_grid', 'transform_coordinates', 'generate_latent_grid', 'generate_latent_grid_traversal',
           'get_sampler', 'init_dataloader', 'init_ssvae_dataloaders', 'iter_batches', 'average_weights',
           'get_activation', 'get_bnorm', 'get_conv', 'get_maxpool',
           'to_onehot', 'set_deterministic_mode', 'Concat',
           'plot_img_grid', 'plot_spect_grid', 'plot_grid_traversal', 'gp_model']
