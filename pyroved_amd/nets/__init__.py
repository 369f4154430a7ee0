"""Fully-connected encoder / decoder modules of the SVI hot path."""
from .fc import fcDecoderNet, fcEncoderNet, sDecoderNet, coord_latent, make_fc_layers

__all__ = ["fcEncoderNet", "fcDecoderNet", "sDecoderNet"]
