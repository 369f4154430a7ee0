"""Fully-connected encoder / decoder modules of the SVI hot path."""
from .fc import fcDecoderNet, fcEncoderNet, jfcEncoderNet, sDecoderNet, coord_latent, make_fc_layers

__all__ = ["fcEncoderNet", "jfcEncoderNet", "fcDecoderNet", "sDecoderNet"]
