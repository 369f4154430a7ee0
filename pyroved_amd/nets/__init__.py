"""Fully-connected encoder / decoder modules of the SVI hot path."""
from .fc import (fcDecoderNet, fcEncoderNet, jfcEncoderNet, sDecoderNet, coord_latent, make_fc_layers,
                 fcClassifierNet, fcRegressorNet)

from .conv import convEncoderNet, convDecoderNet, FeatureExtractor, Upsampler, UpsampleBlock, features_to_latent, latent_to_features

__all__ = ["fcEncoderNet", "jfcEncoderNet", "fcDecoderNet", "sDecoderNet", "convEncoderNet", "convDecoderNet",
           "fcClassifierNet", "fcRegressorNet"]
