"""Variational autoencoder models of the SVI hot path."""
from .base import baseVAE
from .ivae import iVAE

__all__ = ['iVAE']
