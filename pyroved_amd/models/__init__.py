"""Variational autoencoder models of the SVI hot path."""
from .base import baseVAE
from .ivae import iVAE
from .jivae import jiVAE
from .ved import VED
from .ssivae import ssiVAE, ss_reg_iVAE

__all__ = ['iVAE', 'jiVAE', 'VED', 'ssiVAE', 'ss_reg_iVAE']
