"""
pyroved_amd — MI355X-native implementation of pyroVED's SVI training hot path
(iVAE encoder -> reparameterise -> coordinate-grid decoder -> ELBO -> Adam), behind the
reference's own `models.iVAE` / `trainers.SVItrainer` API.  Compute runs in hand-written
HIP kernels (libpyroved_amd.so, C ABI in include/pyroved_amd.h); there is no CPU fallback.
"""
from . import models, trainers, nets, utils
from .__version__ import version as __version__

__all__ = ['models', 'trainers', 'nets', 'utils', '__version__']
