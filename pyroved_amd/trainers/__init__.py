"""Training loops with stochastic variational inference on the HIP path."""
from .svi import SVItrainer

__all__ = ['SVItrainer']
