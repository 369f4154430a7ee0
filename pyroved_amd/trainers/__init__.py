"""Training loops with stochastic variational inference on the HIP path."""
from .svi import SVItrainer
from .auxsvi import auxSVItrainer

__all__ = ['SVItrainer', 'auxSVItrainer']
