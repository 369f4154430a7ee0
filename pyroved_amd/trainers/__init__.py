"""Training loops with stochastic variational inference on the HIP path (same names as pyroved.trainers)."""
from . import auxsvi as _aux, svi as _svi

SVItrainer, auxSVItrainer = _svi.SVItrainer, _aux.auxSVItrainer
__all__ = ("SVItrainer", "auxSVItrainer")
