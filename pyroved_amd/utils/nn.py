"""Host-side helpers mirroring pyroved/utils/nn.py:37-124."""
from typing import List, Type, Union

import torch
import torch.nn as nn


def to_onehot(idx: torch.Tensor, n: int) -> torch.Tensor:
    """One-hot encoding of a label (pyroved/utils/nn.py:37-48)."""
    if torch.max(idx).item() >= n:
        raise AssertionError(
            "Labelling must start from 0 and "
            "maximum label value must be less than total number of classes")
    if idx.dim() == 1:
        idx = idx.unsqueeze(1)
    onehot = torch.zeros(idx.size(0), n)
    return onehot.scatter_(1, idx, 1)


class Concat(nn.Module):
    """Broadcast-concatenation of a list of tensors along the last dim; a tensor passes
    through unchanged (pyroved/utils/nn.py:51-74).  Pure data movement."""
    def __init__(self, allow_broadcast: bool = True):
        self.allow_broadcast = allow_broadcast
        super().__init__()

    def forward(self, input_args: Union[List[torch.Tensor], torch.Tensor]) -> torch.Tensor:
        if torch.is_tensor(input_args):
            return input_args
        input_args = [a.flatten(1) if a.ndim >= 4 else a for a in input_args]
        if self.allow_broadcast:
            shape = torch.broadcast_shapes(*[s.shape[:-1] for s in input_args]) + (-1,)
            input_args = [s.expand(shape) for s in input_args]
        return torch.cat(input_args, dim=-1)


def _to_device(input_data, **kwargs):
    device = kwargs.get("device", 'cuda' if torch.cuda.is_available() else 'cpu')
    if len(input_data) == 1:
        return input_data[0].to(device)
    return [t.to(device) for t in input_data]


def set_deterministic_mode(seed: int) -> None:
    """Sets all torch manual seeds (pyroved/utils/nn.py:87-100)."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
        torch.cuda.manual_seed_all(seed)
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False


def get_bnorm(dim: int) -> Type[nn.Module]:
    return {1: nn.BatchNorm1d, 2: nn.BatchNorm2d, 3: nn.BatchNorm3d}[dim]


def get_conv(dim: int) -> Type[nn.Module]:
    return {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}[dim]


def get_maxpool(dim: int) -> Type[nn.Module]:
    return {1: nn.MaxPool1d, 2: nn.MaxPool2d, 3: nn.MaxPool3d}[dim]


def get_activation(activation: str) -> Type[nn.Module]:
    """Activation module class by name (pyroved/utils/nn.py:118-124).  The modules only
    mark the layer type in the nn.Sequential (so state_dict keys match the reference);
    the arithmetic runs in the HIP kernels (enum pv_act)."""
    if activation is None:
        return
    activations = {"lrelu": nn.LeakyReLU, "tanh": nn.Tanh,
                   "softplus": nn.Softplus, "relu": nn.ReLU,
                   "gelu": nn.GELU}
    return activations[activation]


def activation_name(module: nn.Module) -> str:
    for name, cls in (("lrelu", nn.LeakyReLU), ("tanh", nn.Tanh), ("softplus", nn.Softplus),
                      ("relu", nn.ReLU), ("gelu", nn.GELU), ("sigmoid", nn.Sigmoid)):
        if isinstance(module, cls):
            return name
    raise NotImplementedError("unsupported activation module %r" % (module,))


def average_weights(ensemble):
    """Averages the weights of all state_dicts in the ensemble {epoch: state_dict} (pyroved/utils/nn.py:11-34);
    batch-norm statistics are taken from the first one."""
    from copy import deepcopy as dc
    ensemble = {k - min(ensemble.keys()): v for (k, v) in ensemble.items()}
    out = dc(ensemble[0])
    for name in [n for n in out.keys() if n.split('_')[-1] not in ["mean", "var", "tracked"]]:
        ws = [dc(sd[name]) for sd in ensemble.values() if name in sd]
        out[name].copy_(sum(ws) / float(len(ws)))
    return out
