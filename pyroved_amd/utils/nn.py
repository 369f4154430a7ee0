"""Host-side helpers mirroring pyroved/utils/nn.py:37-124."""
from typing import List, Type, Union

import torch
import torch.nn as nn


_ACTIVATIONS = (("lrelu", nn.LeakyReLU), ("tanh", nn.Tanh), ("softplus", nn.Softplus),
                ("relu", nn.ReLU), ("gelu", nn.GELU))
_BY_DIM = {"conv": (nn.Conv1d, nn.Conv2d, nn.Conv3d),
           "bnorm": (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d),
           "maxpool": (nn.MaxPool1d, nn.MaxPool2d, nn.MaxPool3d)}


def to_onehot(idx: torch.Tensor, n: int) -> torch.Tensor:
    """(len(idx), n) float one-hot rows of integer labels 0..n-1 (pyroved/utils/nn.py:37-48: CPU float32 result,
    AssertionError for a label >= n)."""
    labels = idx.reshape(idx.shape[0], -1).long()          # (N,) or (N, 1); several columns give multi-hot rows
    if int(labels.max()) >= n:
        raise AssertionError("Labelling must start from 0 and maximum label value must be less than "
                             "total number of classes")
    return (labels.cpu()[:, :, None] == torch.arange(n)).any(1).to(torch.float32)


class Concat(nn.Module):
    """Joins a list of tensors along the last axis after broadcasting their leading axes against each other (tensors
    with >= 4 axes are flattened from axis 1 first); a bare tensor passes through (pyroved/utils/nn.py:51-74).  Pure
    data movement: the networks call it on `[x, y]` conditioning inputs."""
    def __init__(self, allow_broadcast: bool = True):
        super().__init__()
        self.allow_broadcast = allow_broadcast

    def forward(self, input_args: Union[List[torch.Tensor], torch.Tensor]) -> torch.Tensor:
        if torch.is_tensor(input_args):
            return input_args
        parts = [t.flatten(1) if t.ndim >= 4 else t for t in input_args]
        if self.allow_broadcast:
            lead = torch.broadcast_shapes(*(t.shape[:-1] for t in parts))
            parts = [t.expand(*lead, t.shape[-1]) for t in parts]
        return torch.cat(parts, dim=-1)


def _to_device(input_data, **kwargs):
    """One tensor for a 1-element sequence, a list otherwise, moved to kwargs['device'] (default: the GPU if there is
    one — the reference ignores the model's own device here, pyroved/utils/nn.py:77-84)."""
    device = kwargs.get("device") or ("cuda" if torch.cuda.is_available() else "cpu")
    moved = [t.to(device) for t in input_data]
    return moved[0] if len(moved) == 1 else moved


def set_deterministic_mode(seed: int) -> None:
    """Seeds the CPU generator and every GPU generator (pyroved/utils/nn.py:87-100).  The model constructors and the
    trainers call it, which fixes the stream parameter initialisation, shuffling and eps draws come from."""
    torch.manual_seed(seed)
    if not torch.cuda.is_available():
        return
    torch.cuda.empty_cache()
    torch.cuda.manual_seed_all(seed)
    # the reference's cudnn switches; MIOpen honours the same torch flags (only stand-alone nets.conv modules use it)
    torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False


def get_bnorm(dim: int) -> Type[nn.Module]:
    return _BY_DIM["bnorm"][dim - 1]


def get_conv(dim: int) -> Type[nn.Module]:
    return _BY_DIM["conv"][dim - 1]


def get_maxpool(dim: int) -> Type[nn.Module]:
    return _BY_DIM["maxpool"][dim - 1]


def get_activation(activation: str) -> Type[nn.Module]:
    """Activation module class by name, None for None (pyroved/utils/nn.py:118-124; KeyError for unknown names).  The
    modules only mark the layer type in the nn.Sequential (so state_dict keys match the reference); inside the models
    the arithmetic runs in the HIP kernels (enum pv_act)."""
    return None if activation is None else dict(_ACTIVATIONS)[activation]


def activation_name(module: nn.Module) -> str:
    for name, cls in _ACTIVATIONS + (("sigmoid", nn.Sigmoid),):
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
