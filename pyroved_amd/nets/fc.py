"""
Fully-connected encoder / decoder modules — host-side mirror of pyroved/nets/fc.py.

Same class names, constructor signatures, sub-module names (hence `state_dict`
keys) and parameter initialisation order as the reference, so checkpoints and
seeds interchange.  The modules only *hold* parameters: the arithmetic of
`forward` runs in HIP kernels through the C ABI (GPU tensors only — there is no
CPU path), and training does not call `forward` at all (engine.py drives the
fused kernels over a flat parameter buffer these parameters are views of).
"""
from typing import List, Tuple, Type

import torch
import torch.nn as nn

from ..utils import get_activation, Concat
from .. import ops

tt = torch.tensor


def _prod(dims) -> int:
    n = 1
    for d in dims:
        n *= int(d)
    return n


def make_fc_layers(in_dim: int, hidden_dim: List[int], activation: str = "tanh") -> Type[nn.Module]:
    """Stack of Linear + activation pairs (pyroved/nets/fc.py:307-324): indices 0, 2, 4...
    are the Linear layers."""
    if isinstance(hidden_dim, tuple):
        hidden_dim = list(hidden_dim)
    dims = [in_dim] + hidden_dim
    layers = []
    for i in range(1, len(hidden_dim) + 1):
        layers.extend([nn.Linear(dims[i - 1], dims[i]), get_activation(activation)()])
    return nn.Sequential(*layers)


def _run_stack(fc_layers: nn.Sequential, activation: str, h: torch.Tensor) -> torch.Tensor:
    for m in fc_layers:
        if isinstance(m, nn.Linear):
            h = ops.linear_act(h, m.weight, m.bias, activation)
    return h


class fcEncoderNet(nn.Module):
    """Standard fully-connected encoder: outputs the mean and (softplus) standard
    deviation of the encoded distribution (pyroved/nets/fc.py:19-61)."""
    def __init__(self, in_dim: Tuple[int], latent_dim: int = 2, c_dim: int = 0,
                 hidden_dim: List[int] = None, activation: str = 'tanh',
                 softplus_out: bool = True, flat: bool = True) -> None:
        super(fcEncoderNet, self).__init__()
        if len(in_dim) not in [1, 2, 3]:
            raise ValueError("in_dim must be (h, w), (h, w, c), or (l,)")
        self.in_dim = _prod(in_dim) + c_dim
        if hidden_dim is None:
            hidden_dim = [128, 128]
        self.flat = flat
        self.activation = activation
        self.softplus_out = softplus_out
        self.concat = Concat()
        self.fc_layers = make_fc_layers(self.in_dim, hidden_dim, activation)
        self.fc11 = nn.Linear(hidden_dim[-1], latent_dim)
        self.fc12 = nn.Linear(hidden_dim[-1], latent_dim)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor]:
        x = self.concat(x)
        if self.flat:
            x = x.reshape(-1, self.in_dim)
        h = _run_stack(self.fc_layers, self.activation, x)
        mu = ops.linear_act(h, self.fc11.weight, self.fc11.bias, None)
        sigma = ops.linear_act(h, self.fc12.weight, self.fc12.bias, "softplus" if self.softplus_out else None)
        return mu, sigma


class jfcEncoderNet(nn.Module):
    """Fully-connected encoder for the joint VAE: mean, (softplus) standard deviation and class
    probabilities softmax(fc13(h)) (pyroved/nets/fc.py:64-108)."""
    def __init__(self, in_dim: Tuple[int], latent_dim: int = 2, discrete_dim: int = 0,
                 hidden_dim: List[int] = None, activation: str = 'tanh',
                 softplus_out: bool = True, flat: bool = True) -> None:
        super(jfcEncoderNet, self).__init__()
        if len(in_dim) not in [1, 2, 3]:
            raise ValueError("in_dim must be (h, w), (h, w, c), or (l,)")
        self.in_dim = _prod(in_dim)
        if hidden_dim is None:
            hidden_dim = [128, 128]
        self.flat = flat
        self.activation = activation
        self.softplus_out = softplus_out
        self.discrete_dim = discrete_dim
        self.concat = Concat()
        self.fc_layers = make_fc_layers(self.in_dim, hidden_dim, activation)
        self.fc11 = nn.Linear(hidden_dim[-1], latent_dim)
        self.fc12 = nn.Linear(hidden_dim[-1], latent_dim)
        self.fc13 = nn.Linear(hidden_dim[-1], discrete_dim)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor]:
        x = self.concat(x)
        if self.flat:
            x = x.reshape(-1, self.in_dim)
        h = _run_stack(self.fc_layers, self.activation, x)
        mu = ops.linear_act(h, self.fc11.weight, self.fc11.bias, None)
        sigma = ops.linear_act(h, self.fc12.weight, self.fc12.bias, "softplus" if self.softplus_out else None)
        alpha = torch.softmax(ops.linear_act(h, self.fc13.weight, self.fc13.bias, None), dim=-1)
        return mu, sigma, alpha


class fcDecoderNet(nn.Module):
    """Standard fully-connected decoder (pyroved/nets/fc.py:111-152)."""
    def __init__(self, out_dim: Tuple[int], latent_dim: int, c_dim: int = 0,
                 hidden_dim: List[int] = None, activation: str = 'tanh',
                 sigmoid_out: bool = True, unflat: bool = True) -> None:
        super(fcDecoderNet, self).__init__()
        if len(out_dim) not in [1, 2, 3]:
            raise ValueError("in_dim must be (h, w), (h, w, c), or (l,)")
        self.unflat = unflat
        if self.unflat:
            self.reshape = out_dim
        out_dim = _prod(out_dim)
        if hidden_dim is None:
            hidden_dim = [128, 128]
        self.activation = activation
        self.sigmoid_out = sigmoid_out
        self.concat = Concat()
        self.fc_layers = make_fc_layers(latent_dim + c_dim, hidden_dim, activation)
        self.out = nn.Linear(hidden_dim[-1], out_dim)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        z = self.concat(z)
        h = _run_stack(self.fc_layers, self.activation, z)
        x = ops.linear_act(h, self.out.weight, self.out.bias, "sigmoid" if self.sigmoid_out else None)
        if self.unflat:
            return x.view(-1, *self.reshape)
        return x


class coord_latent(nn.Module):
    """The "spatial" first layer of the invariant decoder: tanh(fc_coord(x') + fc_latent(z))
    (pyroved/nets/fc.py:202-237; based on arXiv:1909.11663)."""
    def __init__(self, latent_dim: int, out_dim: int, ndim: int = 2, activation_out: bool = True) -> None:
        super(coord_latent, self).__init__()
        self.fc_coord = nn.Linear(ndim, out_dim)
        self.fc_latent = nn.Linear(latent_dim, out_dim, bias=False)
        self.activation = nn.Tanh() if activation_out else None

    def forward(self, x_coord: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
        """tanh(fc_coord(x_coord) + fc_latent(z)[:, None, :]) flattened to (B*N, out_dim) (fc.py:220-237)."""
        batch_dim, n = x_coord.shape[0], x_coord.shape[1]
        h_x = ops.linear_act(x_coord.reshape(batch_dim * n, -1), self.fc_coord.weight, self.fc_coord.bias, None)
        h_z = ops.linear_act(z.reshape(batch_dim, -1), self.fc_latent.weight, None, None)
        h = (h_x.view(batch_dim, n, -1) + h_z.unsqueeze(1)).reshape(batch_dim * n, -1)
        return self.activation(h) if self.activation is not None else h


class sDecoderNet(nn.Module):
    """Spatial generator (decoder): a per-pixel MLP over transformed coordinates and the
    latent code (pyroved/nets/fc.py:155-199)."""
    def __init__(self, out_dim: Tuple[int], latent_dim: int, c_dim: int = 0,
                 hidden_dim: List[int] = None, activation: str = 'tanh',
                 sigmoid_out: bool = True, unflat: bool = True) -> None:
        super(sDecoderNet, self).__init__()
        if len(out_dim) not in [1, 2, 3]:
            raise ValueError("in_dim must be (h, w), (h, w, c), or (l,)")
        self.unflat = unflat
        if self.unflat:
            self.reshape = out_dim
        if hidden_dim is None:
            hidden_dim = [128, 128]
        coord_dim = 1 if len(out_dim) < 2 else 2
        self.activation = activation
        self.sigmoid_out = sigmoid_out
        self.concat = Concat()
        self.coord_latent = coord_latent(latent_dim + c_dim, hidden_dim[0], coord_dim)
        self.fc_layers = make_fc_layers(hidden_dim[0], hidden_dim, activation)
        self.out = nn.Linear(hidden_dim[-1], 1)

    def forward(self, x_coord: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
        """x_coord: (B, N, coord_dim) transformed coordinates, z: (B, latent_dim [+ c_dim]) or the list the
        reference's Concat takes -> (B, *out_dim) (pyroved/nets/fc.py:189-199, 220-237).  The unfused operator
        path: every Linear(+activation) is one ops.linear_act (HIP GEMMs, differentiable); SVItrainer and
        iVAE.decode use the fused kernels behind pv_ivae_loss_and_grads / pv_ivae_decode instead."""
        z = self.concat(z)
        h = self.coord_latent(x_coord, z)
        h = _run_stack(self.fc_layers, self.activation, h)
        x = ops.linear_act(h, self.out.weight, self.out.bias, "sigmoid" if self.sigmoid_out else None)
        if self.unflat:
            return x.view(-1, *self.reshape)
        return x


class fcClassifierNet(nn.Module):
    """Fully-connected classifier: softmax(out(fc_layers(x))) (pyroved/nets/fc.py:240-271)."""
    def __init__(self, in_dim: Tuple[int], num_classes: int, hidden_dim: List[int] = None,
                 activation: str = 'tanh') -> None:
        super(fcClassifierNet, self).__init__()
        if len(in_dim) not in [1, 2, 3]:
            raise ValueError("in_dim must be (h, w), (h, w, c), or (l,)")
        self.in_dim = _prod(in_dim)
        if hidden_dim is None:
            hidden_dim = [128, 128]
        self.activation = activation
        self.fc_layers = make_fc_layers(self.in_dim, hidden_dim, activation)
        self.out = nn.Linear(hidden_dim[-1], num_classes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x.reshape(-1, self.in_dim)
        h = _run_stack(self.fc_layers, self.activation, x)
        return torch.softmax(ops.linear_act(h, self.out.weight, self.out.bias, None), dim=-1)


class fcRegressorNet(nn.Module):
    """Fully-connected regressor: out(fc_layers(x)) (pyroved/nets/fc.py:274-304)."""
    def __init__(self, in_dim: Tuple[int], c_dim: int, hidden_dim: List[int] = None,
                 activation: str = 'tanh') -> None:
        super(fcRegressorNet, self).__init__()
        if len(in_dim) not in [1, 2, 3]:
            raise ValueError("in_dim must be (h, w), (h, w, c), or (l,)")
        self.in_dim = _prod(in_dim)
        if hidden_dim is None:
            hidden_dim = [128, 128]
        self.activation = activation
        self.fc_layers = make_fc_layers(self.in_dim, hidden_dim, activation)
        self.out = nn.Linear(hidden_dim[-1], c_dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x.reshape(-1, self.in_dim)
        h = _run_stack(self.fc_layers, self.activation, x)
        return ops.linear_act(h, self.out.weight, self.out.bias, None)
