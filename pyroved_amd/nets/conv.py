"""
Convolutional encoder / decoder modules — host-side mirror of pyroved/nets/conv.py.

Same class names, constructor signatures, sub-module names (hence `state_dict` keys) and parameter
initialisation order as the reference, so checkpoints and seeds interchange.  The modules only *hold*
parameters and describe the layer sequence: the arithmetic runs in HIP kernels through the C ABI (inside a
model the owning engine walks the layer lists into its plan; a stand-alone FeatureExtractor / Upsampler called on
a float32 CUDA tensor runs as one autograd Function over pv_convnet_forward / pv_convnet_backward, ops.conv_stack).
3-D stacks and CPU tensors compose the torch modules, as the reference does.

Scope of this build: 1-D and 2-D data, kernel 3 / stride 1 / padding 1 convolutions, 2x max-pooling,
2x upsampling (nearest — the reference's own fallback for 1-D decoders, nets/conv.py:126-130 — or bilinear in 2-D),
optional batch normalisation after every activation (the reference's order conv -> activation -> BN).
"""
from typing import List, Tuple, Union
from warnings import warn

import torch
import torch.nn as nn

from ..utils import get_activation

tt = torch.tensor


def get_conv(dim: int):
    return {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}[dim]


def get_maxpool(dim: int):
    return {1: nn.MaxPool1d, 2: nn.MaxPool2d, 3: nn.MaxPool3d}[dim]


def get_bnorm(dim: int):
    return {1: nn.BatchNorm1d, 2: nn.BatchNorm2d, 3: nn.BatchNorm3d}[dim]


class UpsampleBlock(nn.Module):
    """2x interpolation followed by a 1-by-1 convolution (pyroved/nets/conv.py:105-147)."""
    def __init__(self, ndim: int, input_channels: int, output_channels: int,
                 scale_factor: int = 2, mode: str = "bilinear") -> None:
        super(UpsampleBlock, self).__init__()
        if mode not in ("bilinear", "nearest"):
            raise NotImplementedError("Use 'bilinear' or 'nearest' for upsampling mode")
        if not 0 < ndim < 4:
            raise AssertionError("ndim must be equal to 1, 2 or 3")
        if mode == "bilinear" and ndim in (3, 1):
            warn("'bilinear' mode is not supported for 1D and 3D; switching to 'nearest' mode", category=UserWarning)
            mode = "nearest"
        self.mode = mode
        self.scale_factor = scale_factor
        self.conv = get_conv(ndim)(input_channels, output_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Stand-alone module forward (conv.py:141-147): interpolate, then the 1-by-1 convolution.  Inside models.VED
        the block runs as one op of the HIP conv stack (pv_convstack.h) instead."""
        x = nn.functional.interpolate(x, scale_factor=self.scale_factor, mode=self.mode)
        return self.conv(x)


def _conv_stack(ndim, input_channels, conv_filters, kernel_size, stride, padding, batchnorm, activation, tail):
    """The reference's layer-list construction shared by FeatureExtractor and Upsampler (conv.py:171-193, 238-259):
    per block [conv + activation (+ batchnorm)] per filter, then `tail(block_index, last_channels, n_convs_so_far)`."""
    act = get_activation(activation)
    layers, ch_in, j = [], input_channels, 0
    for i, cblock in enumerate(conv_filters):
        for ch in cblock:
            layers.append(get_conv(ndim)(ch_in, ch, kernel_size, stride, padding))
            if act is not None:
                layers.append(act())
            if batchnorm:
                layers.append(get_bnorm(ndim)(ch))
            ch_in = ch
            j += 1
        layers.extend(tail(i, ch_in, j))
    return layers, ch_in


def _stack_forward(stack, x: torch.Tensor) -> torch.Tensor:
    """forward of a stand-alone FeatureExtractor / Upsampler (conv.py:195-213, 261-262).  A float32 CUDA input of a 1-D / 2-D
    stack runs as ONE autograd Function over the library's conv-stack executor (ops.conv_stack: the kernels the VED and
    conv-encoder steps use, differentiable in x and every parameter); 3-D data, CPU tensors and layer settings the
    executor does not cover compose the torch modules as the reference does."""
    from .. import ops
    if ops.conv_stack_supported(stack, x):
        return ops.conv_stack(stack, x)
    return stack.layers(x)


class FeatureExtractor(nn.Sequential):
    """Convolutional feature extractor (pyroved/nets/conv.py:150-213)."""
    def __init__(self, ndim: int, input_channels: int = 1, conv_filters: List[int] = None,
                 kernel_size: Union[Tuple[int], int] = 3, stride: Union[Tuple[int], int] = 1,
                 padding: Union[Tuple[int], int] = 1, batchnorm: bool = False, activation: str = "lrelu",
                 pool_last: bool = True) -> None:
        super(FeatureExtractor, self).__init__()
        if not 0 < ndim < 4:
            raise AssertionError("ndim must be equal to 1, 2 or 3")
        if conv_filters is None:
            conv_filters = [(32,), (64, 64), (128, 128)]
        total = sum(len(c) for c in conv_filters)

        def tail(i, ch, j):
            return [get_maxpool(ndim)(2, 2)] if (j + 1 < total or pool_last) else []
        layers, _ = _conv_stack(ndim, input_channels, conv_filters, kernel_size, stride, padding, batchnorm,
                                activation, tail)
        self.layers = nn.Sequential(*layers)
        self.ndim, self.activation, self.batchnorm = ndim, activation, batchnorm

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _stack_forward(self, x)


class Upsampler(nn.Sequential):
    """Convolutional upsampler (pyroved/nets/conv.py:216-262)."""
    def __init__(self, ndim: int, input_channels: int = 128, conv_filters: List[int] = None,
                 output_channels: int = 1, kernel_size: Union[Tuple[int], int] = 3,
                 stride: Union[Tuple[int], int] = 1, padding: Union[Tuple[int], int] = 1,
                 batchnorm: bool = False, activation: str = "lrelu", upsampling_mode: str = "bilinear") -> None:
        super(Upsampler, self).__init__()
        if not 0 < ndim < 4:
            raise AssertionError("ndim must be equal to 1, 2 or 3")
        if conv_filters is None:
            conv_filters = [(128, 128), (64, 64), (32,)]

        def tail(i, ch, j):
            return [UpsampleBlock(ndim, ch, ch, mode=upsampling_mode)]
        layers, ch = _conv_stack(ndim, input_channels, conv_filters, kernel_size, stride, padding, batchnorm,
                                 activation, tail)
        layers.append(get_conv(ndim)(ch, output_channels, 1, 1, 0))
        self.layers = nn.Sequential(*layers)
        self.ndim, self.activation, self.batchnorm = ndim, activation, batchnorm

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _stack_forward(self, x)


class features_to_latent(nn.Module):
    """Maps conv features to the latent space (pyroved/nets/conv.py:265-277): flatten (C, spatial...) + Linear."""
    def __init__(self, input_dim: Tuple[int], latent_dim: int = 2) -> None:
        super(features_to_latent, self).__init__()
        self.reshape_ = int(torch.prod(tt(input_dim)))
        self.fc_latent = nn.Linear(self.reshape_, latent_dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from .. import ops
        return ops.linear_act(x.reshape(-1, self.reshape_), self.fc_latent.weight, self.fc_latent.bias, None)


class latent_to_features(nn.Module):
    """Maps a latent vector to the feature space (pyroved/nets/conv.py): Linear + view (C, spatial...)."""
    def __init__(self, latent_dim: int, out_dim: Tuple[int]) -> None:
        super(latent_to_features, self).__init__()
        self.reshape_ = out_dim
        self.fc = nn.Linear(latent_dim, int(torch.prod(tt(out_dim)).item()))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from .. import ops
        return ops.linear_act(x, self.fc.weight, self.fc.bias, None).view(-1, *self.reshape_)


class convEncoderNet(nn.Module):
    """Standard convolutional encoder (pyroved/nets/conv.py:24-64)."""
    def __init__(self, input_dim: Tuple[int], latent_dim: int = 2, input_channels: int = 1,
                 hidden_dim: List[int] = None, batchnorm: bool = False, activation: str = "lrelu",
                 softplus_out: bool = True, pool_last: bool = False) -> None:
        super(convEncoderNet, self).__init__()
        if hidden_dim is None:
            hidden_dim = [(32,), (64, 64), (128, 128)]
        dim_denom = 2 ** len(hidden_dim) if pool_last else 2 ** (len(hidden_dim) - 1)
        output_dim = torch.div(tt(input_dim), dim_denom).int().tolist()
        output_channels = hidden_dim[-1][-1]
        self.latent_dim = latent_dim
        self.input_dim, self.input_channels = tuple(int(d) for d in input_dim), input_channels
        self.softplus_out = softplus_out
        self.feature_extractor = FeatureExtractor(
            len(input_dim), input_channels, hidden_dim, batchnorm=batchnorm, activation=activation,
            pool_last=pool_last)
        self.features2latent = features_to_latent([output_channels, *output_dim], 2 * latent_dim)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor]:
        """Inside a model (models.VED, iVAE.set_encoder) the HIP conv stack of the owning engine runs
        (pv_ved_encode / the plan's encoder ops).  A stand-alone net — any ndim, any number of input channels, as the
        reference's own unit tests build them — composes its sub-modules (conv.py:57-64): the feature extractor's
        layers, then the latent Linear through ops.linear_act."""
        if getattr(self, "_pv_engine", None) is not None:
            from ..engine_ved import conv_encoder_forward
            return conv_encoder_forward(self, x)
        encoded = self.features2latent(self.feature_extractor(x))
        mu, sigma = encoded.split(self.latent_dim, 1)
        return mu, (nn.functional.softplus(sigma) if self.softplus_out else sigma)


class convDecoderNet(nn.Module):
    """Standard convolutional decoder (pyroved/nets/conv.py:67-102)."""
    def __init__(self, latent_dim: int, output_dim: int, output_channels: int = 1,
                 hidden_dim: List[int] = None, batchnorm: bool = False, activation: str = "lrelu",
                 sigmoid_out: bool = True, upsampling_mode: str = "bilinear") -> None:
        super(convDecoderNet, self).__init__()
        if hidden_dim is None:
            hidden_dim = [(128, 128), (64, 64), (32,)]
        input_dim = torch.div(tt(output_dim), 2 ** len(hidden_dim)).int().tolist()
        self.latent_dim = latent_dim
        self.output_dim, self.output_channels = tuple(int(d) for d in output_dim), output_channels
        self.sigmoid_out = sigmoid_out
        self.latent2features = latent_to_features(latent_dim, [hidden_dim[0][0], *input_dim])
        self.upsampler = Upsampler(
            len(output_dim), hidden_dim[0][0], hidden_dim, output_channels, batchnorm=batchnorm,
            activation=activation, upsampling_mode=upsampling_mode)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        """As convEncoderNet.forward: the owning engine's HIP conv stack inside a model, the sub-modules' own
        composition (conv.py:97-102) for a stand-alone net."""
        if getattr(self, "_pv_engine", None) is not None:
            from ..engine_ved import conv_decoder_forward
            return conv_decoder_forward(self, z)
        x = self.upsampler(self.latent2features(z))
        return torch.sigmoid(x) if self.sigmoid_out else x
