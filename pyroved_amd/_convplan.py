"""
_convplan.py — reads a nets/conv.py layer stack (FeatureExtractor.layers / Upsampler.layers) into the op sequence
(pv_op[]) of the C ABI.  Shared by engine_ved.VEDEngine and engine.IVAEEngine (convolutional encoder).
"""
import os
from typing import Optional

import torch.nn as nn

from . import _abi
from .nets.conv import UpsampleBlock


class UnsupportedModel(NotImplementedError):
    pass


def _conv_types():
    return (nn.Conv1d, nn.Conv2d)


def conv_ops(layers: nn.Sequential, activation, prefix: Optional[str] = None):
    """nn.Sequential of nets/conv.py -> list of (kind, conv module | None, act, state_dict key prefix)."""
    mods = list(layers)
    ops, i = [], 0
    while i < len(mods):
        mod, pos = mods[i], i
        if isinstance(mod, _conv_types()):
            k = mod.kernel_size[0]
            if (any(v != k for v in mod.kernel_size) or k not in (1, 3) or any(v != 1 for v in mod.stride)
                    or any(v != k // 2 for v in mod.padding) or any(v != 1 for v in mod.dilation) or mod.groups != 1):
                raise UnsupportedModel("conv layers must be kernel 3 / padding 1 or kernel 1, stride 1")
            act = None
            bn = (nn.BatchNorm1d, nn.BatchNorm2d)
            if i + 1 < len(mods) and not isinstance(mods[i + 1], _conv_types() + bn + (UpsampleBlock, nn.MaxPool1d,
                                                                                        nn.MaxPool2d)):
                act = activation
                i += 1
            ops.append(("conv", mod, act, None if prefix is None else "%s.%d" % (prefix, pos)))
            if i + 1 < len(mods) and isinstance(mods[i + 1], bn):          # conv -> activation -> batch norm
                i += 1
                b_ = mods[i]
                if (not b_.affine or not b_.track_running_stats or b_.momentum != 0.1 or b_.eps != 1e-5
                        or b_.num_features != mod.out_channels):
                    raise UnsupportedModel("batch-norm layers must be the reference's nn.BatchNormNd(channels) defaults")
                ops.append(("batchnorm", b_, None, None if prefix is None else "%s.%d" % (prefix, i)))
        elif isinstance(mod, (nn.MaxPool1d, nn.MaxPool2d)):
            def _all(v, want):
                return all(int(u) == want for u in (v if isinstance(v, (tuple, list)) else (v,)))
            if not (_all(mod.kernel_size, 2) and _all(mod.stride if mod.stride is not None else mod.kernel_size, 2)
                    and _all(mod.padding, 0) and _all(mod.dilation, 1) and not mod.ceil_mode and not mod.return_indices):
                raise UnsupportedModel("max pooling must be the reference's 2x window, stride 2, no padding")
            ops.append(("maxpool2", None, None, None))
        elif isinstance(mod, UpsampleBlock):
            if mod.scale_factor != 2 or mod.mode not in ("nearest", "bilinear"):
                raise UnsupportedModel("upsampling must be 2x nearest or bilinear")
            # interpolate, then the 1-by-1 convolution (conv.py:141-147) == the 1-by-1 convolution, then interpolate: both are
            # linear, act on different axes, and the interpolation weights sum to one (so the bias commutes too).  The plan
            # runs the convolution FIRST — on a quarter (2-D) / half (1-D) of the pixels, forward and backward.
            pair = [("conv", mod.conv, None, None if prefix is None else "%s.%d.conv" % (prefix, pos)),
                    ("upsample2" if mod.mode == "nearest" else "upsample2_bilinear", None, None, None)]
            ops.extend(pair[::-1] if os.environ.get("PV_UPSAMPLE_FIRST") else pair)      # (the reference's order, for A/B runs)
        else:
            raise UnsupportedModel("unsupported layer %s in a conv stack" % type(mod).__name__)
        i += 1
    if len(ops) > _abi.PV_MAX_OPS:
        raise UnsupportedModel("more than %d ops in a conv stack" % _abi.PV_MAX_OPS)
    return ops


def fill_ops(arr, ops, layout) -> int:
    """Writes `ops` into a ctypes pv_op array; offsets from the flat-buffer layout."""
    for j, (kind, mod, act, key) in enumerate(ops):
        o = arr[j]
        o.kind = _abi.OP[kind]
        if kind == "conv":
            o.cin, o.cout, o.ksize = mod.in_channels, mod.out_channels, mod.kernel_size[0]
            o.act = _abi.ACT[act]
            o.w_off = layout[key + ".weight"]
            o.b_off = layout[key + ".bias"] if mod.bias is not None else -1
        elif kind == "batchnorm":
            o.cin = o.cout = mod.num_features
            o.w_off, o.b_off = layout[key + ".weight"], layout[key + ".bias"]
            o.aux0_off, o.aux1_off = layout[key + ".running_mean"], layout[key + ".running_var"]
    return len(ops)


def bn_modules(ops):
    """The batch-norm modules of an op list (their num_batches_tracked counters are kept by the host)."""
    return [mod for kind, mod, _, _ in ops if kind == "batchnorm"]
