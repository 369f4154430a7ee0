"""
ops.py — thin tensor-level wrappers over the C ABI building blocks, used by the
standalone `nets.*.forward` calls.  (Training and iVAE.encode/decode go through the
plan-based entry points in engine.py instead.)
"""
import torch

from . import _abi


def linear_act(x: torch.Tensor, weight: torch.Tensor, bias, act: str = None) -> torch.Tensor:
    """y = act(x @ weight.T + bias) on the GPU (pv_linear_fwd; nn.Linear + activation of
    pyroved/nets/fc.py:307-324)."""
    _abi.require_device(x, "x")
    _abi.require_device(weight, "weight")
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    m, k = x2.shape
    n = weight.shape[0]
    if weight.shape[1] != k:
        raise ValueError("linear_act: x has %d features, weight expects %d" % (k, weight.shape[1]))
    w = weight.detach().contiguous()
    b = None if bias is None else bias.detach().contiguous()
    y = torch.empty(m, n, device=x.device, dtype=torch.float32)
    L = _abi.lib()
    nbytes = L.pv_linear_workspace_bytes(m, k, n)
    ws = torch.empty(max(int(nbytes), 256), device=x.device, dtype=torch.uint8)
    _abi.check(L.pv_linear_fwd(_abi.ptr(x2), k, _abi.ptr(w), _abi.ptr(b), _abi.ptr(y), None, n, m, k, n,
                               _abi.ACT[act], _abi.ptr(ws), ws.numel(), _abi.current_stream()), "pv_linear_fwd")
    return y.reshape(*lead, n)
