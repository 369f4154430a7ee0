"""
ops.py — tensor-level operators over the C ABI building blocks, used by the standalone
`nets.*.forward` calls (the "operator boundary" of SURVEY §8b: the modules stay nn.Modules
holding ordinary nn.Parameters, and each Linear+activation is one autograd Function whose
forward and backward are the library's HIP GEMMs).  Training with SVItrainer and
iVAE.encode/decode go through the plan-based entry points in engine.py instead.
"""
import torch

from . import _abi


def _ws(m: int, k: int, n: int, dev) -> torch.Tensor:
    nbytes = _abi.lib().pv_linear_workspace_bytes(m, k, n)
    return torch.empty(max(int(nbytes), 256), device=dev, dtype=torch.uint8)


def _act_grad(act, y: torch.Tensor, pre) -> torch.Tensor:
    """act'(pre) written through the layer's output y where that is possible (utils/nn.py:77-84's table)."""
    if act == "tanh":
        return 1.0 - y * y
    if act == "sigmoid":
        return y * (1.0 - y)
    if act == "softplus":
        return 1.0 - torch.exp(-y)                    # sigmoid(pre) = 1 - exp(-softplus(pre))
    if act == "relu":
        return (y > 0).to(y.dtype)
    if act == "lrelu":
        return torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.01))      # nn.LeakyReLU() default slope
    if act == "gelu":
        cdf = 0.5 * (1.0 + torch.erf(pre * 0.7071067811865476))
        return cdf + pre * torch.exp(-0.5 * pre * pre) * 0.3989422804014327
    raise ValueError("unknown activation %r" % (act,))


class _LinearAct(torch.autograd.Function):
    """y = act(x W^T + b): pv_linear_fwd forward, pv_linear_bwd backward (nn.Linear + activation of
    pyroved/nets/fc.py:307-324)."""

    @staticmethod
    def forward(ctx, x2, weight, bias, act):
        m, k = x2.shape
        n = weight.shape[0]
        L = _abi.lib()
        y = torch.empty(m, n, device=x2.device, dtype=torch.float32)
        pre = torch.empty_like(y) if act == "gelu" else None
        ws = _ws(m, k, n, x2.device)
        with _abi.device_of(x2.device):
            _abi.check(L.pv_linear_fwd(_abi.ptr(x2), k, _abi.ptr(weight), _abi.ptr(bias), _abi.ptr(y), _abi.ptr(pre), n,
                                       m, k, n, _abi.ACT[act], _abi.ptr(ws), ws.numel(), _abi.current_stream()),
                       "pv_linear_fwd")
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x2, weight, y, pre)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, weight, y, pre = ctx.saved_tensors
        m, k = x2.shape
        n = weight.shape[0]
        dpre = dy.contiguous() if ctx.act in (None, "none") else (dy * _act_grad(ctx.act, y, pre)).contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty_like(x2) if need_x else None
        dw = torch.empty_like(weight) if need_w else None
        db = torch.empty(n, device=dy.device, dtype=torch.float32) if need_b else None
        ws = _ws(m, k, n, dy.device)
        with _abi.device_of(dy.device):
            _abi.check(_abi.lib().pv_linear_bwd(_abi.ptr(dpre), n, _abi.ptr(x2), k, _abi.ptr(weight), _abi.ptr(dx), k,
                                                None, None, 0, 0, _abi.ptr(dw), _abi.ptr(db), m, k, n,
                                                _abi.ptr(ws), ws.numel(), _abi.current_stream()), "pv_linear_bwd")
        return dx, dw, db, None


def linear_act(x: torch.Tensor, weight: torch.Tensor, bias, act: str = None) -> torch.Tensor:
    """y = act(x @ weight.T + bias) on the GPU, differentiable in x, weight and bias."""
    _abi.require_device(x, "x")
    _abi.require_device(weight, "weight")
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1]).to(torch.float32).contiguous()
    if weight.shape[1] != x2.shape[1]:
        raise ValueError("linear_act: x has %d features, weight expects %d" % (x2.shape[1], weight.shape[1]))
    w = weight.contiguous()
    b = None if bias is None else bias.contiguous()
    y = _LinearAct.apply(x2, w, b, act)
    return y.reshape(*lead, weight.shape[0])
