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


# ------------------------------------------------------------------------------------------------------------------
# A convolutional stack (nets.conv.FeatureExtractor / Upsampler used OUTSIDE a model: the reference's
# nets/conv.py:150-262 forward) as one autograd Function over pv_convnet_forward / pv_convnet_backward — the same op
# executor and kernels the VED and conv-encoder steps run (csrc/pv_convstack.h).

def _stack_tensors(ops_list):
    """The stack's parameter tensors (and batch-norm running statistics) in plan order, with their keys."""
    out = []
    for kind, mod, _, key in ops_list:
        if kind == "conv":
            out.append((key + ".weight", mod.weight, True))
            if mod.bias is not None:
                out.append((key + ".bias", mod.bias, True))
        elif kind == "batchnorm":
            out.append((key + ".weight", mod.weight, True))
            out.append((key + ".bias", mod.bias, True))
            out.append((key + ".running_mean", mod.running_mean, False))
            out.append((key + ".running_var", mod.running_var, False))
    return out


def conv_stack_supported(stack, x: torch.Tensor) -> bool:
    """CUDA fp32 input of a 1-D / 2-D stack whose layers the op executor covers (kernel 3 / 1, stride 1, 2x pooling,
    2x nearest / bilinear upsampling, the reference's batch norm) — anything else stays on the modules' own forward."""
    from ._convplan import conv_ops, UnsupportedModel
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and stack.ndim in (1, 2)
            and x.dim() == stack.ndim + 2 and x.shape[0] > 0):
        return False
    try:
        ops_list = conv_ops(stack.layers, stack.activation, "L")
    except UnsupportedModel:
        return False
    first = next((m for kind, m, _, _ in ops_list if kind == "conv"), None)
    if first is None or x.shape[1] != first.in_channels:
        return False                                   # (the modules' own forward raises torch's shape error)
    return _stack_plan(stack, x, ops_list)[1] >= 0     # geometry the executor rejects: the torch modules run instead


def _stack_plan(stack, x, ops_list, layout=None):
    """(plan, workspace bytes or a negative library code) of the stack on input x; offsets from `layout` when given."""
    import ctypes as C
    from ._convplan import fill_ops
    if layout is None:
        layout, off = {}, 0
        for key, t, _ in _stack_tensors(ops_list):
            layout[key] = off
            off += (t.numel() + 63) // 64 * 64
    p = _abi.pv_convnet_plan()
    p.batch, p.ndim, p.in_ch = x.shape[0], stack.ndim, x.shape[1]
    for i, d in enumerate(x.shape[2:]):
        p.in_dim[i] = d
    p.n_ops = fill_ops(p.ops, ops_list, layout)
    p.bn_eval = int(not stack.training)
    p.conv_bf16 = 0
    p.need_dx = int(x.requires_grad)
    with _abi.device_of(x.device):
        need = int(_abi.lib().pv_convnet_workspace_bytes(C.byref(p)))
    return p, need


class _ConvStack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stack, *tensors):
        import ctypes as C
        from ._convplan import conv_ops, bn_modules
        ops_list = conv_ops(stack.layers, stack.activation, "L")
        named = _stack_tensors(ops_list)
        assert len(named) == len(tensors)
        layout, off = {}, 0
        for (key, t, _), _t in zip(named, tensors):
            layout[key] = off
            off += (t.numel() + 63) // 64 * 64                       # (16-byte aligned pieces for the vector loads)
        flat = torch.zeros(max(off, 1), device=x.device, dtype=torch.float32)
        for (key, t, _), tt_ in zip(named, tensors):
            flat[layout[key]:layout[key] + t.numel()].copy_(tt_.detach().reshape(-1))
        p, need = _stack_plan(stack, x, ops_list, layout)
        L = _abi.lib()
        with _abi.device_of(x.device):
            if need < 0:
                raise _abi.PvError("pyroved_amd: unsupported conv stack (pv_convnet_workspace_bytes -> %d)" % need)
            shp = (C.c_int32 * 3)()
            _abi.check(L.pv_convnet_out_shape(C.byref(p), shp), "pv_convnet_out_shape")
            ws = torch.empty(int(need), device=x.device, dtype=torch.uint8)
            xc = x.detach().contiguous()
            out = torch.empty((x.shape[0], shp[0]) + tuple(shp[1:1 + stack.ndim]), device=x.device, dtype=torch.float32)
            p.params, p.ws, p.ws_bytes = flat.data_ptr(), ws.data_ptr(), ws.numel()
            _abi.check(L.pv_convnet_forward(C.byref(p), _abi.ptr(xc), _abi.ptr(out), _abi.current_stream()), "pv_convnet_forward")
        bns = bn_modules(ops_list)
        if bns and stack.training:                                   # the kernels updated the running statistics in `flat`
            with torch.no_grad():
                for (key, t, is_param) in named:
                    if not is_param:
                        t.copy_(flat[layout[key]:layout[key] + t.numel()].view_as(t))
                for b_ in bns:
                    b_.num_batches_tracked += 1
        ctx.plan, ctx.named, ctx.layout = p, named, layout
        ctx.save_for_backward(xc, flat)         # (version-checked: an in-place edit of either before backward raises)
        ctx.ws = ws
        return out

    @staticmethod
    def backward(ctx, dout):
        import ctypes as C
        p = ctx.plan
        xc, flat = ctx.saved_tensors
        ws = ctx.ws
        p.params, p.ws, p.ws_bytes = flat.data_ptr(), ws.data_ptr(), ws.numel()
        grads = torch.zeros_like(flat)
        dx = torch.empty_like(xc) if (ctx.needs_input_grad[0] and p.need_dx) else None
        p.grads = grads.data_ptr()
        dout = dout.contiguous()
        with _abi.device_of(xc.device):
            _abi.check(_abi.lib().pv_convnet_backward(C.byref(p), _abi.ptr(xc), _abi.ptr(dout), _abi.ptr(dx), _abi.current_stream()),
                       "pv_convnet_backward")
        outs = []
        for k, (key, t, is_param) in enumerate(ctx.named):
            if is_param and ctx.needs_input_grad[2 + k]:
                o = ctx.layout[key]
                outs.append(grads[o:o + t.numel()].view_as(t).clone())
            else:
                outs.append(None)
        return (dx, None) + tuple(outs)


def conv_stack(stack, x: torch.Tensor) -> torch.Tensor:
    """stack.layers applied to x (B, C, *dims) on the GPU through the library, differentiable in x and the parameters."""
    from ._convplan import conv_ops
    named = _stack_tensors(conv_ops(stack.layers, stack.activation, "L"))
    return _ConvStack.apply(x, stack, *[t for _, t, _ in named])
