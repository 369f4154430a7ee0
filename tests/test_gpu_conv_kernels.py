"""
Kernel-level numerics of the convolution path behind `models.VED` / `iVAE.set_encoder(convEncoderNet)` (reference:
pyroved/nets/conv.py:24-64, 146-249), through the library's C-ABI test hooks, against plain PyTorch in float64 on the
same inputs:
  * pv_conv3_sp (pv_conv_sp.hip): 2-D kernel-3 convolution, forward and input gradient, operands split exactly into
    three bf16 pieces (fp32-class, tolerance 2e-6 relative l2) or two (mixed precision, 2e-5), odd image sizes, channel
    counts that are not multiples of the 64-channel tile, fused bias / activation / activation-derivative epilogues;
  * pv_conv3_sp_wgrad: its weight and bias gradients;
  * pv_c1_convpool (pv_conv_c1.hip): first block conv(1 -> C) + activation + 2x max-pool, forward and backward;
  * pv_convhead (pv_convhead.hip): the Linear head over a channels-last feature map.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pyroved_amd import _abi

pytestmark = pytest.mark.gpu
P = C.c_void_p
ACTS = {"none": 0, "tanh": 1, "relu": 2, "lrelu": 3, "softplus": 4, "sigmoid": 6}


def lib():
    handle = C.CDLL(_abi.LIB_PATH)
    handle.pv_debug_conv3_wgrad_ws.restype = C.c_longlong
    handle.pv_debug_c1_convpool_ws.restype = C.c_longlong
    handle.pv_debug_convhead_ws.restype = C.c_longlong
    return handle


def ptr(t):
    return P(t.data_ptr()) if t is not None else P(0)


def stream():
    return P(torch.cuda.current_stream().cuda_stream)


def rel_l2(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def act_fn(name, v):
    return {"none": lambda t: t, "tanh": torch.tanh, "relu": torch.relu, "lrelu": lambda t: F.leaky_relu(t, 0.01),
            "softplus": F.softplus, "sigmoid": torch.sigmoid}[name](v)


def act_grad_of_output(name, y):
    """the activation derivative as a function of the OUTPUT (what the backward kernels use)"""
    if name == "tanh":
        return 1 - y * y
    if name == "relu":
        return (y > 0).to(y.dtype)
    if name == "lrelu":
        return torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.01))
    if name == "softplus":
        return 1 - torch.exp(-y)
    if name == "sigmoid":
        return y * (1 - y)
    return torch.ones_like(y)


CONV_CASES = [  # (B, H, W, Cin, Cout, act)
    (3, 16, 16, 32, 64, "lrelu"), (2, 13, 21, 64, 40, "tanh"), (1, 5, 7, 32, 8, "none"), (2, 32, 32, 64, 128, "relu"),
    (2, 8, 8, 96, 72, "softplus"), (5, 4, 4, 128, 128, "sigmoid"), (1, 33, 17, 32, 32, "lrelu")]


# mode 7 (round 4): ONE fp16 piece per operand, one product — the throughput precision: per-element relative error 2^-12 on
# both operands, ~3e-4 on a 288..1152-term dot product of random data
# mode 8 (round 5, "f16w2"): the weights two exact fp16 pieces, the patch ONE piece (round to nearest), two products: what is left is
# the patch's rounding alone — 2^-12 per element, independent from element to element: ~2e-4 on a dot product of random data, and
# it averages out of everything a training step sums (the full-size gradient tests hold the mode to the fp32-class bars)
@pytest.mark.parametrize("mode,tol", [(3, 2e-6), (2, 2e-5), (7, 6e-4), (8, 3e-4)])
@pytest.mark.parametrize("B,H,W,Ci,Co,act", CONV_CASES)
def test_split_operand_conv_forward_and_input_gradient(gpu_device, mode, tol, B, H, W, Ci, Co, act):
    g = torch.Generator().manual_seed(B * 1000 + H * 31 + Co)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).cuda()
    bias = (torch.randn(Co, generator=g) * 0.1).cuda()
    x = torch.randn(B, H, W, Ci, generator=g).cuda()
    n = max(Co, Ci)
    scratch = torch.empty(((n + 63) // 64) * 64 * n * 9 * 6 + 4096, dtype=torch.uint8, device="cuda")
    out = torch.full((B, H, W, Co), float("nan"), device="cuda")
    rc = lib().pv_debug_conv3(mode, ptr(x), B, H, W, 2, ptr(w), Co, Ci, 0, ptr(bias), ptr(out), ACTS[act], ptr(scratch), P(0), 0,
                              stream())
    assert rc == 0
    ref = act_fn(act, F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), bias.double(), padding=1)).permute(0, 2, 3, 1)
    assert rel_l2(out, ref) < tol
    # input gradient of the same convolution, times the producing layer's activation derivative (a function of `eg_y`)
    dy = torch.randn(B, H, W, Co, generator=g).cuda()
    eg_y = torch.tanh(torch.randn(B, H, W, Ci, generator=g)).cuda()
    din = torch.full((B, H, W, Ci), float("nan"), device="cuda")
    rc = lib().pv_debug_conv3(mode, ptr(dy), B, H, W, 2, ptr(w), Co, Ci, 1, P(0), ptr(din), 0, ptr(scratch), ptr(eg_y), ACTS[act],
                              stream())
    if Co % 32:                   # the input gradient contracts over Cout: not this kernel's case (the stack falls back)
        assert rc == -1
        return
    assert rc == 0
    refd = F.conv_transpose2d(dy.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    refd = refd * act_grad_of_output(act, eg_y.double())
    assert rel_l2(din, refd) < tol


@pytest.mark.parametrize("mode,tol", [(3, 2e-6), (2, 2e-5), (7, 6e-4)])
@pytest.mark.parametrize("B,H,W,Ci,Co,act", CONV_CASES)
def test_split_operand_conv_weight_gradient(gpu_device, mode, tol, B, H, W, Ci, Co, act):
    g = torch.Generator().manual_seed(7 + B * 1000 + H * 31 + Co)
    x = torch.randn(B, H, W, Ci, generator=g).cuda()
    dy = torch.randn(B, H, W, Co, generator=g).cuda()
    dw = torch.full((Co, Ci, 3, 3), float("nan"), device="cuda")
    db = torch.full((Co,), float("nan"), device="cuda")
    nb = lib().pv_debug_conv3_wgrad_ws(mode, B, H, W, Ci, Co, 2)
    ws = torch.empty(max(int(nb), 256), dtype=torch.uint8, device="cuda")
    rc = lib().pv_debug_conv3_wgrad(mode, ptr(dy), ptr(x), B, H, W, Ci, 2, ptr(dw), ptr(db), Co, ptr(ws), C.c_longlong(ws.numel()),
                                    stream())
    assert rc == 0
    wd = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    y = F.conv2d(x.permute(0, 3, 1, 2).double(), wd, None, padding=1)
    (gw,) = torch.autograd.grad(y, wd, dy.permute(0, 3, 1, 2).double())
    assert rel_l2(dw, gw) < tol
    assert rel_l2(db, dy.double().sum((0, 1, 2))) < tol


@pytest.mark.parametrize("B,H,W,Co,act", [(3, 16, 16, 32, "lrelu"), (2, 13, 9, 8, "tanh"), (1, 64, 64, 32, "relu"),
                                          (4, 6, 10, 64, "softplus"), (2, 7, 7, 12, "none"), (9, 2, 2, 4, "sigmoid")])
def test_fused_first_block_forward_and_backward(gpu_device, B, H, W, Co, act):
    """conv(1 -> Co, k3) + activation + 2x max-pool as one kernel each way (odd sizes drop the last line / column like
    nn.MaxPool2d); the backward takes dL/d(pooled) and returns the convolution's weight and bias gradients."""
    g = torch.Generator().manual_seed(B + 10 * H + Co)
    x = torch.randn(B, H, W, generator=g).cuda()
    w = (torch.randn(Co, 1, 3, 3, generator=g) / 3).cuda()
    bias = (torch.randn(Co, generator=g) * 0.1).cuda()
    Hp, Wp = H // 2, W // 2
    y = torch.full((B, Hp, Wp, Co), float("nan"), device="cuda")
    code = torch.zeros(B, Hp, Wp, Co, dtype=torch.uint8, device="cuda")
    L = lib()
    rc = L.pv_debug_c1_convpool(0, ptr(x), B, H, W, ptr(w), ptr(bias), Co, ACTS[act], ptr(y), ptr(code), P(0), P(0), P(0), P(0),
                                C.c_longlong(0), stream())
    assert rc == 0
    wd = w.double().clone().requires_grad_(True)
    bd = bias.double().clone().requires_grad_(True)
    pre = F.conv2d(x.double().unsqueeze(1), wd, bd, padding=1)
    ref = F.max_pool2d(act_fn(act, pre), 2)
    assert rel_l2(y, ref.permute(0, 2, 3, 1)) < 2e-6
    gy = torch.randn(B, Hp, Wp, Co, generator=g).cuda()
    gw, gb = torch.autograd.grad(ref, (wd, bd), gy.permute(0, 3, 1, 2).double())
    dw = torch.full((Co, 1, 3, 3), float("nan"), device="cuda")
    db = torch.full((Co,), float("nan"), device="cuda")
    ws = torch.empty(int(L.pv_debug_c1_convpool_ws(B, H, W, Co)), dtype=torch.uint8, device="cuda")
    rc = L.pv_debug_c1_convpool(1, ptr(x), B, H, W, ptr(w), ptr(bias), Co, ACTS[act], ptr(y), ptr(code), ptr(gy), ptr(dw), ptr(db),
                                ptr(ws), C.c_longlong(ws.numel()), stream())
    assert rc == 0
    assert rel_l2(dw, gw) < 5e-6
    assert rel_l2(db, gb) < 5e-6


@pytest.mark.parametrize("B,S,Cc,out,act", [(5, 16, 32, 4, "lrelu"), (3, 9, 20, 12, "tanh"), (17, 256, 128, 4, "relu"),
                                            (2, 4, 8, 16, "none"), (40, 64, 64, 7, "softplus")])
def test_conv_head_without_transposes(gpu_device, B, S, Cc, out, act):
    """features2latent over a channels-last feature map a[b][s][c] with the torch weight W[j][c*S + s]: forward, the
    feature gradient (times the activation derivative of the layer that produced a) and the weight / bias gradients."""
    g = torch.Generator().manual_seed(B + S + out)
    Fdim = S * Cc
    w = (torch.randn(out, Fdim, generator=g) / Fdim ** 0.5).cuda()
    bias = torch.randn(out, generator=g).cuda()
    a = torch.tanh(torch.randn(B, S, Cc, generator=g)).cuda()
    dhead = torch.randn(B, out, generator=g).cuda()
    L = lib()
    ws = torch.empty(int(L.pv_debug_convhead_ws(B, C.c_longlong(Fdim), out)), dtype=torch.uint8, device="cuda")
    wt = torch.empty(out, Fdim, device="cuda")
    head = torch.full((B, out), float("nan"), device="cuda")
    gout = torch.full((B, S, Cc), float("nan"), device="cuda")
    dw = torch.full((out, Fdim), float("nan"), device="cuda")
    db = torch.full((out,), float("nan"), device="cuda")
    for what in range(4):
        rc = L.pv_debug_convhead(what, ptr(w), ptr(wt), ptr(bias), ptr(a), ptr(head), ptr(dhead), ptr(gout), ptr(dw), ptr(db), B, S, Cc,
                                 out, ACTS[act], ptr(ws), C.c_longlong(ws.numel()), stream())
        assert rc == 0
    feat = a.double().permute(0, 2, 1).reshape(B, Fdim)                       # torch's flatten of (B, C, spatial)
    assert rel_l2(head, feat @ w.double().t() + bias.double()) < 2e-6
    dfeat = (dhead.double() @ w.double()).reshape(B, Cc, S).permute(0, 2, 1)
    assert rel_l2(gout, dfeat * act_grad_of_output(act, a.double())) < 2e-6
    assert rel_l2(dw, dhead.double().t() @ feat) < 2e-6
    assert rel_l2(db, dhead.double().sum(0)) < 2e-6


# ---------------------------------------------------------------------------------------------------------------------
# stand-alone nets.conv stacks (FeatureExtractor / Upsampler outside a model: the reference's nets/conv.py:150-262 forward)
# run on the library's conv-stack executor as one autograd Function (ops.conv_stack); reference = the same modules'
# torch composition in float64 on the CPU.
STACK_CASES = [
    ("fe2d_default", dict(kind="fe", ndim=2, in_ch=1, filters=None, bn=False, act="lrelu"), (3, 1, 32, 32), False),
    ("fe2d_default_dx", dict(kind="fe", ndim=2, in_ch=1, filters=None, bn=False, act="lrelu"), (2, 1, 16, 16), True),
    ("fe2d_rgb_tanh", dict(kind="fe", ndim=2, in_ch=3, filters=[(8,), (16, 16)], bn=False, act="tanh"), (2, 3, 12, 20), True),
    ("fe1d", dict(kind="fe", ndim=1, in_ch=2, filters=[(16,), (32, 32)], bn=False, act="lrelu"), (4, 2, 40), True),
    ("fe2d_bn", dict(kind="fe", ndim=2, in_ch=1, filters=[(8,), (16,)], bn=True, act="lrelu"), (5, 1, 8, 8), False),
    ("up1d", dict(kind="up", ndim=1, in_ch=32, filters=[(32, 32), (16,)], bn=False, act="lrelu", out_ch=1), (3, 32, 8), True),
    ("up2d_bilinear", dict(kind="up", ndim=2, in_ch=16, filters=[(16,), (8,)], bn=False, act="tanh", out_ch=2), (2, 16, 4, 4), True),
    # kernel-1 family with a transcendental activation derivative in the input-gradient epilogue, ragged channel counts
    # (scalar path of the kernel-1 kernels), batch norm between the blocks, and the 2-D nearest upsample (not fused)
    ("up1d_tanh", dict(kind="up", ndim=1, in_ch=32, filters=[(32, 32), (16, 16)], bn=False, act="tanh", out_ch=1), (5, 32, 8), True),
    ("up1d_ragged_softplus", dict(kind="up", ndim=1, in_ch=12, filters=[(20,), (6,)], bn=False, act="softplus", out_ch=3), (3, 12, 9), True),
    ("up1d_bn", dict(kind="up", ndim=1, in_ch=16, filters=[(16,), (8,)], bn=True, act="lrelu", out_ch=1), (6, 16, 8), True),
    ("up2d_nearest", dict(kind="up", ndim=2, in_ch=16, filters=[(16,), (8,)], bn=False, act="lrelu", out_ch=1, mode="nearest"), (2, 16, 4, 6), True),
]


@pytest.mark.parametrize("name,spec,shape,need_dx", STACK_CASES, ids=[c[0] for c in STACK_CASES])
def test_standalone_conv_stacks_run_on_the_library(gpu_device, name, spec, shape, need_dx):
    import copy
    import warnings
    from pyroved_amd.nets.conv import FeatureExtractor, Upsampler
    warnings.filterwarnings("ignore")
    torch.manual_seed(3)
    if spec["kind"] == "fe":
        net = FeatureExtractor(spec["ndim"], spec["in_ch"], spec["filters"], batchnorm=spec["bn"], activation=spec["act"])
    else:
        net = Upsampler(spec["ndim"], spec["in_ch"], spec["filters"], spec["out_ch"], batchnorm=spec["bn"],
                        activation=spec["act"], upsampling_mode=spec.get("mode", "bilinear" if spec["ndim"] == 2 else "nearest"))
    ref = copy.deepcopy(net).double()
    net = net.cuda()
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(1))
    xg = x.cuda().requires_grad_(need_dx)
    out = net(xg)
    assert "ConvStack" in type(out.grad_fn).__name__                 # the library path, not the modules' torch forward
    xr = x.double().requires_grad_(need_dx)
    want = ref.layers(xr)
    assert out.shape == want.shape
    assert rel_l2(out, want) < 5e-6
    gout = torch.randn(*want.shape, generator=torch.Generator().manual_seed(2))
    out.backward(gout.cuda())
    want.backward(gout.double())
    for (n1, p1), (_, p2) in zip(net.named_parameters(), ref.named_parameters()):
        assert rel_l2(p1.grad, p2.grad) < 2e-5, n1
    if need_dx:
        assert rel_l2(xg.grad, xr.grad) < 2e-5
    if spec["bn"]:                                                    # running statistics written back to the modules
        for (n1, b1), (_, b2) in zip(net.named_buffers(), ref.named_buffers()):
            if b1.dtype.is_floating_point:
                assert rel_l2(b1, b2) < 1e-5, n1
            else:
                assert int(b1) == int(b2), n1


@pytest.mark.parametrize("scale", [1.0, 1e-6, 3e4])
@pytest.mark.parametrize("B,H,W,Ci,Co,act", CONV_CASES[:4])
def test_fp16_two_piece_conv_is_fp32_class_at_any_magnitude(gpu_device, scale, B, H, W, Ci, Co, act):
    """mode 4: fp16 pieces with exact power-of-two scaling per 32-channel chunk of a tile's patch — inputs of size 1, 1e-6
    (gradient-like) and 3e4, with one channel block 1e-9 of the rest and one exactly zero."""
    g = torch.Generator().manual_seed(B * 1000 + H * 31 + Co)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).cuda()
    x = torch.randn(B, H, W, Ci, generator=g) * scale
    x[..., :8] *= 1e-9
    if Ci >= 64:
        x[..., 32:64] = 0.0
    x = x.cuda()
    n = max(Co, Ci)
    scratch = torch.empty(((n + 63) // 64) * 64 * n * 9 * 6 + 4096, dtype=torch.uint8, device="cuda")
    out = torch.full((B, H, W, Co), float("nan"), device="cuda")
    rc = lib().pv_debug_conv3(4, ptr(x), B, H, W, 2, ptr(w), Co, Ci, 0, P(0), ptr(out), 0, ptr(scratch), P(0), 0, stream())
    assert rc == 0
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, padding=1).permute(0, 2, 3, 1)
    assert rel_l2(out, ref) < 2e-6


@pytest.mark.parametrize("sx,sdy", [(1.0, 1.0), (1.0, 1e-7), (3e3, 1e-4)])
@pytest.mark.parametrize("B,H,W,Ci,Co,act", CONV_CASES[:5])
def test_fp16_two_piece_weight_gradient_is_fp32_class_at_any_magnitude(gpu_device, sx, sdy, B, H, W, Ci, Co, act):
    """mode 4 of the weight gradient: every staged dY tile and patch tile carries its own exact power-of-two scale; inputs
    and gradients of very different sizes, one image all zero, one image 1e6 times the others."""
    g = torch.Generator().manual_seed(11 + B * 1000 + H * 31 + Co)
    x = torch.randn(B, H, W, Ci, generator=g) * sx
    dy = torch.randn(B, H, W, Co, generator=g) * sdy
    if B > 1:
        dy[0] = 0.0
        x[-1] *= 1e6
    x, dy = x.cuda(), dy.cuda()
    dw = torch.full((Co, Ci, 3, 3), float("nan"), device="cuda")
    db = torch.full((Co,), float("nan"), device="cuda")
    nb = lib().pv_debug_conv3_wgrad_ws(4, B, H, W, Ci, Co, 2)
    ws = torch.empty(max(int(nb), 256), dtype=torch.uint8, device="cuda")
    rc = lib().pv_debug_conv3_wgrad(4, ptr(dy), ptr(x), B, H, W, Ci, 2, ptr(dw), ptr(db), Co, ptr(ws), C.c_longlong(ws.numel()),
                                    stream())
    assert rc == 0
    wd = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    y = F.conv2d(x.permute(0, 3, 1, 2).double(), wd, None, padding=1)
    (gw,) = torch.autograd.grad(y, wd, dy.permute(0, 3, 1, 2).double())
    assert rel_l2(dw, gw) < 2e-6
    assert rel_l2(db, dy.double().sum((0, 1, 2))) < 2e-6


@pytest.mark.parametrize("scale", [1.0, 1e-6, 3e4])
@pytest.mark.parametrize("B,dims,Ci,Co", [(3, (40,), 32, 64), (2, (16,), 128, 128), (4, (64,), 64, 32), (2, (12, 20), 64, 40),
                                         (9, (16,), 64, 64), (5, (24,), 32, 40), (17, (8,), 32, 32), (3, (32,), 64, 128), (7, (11,), 32, 16),
                                         # more than 1024 one-sample tiles: several short signals share a 64-position tile
                                         (1100, (16,), 32, 32), (1030, (24,), 32, 40), (1500, (8,), 32, 16), (1027, (11,), 32, 64)])
def test_fp16_two_piece_tile_kernel_1d_and_2d(gpu_device, scale, B, dims, Ci, Co):
    """mode 5: the round-1 tile kernel (the 1-D decoder layers of VED run on it) with fp16 two-piece operands."""
    nd = len(dims)
    g = torch.Generator().manual_seed(B + Ci + Co + dims[0])
    w = torch.randn(Co, Ci, *([3] * nd), generator=g) / (3 * Ci ** 0.5)
    x = torch.randn(B, Ci, *dims, generator=g) * scale
    x[:, :8] *= 1e-9
    bias = torch.randn(Co, generator=g) * 0.1 * scale
    H, W = (dims[0], 1) if nd == 1 else dims
    xcl = (x.permute(0, 2, 1) if nd == 1 else x.permute(0, 2, 3, 1)).contiguous().cuda()
    n = max(Co, Ci)
    scratch = torch.empty(((n + 63) // 64) * 64 * n * 9 * 6 + 4096, dtype=torch.uint8, device="cuda")
    out = torch.full((B, H, W, Co), float("nan"), device="cuda")
    wc, bc = w.cuda(), bias.cuda()
    rc = lib().pv_debug_conv3(5, ptr(xcl), B, H, W, nd, ptr(wc), Co, Ci, 0, ptr(bc), ptr(out), 0, ptr(scratch), P(0), 0, stream())
    assert rc == 0
    conv = F.conv1d if nd == 1 else F.conv2d
    ref = conv(x.double(), w.double(), bias.double(), padding=1)
    ref = (ref.permute(0, 2, 1) if nd == 1 else ref.permute(0, 2, 3, 1)).reshape(B, H, W, Co)
    assert rel_l2(out, ref) < 2e-6


@pytest.mark.parametrize("mode,nd,B,H,W,Ci,Co", [(5, 1, 256, 16, 1, 128, 128), (4, 2, 64, 16, 16, 128, 128)])
def test_fp16_kernels_are_bit_reproducible(gpu_device, mode, nd, B, H, W, Ci, Co):
    """The per-chunk scales travel between waves through LDS: a missing wait before the barrier once let 1 workgroup in ~3000
    read a stale one (15 % of launches at this size).  30 launches must agree bit for bit."""
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(Co, Ci, *([3] * nd), generator=g) / (3 * Ci ** 0.5)).cuda()
    x = (torch.randn(B, H, W, Ci, generator=g) * 1e-3).cuda()
    scratch = torch.empty(4 << 20, dtype=torch.uint8, device="cuda")
    first = None
    for _ in range(30):
        out = torch.full((B, H, W, Co), float("nan"), device="cuda")
        assert lib().pv_debug_conv3(mode, ptr(x), B, H, W, nd, ptr(w), Co, Ci, 0, P(0), ptr(out), 0, ptr(scratch), P(0), 0, stream()) == 0
        if first is None:
            first = out.clone()
        else:
            assert torch.equal(first, out)


@pytest.mark.parametrize("mode", [4, 3, 2])
@pytest.mark.parametrize("B,H,W,Ci,Co,act", [(3, 16, 16, 32, 64, "lrelu"), (2, 12, 20, 64, 40, "tanh"), (1, 6, 34, 32, 8, "relu"),
                                             (2, 32, 32, 64, 128, "none"), (5, 4, 4, 128, 72, "softplus")])
def test_convolution_with_fused_max_pool(gpu_device, mode, B, H, W, Ci, Co, act):
    """The 2x max-pool in the convolution's epilogue (pooled values + one winner byte each, no full-resolution output) and its
    backward from the bytes: dL/d(pre-activation) = g * act'(pooled) at the winner, zero elsewhere."""
    g = torch.Generator().manual_seed(B + H + Co)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).cuda()
    bias = (torch.randn(Co, generator=g) * 0.1).cuda()
    x = torch.randn(B, H, W, Ci, generator=g).cuda()
    n = max(Co, Ci)
    scratch = torch.empty(((n + 63) // 64) * 64 * n * 9 * 6 + 4096, dtype=torch.uint8, device="cuda")
    pooled = torch.full((B, H // 2, W // 2, Co), float("nan"), device="cuda")
    code = torch.full((B, H // 2, W // 2, Co), 255, dtype=torch.uint8, device="cuda")
    L = lib()
    rc = L.pv_debug_conv3_pool(mode, ptr(x), B, H, W, ptr(w), Co, Ci, ptr(bias), ACTS[act], ptr(pooled), ptr(code), ptr(scratch), P(0),
                               P(0), stream())
    assert rc == 0
    pre = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), bias.double(), padding=1).requires_grad_(True)
    y = act_fn(act, pre)
    ref = F.max_pool2d(y, 2)
    tol = 2e-6 if mode != 2 else 2e-5
    assert rel_l2(pooled, ref.permute(0, 2, 3, 1)) < tol
    assert int(code.max()) <= 3
    gp = torch.randn(B, H // 2, W // 2, Co, generator=g).cuda()
    din = torch.full((B, H, W, Co), float("nan"), device="cuda")
    rc = L.pv_debug_conv3_pool(mode, P(0), B, H, W, P(0), Co, Ci, P(0), ACTS[act], ptr(pooled), ptr(code), P(0), ptr(gp), ptr(din), stream())
    assert rc == 0
    (gpre,) = torch.autograd.grad(ref, pre, gp.permute(0, 3, 1, 2).double())
    # (ties between fp32 and fp64 winners are measure-zero for random inputs; compare as tensors)
    assert rel_l2(din, gpre.permute(0, 2, 3, 1)) < max(tol, 1e-5)


@pytest.mark.parametrize("B,S,Cc,zd", [(5, 16, 32, 2), (3, 9, 20, 5), (37, 16, 128, 2), (2, 4, 8, 16), (70, 64, 64, 7)])
def test_latent_to_features_without_transposes(gpu_device, B, S, Cc, zd):
    """latent2features: Linear(z_dim -> C*S) + view(C, S) producing the channels-last map a[b][s][c] directly, its weight / bias
    gradients and the latent gradient from a channels-last gradient map."""
    g = torch.Generator().manual_seed(B + S + zd)
    Fdim = S * Cc
    w = (torch.randn(Fdim, zd, generator=g) / zd ** 0.5).cuda()
    bias = torch.randn(Fdim, generator=g).cuda()
    z = torch.randn(B, zd, generator=g).cuda()
    gmap = torch.randn(B, S, Cc, generator=g).cuda()
    L = lib()
    ws = torch.empty(int(L.pv_debug_convhead_ws(B, C.c_longlong(Fdim), zd)), dtype=torch.uint8, device="cuda")
    wt = torch.empty(zd, Fdim, device="cuda")
    a = torch.full((B, S, Cc), float("nan"), device="cuda")
    dw = torch.full((Fdim, zd), float("nan"), device="cuda")
    db = torch.full((Fdim,), float("nan"), device="cuda")
    dz = torch.full((B, zd), float("nan"), device="cuda")
    for what in range(4):
        rc = L.pv_debug_l2f(what, ptr(w), ptr(wt), ptr(bias), ptr(z), ptr(a), ptr(gmap), ptr(dw), ptr(db), ptr(dz), B, S, Cc, zd,
                            ptr(ws), C.c_longlong(ws.numel()), stream())
        assert rc == 0
    f0 = z.double() @ w.double().t() + bias.double()                          # (B, C*S), channels-first flatten
    assert rel_l2(a, f0.reshape(B, Cc, S).permute(0, 2, 1)) < 2e-6
    gf = gmap.double().permute(0, 2, 1).reshape(B, Fdim)
    assert rel_l2(dw, gf.t() @ z.double()) < 2e-6
    assert rel_l2(db, gf.sum(0)) < 2e-6
    assert rel_l2(dz, gf @ w.double()) < 2e-6


K1_CASES = [  # (rows, Cin, Cout, act): the 1-D decoder's shapes of BASELINE config 5 and ragged / unaligned ones
    (4096, 128, 128, "none"), (8192, 64, 64, "lrelu"), (16384, 32, 32, "tanh"), (32768, 32, 1, "none"), (50, 48, 20, "relu"),
    (33, 7, 5, "sigmoid"), (1000, 256, 96, "softplus"), (17, 16, 4, "lrelu"), (70000, 1, 3, "none")]


@pytest.mark.parametrize("rows,Ci,Co,act", K1_CASES)
def test_kernel1_convolution_family(gpu_device, rows, Ci, Co, act):
    """pv_conv_k1.hip: the kernel-1 convolution (a Linear over the pixels of a channels-last map) forward, input gradient
    (with the producing layer's activation derivative folded in) and weight / bias gradient, against float64."""
    g = torch.Generator().manual_seed(rows + Ci + Co)
    x = torch.randn(rows, Ci, generator=g).cuda()
    w = (torch.randn(Co, Ci, generator=g) / Ci ** 0.5).cuda()
    bias = torch.randn(Co, generator=g).cuda()
    gy = torch.randn(rows, Co, generator=g).cuda()
    yprev = act_fn(act, torch.randn(rows, Ci, generator=g)).cuda()        # the producing layer's output (for act')
    L = lib()
    L.pv_debug_k1_ws.restype = C.c_longlong
    ws = torch.empty(max(int(L.pv_debug_k1_ws(C.c_longlong(rows), Ci, Co)), 256), dtype=torch.uint8, device="cuda")
    out = torch.full((rows, Co), float("nan"), device="cuda")
    gin = torch.full((rows, Ci), float("nan"), device="cuda")
    dw = torch.full((Co, Ci), float("nan"), device="cuda")
    db = torch.full((Co,), float("nan"), device="cuda")
    args = (C.c_longlong(rows), Ci, Co)
    tail = (ptr(ws), C.c_longlong(ws.numel()), stream())
    assert L.pv_debug_k1(0, ptr(x), ptr(w), ptr(bias), ptr(out), P(0), *args, ACTS[act], P(0), *tail) == 0
    assert L.pv_debug_k1(1, ptr(gy), ptr(w), P(0), ptr(gin), P(0), *args, ACTS[act], ptr(yprev), *tail) == 0
    assert L.pv_debug_k1(2, ptr(gy), ptr(x), P(0), ptr(dw), ptr(db), *args, 0, P(0), *tail) == 0
    ref = act_fn(act, x.double() @ w.double().t() + bias.double())
    assert rel_l2(out, ref) < 2e-6
    assert rel_l2(gin, (gy.double() @ w.double()) * act_grad_of_output(act, yprev.double())) < 2e-6
    assert rel_l2(dw, gy.double().t() @ x.double()) < 2e-6
    assert rel_l2(db, gy.double().sum(0)) < 2e-6


@pytest.mark.parametrize("rows,Ci,Co,act", [(4096, 128, 128, "lrelu"), (1024, 64, 64, "tanh"), (37, 24, 10, "relu"), (500, 32, 32, "none")])
def test_kernel1_convolution_with_fused_upsample(gpu_device, rows, Ci, Co, act):
    """the 1-D nearest 2x upsample that follows an UpsampleBlock's convolution, fused: forward rows stored twice, backward
    kernels reading g[2p] + g[2p + 1]."""
    g = torch.Generator().manual_seed(rows + Ci + Co + 1)
    x = torch.randn(rows, Ci, generator=g).cuda()
    w = (torch.randn(Co, Ci, generator=g) / Ci ** 0.5).cuda()
    bias = torch.randn(Co, generator=g).cuda()
    gup = torch.randn(2 * rows, Co, generator=g).cuda()
    yprev = act_fn(act, torch.randn(rows, Ci, generator=g)).cuda()
    L = lib()
    L.pv_debug_k1_ws.restype = C.c_longlong
    ws = torch.empty(max(int(L.pv_debug_k1_ws(C.c_longlong(rows), Ci, Co)), 256), dtype=torch.uint8, device="cuda")
    out = torch.full((2 * rows, Co), float("nan"), device="cuda")
    gin = torch.full((rows, Ci), float("nan"), device="cuda")
    dw = torch.full((Co, Ci), float("nan"), device="cuda")
    db = torch.full((Co,), float("nan"), device="cuda")
    args = (C.c_longlong(rows), Ci, Co)
    tail = (ptr(ws), C.c_longlong(ws.numel()), stream())
    assert L.pv_debug_k1(4, ptr(x), ptr(w), ptr(bias), ptr(out), P(0), *args, 0, P(0), *tail) == 0
    assert L.pv_debug_k1(5, ptr(gup), ptr(w), P(0), ptr(gin), P(0), *args, ACTS[act], ptr(yprev), *tail) == 0
    assert L.pv_debug_k1(6, ptr(gup), ptr(x), P(0), ptr(dw), ptr(db), *args, 0, P(0), *tail) == 0
    ref = (x.double() @ w.double().t() + bias.double()).repeat_interleave(2, dim=0)
    assert rel_l2(out, ref) < 2e-6
    gy = gup.double().reshape(rows, 2, Co).sum(1)
    assert rel_l2(gin, (gy @ w.double()) * act_grad_of_output(act, yprev.double())) < 2e-6
    assert rel_l2(dw, gy.t() @ x.double()) < 2e-6
    assert rel_l2(db, gy.sum(0)) < 2e-6


@pytest.mark.parametrize("B,Ln,Ci,Co", [(256, 16, 128, 128), (64, 32, 64, 64), (5, 24, 20, 12), (3, 7, 32, 8), (2, 1, 16, 16), (9, 128, 32, 1)])
def test_kernel3_1d_weight_gradient_on_the_register_fed_kernel(gpu_device, B, Ln, Ci, Co):
    """the kernel-3, padding-1 Conv1d weight gradient as three shifted, boundary-masked kernel-1 problems, against float64."""
    g = torch.Generator().manual_seed(B + Ln + Ci + Co)
    x = torch.randn(B, Ln, Ci, generator=g).cuda()
    gy = torch.randn(B, Ln, Co, generator=g).cuda()
    L = lib()
    L.pv_debug_k1_ws.restype = C.c_longlong
    rows = B * Ln
    ws = torch.empty(max(int(L.pv_debug_k1_ws(C.c_longlong(rows), Ci, Co)), 256), dtype=torch.uint8, device="cuda")
    dw = torch.full((Co, Ci, 3), float("nan"), device="cuda")
    db = torch.full((Co,), float("nan"), device="cuda")
    assert L.pv_debug_k1(3, ptr(gy), ptr(x), P(0), ptr(dw), ptr(db), C.c_longlong(rows), Ci, Co, Ln, P(0), ptr(ws),
                         C.c_longlong(ws.numel()), stream()) == 0
    xd = x.double().permute(0, 2, 1)
    wd = torch.zeros(Co, Ci, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    y = F.conv1d(xd, wd, None, padding=1)
    (gw,) = torch.autograd.grad(y, wd, gy.double().permute(0, 2, 1))
    assert rel_l2(dw, gw) < 2e-6
    assert rel_l2(db, gy.double().sum((0, 1))) < 2e-6


@pytest.mark.parametrize("B,Ln,Ci,Co", [(256, 16, 128, 128), (64, 32, 64, 32), (5, 24, 20, 40), (3, 7, 32, 8), (9, 128, 32, 64), (4, 10, 48, 17),
                                        (1, 16, 32, 32), (8, 16, 32, 32), (2, 300, 24, 24)])   # (small tensors: the first pixels' left tap weighs most)
def test_batched_weight_gradients_one_launch(gpu_device, B, Ln, Ci, Co):
    """PvK1Batch: a kernel-1 and a Conv1d kernel-3 weight gradient recorded and run by ONE table-driven launch (the 32 x 32 tile
    form with all taps in the workgroup when both channel counts exceed 16), then one reduction launch; against float64."""
    g = torch.Generator().manual_seed(B + Ln + Ci + Co + 7)
    x = torch.randn(B, Ln, Ci, generator=g).cuda()
    gy = torch.randn(B, Ln, Co, generator=g).cuda()
    L = lib()
    L.pv_debug_k1_ws.restype = C.c_longlong
    rows = B * Ln
    ws = torch.empty(2 * int(L.pv_debug_k1_ws(C.c_longlong(rows), Ci, Co)) + 4096, dtype=torch.uint8, device="cuda")
    dw1 = torch.full((Co, Ci), float("nan"), device="cuda")
    db1 = torch.full((Co,), float("nan"), device="cuda")
    dw3 = torch.full((Co, Ci, 3), float("nan"), device="cuda")
    db3 = torch.full((Co,), float("nan"), device="cuda")
    assert L.pv_debug_k1_batch(ptr(gy), ptr(x), C.c_longlong(rows), Ln, Ci, Co, ptr(dw1), ptr(db1), ptr(dw3), ptr(db3), ptr(ws),
                               C.c_longlong(ws.numel()), stream()) == 0
    g2, x2 = gy.double().reshape(rows, Co), x.double().reshape(rows, Ci)
    assert rel_l2(dw1, g2.t() @ x2) < 2e-6
    assert rel_l2(db1, g2.sum(0)) < 2e-6
    wd = torch.zeros(Co, Ci, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    y = F.conv1d(x.double().permute(0, 2, 1), wd, None, padding=1)
    (gw,) = torch.autograd.grad(y, wd, gy.double().permute(0, 2, 1))
    assert rel_l2(dw3, gw) < 2e-6
    assert rel_l2(db3, g2.sum(0)) < 2e-6
