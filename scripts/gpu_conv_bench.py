"""Accuracy and time of the kernel-3 convolution kernels on the conv layer shapes of BASELINE configs C4 / C5, through
the library's test hook pv_debug_conv3 / pv_debug_conv3_wgrad (mode 0: f32-input MFMA direct kernel, 1: its bf16
two-piece form, 2 / 3: the split-operand bf16 kernels of pv_conv_sp.hip with 2 / 3 pieces).
    python scripts/gpu_conv_bench.py [fwd|wgrad|all] [B]
Accuracy: relative l2 error against an fp64 convolution (torch, on the GPU) at batch 4.  Time: HIP events over 20 calls."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from pyroved_amd import _abi

lib = C.CDLL(_abi.LIB_PATH)
P = C.c_void_p


def ptr(t):
    return P(t.data_ptr()) if t is not None else P(0)


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def conv_call(mode, x, w, bias, flip, act=0):
    """x: (B, H, W, Cin') channels-last; w: (Co, Ci, 3, 3).  flip=0: out (B,H,W,Co); flip=1: x has Co channels, out Ci."""
    B, H, W_, _ = x.shape
    Co, Ci = w.shape[:2]
    N = Ci if flip else Co
    out = torch.empty(B, H, W_, N, device="cuda")
    n = max(Co, Ci)
    scratch = torch.empty(((n + 63) // 64) * 64 * n * 9 * 6 + 4096, dtype=torch.uint8, device="cuda")
    def run():
        rc = lib.pv_debug_conv3(mode, ptr(x), B, H, W_, 2, ptr(w), Co, Ci, flip, ptr(bias), ptr(out), act, ptr(scratch),
                                P(0), 0, P(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    return out, run


def wgrad_call(mode, dy, x):
    B, H, W_, Co = dy.shape
    Ci = x.shape[-1]
    dw = torch.empty(Co, Ci, 3, 3, device="cuda")
    db = torch.empty(Co, device="cuda")
    lib.pv_debug_conv3_wgrad_ws.restype = C.c_longlong
    nb = lib.pv_debug_conv3_wgrad_ws(mode, B, H, W_, Ci, Co, 2)
    ws = torch.empty(max(int(nb), 256), dtype=torch.uint8, device="cuda")
    def run():
        rc = lib.pv_debug_conv3_wgrad(mode, ptr(dy), ptr(x), B, H, W_, Ci, 2, ptr(dw), ptr(db), Co, ptr(ws), C.c_longlong(ws.numel()),
                                      P(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    return dw, db, run


def time_of(run, n=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


SHAPES = [  # (H, W, Cin, Cout): the default convEncoderNet stack at 64x64 input (after its first layer)
    (32, 32, 32, 64), (32, 32, 64, 64), (16, 16, 64, 128), (16, 16, 128, 128)]


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    modes = [int(m) for m in os.environ.get("MODES", "0,1,2,3").split(",")]
    g = torch.Generator().manual_seed(0)
    for (H, W_, Ci, Co) in SHAPES:
        w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).cuda()
        bias = torch.randn(Co, generator=g).cuda() * 0.1
        xs = torch.randn(4, H, W_, Ci, generator=g).cuda()
        xb = torch.randn(B, H, W_, Ci, generator=g).cuda()
        gmac = B * H * W_ * 9 * Ci * Co / 1e9
        if what in ("fwd", "all"):
            ref = F.conv2d(xs.permute(0, 3, 1, 2).double(), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
            dys = torch.randn(4, H, W_, Co, generator=g).cuda()
            dyb = torch.randn(B, H, W_, Co, generator=g).cuda()
            refd = F.conv_transpose2d(dys.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
            for m in modes:
                out, run = conv_call(m, xs, w, bias, 0); run(); e = rel(out, ref)
                outd, rund = conv_call(m, dys, w, None, 1); rund(); ed = rel(outd, refd)
                _, runb = conv_call(m, xb, w, bias, 0); tf = time_of(runb)
                _, runbd = conv_call(m, dyb, w, None, 1); td = time_of(runbd)
                print("fwd  %3dx%-3d %3d->%-3d B=%d mode %d: err %.1e / dgrad %.1e | %7.1f us (%5.1f TF) / dgrad %7.1f us (%5.1f TF)"
                      % (H, W_, Ci, Co, B, m, e, ed, tf, 2 * gmac / tf * 1e3, td, 2 * gmac / td * 1e3), flush=True)
        if what in ("wgrad", "all"):
            dys = torch.randn(4, H, W_, Co, generator=g).cuda()
            dyb = torch.randn(B, H, W_, Co, generator=g).cuda()
            xd = xs.permute(0, 3, 1, 2).double().requires_grad_(False)
            wd = w.double().requires_grad_(True)
            y = F.conv2d(xd, wd, None, padding=1)
            (gw,) = torch.autograd.grad(y, wd, dys.permute(0, 3, 1, 2).double())
            gb = dys.double().sum((0, 1, 2))
            for m in modes:
                dw, db, run = wgrad_call(m, dys, xs); run(); e = rel(dw, gw); eb = rel(db, gb)
                _, _, runb = wgrad_call(m, dyb, xb); tw = time_of(runb)
                print("wgrad %3dx%-3d %3d->%-3d B=%d mode %d: err %.1e (bias %.1e) | %7.1f us (%5.1f TF)"
                      % (H, W_, Ci, Co, B, m, e, eb, tw, 2 * gmac / tw * 1e3), flush=True)


main()
