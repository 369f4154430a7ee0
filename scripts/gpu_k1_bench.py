"""Time of the 1x1-convolution (= Linear over pixels) forward / backward on the shapes of VED C5's 1-D decoder, through
ops.linear_act (pv_linear_fwd / pv_linear_bwd):  python scripts/gpu_k1_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyroved_amd import ops

SHAPES = [(4096, 128, 128), (8192, 64, 64), (16384, 32, 32), (32768, 32, 1)]   # rows, Cin, Cout


def time_of(run, n=30):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rows, ci, co in SHAPES:
    x = torch.randn(rows, ci, device="cuda", requires_grad=True)
    w = (torch.randn(co, ci, device="cuda") / ci ** 0.5).requires_grad_(True)
    b = torch.zeros(co, device="cuda", requires_grad=True)
    g = torch.randn(rows, co, device="cuda")
    with torch.no_grad():
        tf = time_of(lambda: ops.linear_act(x, w, b, None))
    y = ops.linear_act(x, w, b, None)
    tb = time_of(lambda: torch.autograd.grad(y, (x, w, b), g, retain_graph=True))
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    err = ((y.double() - ref).norm() / ref.norm()).item()
    print("rows %6d  %3d -> %-3d : fwd %6.1f us   bwd (dgrad + wgrad) %6.1f us   err %.1e" % (rows, ci, co, tf, tb, err), flush=True)
