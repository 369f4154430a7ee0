"""Time of the 1-D kernel-3 convolution kernels on VED's decoder layer shapes (batch 256): forward (modes 0 f32, 1 bf16 two-piece,
5 fp16 two-piece) and weight gradient (modes 0, 1), through the debug hooks."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyroved_amd import _abi
lib = C.CDLL(_abi.LIB_PATH); P = C.c_void_p
lib.pv_debug_conv3_wgrad_ws.restype = C.c_longlong
ptr = lambda t: P(t.data_ptr()) if t is not None else P(0)
st = lambda: P(torch.cuda.current_stream().cuda_stream)
def timeit(run, n=30):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 256
g = torch.Generator().manual_seed(0)
for (L, Ci, Co) in [(16, 128, 128), (32, 128, 64), (32, 64, 64), (64, 64, 32)]:
    w = (torch.randn(Co, Ci, 3, generator=g) / (3 * Ci ** 0.5)).cuda()
    x = torch.randn(B, L, 1, Ci, generator=g).cuda()
    dy = torch.randn(B, L, 1, Co, generator=g).cuda()
    out = torch.empty(B, L, 1, Co, device="cuda")
    scratch = torch.empty(4 << 20, dtype=torch.uint8, device="cuda")
    line = "L=%d %d->%d:" % (L, Ci, Co)
    for m in (0, 1, 5):
        t = timeit(lambda: lib.pv_debug_conv3(m, ptr(x), B, L, 1, 1, ptr(w), Co, Ci, 0, P(0), ptr(out), 0, ptr(scratch), P(0), 0, st()))
        line += " fwd m%d %.1f us" % (m, t)
    for m in (0, 1):
        dw = torch.empty(Co, Ci, 3, device="cuda"); db = torch.empty(Co, device="cuda")
        ws = torch.empty(max(int(lib.pv_debug_conv3_wgrad_ws(m, B, L, 1, Ci, Co, 1)), 256), dtype=torch.uint8, device="cuda")
        t = timeit(lambda: lib.pv_debug_conv3_wgrad(m, ptr(dy), ptr(x), B, L, 1, Ci, 1, ptr(dw), ptr(db), Co, ptr(ws), C.c_longlong(ws.numel()), st()))
        line += " | wgrad m%d %.1f us" % (m, t)
    print(line, flush=True)
