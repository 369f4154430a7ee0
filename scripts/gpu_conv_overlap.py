"""Does a layer's weight gradient overlap with its input gradient when both run at once?  Two streams, no events inside
the timed region: 20 x pv_conv3_sp_wgrad on one, 20 x pv_conv3_sp (input-gradient form) on the other, against each alone."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyroved_amd import _abi
lib = C.CDLL(_abi.LIB_PATH)
lib.pv_debug_conv3_wgrad_ws.restype = C.c_longlong
P = C.c_void_p
ptr = lambda t: P(t.data_ptr()) if t is not None else P(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for (H, W, Ci, Co) in [(32, 32, 64, 64), (16, 16, 128, 128)]:
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).cuda()
    x = torch.randn(B, H, W, Ci, generator=g).cuda()
    dy = torch.randn(B, H, W, Co, generator=g).cuda()
    din = torch.empty(B, H, W, Ci, device="cuda")
    dw = torch.empty(Co, Ci, 3, 3, device="cuda"); db = torch.empty(Co, device="cuda")
    n = max(Co, Ci)
    scratch = torch.empty(((n + 63) // 64) * 64 * n * 9 * 6 + 4096, dtype=torch.uint8, device="cuda")
    ws = torch.empty(int(lib.pv_debug_conv3_wgrad_ws(3, B, H, W, Ci, Co, 2)), dtype=torch.uint8, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    def dgrad(s):
        assert lib.pv_debug_conv3(3, ptr(dy), B, H, W, 2, ptr(w), Co, Ci, 1, P(0), ptr(din), 0, ptr(scratch), P(0), 0, P(s.cuda_stream)) == 0
    def wgrad(s):
        assert lib.pv_debug_conv3_wgrad(3, ptr(dy), ptr(x), B, H, W, Ci, 2, ptr(dw), ptr(db), Co, ptr(ws), C.c_longlong(ws.numel()), P(s.cuda_stream)) == 0
    def run(fa, fb, n=20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            if fa: fa(sa)
            if fb: fb(sb)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    for _ in range(2): run(dgrad, wgrad, 5)
    td, tw, tb = run(dgrad, None), run(None, wgrad), run(dgrad, wgrad)
    print("%dx%d %d->%d B=%d: dgrad %.1f us, wgrad %.1f us, sum %.1f, both at once %.1f us" % (H, W, Ci, Co, B, td, tw, td + tw, tb), flush=True)
