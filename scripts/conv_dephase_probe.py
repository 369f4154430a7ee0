"""Would de-phased convolution launches run faster?  The four kernel-3 layers of the C5 encoder (independent inputs here) one after
the other on one stream, against the same four launches on four streams at once (workgroups of different layers, in different
phases, share the CUs).  mode 4: two fp16 pieces, 7: one.   python scripts/conv_dephase_probe.py [mode] [B]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyroved_amd import _abi
lib = C.CDLL(_abi.LIB_PATH)
P = C.c_void_p
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = torch.Generator().manual_seed(0)
layers = []
for (H, W, Ci, Co) in [(32, 32, 32, 64), (32, 32, 64, 64), (16, 16, 64, 128), (16, 16, 128, 128)]:
    x = torch.randn(B, H, W, Ci, generator=g).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).cuda()
    out = torch.empty(B, H, W, Co, device="cuda")
    n = max(Co, Ci)
    scratch = torch.empty(((n + 63) // 64) * 64 * n * 9 * 6 + 4096, dtype=torch.uint8, device="cuda")
    layers.append((x, w, out, scratch, H, W, Ci, Co))
streams = [torch.cuda.Stream() for _ in layers]

def launch(l, s):
    x, w, out, scratch, H, W, Ci, Co = l
    rc = lib.pv_debug_conv3(mode, P(x.data_ptr()), B, H, W, 2, P(w.data_ptr()), Co, Ci, 0, P(0), P(out.data_ptr()), 0, P(scratch.data_ptr()),
                            P(0), 0, P(s.cuda_stream))
    assert rc == 0

def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

cur = torch.cuda.current_stream()
def seq():
    for l in layers:
        launch(l, cur)
def par():
    ev = torch.cuda.Event()
    ev.record(cur)
    for l, s in zip(layers, streams):
        s.wait_event(ev)
        launch(l, s)
    for s in streams:
        cur.wait_stream(s)
print("mode %d batch %d: four layers on one stream %.1f us, on four streams at once %.1f us" % (mode, B, timed(seq), timed(par)))
