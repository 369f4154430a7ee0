"""Time of the batched register-fed weight-gradient launch (pv_debug_k1_batch: a kernel-1 and a Conv1d kernel-3 problem of the
same shape, one table launch + one reduction launch) on VED C5's decoder shapes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyroved_amd import _abi
lib = C.CDLL(_abi.LIB_PATH); P = C.c_void_p
lib.pv_debug_k1_ws.restype = C.c_longlong
for (B, Ln, Ci, Co) in [(256, 16, 128, 128), (256, 32, 128, 64), (256, 32, 64, 64), (256, 64, 64, 32), (256, 128, 32, 1)]:
    rows = B * Ln
    x = torch.randn(B, Ln, Ci, device="cuda"); gy = torch.randn(B, Ln, Co, device="cuda")
    ws = torch.empty(2 * int(lib.pv_debug_k1_ws(C.c_longlong(rows), Ci, Co)) + 4096, dtype=torch.uint8, device="cuda")
    dw1 = torch.empty(Co, Ci, device="cuda"); db1 = torch.empty(Co, device="cuda")
    dw3 = torch.empty(Co, Ci, 3, device="cuda"); db3 = torch.empty(Co, device="cuda")
    def run():
        rc = lib.pv_debug_k1_batch(P(gy.data_ptr()), P(x.data_ptr()), C.c_longlong(rows), Ln, Ci, Co, P(dw1.data_ptr()), P(db1.data_ptr()),
                                   P(dw3.data_ptr()), P(db3.data_ptr()), P(ws.data_ptr()), C.c_longlong(ws.numel()),
                                   P(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize()
    print("B %d L %d %d->%d: %.1f us per (table launch + reduction)" % (B, Ln, Ci, Co, e0.elapsed_time(e1) / 100 * 1e3), flush=True)
