"""Phase timing of one workgroup of the kernel-3 convolution weight-gradient kernel (a -DSW_TRACE build:
scripts/mkvariant_file.sh swtrace pv_conv_sp.hip -DSW_TRACE; PV_LIB_PATH=pyroved_amd/variants/lib_swtrace.so python scripts/gpu_trace_sw.py [mode])."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyroved_amd import _abi
lib = C.CDLL(_abi.LIB_PATH)
P = C.c_void_p
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 4
names = ["barrier A (previous reads done)", "scales + split + LDS stores", "barrier B", "fetch issue (next tile)", "fragment reads + MFMAs", "load wait + wave max"]
for (H, W, Ci, Co) in ([] if len(sys.argv) > 2 else [(32, 32, 32, 64), (32, 32, 64, 64), (16, 16, 64, 128), (16, 16, 128, 128)]):
    B = 256
    g = torch.Generator().manual_seed(0)
    dy = torch.randn(B, H, W, Co, generator=g).cuda()
    x = torch.randn(B, H, W, Ci, generator=g).cuda()
    dw = torch.empty(Co, Ci, 3, 3, device="cuda"); db = torch.empty(Co, device="cuda")
    lib.pv_debug_conv3_wgrad_ws.restype = C.c_longlong
    nb = lib.pv_debug_conv3_wgrad_ws(mode, B, H, W, Ci, Co, 2)
    ws = torch.empty(max(int(nb), 256), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        rc = lib.pv_debug_conv3_wgrad(mode, P(dy.data_ptr()), P(x.data_ptr()), B, H, W, Ci, 2, P(dw.data_ptr()), P(db.data_ptr()), Co,
                                      P(ws.data_ptr()), C.c_longlong(ws.numel()), P(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    torch.cuda.synchronize()
    buf = (C.c_longlong * 16)()
    assert lib.pv_debug_read_trace_sw(buf, 16) == 0
    n = max(buf[6], 1)
    print("wgrad %dx%d %d->%d mode %d: %d tiles, %d cycles in all (shader clock ticks), per tile:" % (H, W, Ci, Co, mode, buf[6], buf[7]))
    for k in range(6):
        print("   %-34s %8.0f" % (names[k], buf[k] / n))

# forward / input-gradient kernel (-DSP_TRACE build): python scripts/gpu_trace_sw.py <mode> fwd
if len(sys.argv) > 2 and sys.argv[2] == "fwd":
    fn = ["prologue (addresses, first requests, first data + wave max)", "barrier A (per chunk)", "scale + split + LDS stores (patch, first weight stage)",
          "barrier B", "3 weight stages: fragment reads + MFMAs + 2 barriers each", "next chunk's data wait + wave max", "epilogue (bias, activation, stores issued)"]
    for (H, W, Ci, Co) in [(32, 32, 32, 64), (32, 32, 64, 64), (16, 16, 64, 128), (16, 16, 128, 128)]:
        B = 256
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, H, W, Ci, generator=g).cuda()
        w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).cuda()
        out = torch.empty(B, H, W, Co, device="cuda")
        n = max(Co, Ci)
        scratch = torch.empty(((n + 63) // 64) * 64 * n * 9 * 6 + 4096, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            rc = lib.pv_debug_conv3(mode, P(x.data_ptr()), B, H, W, 2, P(w.data_ptr()), Co, Ci, 0, P(0), P(out.data_ptr()), 0, P(scratch.data_ptr()),
                                    P(0), 0, P(torch.cuda.current_stream().cuda_stream))
            assert rc == 0
        torch.cuda.synchronize()
        buf = (C.c_longlong * 16)()
        assert lib.pv_debug_read_trace_sp(buf, 16) == 0
        nch = max(buf[7], 1)
        print("fwd %dx%d %d->%d mode %d: %d chunks, %d cycles in all:" % (H, W, Ci, Co, mode, buf[7], buf[8]))
        for k in range(7):
            print("   %-62s %8.0f%s" % (fn[k], buf[k] / (nch if 1 <= k <= 5 else 1), " per chunk" if 1 <= k <= 5 else ""))
