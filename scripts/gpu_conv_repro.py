import ctypes as C, torch, sys
sys.path.insert(0, "/root/repo")
from pyroved_amd import _abi
lib = C.CDLL(_abi.LIB_PATH); P = C.c_void_p
lib.pv_debug_conv3_wgrad_ws.restype = C.c_longlong
ptr = lambda t: P(t.data_ptr()) if t is not None else P(0)
st = lambda: P(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(0)
for mode, nd, B, H, W, Ci, Co in [(5, 1, 256, 16, 1, 128, 128), (5, 1, 256, 32, 1, 128, 64), (5, 1, 256, 64, 1, 64, 32), (5, 1, 256, 32, 1, 64, 64),
                                  (4, 2, 256, 32, 32, 64, 64), (4, 2, 256, 16, 16, 128, 128)]:
    w = (torch.randn(Co, Ci, *([3] * nd), generator=g) / (3 * Ci ** 0.5)).cuda()
    x = (torch.randn(B, H, W, Ci, generator=g) * 1e-3).cuda()
    n = max(Co, Ci)
    scratch = torch.empty(((n + 63) // 64) * 64 * n * 9 * 6 + 4096, dtype=torch.uint8, device="cuda")
    outs = []
    for flip in (0, 1):
        xin = x if not flip else (torch.randn(B, H, W, Co, generator=g) * 1e-5).cuda()
        res = []
        for rep in range(6):
            out = torch.full((B, H, W, Ci if flip else Co), float("nan"), device="cuda")
            rc = lib.pv_debug_conv3(mode, ptr(xin), B, H, W, nd, ptr(w), Co, Ci, flip, P(0), ptr(out), 0, ptr(scratch), P(0), 0, st())
            assert rc == 0
            res.append(out.clone())
        print(mode, nd, (H, W, Ci, Co), "flip", flip, "identical:", all(torch.equal(res[0], r) for r in res[1:]), "nan:", bool(torch.isnan(res[0]).any()))
    if nd == 2:
        dy = (torch.randn(B, H, W, Co, generator=g) * 1e-5).cuda()
        ws = torch.empty(int(lib.pv_debug_conv3_wgrad_ws(4, B, H, W, Ci, Co, 2)), dtype=torch.uint8, device="cuda")
        res = []
        for rep in range(6):
            dw = torch.empty(Co, Ci, 3, 3, device="cuda"); db = torch.empty(Co, device="cuda")
            assert lib.pv_debug_conv3_wgrad(4, ptr(dy), ptr(x), B, H, W, Ci, 2, ptr(dw), ptr(db), Co, ptr(ws), C.c_longlong(ws.numel()), st()) == 0
            res.append((dw.clone(), db.clone()))
        print("wgrad identical:", all(torch.equal(res[0][0], r[0]) and torch.equal(res[0][1], r[1]) for r in res[1:]))
