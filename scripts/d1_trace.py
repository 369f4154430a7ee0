"""Per-step cycle stamps of the fused 1-D decoder kernels (library built with -DD1_TRACE): workgroup 0, thread 0."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv
from pyroved_amd import _abi
lib = C.CDLL(_abi.LIB_PATH)
B = 256
m = pv.models.VED((64, 64), (128,), latent_dim=2, seed=1, device="cuda")
eng = m.engine(fused=2)
g = torch.Generator().manual_seed(0)
x, y, eps = torch.rand(B, 1, 64, 64, generator=g).cuda(), torch.rand(B, 1, 128, generator=g).cuda(), torch.randn(B, 2, generator=g).cuda()
for _ in range(5):
    eng.loss_and_grads(x, eps, 1.0, y)
torch.cuda.synchronize()
buf = (C.c_longlong * 128)()
assert lib.pv_debug_read_d1_trace(buf) == 0
for k, name in ((0, "forward"), (1, "backward")):
    t = [buf[64 * k + i] for i in range(14)]
    print(name, "cycles per stage (stage 0 = operand request + staging):", [t[i + 1] - t[i] for i in range(11) if t[i + 1] > t[i]], "total", max(t) - t[0])
for k, name in ((0, "forward"), (1, "backward")):
    for s_, what in ((32, "step 1"), (40, "step 8")):
        t = [buf[64 * k + s_ + j] for j in range(7)]
        if min(t) > 0:
            print(name, what, "entry->halo %d  ->mma start %d  mma %d  epilogue %d  next requests %d  barrier %d" % tuple(t[j + 1] - t[j] for j in range(6)))
t = [buf[0], buf[48], buf[49], buf[50], buf[51], buf[1]]
if min(t) > 0:
    print("forward stage 0: first requests %d  head finish %d  sample + KL %d  latent_to_features %d  halo + barrier %d" % tuple(t[j + 1] - t[j] for j in range(5)))
