"""Phase timing of the fused bf16 kernel (PV_FD_ABLATE=256): prints cycles per phase for workgroup 0 / wave 0."""
import ctypes as C, os, sys
os.environ["PV_FD_ABLATE"] = os.environ.get("PV_TRACE_MASK", "256")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv
from pyroved_amd import _abi
fused = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
eng = model.engine(fused=fused)
x = torch.rand(256, 28, 28, generator=torch.Generator().manual_seed(0)).cuda()
eps = torch.randn(256, model.z_dim).cuda()
for _ in range(3):
    eng.loss_and_grads(x, eps)
torch.cuda.synchronize()
lib = C.CDLL(_abi.LIB_PATH)
buf = (C.c_longlong * 64)()
print("rc", lib.pv_debug_read_trace(buf, 64))
names = ["start", "coord+loads", "fwd L1", "fwd L2(+tanh)", "tanh+loss+dwo", "presplit", "exch L2", "dgrad L2", "dgrad L1(+h0)", "-", "presplit+exch L1", "rowlocal", "exch 0"]
for t in range(4):
    st = [buf[t * 16 + k] for k in range(13)]
    if st[0] == 0: continue
    prev = st[0]; out = []
    for k in range(1, 13):
        if st[k]:
            out.append("%s=%d" % (names[k], st[k] - prev)); prev = st[k]
    print("tile", t, "total", prev - st[0], " ".join(out))
if int(os.environ.get("PV_FD_ABLATE", "0")) & 512:
    buf2 = (C.c_longlong * 160)()
    lib.pv_debug_read_trace(buf2, 160)
    base = buf2[128]
    print("exchange L2 chunks (cycles since first stamp): store_done, after_bar1, after_consume, after_bar2")
    for c in range(4):
        print(" chunk", c, [buf2[128 + 4 * c + k] - base for k in range(4)])
