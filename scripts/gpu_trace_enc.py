"""Phase stamps of the encoder launch (library built with -DEN_TRACE: pv_encoder.hip): cycles per phase of the first row block's
workgroup; PV_ENC_TWO=1 for the two-launch form."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv
from pyroved_amd import _abi
import sys as _s
SH = (64, 64) if len(_s.argv) > 1 and _s.argv[1] == "c4fc" else (28, 28)
BB = 128 if SH[0] == 64 else 256
model = pv.models.iVAE(SH, 2, ["r", "t", "s"] if SH[0] == 64 else ["r", "t"], seed=1, device="cuda")
eng = model.engine(fused=2 if SH[0] == 64 else 3)
eng.enc_fold = False
x = torch.rand(BB, SH[0], SH[1], generator=torch.Generator().manual_seed(0)).cuda()
eps = torch.randn(BB, model.z_dim).cuda()
for _ in range(5):
    eng.loss_and_grads(x, eps)
torch.cuda.synchronize()
lib = C.CDLL(_abi.LIB_PATH)
buf = (C.c_longlong * 64)()
print("rc", lib.pv_debug_read_enc_trace(buf, 64))
t = [buf[i] for i in range(7)]
names = ["l0 load + prefetch requests + LDS", "hidden layer 1", "head", "z / KL terms", "block sums", "split latent", "?"]
print([(names[i], t[i + 1] - t[i]) for i in range(6)], "total", t[6] - t[0])

bp = (C.c_longlong * 16)()
if hasattr(lib, "pv_debug_read_enc_trace_p") and lib.pv_debug_read_enc_trace_p(bp) == 0 and bp[0]:
    print("producer tile (0, 0): first requests issued %d | K loop done %d | tile stored %d (cycles from its start); consumer row block 0 started %d cycles %s that producer"
          % (bp[1] - bp[0], bp[2] - bp[0], bp[3] - bp[0], abs(buf[0] - bp[0]), "after" if buf[0] >= bp[0] else "before"))
