"""Phase timing of pv_enc_fwd_kernel (library built with -DEN_TRACE): cycles between the stamps of workgroup 0."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv
from pyroved_amd import _abi
model = pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
eng = model.engine(fused=3)
x = torch.rand(256, 28, 28, generator=torch.Generator().manual_seed(0)).cuda()
eps = torch.randn(256, model.z_dim).cuda()
for _ in range(5):
    eng.loss_and_grads(x, eps); eng.adam_step()
torch.cuda.synchronize()
lib = C.CDLL(_abi.LIB_PATH)
buf = (C.c_longlong * 64)()
print("rc", lib.pv_debug_read_enc_trace(buf, 64))
names = ["entry", "prefetch issued + layer-0 output in LDS", "hidden layer 1", "head", "z / KL elementwise", "block sums", "split_latent",
         ]
st = [buf[k] for k in range(7)]
for k in range(1, 7):
    print("%8d  %s" % (st[k] - st[k - 1], names[k]))
print("total to stamp 6:", st[6] - st[0], "(+ the hz tail)")
