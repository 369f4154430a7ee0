"""A few SVI steps of one secondary configuration, for rocprofv3 --kernel-trace --stats:  prof_cfg.py 64x64|c3|b1024 [fused]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv
which = sys.argv[1]; fused = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if which == "64x64":
    model, B, dd = pv.models.iVAE((64, 64), 2, ["r", "t", "s"], seed=1, device="cuda"), 128, (64, 64)
elif which == "c3":
    model, B, dd = pv.models.jiVAE((28, 28), 2, 10, ["r"], seed=1, device="cuda"), 512, (28, 28)
else:
    model, B, dd = pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda"), 1024, (28, 28)
eng = model.engine(fused=fused)
x = torch.rand(B, *dd).cuda(); eps = torch.randn(B, model.z_dim).cuda()
for i in range(25):
    eng.loss_and_grads(x, eps); eng.adam_step()
torch.cuda.synchronize()
