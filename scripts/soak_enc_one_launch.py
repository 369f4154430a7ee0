"""The one-launch encoder against the two-launch form on the same inputs, many times (a hand-off that can race shows up as a rare
mismatch): python scripts/soak_enc_one_launch.py [steps] [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv
from pyroved_amd import _abi
dbg = C.CDLL(_abi.LIB_PATH)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
b = int(sys.argv[2]) if len(sys.argv) > 2 else 256
m = pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
eng = m.engine(fused=3)
g = torch.Generator(device="cuda").manual_seed(0)
bad = 0
for i in range(steps):
    x = torch.rand(b, 28, 28, generator=g, device="cuda")
    eps = torch.randn(b, m.z_dim, generator=g, device="cuda")
    out = []
    for two in (0, 1):
        dbg.pv_debug_enc_two(two)
        eng.loss_and_grads(x, eps)
        out.append((eng.scalars.clone(), eng.grad.clone()))
    if not (torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])):
        bad += 1
        if bad < 5:
            print("step %d differs: %s vs %s" % (i, out[0][0].tolist(), out[1][0].tolist()))
    if i % 64 == 0:
        eng.adam_step()                                  # (let the weights move)
dbg.pv_debug_enc_two(-1)
torch.cuda.synchronize()
print("%d steps at batch %d: %d mismatches" % (steps, b, bad))
