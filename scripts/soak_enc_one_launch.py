"""The one-launch encoder against the two-launch form on the same inputs, many times (a hand-off that can race shows up as a rare
mismatch): python scripts/soak_enc_one_launch.py [steps] [batch] [--contend]

--contend: a second stream keeps the GPU busy with long matrix multiplications of varying size while the steps run (the
consumers' producers are then dispatched late or not at all before the consumers' poll limit: the exact fallback of
csrc/pv_encoder.hip takes over), and every 8th step runs with the poll limit forced to 0.  Prints the mismatches (must be 0)
and how many consumers computed their own tiles."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv
from pyroved_amd import _abi
dbg = C.CDLL(_abi.LIB_PATH)
dbg.pv_debug_enc_late_count.restype = C.c_longlong
args = [a for a in sys.argv[1:] if not a.startswith("--")]
contend = "--contend" in sys.argv
steps = int(args[0]) if len(args) > 0 else 3000
b = int(args[1]) if len(args) > 1 else 256
m = pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
eng = m.engine(fused=3)
g = torch.Generator(device="cuda").manual_seed(0)
side = torch.cuda.Stream()
big = [torch.randn(n, n, device="cuda") for n in (1024, 3072, 6144)]
bad = 0
late0 = dbg.pv_debug_enc_late_count()
for i in range(steps):
    x = torch.rand(b, 28, 28, generator=g, device="cuda")
    eps = torch.randn(b, m.z_dim, generator=g, device="cuda")
    out = []
    for two in (0, 1):
        eng.enc_two_launch = bool(two)                   # plan flags (ABI v14 / v15)
        eng.enc_per_image = False                        # (round 6: this soak is about the TILED encoder's hand-off)
        eng.enc_no_wait = bool(contend and two == 0 and i % 8 == 7)
        if contend:
            with torch.cuda.stream(side):
                a = big[i % 3]
                for _ in range(1 + i % 3):
                    a = a @ big[i % 3]
        eng.loss_and_grads(x, eps)
        out.append((eng.scalars.clone(), eng.grad.clone()))
    if not (torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])):
        bad += 1
        if bad < 5:
            print("step %d differs: %s vs %s" % (i, out[0][0].tolist(), out[1][0].tolist()))
    if i % 64 == 0:
        eng.adam_step()                                  # (let the weights move)
torch.cuda.synchronize()
print("%d steps at batch %d%s: %d mismatches, %d consumer fallbacks" % (steps, b, " under contention" if contend else "", bad,
                                                                       dbg.pv_debug_enc_late_count() - late0))
