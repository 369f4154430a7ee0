"""Phase timing of the 4-wave fp32-class kernel's column-parallel tail (FB_TRACE build, PV_FD_ABLATE=256): cycles of workgroup 0 /
thread 0 between the tail's stamps, next to the kernel-level stamps (prologue / tile loop + tail / record)."""
import ctypes as C, os, sys
os.environ["PV_FD_ABLATE"] = os.environ.get("PV_TRACE_MASK", "256")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv
from pyroved_amd import _abi
model = pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
eng = model.engine(fused=2)
x = torch.rand(256, 28, 28, generator=torch.Generator().manual_seed(0)).cuda()
eps = torch.randn(256, model.z_dim).cuda()
for _ in range(3):
    eng.loss_and_grads(x, eps)
torch.cuda.synchronize()
lib = C.CDLL(_abi.LIB_PATH)
kb = (C.c_longlong * 256)()
lib.pv_debug_read_trace(kb, 256)
names = ["tail start (inputs landed, coords)", "coord layer own + put h0 (+ W2l reload issue)", "barrier 0", "get h0, fwd L1 own, tanh, put h1",
         "wait W2l", "barrier 1", "get h1, fwd L2 own, tanh, logit partial", "barrier 2", "lik, d(wo) sums, dpre2, split, put", "barrier 3",
         "get dpre2, stage (wave 0)", "barrier 4", "consume16 (W2)", "dgrad L2 own, split, put", "barrier 5", "get dpre1, dgrad L1 own",
         "stage, column sums, row partials", "barrier 6", "consume16 (W1)"]
st = [kb[160 + k] for k in range(19)]
if st[0]:
    for k in range(1, 19):
        print("%7d  %s" % (st[k] - st[k - 1], names[k]))
    print("tail total (to the last consume): %d cycles" % (st[18] - st[0]))
if kb[200]:
    print("kernel-level (workgroup 0, wave 0): prologue %d cycles, tiles + tail %d, epilogue (record write) %d; whole %d"
          % (kb[201] - kb[200], kb[202] - kb[201], kb[203] - kb[202], kb[203] - kb[200]))
for t in range(2):
    a, b = kb[t * 32 + 0], kb[t * 32 + 13]
    if a and b:
        print("row-parallel tile %d: %d cycles" % (t, b - a))
