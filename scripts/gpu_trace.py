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
names = ["start", "coord+split", "fwd L1+tanh+split", "W2 wait+bar", "fwd L2..lik+dpre2", "stage L2+bar", "wgrad L2+bar",
         "dgrad L2+split", "W1 wait+bar", "dgrad L1+split", "stage L1+bar", "wgrad L1+rowlocal+bar", "split+stage dpre0+bar",
         "coord sums (MFMA)+bar"]
buf = (C.c_longlong * 128)()
lib.pv_debug_read_trace(buf, 128)
sub = {16: "fwdL2 mfma", 17: "tanh8", 18: "logit+lik", 19: "dwo", 20: "dpre2", 21: "wgrad L1 consume", 22: "rowlocal"}
for t in range(4):
    st = [buf[t * 32 + k] for k in range(14)]
    fine = {k: buf[t * 32 + k] for k in sub}
    if st[0]:
        seq = [3, 16, 17, 18, 19, 20, 4]
        print("  tile", t, "fwdL2..dpre2 detail:", " ".join("%s=%d" % (sub.get(b, "presplit"), (fine.get(b) or st[b]) - (fine.get(a) or st[a])) for a, b in zip(seq, seq[1:])),
              "| tail: consume=%d rowlocal=%d bar=%d" % (fine[21] - st[10], fine[22] - fine[21], st[11] - fine[22]))
    if st[0] == 0: continue
    prev = st[0]; out = []
    for k in range(1, 14):
        if st[k]:
            out.append("%s=%d" % (names[k], st[k] - prev)); prev = st[k]
    print("tile", t, "total", prev - st[0], " ".join(out))
# raw view: every stamp of tile 1 sorted by time (the phase names above assume the x3 kernel's barrier structure)
labels = {0: "tile start", 1: "coord layer done", 2: "fwd L1+tanh+cvt done", 3: "(x3 barrier) before fwd L2", 16: "fwd L2 mfma done",
          17: "tanh8 done", 18: "logit+lik done", 19: "d(wo) done", 20: "dpre2 done", 4: "cvt dpre2 done", 5: "stage L2 + barrier done",
          6: "(x3) consume2 barrier", 7: "dgrad L2 + cvt done", 8: "(x3 barrier) before dgrad L1", 9: "dgrad L1 + coord sums + cvt done",
          10: "stage L1 + barrier done", 21: "wgrad L1 consume done", 22: "row-local done", 13: "tile end"}
for t in (1, 2):
    st = sorted((buf[t * 32 + k], k) for k in labels if buf[t * 32 + k])
    if not st: continue
    print("tile", t, "raw:")
    for (a, ka), (b, kb) in zip(st, st[1:]):
        print("   %6d  -> %s" % (b - a, labels[kb]))

kb = (C.c_longlong * 256)()
lib.pv_debug_read_trace(kb, 256)
if kb[200]:
    print("kernel-level (workgroup 0, wave 0): prologue %d cycles, tiles %d, epilogue (record write) %d; whole %d"
          % (kb[201] - kb[200], kb[202] - kb[201], kb[203] - kb[202], kb[203] - kb[200]))
