"""Phase timing of the 8-wave plain-bf16 decoder kernel (a -DW8_TRACE build: scripts/mkvariant_file.sh trace pv_sdec_fused_w8.hip -DW8_TRACE;
PV_LIB_PATH=pyroved_amd/variants/lib_trace.so python scripts/gpu_trace_w8.py): cycles per phase, workgroup 0, waves 0 / 7."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv
from pyroved_amd import _abi
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model = pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
eng = model.engine(fused=3)
eng.enc_fold = os.environ.get('PV_TRACE_FOLD', '1') != '0'
x = torch.rand(B, 28, 28, generator=torch.Generator().manual_seed(0)).cuda()
eps = torch.randn(B, model.z_dim).cuda()
for _ in range(3):
    eng.loss_and_grads(x, eps)
torch.cuda.synchronize()
lib = C.CDLL(_abi.LIB_PATH)
buf = (C.c_longlong * 512)()
print("rc", lib.pv_debug_read_trace_w8(buf, 512))
# stamps in program order (index 12 — the row-local coordinate backward — sits between barrier 2 and the round-2 staging)
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 10, 11, 13]
names = {1: "coord layer", 2: "fwd L1+tanh", 3: "fwd L2+logit+lik", 4: "dwo colsum+dpre2", 5: "stage1+bar1", 6: "consume wgrad2",
         7: "dgrad2+dtanh", 8: "dgrad1+dtanh", 9: "bar2", 12: "rowlocal+colsum dpre0(+flush)", 10: "stage2+bar3", 11: "consume wgrad1",
         13: "bar4"}
for w, base in ((0, 0), (7, 256)):
    for t in range(8):
        st = [buf[base + t * 16 + k] for k in range(15)]
        if not st[0]:
            continue
        out, prev = [], st[0]
        for k in order[1:]:
            if st[k]:
                out.append("%s=%d" % (names[k], st[k] - prev)); prev = st[k]
        print("wave", w, "tile", t, "total", prev - st[0], " ".join(out))
    nxt = [buf[base + t * 16] for t in range(8) if buf[base + t * 16]]
    print("wave", w, "tile starts (delta):", [b - a for a, b in zip(nxt, nxt[1:])])
if buf[128 + 4]:
    kk = [buf[128 + i] for i in (0, 4, 5, 6, 7, 1)]
    print("folded guide (workgroup 0, wave 0): entry -> layer 0 %d, layer 1 %d, head %d, sample + split + fc_latent %d, images + barrier %d cycles"
          % tuple(b - a for a, b in zip(kk, kk[1:])))
k = [buf[128 + i] for i in range(4)]
if k[0]:
    print("launch (workgroup 0, wave 0): prologue %d, tile loop %d, epilogue (flush, record, column sums) %d cycles" % (k[1] - k[0], k[2] - k[1], k[3] - k[2]))
    rt = [buf[136 + i] for i in range(4)]
    us = (rt[3] - rt[0]) / 100.0
    print("  entry -> record written: %d cycles in %.2f us of the constant 100 MHz counter = %.3f GHz" % (k[3] - k[0], us, (k[3] - k[0]) / us / 1e3))
e = [buf[144 + i] for i in range(10)]
if e[0] and k[2]:
    nm = ["records stored", "barrier", "prefetch + scr stores", "barrier", "column sums, row/dz partials", "barrier", "head bwd + edp1 + barrier", "edp0 partials + barrier",
          "-", "-"]
    prev = k[2]
    out = []
    for i in range(10):
        if e[i]:
            out.append("%s=%d" % (nm[i], e[i] - prev)); prev = e[i]
    out.append("end=%d" % (k[3] - prev))
    print("epilogue phases (cycles):", " ".join(out))
x = [buf[144 + i] for i in (1, 10, 11, 12, 2)]
if all(x):
    print("  inside 'prefetch + scr stores': row-sum loads issued %d, other loads issued %d, record stores issued %d, scr stores + wave sum %d" % tuple(b - a for a, b in zip(x, x[1:])))
c = [buf[128]] + [buf[144 + i] for i in range(20, 27)] + [buf[128 + 4]]
if all(c[1:8]):
    print("  coop layer 0 (cycles): entry -> rows + loads requested %d, staged %d, multiply-adds %d, transpose + stores issued %d, stores acknowledged %d, barrier + tag %d, producers' tags seen %d, h1 read %d"
          % tuple(b - a for a, b in zip(c, c[1:])))
