"""Phase timing of the latent-backward workgroup of sample 0 (a -DLB_TRACE build: scripts/mkvariant_file.sh lbtrace
pv_elementwise.hip -DLB_TRACE; PV_LIB_PATH=pyroved_amd/variants/lib_lbtrace.so python scripts/gpu_trace_lb.py)."""
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
    eng.loss_and_grads(x, eps)
torch.cuda.synchronize()
lib = C.CDLL(_abi.LIB_PATH)
buf = (C.c_longlong * 32)()
print("rc", lib.pv_debug_read_trace_lb(buf, 32))
names = ["start (after prefetch issue)", "row sums", "part_hz sums", "dzc = dhz Wz", "head bwd", "chain: head phase", "chain: layers"]
prev = buf[0]
for k in range(1, 7):
    print("%-28s %7d cycles" % (names[k], buf[k] - prev)); prev = buf[k]
print("total", buf[6] - buf[0], " | row loads landed + summed per thread at", buf[7] - buf[0])
if buf[8]:
    print("   first phase: row loads issued at", buf[8] - buf[0], "| the chain's operand requests issued at", buf[9] - buf[0])
