"""Captures the data-parallel step (pv_ivae_dp_step: gradient launches -> ncclAllReduce -> Adam, one library call on one stream)
in a hipGraph at world size 1 and replays it: the replayed steps must equal eager steps bit for bit.  (DESIGN.md section 6:
"capturable as a graph".)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyroved_amd as pv
from pyroved_amd import dist as pvdist

torch.cuda.set_device(0)
comm = pvdist.native_comm(torch.device("cuda", 0))
mk = lambda: pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
ma, mb = mk(), mk()
ea, eb = ma.engine(fused=3), mb.engine(fused=3)
g = torch.Generator().manual_seed(0)
x = torch.rand(256, 28, 28, generator=g).cuda()
eps = torch.randn(256, ma.z_dim, generator=g).cuda()
ha, hb = torch.zeros(4, device="cuda"), torch.zeros(4, device="cuda")
# warm both (module load, RCCL's first call) with one eager step each
ea.loss_and_grads(x, eps, step=True, comm=comm, hist_out=ha)
eb.loss_and_grads(x, eps, step=True, comm=comm, hist_out=hb)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(graph, stream=s):
        eb.loss_and_grads(x, eps, step=True, comm=comm, hist_out=hb)    # adam_step = 2 is baked into the captured launch
torch.cuda.synchronize()
eb.adam_t -= 1                      # capture does not execute
graph.replay(); eb.adam_t += 1
ea.loss_and_grads(x, eps, step=True, comm=comm, hist_out=ha)
torch.cuda.synchronize()
print("replay == eager step 2:", torch.equal(ea.flat, eb.flat), torch.equal(ha, hb), ha.tolist())
