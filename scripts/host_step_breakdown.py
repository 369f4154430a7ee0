"""Where does the host time of one C2 step go?  perf_counter_ns around the pieces of IVAEEngine.loss_and_grads(step=True)
(steady state: the queue is drained every 64 steps so that enqueueing never blocks).   python scripts/host_step_breakdown.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pyroved_amd as pv
from pyroved_amd import _abi, engine as E

cfg = dict(bench.CONFIGS["C2"]); dev = torch.device("cuda:0")
model = bench.make_model(pv, cfg, dev); eng = model.engine(fused=3)
B = cfg["batch"]
x = bench.make_data(cfg, B, torch.Generator().manual_seed(0))[0].to(dev)
eps = torch.randn(B, model.z_dim, device=dev); hist = torch.zeros(4, device=dev)
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter_ns(); r = fn(*a, **k); acc[name] = acc.get(name, 0) + time.perf_counter_ns() - t; return r
    return w
eng.ensure_bound = timed("ensure_bound", eng.ensure_bound)
eng._plan = timed("_plan", eng._plan)
eng._prep = timed("_prep (x4)", eng._prep)
eng._count_bn = timed("_count_bn", eng._count_bn)
lib = _abi.lib()
real_step = lib.pv_ivae_step
class L:                                             # a proxy that times the one library call of the step
    def __getattr__(self, k): return getattr(lib, k)
    def pv_ivae_step(self, *a):
        t = time.perf_counter_ns(); r = real_step(*a); acc["pv_ivae_step (library: 3 launches)"] = acc.get("pv_ivae_step (library: 3 launches)", 0) + time.perf_counter_ns() - t; return r
proxy = L()
_abi.lib = lambda: proxy
for i in range(50): eng.loss_and_grads(x, eps, scalars_out=hist, step=True)
torch.cuda.synchronize(); acc.clear()
N = 2000; tot = 0
for i in range(N):
    if i % 64 == 0: torch.cuda.synchronize()
    t = time.perf_counter_ns(); eng.loss_and_grads(x, eps, scalars_out=hist, step=True); tot += time.perf_counter_ns() - t
torch.cuda.synchronize()
print("host time per step: %.2f us" % (tot / N / 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print("   %-40s %6.2f us" % (k, v / N / 1e3))
print("   %-40s %6.2f us" % ("everything else in loss_and_grads", (tot - sum(acc.values())) / N / 1e3))
