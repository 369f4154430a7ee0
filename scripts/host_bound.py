"""Is a config's step enqueue-bound?  Times the HOST side of K back-to-back steps (no synchronisation) next to the whole
region (enqueue + drain), on the GPU box:  python scripts/host_bound.py C5 [fused] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pyroved_amd as pv

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
fused = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = int(sys.argv[3]) if len(sys.argv) > 3 else 200
cfg = dict(bench.CONFIGS[name]); dev = torch.device("cuda:0")
model = bench.make_model(pv, cfg, dev); eng = model.engine(fused=fused)
B, ring = cfg["batch"], cfg["ring"]
data = [t.view(ring, B, *t.shape[1:]).to(dev) for t in bench.make_data(cfg, ring * B, torch.Generator().manual_seed(0))]
eps = torch.randn(8, B, model.z_dim, device=dev); hist = torch.zeros(8, 4, device=dev)
ved = cfg["kind"] == "ved"
eng._CONV_W_EVERY = 1 << 40      # (the weight-range check reads a scalar back every 64th call: a synchronisation, not enqueue cost)
eng._check_conv_weight_range(force=True) if hasattr(eng, "_check_conv_weight_range") else None
def step(i):
    if ved:
        eng.loss_and_grads(data[0][i % ring], eps[i % 8], 1.0, data[1][i % ring], scalars_out=hist[i % 8]); eng.adam_step()
    elif getattr(eng, "supports_step", False):
        eng.loss_and_grads(data[0][i % ring], eps[i % 8], scalars_out=hist[i % 8], step=True)
    else:
        eng.loss_and_grads(data[0][i % ring], eps[i % 8], scalars_out=hist[i % 8]); eng.adam_step()
for i in range(20): step(i)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(K): step(i)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%s fused=%d: host enqueue %.4f ms/step, whole region %.4f ms/step (drain after the last enqueue %.3f ms)"
          % (name, fused, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3, (t2 - t1) * 1e3))
