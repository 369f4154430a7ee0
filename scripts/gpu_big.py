"""Large-shape sanity runs (no oracle: finite losses, decreasing over a few steps, wall time): index-width / workspace checks."""
import os, sys, time, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv

def run(name, model, B, dd, fused, steps=4, y=None):
    eng = model.engine(fused=fused)
    x = torch.rand(B, *dd, device="cuda")
    hist = torch.zeros(steps, 4, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        eps = torch.randn(B, model.z_dim, device="cuda")
        eng.loss_and_grads(x, eps, 1.0, *(() if y is None else (y,)), scalars_out=hist[i]) if getattr(eng, "supports_scalars_out", False) else eng.loss_and_grads(x, eps, 1.0)
        eng.adam_step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    h = (hist[:, 0] / B).tolist()
    ok = all(v == v and abs(v) < 1e9 for v in h)
    print("%-46s B=%-6d fused=%d  %8.2f ms/step  loss/img %s  ws %.2f GB  %s" % (
        name, B, fused, dt * 1e3, ["%.2f" % v for v in h], eng.ws.numel() / 2**30, "ok" if ok else "NON-FINITE"), flush=True)
    del eng, model, x
    torch.cuda.empty_cache()

run("iVAE 28x28 rt", pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda"), 32768, (28, 28), 3)
run("iVAE 28x28 rt", pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda"), 32768, (28, 28), 2)
run("iVAE 128x128 rts", pv.models.iVAE((128, 128), 2, ["r", "t", "s"], seed=1, device="cuda"), 256, (128, 128), 3)
run("iVAE 128x128 rts layered (hid 64)", pv.models.iVAE((128, 128), 2, ["r", "t", "s"], hidden_dim_d=[64, 64], seed=1, device="cuda"), 64, (128, 128), 0)
run("jiVAE K=10 28x28 r", pv.models.jiVAE((28, 28), 2, 10, ["r"], seed=1, device="cuda"), 4096, (28, 28), 3)
run("iVAE 1-D 4096 t", pv.models.iVAE((4096,), 2, ["t"], seed=1, device="cuda"), 1024, (4096,), 3)
run("iVAE 28x28 none (vanilla)", pv.models.iVAE((28, 28), 2, None, seed=1, device="cuda"), 65536, (28, 28), 2)
