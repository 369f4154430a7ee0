import time, torch, sys, os
sys.path.insert(0, os.getcwd())
import pyroved_amd as pv
dev = torch.device("cuda")
x = torch.rand(60000, 28, 28, generator=torch.Generator().manual_seed(0))
for B in (32, 100, 256):
    for prec in ("bf16", None):
        m = pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device=dev)
        kw = {"precision": prec} if prec else {}
        tr = pv.trainers.SVItrainer(m, seed=1, **kw)
        loader = pv.utils.init_dataloader(x, batch_size=B)
        tr.step(loader)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2): tr.step(loader)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        n = 2 * len(loader)
        print("trainer B=%d %s: %.1f us/step, %.0f images/s" % (B, prec or "fp32-class", 1e6 * dt / n, 2 * 60000 / dt))
