"""Long training run in both precisions from the same seeds: the ELBO trajectory of the bf16 mode against the fp32-class
one (synthetic blob images, batch 256, 3000 steps).  Prints the per-image ELBO at checkpoints and the largest gap."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv
steps = int(os.environ.get("STEPS", 3000))
g = torch.Generator().manual_seed(0)
# blob-like images: a few bright pixels each, so that there is something to learn
n = 4096
x = torch.rand(n, 28, 28, generator=g)
x = ((x > 0.85).float() * torch.rand(n, 28, 28, generator=g)).cuda()
curves = {}
for fused in (2, 3):
    model = pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
    eng = model.engine(fused=fused)
    torch.manual_seed(1)
    hist = torch.zeros(steps, 4, device="cuda")
    for i in range(steps):
        idx = (i * 256) % n
        eps = torch.empty(256, model.z_dim).normal_().cuda()
        eng.loss_and_grads(x[idx:idx + 256], eps, scalars_out=hist[i])
        eng.adam_step()
    h = hist[:, 0].cpu() / 256
    assert torch.isfinite(h).all()
    curves[fused] = h
    print("fused=%d" % fused, " ".join("%d:%.3f" % (k, h[k - 16:k].mean()) for k in (16, 100, 300, 1000, 2000, steps)))
a, b = curves[2], curves[3]
sm = lambda t: t.unfold(0, 64, 64).mean(-1)
gap = ((sm(a) - sm(b)).abs() / sm(a).abs()).max().item()
print("largest relative gap between the 64-step moving averages: %.2e" % gap)
