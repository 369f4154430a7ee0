"""Stability soak on the GPU: every model family trains for a few hundred SVI steps on synthetic blob data through the
reference API (both precision modes where they exist); losses must stay finite and end below where they started.
    python scripts/soak_families.py [epochs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
g = torch.Generator().manual_seed(0)


def blobs(n, h, w):
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing="ij")
    c = 0.5 * (torch.rand(n, 2, generator=g) - 0.5)
    s = 0.15 + 0.2 * torch.rand(n, 1, 1, generator=g)
    a = torch.rand(n, generator=g) * 3.14159
    u = (xx[None] - c[:, 0, None, None]) * torch.cos(a)[:, None, None] + (yy[None] - c[:, 1, None, None]) * torch.sin(a)[:, None, None]
    v = -(xx[None] - c[:, 0, None, None]) * torch.sin(a)[:, None, None] + (yy[None] - c[:, 1, None, None]) * torch.cos(a)[:, None, None]
    return torch.exp(-(u ** 2 / (2 * s ** 2) + v ** 2 / (2 * (0.5 * s) ** 2))).clamp(0, 1)


def report(name, hist, t0):
    ok = all(v == v and abs(v) < 1e9 for v in hist) and hist[-1] < hist[0]
    print("%-44s %s  loss %.3f -> %.3f  (%.1f s)" % (name, "ok  " if ok else "FAIL", hist[0], hist[-1], time.time() - t0), flush=True)
    return ok


bad = 0
x = blobs(4096, 28, 28)
for precision in ("fp32", "bf16"):
    for inv in (["r", "t"], ["r", "t", "s"], None):
        t0 = time.time()
        m = pv.models.iVAE((28, 28), 2, inv, seed=1, device="cuda")
        tr = pv.trainers.SVItrainer(m, seed=1, precision=precision)
        ld = pv.utils.init_dataloader(x, batch_size=128)
        for _ in range(epochs):
            tr.step(ld, scale_factor=1.0)
        bad += not report("iVAE %s inv=%s" % (precision, inv), tr.loss_history["training_loss"], t0)
    t0 = time.time()
    m = pv.models.jiVAE((28, 28), 2, 4, ["r", "t"], seed=1, device="cuda")
    tr = pv.trainers.SVItrainer(m, seed=1, enumerate_parallel=True, precision=precision)
    ld = pv.utils.init_dataloader(x[:2048], batch_size=128)
    for _ in range(epochs):
        tr.step(ld, scale_factor=[1.0, 3.0])
    bad += not report("jiVAE K=4 %s" % precision, tr.loss_history["training_loss"], t0)
    for bn in (False, True):
        t0 = time.time()
        m = pv.models.VED((28, 28), (28, 28), hidden_dim_e=[(16,), (32, 32)], hidden_dim_d=[(32, 32), (16,)], batchnorm=bn,
                          seed=1, device="cuda")
        tr = pv.trainers.SVItrainer(m, seed=1, precision=precision)
        ld = pv.utils.init_dataloader(x[:2048, None], x[:2048, None].flip(-1), batch_size=64)
        for _ in range(epochs):
            tr.step(ld)
        bad += not report("VED bn=%s %s" % (bn, precision), tr.loss_history["training_loss"], t0)
# semi-supervised (fp32-class)
labels = (torch.rand(4096, generator=g) * 3).long()
xs = (x * (0.5 + 0.25 * labels[:, None, None].float())).clamp(0, 1)
for task in ("classification", "regression"):
    t0 = time.time()
    if task == "classification":
        m = pv.models.ssiVAE((28, 28), 2, 3, ["r", "t"], seed=1, device="cuda")
        ys = pv.utils.to_onehot(labels, 3)
    else:
        m = pv.models.ss_reg_iVAE((28, 28), 2, 1, ["r", "t"], seed=1, device="cuda")
        ys = labels[:, None].float() / 2
    lu, ls, lv = pv.utils.init_ssvae_dataloaders(xs[:3072], (xs[3072:3584], ys[3072:3584]), (xs[3584:], ys[3584:]), batch_size=128)
    tr = pv.trainers.auxSVItrainer(m, task=task, seed=1)
    for _ in range(epochs):
        tr.step(lu, ls, lv, aux_loss_multiplier=20)
    bad += not report("ss %s (validation %s %.3f -> %.3f)" % (task, "accuracy" if task == "classification" else "mse",
                                                             tr.history["test"][0], tr.history["test"][-1]),
                      tr.history["training_loss"], t0)
print("failures:", bad)
sys.exit(1 if bad else 0)
