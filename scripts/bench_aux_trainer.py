"""auxSVItrainer.step throughput through the reference API (ssiVAE, 28x28 ['r','t'], 3 classes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv
g = torch.Generator().manual_seed(0)
n, B = 16384, 256
x = torch.rand(n, 784, generator=g)
labels = pv.utils.to_onehot(torch.randint(0, 3, (2048,), generator=g), 3)
lu, ls, lv = pv.utils.init_ssvae_dataloaders(x, (x[:2048], labels), (x[:512], labels[:512]), batch_size=B)
m = pv.models.ssiVAE((28, 28), 2, 3, ["r", "t"], seed=1, device="cuda")
tr = pv.trainers.auxSVItrainer(m, seed=1)
tr.step(lu, ls)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    tr.step(lu, ls)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("auxSVItrainer: %d unlabeled images x 3 epochs in %.3f s -> %.0f images/s (%.3f ms per unlabeled batch of %d); loss %s"
      % (n, dt, 3 * n / dt, 1e3 * dt / (3 * len(lu)), B, ["%.3f" % v for v in tr.history["training_loss"]]))
