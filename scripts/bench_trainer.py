"""End-to-end SVItrainer.step(train_loader) throughput through the reference API (CPU DataLoader handed over):
device-resident feed vs per-batch host iteration.  Secondary number for DESIGN.md §5."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv
n, B = int(os.environ.get("N", 61440)), 256
x = torch.rand(n, 28, 28, generator=torch.Generator().manual_seed(0))
for feed, precision in ((True, "fp32"), (True, "bf16"), (False, "fp32")):
    model = pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
    loader = pv.utils.init_dataloader(x, batch_size=B)
    tr = pv.trainers.SVItrainer(model, seed=1, device_feed=feed, precision=precision)
    tr.step(loader)                       # warm-up epoch (also uploads the dataset)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    epochs = 3
    for _ in range(epochs):
        tr.step(loader)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("device_feed=%-5s precision=%s %d images x %d epochs: %.3f s  -> %.0f images/s (%.3f ms per step of %d)  loss %s" % (
        feed, precision, n, epochs, dt, n * epochs / dt, 1e3 * dt / (epochs * (n // B)), B,
        ["%.4f" % v for v in tr.loss_history["training_loss"]]), flush=True)
