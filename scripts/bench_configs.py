"""Throughput of the other BASELINE.json configurations the HIP path covers (secondary numbers for DESIGN.md §5;
bench.py stays the contract for the headline config).  One GPU, inputs resident in HBM, eps pre-generated."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv

CASES = [
    ("C1  iVAE 28x28 ['r'] B=128", lambda: pv.models.iVAE((28, 28), 2, ["r"], seed=1, device="cuda"), 128, (28, 28), 1),
    ("C2  iVAE 28x28 ['r','t'] B=256", lambda: pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda"), 256, (28, 28), 1),
    ("C3  jiVAE K=10 28x28 ['r'] B=512", lambda: pv.models.jiVAE((28, 28), 2, 10, ["r"], seed=1, device="cuda"), 512, (28, 28), 10),
    ("C4' iVAE 64x64 ['r','t','s'] fc encoder B=128/GPU", lambda: pv.models.iVAE((64, 64), 2, ["r", "t", "s"], seed=1, device="cuda"), 128, (64, 64), 1),
    ("    iVAE 28x28 ['r','t'] B=1024", lambda: pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda"), 1024, (28, 28), 1),
]
steps, warm = int(os.environ.get("STEPS", 60)), 10
FUSED = int(os.environ.get("FUSED", 2))          # 2: fp32-class (default), 3: bf16 operands
print("decoder path fused=%d" % FUSED)
ONLY = os.environ.get("ONLY")                   # substring filter on the case names
if ONLY is None:
    # one process per case: a case that follows a much larger one in the same process inherits its freed (fragmented)
    # device memory and runs up to 2x slower — an artefact of the sequence, not of the case
    import subprocess
    for tag in ["C1 ", "C2 ", "C3 ", "C4'", "B=1024", "C5", "C4 "]:
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, ONLY=tag))
    sys.exit(0)
for name, make, B, dd, K in CASES:
    if ONLY and ONLY not in name:
        continue
    model = make()
    eng = model.engine(fused=FUSED)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(4, B, *dd, generator=g).cuda()
    eps = torch.randn(steps + warm, B, model.z_dim, generator=g).cuda()
    hist = torch.zeros(steps + warm, 4, device="cuda")
    def step(i):
        eng.loss_and_grads(x[i % 4], eps[i], 1.0, scalars_out=hist[i])
        eng.adam_step()
    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warm + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n_pix = dd[0] * dd[1]
    fl = 3 * n_pix * 66304 * K * B          # algorithmic decoder FLOPs of the step, K decoder passes
    print("%-52s %8.3f ms/step  %9.0f images/s  %6.1f TF algorithmic  loss/img %.3f -> %.3f" % (
        name, dt * 1e3, B / dt, fl / dt / 1e12, hist[warm, 0].item() / B, hist[-1, 0].item() / B), flush=True)
    del eng, model
    torch.cuda.empty_cache()

import warnings; warnings.filterwarnings("ignore")
g = torch.Generator().manual_seed(0)


def timed(eng, B, z_dim, make_args):
    eps = torch.randn(steps + warm, B, z_dim, generator=g).cuda()
    hist = torch.zeros(steps + warm, 4, device="cuda")
    def one(i):
        eng.loss_and_grads(*make_args(i, eps[i]), scalars_out=hist[i]); eng.adam_step()
    for i in range(warm): one(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): one(warm + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, hist


if ONLY == "C5":      # VED im2spec 64x64 -> 128, batch 256 per GPU
    model = pv.models.VED((64, 64), (128,), seed=1, device="cuda")
    eng, B = model.engine(fused=FUSED), 256
    x = torch.rand(2, B, 1, 64, 64, generator=g).cuda(); y = torch.rand(2, B, 1, 128, generator=g).cuda()
    dt, hist = timed(eng, B, 2, lambda i, e: (x[i % 2], e, 1.0, y[i % 2]))
    print("%-52s %8.3f ms/step  %9.0f images/s  %6.1f TF algorithmic (0.709 GFLOP/img)  loss/img %.3f -> %.3f" % (
        "C5  VED 64x64 -> 128 B=256/GPU", dt * 1e3, B / dt, 0.709e9 * B / dt / 1e12, hist[warm, 0].item() / B,
        hist[-1, 0].item() / B))
elif ONLY == "C4 ":   # iVAE 64x64 ['r','t','s'] with the convolutional encoder, batch 128 per GPU
    model = pv.models.iVAE((64, 64), 2, ["r", "t", "s"], seed=1, device="cuda")
    model.set_encoder(pv.nets.convEncoderNet((64, 64), latent_dim=model.z_dim))
    eng, B = model.engine(fused=FUSED), 128
    x = torch.rand(2, B, 64, 64, generator=g).cuda()
    dt, hist = timed(eng, B, model.z_dim, lambda i, e: (x[i % 2], e, 1.0))
    print("%-52s %8.3f ms/step  %9.0f images/s  (1.50 GFLOP/img -> %.1f TF algorithmic)  loss/img %.3f -> %.3f" % (
        "C4  iVAE 64x64 ['r','t','s'] conv encoder B=128/GPU", dt * 1e3, B / dt, 1.50e9 * B / dt / 1e12,
        hist[warm, 0].item() / B, hist[-1, 0].item() / B))
