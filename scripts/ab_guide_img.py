"""The guide as one workgroup per image (pv_guide_img.hip, default) against the tiled one-launch encoder (PV_PLAN_ENC_TILED):
ms per SVI step (loss_and_grads + Adam in one call) for iVAE 28x28 ['r','t'] at several batches, both decoder precisions; the
fold (fused=3 at batch 256) is switched off for the comparison so that both arms launch a guide."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyroved_amd as pv

def run(batch, fused, per_image, steps=300, inv=("r", "t"), dims=(28, 28)):
    m = pv.models.iVAE(dims, 2, list(inv), seed=1, device="cuda")
    eng = m.engine(fused=fused)
    eng.enc_per_image = per_image
    eng.enc_fold = False
    g = torch.Generator().manual_seed(0)
    x = torch.rand(batch, *dims, generator=g).cuda()
    eps = torch.randn(batch, m.z_dim, generator=g).cuda()
    for _ in range(30):
        eng.loss_and_grads(x, eps, step=True)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.loss_and_grads(x, eps, step=True)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    return 1e3 * best, eng.scalars[0].item()

for fused in (2, 3):
    for batch in (64, 128, 256, 512):
        a, la = run(batch, fused, True)
        b, lb = run(batch, fused, False)
        print("fused=%d batch %4d   per-image %.4f ms   tiled %.4f ms   (%+.1f %%)   loss %.3f / %.3f" % (fused, batch, a, b, 100 * (a / b - 1), la, lb))
