import matplotlib
matplotlib.use('Agg')
"""Calls every public inference / utility method of every model family once on the GPU (shapes, finiteness, odd batch sizes):
a crash detector for the host-side mirror of the reference API, not a numerics test."""
import os, sys, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv

def ok(name, t):
    ts = t if isinstance(t, (tuple, list)) else (t,)
    for v in ts:
        if torch.is_tensor(v) and v.dtype.is_floating_point:
            assert torch.isfinite(v).all(), name
    print("ok  %-42s %s" % (name, [tuple(v.shape) if torch.is_tensor(v) else v for v in ts]))

g = torch.Generator().manual_seed(0)
x = torch.rand(23, 16, 16, generator=g)
# ---- iVAE
for inv in (None, ["r"], ["r", "t", "s"]):
    m = pv.models.iVAE((16, 16), 2, inv, seed=1, device="cuda")
    tr = pv.trainers.SVItrainer(m, seed=1)
    ld = pv.utils.init_dataloader(x, batch_size=7); lt = pv.utils.init_dataloader(x[:9], batch_size=4)
    tr.step(ld, lt, scale_factor=2.0); tr.step(ld); tr.print_statistics()
    ok("iVAE%s.encode bs=5" % inv, m.encode(x, batch_size=5))
    z = m.encode(x)[0][:, -2:]
    ok("iVAE.decode", m.decode(z, batch_size=6))
    if inv:
        ok("iVAE.decode angle/shift/scale", m.decode(z[:3], angle=0.4, shift=[0.1, -0.1], scale=1.3))
    ok("iVAE.manifold2d", m.manifold2d(4))
    ok("iVAE.elbo_terms", tuple(torch.tensor(v) for v in m.elbo_terms(x[:5].cuda()).values()))
    m.save_weights("/tmp/_w"); m.load_weights("/tmp/_w.pt")
# class-conditioned
m = pv.models.iVAE((16, 16), 2, ["r"], c_dim=3, seed=1, device="cuda")
y = pv.utils.to_onehot(torch.arange(23) % 3, 3)
tr = pv.trainers.SVItrainer(m, seed=1); tr.step(pv.utils.init_dataloader(x, y, batch_size=8))
ok("iVAE(c_dim).encode", m.encode(x, y)); ok("iVAE(c_dim).decode", m.decode(torch.randn(4, 2), y[:4]))
ok("iVAE(c_dim).manifold2d", m.manifold2d(3, y[:1]))
# 1-D
x1 = torch.rand(11, 32, generator=g)
m = pv.models.iVAE((32,), 2, ["t"], seed=1, device="cuda")
pv.trainers.SVItrainer(m, seed=1).step(pv.utils.init_dataloader(x1, batch_size=4))
ok("iVAE 1-D encode/decode", m.decode(m.encode(x1)[0][:, -2:], shift=0.2))
# ---- jiVAE
m = pv.models.jiVAE((16, 16), 2, 3, ["r", "t"], seed=1, device="cuda")
tr = pv.trainers.SVItrainer(m, enumerate_parallel=True, seed=1)
tr.step(pv.utils.init_dataloader(x, batch_size=6), scale_factor=[1.0, 2.0])
ok("jiVAE.encode", m.encode(x, batch_size=10)); ok("jiVAE.encode logits", m.encode(x[:3], logits=True))
ok("jiVAE.decode", m.decode(torch.randn(5, 2), pv.utils.to_onehot(torch.arange(5) % 3, 3)))
ok("jiVAE.manifold2d", m.manifold2d(3, disc_idx=1))
if hasattr(m, "manifold_traversal"):
    ok("jiVAE.manifold_traversal", m.manifold_traversal(3, 0))
# ---- VED
xv, yv = torch.rand(9, 1, 16, 16, generator=g), torch.rand(9, 1, 32, generator=g)
m = pv.models.VED((16, 16), (32,), hidden_dim_e=[(8,), (16, 16)], hidden_dim_d=[(16, 16), (8,)], batchnorm=True, seed=1, device="cuda")
tr = pv.trainers.SVItrainer(m, seed=1); tr.step(pv.utils.init_dataloader(xv, yv, batch_size=4), pv.utils.init_dataloader(xv, yv, batch_size=4))
ok("VED.encode", m.encode(xv, batch_size=4)); ok("VED.decode", m.decode(torch.randn(5, 2)))
ok("VED.predict", m.predict(xv[:3])); ok("VED.manifold2d", m.manifold2d(3))
tr.step(pv.utils.init_dataloader(xv, yv, batch_size=4))            # (eval()-mode training after encode, as in the reference)
# ---- semi-supervised
xf = x.reshape(23, -1)
ys = pv.utils.to_onehot(torch.arange(10) % 3, 3)
m = pv.models.ssiVAE((16, 16), 2, 3, ["r"], seed=1, device="cuda")
tr = pv.trainers.auxSVItrainer(m, seed=1)
lu, ls, lv = pv.utils.init_ssvae_dataloaders(xf, (xf[:10], ys), (xf[:10], ys), batch_size=5)
tr.step(lu, ls, lv, aux_loss_multiplier=30); tr.print_statistics(); tr.save_running_weights("encoder_y"); tr.average_weights("encoder_y")
ok("ssiVAE.classifier", m.classifier(xf, batch_size=6)); ok("ssiVAE.encode", m.encode(xf[:7])); ok("ssiVAE.encode(y idx)", m.encode(xf[:4], torch.tensor([0, 1, 2, 0])))
ok("ssiVAE.decode", m.decode(torch.randn(4, 2), ys[:4])); ok("ssiVAE.manifold2d", m.manifold2d(3, label=2)); ok("ssiVAE.manifold_traversal", m.manifold_traversal(3, 1))
yr = torch.randn(10, 2, generator=g)
m = pv.models.ss_reg_iVAE((16, 16), 2, 2, ["t", "s"], seed=1, device="cuda")
tr = pv.trainers.auxSVItrainer(m, task="regression", seed=1)
lu, ls, lv = pv.utils.init_ssvae_dataloaders(xf, (xf[:10], yr), (xf[:10], yr), batch_size=5)
tr.step(lu, ls, lv); tr.print_statistics()
ok("ss_reg.regressor", m.regressor(xf)); ok("ss_reg.encode", m.encode(xf[:6])); ok("ss_reg.decode", m.decode(torch.randn(3, 2), yr[:3]))
ok("ss_reg.manifold2d", m.manifold2d(3, yr[:1]))      # (y: (1, reg_dim), as the reference expects)
# ---- utils / nets
ok("transform_coordinates", pv.utils.transform_coordinates(pv.utils.generate_grid((8, 8)).expand(3, 64, 2).cuda(), torch.rand(3).cuda(), torch.rand(3, 1, 2).cuda(), torch.ones(3).cuda()))
enc = pv.nets.fcEncoderNet((16, 16), 4).cuda(); ok("fcEncoderNet.forward", enc(x[:3].cuda()))
print("all API calls ran")
