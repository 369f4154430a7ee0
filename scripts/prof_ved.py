import os, sys, time, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv
which = sys.argv[1] if len(sys.argv) > 1 else "ved"
g = torch.Generator().manual_seed(0)
if which == "ved":
    model = pv.models.VED((64, 64), (128,), seed=1, device="cuda"); B = 256
    x = torch.rand(B, 1, 64, 64, generator=g).cuda(); y = torch.rand(B, 1, 128, generator=g).cuda()
    eng = model.engine(fused=int(os.environ.get("FUSED", 2))); eps = torch.randn(B, 2, generator=g).cuda()
    step = lambda: (eng.loss_and_grads(x, eps, 1.0, y), eng.adam_step())
else:
    model = pv.models.iVAE((64, 64), 2, ["r", "t", "s"], seed=1, device="cuda"); B = 128
    model.set_encoder(pv.nets.convEncoderNet((64, 64), latent_dim=model.z_dim))
    x = torch.rand(B, 64, 64, generator=g).cuda(); eng = model.engine(fused=2); eps = torch.randn(B, model.z_dim, generator=g).cuda()
    step = lambda: (eng.loss_and_grads(x, eps), eng.adam_step())
for _ in range(8): step()
torch.cuda.synchronize()
