import sys, os, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, numpy as np
import pyroved_amd as pv
from pyroved_amd import _abi
from oracle import svi_oracle as orc
from conftest import load_golden, make_x, meta_of
torch.set_num_threads(16)
gold = load_golden("ivae_28x28_rt_b256"); meta = meta_of(gold)
x = make_x(meta["xkind"], meta["batch"], meta["data_dim"]); eps = torch.from_numpy(gold["s0.eps"])
cfg = orc.Config(data_dim=meta["data_dim"], latent_dim=2, invariances=meta["invariances"])
lib = C.CDLL(_abi.LIB_PATH)
res = {}
for mode in (0, 1):
    from pyroved_amd.engine import IVAEEngine
    IVAEEngine.dec_kernel = 2 if mode else 1             # pv_ivae_plan.dec_kernel: the 8-wave / the 4-wave plain-bf16 kernel
    model = pv.models.iVAE(meta["data_dim"], 2, meta["invariances"], seed=1, device="cuda")
    eng = model.engine(fused=3)
    if mode == 0:
        o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg); o.step(x, eps)
    eng.loss_and_grads(x.cuda(), eps.cuda())
    res[mode] = (eng.scalars[0].item(), {k: ((eng.grad_of(k).cpu().double() - o.last_grads[k].double()).norm() / o.last_grads[k].double().norm()).item() for k in o.p})
print("oracle loss", o.last["loss"].item())
for mode in (0, 1):
    print("w8" if mode else "4-wave", "loss", res[mode][0], "rel", abs(res[mode][0] - o.last["loss"].item()) / abs(o.last["loss"].item()))
for k in o.p:
    print("%-42s 4-wave %.2e   w8 %.2e" % (k, res[0][1][k], res[1][1][k]))
