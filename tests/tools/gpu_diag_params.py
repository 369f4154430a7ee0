"""Per-key gradient / parameter-after-Adam error vs the CPU oracle for one golden case: python tests/tools/gpu_diag_params.py NAME [fused]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from test_gpu_parity import load_golden, meta_of, build, make_x, rel_l2, orc
name = sys.argv[1]; fused = int(sys.argv[2]) if len(sys.argv) > 2 else 2
gold = load_golden(name); meta = meta_of(gold)
model, cfg, eng = build(meta, fused)
o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
x = make_x(meta["xkind"], meta["batch"], meta["data_dim"]); xg = x.cuda()
for k in range(meta["steps"]):
    eps = torch.from_numpy(gold["s%d.eps" % k])
    eng.loss_and_grads(xg, eps.cuda(), meta["beta"])
    o.step(x, eps, meta["beta"])
    grads = {key: eng.grad_of(key).cpu().clone() for key in o.p}
    eng.adam_step()
    for key, p in model.state_dict().items():
        g, go = grads[key], o.last_grads[key]
        pe = rel_l2(p, o.p[key].detach())
        d = (p.cpu() - o.p[key].detach()).abs()
        print("step %d %-28s grad rel %.2e  |g|max %.2e  max|dg| %.2e  n(|g|<1e-6) %d/%d   param rel %.2e  max|dp| %.2e" % (
            k, key, rel_l2(g, go), go.abs().max(), (g - go).abs().max(), int((go.abs() < 1e-6).sum()), go.numel(), pe, d.max()))
