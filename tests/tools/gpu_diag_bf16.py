"""Plain-bf16 decoder kernel (fused=3) against the CPU oracle on a golden case: ELBO and per-tensor gradient error.
python tests/tools/gpu_diag_bf16.py NAME"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from test_gpu_parity import load_golden, meta_of, build, make_x, rel_l2, orc
name = sys.argv[1]
gold = load_golden(name); meta = meta_of(gold)
for fused in (3,):
    model, cfg, eng = build(meta, fused)
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"]); xg = x.cuda()
    for k in range(meta["steps"]):
        eps = torch.from_numpy(gold["s%d.eps" % k])
        eng.loss_and_grads(xg, eps.cuda(), meta["beta"])
        loss = float(eng.scalars.cpu()[0])
        ref = float(gold["s%d.loss" % k])
        o.step(x, eps, meta["beta"])
        print("fused %d step %d loss %.4f oracle %.4f rel %.2e" % (fused, k, loss, ref, abs(loss - ref) / abs(ref)))
        worst = ("", 0.0)
        for key in o.p:
            g, go = eng.grad_of(key).cpu(), o.last_grads[key]
            e = rel_l2(g, go)
            if e > worst[1]: worst = (key, e)
            if len(sys.argv) > 2:
                print("    %-26s grad rel_l2 %.2e  max|dg|/max|g| %.2e" % (key, e, (g - go).abs().max() / go.abs().max()))
        print("    worst grad rel_l2 %.2e (%s)" % (worst[1], worst[0]))
        eng.adam_step()
        # keep the engine on the oracle's trajectory
        for key, p in model.state_dict().items():
            p.copy_(o.p[key].detach())
