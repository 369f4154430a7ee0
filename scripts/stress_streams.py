"""Race check of the multi-stream conv steps: the same loss_and_grads call N times at the benchmark sizes — every gradient and the
loss scalars must come out bit-identical every time (a missing cross-stream dependency shows up as a run that differs), and equal
to the one-stream step's (engine.side_stream = False).
    python scripts/stress_streams.py [n_repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pyroved_amd as pv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
bad = 0
for name, fused in (("C5", 2), ("C5", 3), ("C4", 2), ("C4", 3)):
    cfg = dict(bench.CONFIGS[name])
    model = bench.make_model(pv, cfg, dev)
    B = cfg["batch"]
    data = [t.to(dev) for t in bench.make_data(cfg, B, torch.Generator().manual_seed(0))]
    eps = torch.randn(B, model.z_dim, generator=torch.Generator().manual_seed(1)).to(dev)
    ved = cfg["kind"] == "ved"
    ref = None
    for mode in ("one-stream", "multi-stream"):
        eng = model.engine(fused=fused)
        eng.side_stream = mode == "multi-stream"
        outs = []
        for it in range(n if mode == "multi-stream" else 2):
            # a little unrelated work on the stream in front of some calls shifts the relative timing of the streams
            if it % 3 == 1:
                torch.empty(1 << (18 + it % 5), device=dev).normal_()
            if ved:
                eng.loss_and_grads(data[0], eps, 1.0, data[1])
            else:
                eng.loss_and_grads(data[0], eps)
            outs.append((eng.grad[:eng.n_flat].clone(), eng.scalars.clone()))
        torch.cuda.synchronize()
        if ref is None:
            ref = outs[0]
        diff = [i for i, (g, s) in enumerate(outs) if not (torch.equal(g, ref[0]) and torch.equal(s, ref[1]))]
        print("%s fused=%d %s: %d calls, %d differ from the one-stream step%s" % (name, fused, mode, len(outs), len(diff),
              (" (first: call %d, max |dg| %.3e)" % (diff[0], (outs[diff[0]][0] - ref[0]).abs().max().item())) if diff else ""))
        bad += len(diff)
print("stress_streams:", "OK" if bad == 0 else "%d MISMATCHES" % bad)
sys.exit(1 if bad else 0)
