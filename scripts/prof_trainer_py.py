"""cProfile of the host side of SVItrainer.step on the headline config (where does a 100 us step spend its Python time?)"""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv
x = torch.rand(20480, 28, 28)
loader = pv.utils.init_dataloader(x, batch_size=256)
model = pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
tr = pv.trainers.SVItrainer(model, seed=1, precision=os.environ.get("PREC", "bf16"))
tr.step(loader); torch.cuda.synchronize()
t0 = time.perf_counter(); tr.step(loader); tr.step(loader); torch.cuda.synchronize(); t1 = time.perf_counter()
print("us per step (wall, 2 epochs of 80 steps): %.1f" % ((t1 - t0) / 160 * 1e6))
pr = cProfile.Profile(); pr.enable(); tr.step(loader); tr.step(loader); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
