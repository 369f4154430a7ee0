"""Worker of test_trainer_data_parallel_two_ranks_one_gpu: trains a small iVAE with SVItrainer on cuda:0 as one rank of a
gloo process group (two ranks share this box's single GPU) and prints the loss history as JSON (rank 0)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as td
import pyroved_amd as pv

td.init_process_group("gloo")
torch.cuda.set_device(0)
g = torch.Generator().manual_seed(0)
x = torch.rand(37, 16, 16, generator=g)
kind = sys.argv[1]
if kind == "ivae":
    model = pv.models.iVAE((16, 16), 2, ["r", "t"], seed=1, device="cuda:0")
    tr = pv.trainers.SVItrainer(model, seed=1)
    for _ in range(2):
        tr.step(pv.utils.init_dataloader(x, batch_size=8), pv.utils.init_dataloader(x[:10], batch_size=5))
    hist = dict(train=tr.loss_history["training_loss"], test=tr.loss_history["test_loss"])
else:
    model = pv.models.ssiVAE((16, 16), 2, 3, ["r"], seed=1, device="cuda:0")
    tr = pv.trainers.auxSVItrainer(model, seed=1)
    xf = x.reshape(37, -1)
    ys = pv.utils.to_onehot(torch.arange(12) % 3, 3)
    lu, ls, lv = pv.utils.init_ssvae_dataloaders(xf, (xf[:12], ys), (xf[:12], ys), batch_size=6)
    for _ in range(2):
        tr.step(lu, ls, lv)
    hist = dict(train=tr.history["training_loss"], test=[float(v) for v in tr.history["test"]])
w = model.state_dict()
hist["wsum"] = float(sum(v.double().sum() for k, v in w.items() if v.dtype.is_floating_point))
if td.get_rank() == 0:
    print("RESULT " + json.dumps(hist))
td.destroy_process_group()
