"""Error table of the fp32-class decoder kernel's precision modes (round 4).

For each fixture and each build of pv_sdec_fused_bf16_kernel — bf16 {3,3,3} (the round 1-3 kernel), fp16 {2,2,1}, {2,2,3},
{3,2,1}, {3,3,3} products for forward / dgrad / wgrad — one loss_and_grads from the fixture's initial parameters against the
FLOAT64 oracle: relative ELBO error and the rel-L2 error of every gradient tensor, next to the fp32 oracle's own distance from
float64 (e32: the noise floor of the bar's reference).  Then the decoder launch's time per mode at batch 256.

    python scripts/fb_prec_table.py [fixture ...]  > profiles/r04_fb_prec_table.txt
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import pyroved_amd as pv  # noqa: E402
from pyroved_amd import _abi  # noqa: E402
from oracle import svi_oracle as orc  # noqa: E402
from conftest import jmeta_of, load_golden, make_x, meta_of  # noqa: E402

torch.set_num_threads(16)
KINDS = [(int(k), n) for k, n in (kv.split(":") for kv in os.environ["PV_TABLE_KINDS"].split(","))] if os.environ.get("PV_TABLE_KINDS") else [
    (0, "bf16 {3,3,3}"), (21, "f16 {2,2,1}"), (23, "f16 {2,2,3}"), (31, "f16 {3,2,1}"), (26, "f16 {2,3|2,1}"), (27, "f16 {2,2|3,1}"),
    (28, "f16 {2,3,1}"), (33, "f16 {3,3,3}")]
FIXTURES = sys.argv[1:] or ["ivae_28x28_r_b128", "ivae_28x28_rt_b256", "ivae_28x28_r_b32_blobs", "ivae_8x8_rts_b6",
                            "ivae_8x8_rts_b6_randn", "ivae_7x9_rts_b3", "ivae_1d16_t_b5"]
lib = C.CDLL(_abi.LIB_PATH)
from pyroved_amd.engine import IVAEEngine  # noqa: E402  (the other precision corners need PV_LIB_PATH=<the experiments build>)


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item()


for name in FIXTURES:
    gold = load_golden(name)
    jiv = name.startswith("jivae")
    meta = jmeta_of(gold) if jiv else meta_of(gold)
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    eps = torch.from_numpy(gold["s0.eps"])
    if jiv:
        cfg = orc.Config(data_dim=meta["data_dim"], latent_dim=meta["latent_dim"], invariances=meta["invariances"],
                         discrete_dim=meta["discrete_dim"])
    else:
        cfg = orc.Config(data_dim=meta["data_dim"], latent_dim=meta["latent_dim"], invariances=meta["invariances"])
    res, o64, o32 = {}, None, None
    for kind, _ in KINDS:
        IVAEEngine.dec_kernel = kind if kind else 1      # pv_ivae_plan.dec_kernel (1: the bf16 three-product kernel)
        if jiv:
            model = pv.models.jiVAE(meta["data_dim"], meta["latent_dim"], meta["discrete_dim"], meta["invariances"], seed=1,
                                    device="cuda")
        else:
            model = pv.models.iVAE(meta["data_dim"], meta["latent_dim"], meta["invariances"], seed=1, device="cuda")
        eng = model.engine(fused=2)
        if o64 is None:
            sd = {k: v.cpu() for k, v in model.state_dict().items()}
            o64 = orc.SVIOracle(sd, cfg, dtype=torch.float64)
            o64.step(x, eps, meta["beta"])
            o32 = orc.SVIOracle(sd, cfg)
            o32.step(x, eps, meta["beta"])
        eng.loss_and_grads(x.cuda(), eps.cuda(), meta["beta"])
        res[kind] = (eng.scalars[0].item(), {k: rel(eng.grad_of(k).cpu(), o64.last_grads[k]) for k in o64.p})
    IVAEEngine.dec_kernel = 0
    l64 = o64.last["loss"].item()
    rows = meta["batch"] * int(torch.tensor(meta["data_dim"]).prod()) * (meta["discrete_dim"] if jiv else 1)
    print("== %s  (%d decoder rows)   float64 ELBO %.6f" % (name, rows, l64))
    print("%-34s %9s" % ("", "fp32 orc") + "".join(" %13s" % n for _, n in KINDS))
    print("%-34s %9.1e" % ("ELBO", abs(o32.last["loss"].item() - l64) / abs(l64))
          + "".join(" %13.1e" % (abs(res[k][0] - l64) / abs(l64)) for k, _ in KINDS))
    worst = {k: 0.0 for k, _ in KINDS}
    for key in o64.p:
        e32 = rel(o32.last_grads[key], o64.last_grads[key])
        print("%-34s %9.1e" % (key, e32) + "".join(" %13.1e" % res[k][1][key] for k, _ in KINDS))
        for k, _ in KINDS:
            worst[k] = max(worst[k], res[k][1][key])
    print("%-34s %9s" % ("worst gradient tensor", "") + "".join(" %13.1e" % worst[k] for k, _ in KINDS))
    print()

# ---- time per mode at batch 256 (the whole loss_and_grads call; the decoder launch is all that differs) ----
gold = load_golden("ivae_28x28_rt_b256")
meta = meta_of(gold)
x = make_x(meta["xkind"], meta["batch"], meta["data_dim"]).cuda()
eps = torch.from_numpy(gold["s0.eps"]).cuda()
print("== loss_and_grads at batch 256 (28x28, ['r','t']), ms per call, median of 5 x 100 calls")
for kind, label in KINDS:
    IVAEEngine.dec_kernel = kind if kind else 1
    model = pv.models.iVAE(meta["data_dim"], 2, meta["invariances"], seed=1, device="cuda")
    eng = model.engine(fused=2)
    for _ in range(20):
        eng.loss_and_grads(x, eps)
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(100):
            eng.loss_and_grads(x, eps)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 100)
    ts.sort()
    print("%-14s %.4f" % (label, ts[2]))
IVAEEngine.dec_kernel = 0
