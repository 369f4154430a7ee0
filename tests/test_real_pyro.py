"""
Whenever a REAL pyro-ppl is importable (it is not in the build container nor on the GPU box, so this module normally
skips), re-run the reference's own SVItrainer under it and compare with the committed fixtures — which were generated
under tests/golden/_minipyro.py, a restatement of the Pyro semantics the hot path relies on (Trace_ELBO assembly and
its score-function term, TraceEnum_ELBO weighting, SVI.step ordering, zero_grads, the evaluate-steps-the-optimizer
quirk).  A disagreement here means the restatement — not the HIP path — misreads Pyro; the assumption list lives in
_minipyro.py's docstring.  Needs the reference checkout (PYROVED_REFERENCE or /root/reference).
"""
import os
import sys
import types

import numpy as np
import pytest
import torch

pyro = pytest.importorskip("pyro")
if getattr(pyro, "_minipyro", False):
    pytest.skip("only the stand-in is installed", allow_module_level=True)

REF = os.environ.get("PYROVED_REFERENCE", "/root/reference")
if not os.path.isdir(os.path.join(REF, "pyroved")):
    pytest.skip("reference checkout not available", allow_module_level=True)

from conftest import load_golden, make_x, meta_of, jmeta_of        # noqa: E402


def _reference():
    if "torchvision" not in sys.modules:
        try:
            import torchvision  # noqa: F401
        except Exception:
            tv, tvu = types.ModuleType("torchvision"), types.ModuleType("torchvision.utils")
            tvu.make_grid = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("stub"))
            tv.utils = tvu
            sys.modules.update({"torchvision": tv, "torchvision.utils": tvu})
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import pyroved
    return pyroved


@pytest.mark.parametrize("name", ["ivae_8x8_rts_b6", "ivae_8x8_rt_b6_beta4"])
def test_ivae_steps_under_real_pyro(name):
    pv_ref = _reference()
    gold = load_golden(name)
    meta = meta_of(gold)
    model = pv_ref.models.iVAE(meta["data_dim"], meta["latent_dim"], meta["invariances"], seed=1, device="cpu")
    trainer = pv_ref.trainers.SVItrainer(model, seed=1, device="cpu")
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    kw = {} if meta["beta"] == 1.0 else {"scale_factor": meta["beta"]}
    for k in range(meta["steps"]):
        loss = trainer.svi.step(x, **kw)
        np.testing.assert_allclose(loss, float(gold["s%d.loss" % k]), rtol=2e-6)
    for n, p in model.named_parameters():
        a = p.detach().double().flatten()
        np.testing.assert_allclose(a.norm().item(), float(gold["s%d.param.%s.l2" % (meta["steps"] - 1, n)]), rtol=2e-5)


@pytest.mark.parametrize("name,enum", [("jivae_8x8_r_k3_b5", True), ("jsivae_8x8_none_k3_b5", False)])
def test_jivae_steps_under_real_pyro(name, enum):
    pv_ref = _reference()
    gold = load_golden(name)
    meta = jmeta_of(gold)
    model = pv_ref.models.jiVAE(meta["data_dim"], meta["latent_dim"], meta["discrete_dim"], meta["invariances"],
                                seed=1, device="cpu")
    trainer = pv_ref.trainers.SVItrainer(model, enumerate_parallel=enum, seed=1, device="cpu")
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    for k in range(meta["steps"]):
        loss = trainer.svi.step(x)
        np.testing.assert_allclose(loss, float(gold["s%d.loss" % k]), rtol=5e-6)


def test_epochs_with_evaluate_under_real_pyro():
    """The evaluate-moves-the-weights quirk (svi.step under no_grad still runs the optimizer on zeroed gradients)."""
    pv_ref = _reference()
    gold = load_golden("epochs_8x8_rts")
    train, test = torch.from_numpy(gold["train"]), torch.from_numpy(gold["test"])
    batch = int(gold["meta.batch"])
    tl = pv_ref.utils.init_dataloader(train, batch_size=batch)
    vl = pv_ref.utils.init_dataloader(test, batch_size=batch)
    model = pv_ref.models.iVAE((8, 8), 2, ["r", "t", "s"], seed=1, device="cpu")
    trainer = pv_ref.trainers.SVItrainer(model, seed=1, device="cpu")
    for _ in range(int(gold["meta.epochs"])):
        trainer.step(tl, vl)
    np.testing.assert_allclose(trainer.loss_history["training_loss"], gold["epochs.training_loss"], rtol=1e-5)
    np.testing.assert_allclose(trainer.loss_history["test_loss"], gold["epochs.test_loss"], rtol=1e-5)
