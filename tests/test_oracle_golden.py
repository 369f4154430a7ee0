"""
Pins the CPU oracle (oracle/svi_oracle.py) to the golden fixtures that were produced by
running the reference's own code (tests/golden/make_golden.py), and pins the product's
host-side construction logic (parameter initialisation order, grid, RNG contract) to the
same fixtures.  CPU only.
"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, make_x, check_digest, meta_of

import pyroved_amd as pv
from oracle import svi_oracle as orc

STEP_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "ivae_*.npz"))
                    if not p.endswith("_fwd.npz"))
FWD_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "ivae_*_fwd.npz")))
EPOCH_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "epochs_*.npz")))


def build(meta):
    model = pv.models.iVAE(meta["data_dim"], meta["latent_dim"], meta["invariances"], seed=1, device="cpu")
    cfg = orc.Config(data_dim=meta["data_dim"], latent_dim=meta["latent_dim"], invariances=meta["invariances"])
    return model, cfg


def test_fixture_inventory():
    assert len(STEP_CASES) >= 14 and len(FWD_CASES) >= 7 and len(EPOCH_CASES) == 3


@pytest.mark.parametrize("name", STEP_CASES)
def test_product_init_matches_reference(name):
    """Same seed => same initial weights as the reference (construction order, models/ivae.py:140-154)."""
    gold = load_golden(name)
    model, _ = build(meta_of(gold))
    keys = [k[len("init."):-len(".sum")] for k in gold if k.startswith("init.") and k.endswith(".sum")]
    assert sorted(keys) == sorted(model.state_dict().keys())
    for k, p in model.state_dict().items():
        check_digest(p, gold, "init." + k, rtol=0, atol=0, what=name)


@pytest.mark.parametrize("name", STEP_CASES)
def test_oracle_steps_match_reference(name):
    gold = load_golden(name)
    meta = meta_of(gold)
    if name in ("ivae_28x28_rt_b256",):
        torch.set_num_threads(8)
    model, cfg = build(meta)
    o = orc.SVIOracle(model.state_dict(), cfg)
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    for k in range(meta["steps"]):
        pre = "s%d" % k
        eps = torch.from_numpy(gold[pre + ".eps"])
        loss = o.step(x, eps, meta["beta"])
        np.testing.assert_allclose(loss, float(gold[pre + ".loss"]), rtol=2e-6)
        np.testing.assert_allclose(o.last["ll"].item(), float(gold[pre + ".term.model.obs"]), rtol=2e-6)
        np.testing.assert_allclose(o.last["logpz"].item(), float(gold[pre + ".term.model.latent"]), rtol=2e-5)
        np.testing.assert_allclose(o.last["logqz"].item(), float(gold[pre + ".term.guide.latent"]), rtol=2e-5)
        np.testing.assert_allclose(o.last["z_loc"].detach().numpy(), gold[pre + ".z_loc"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(o.last["z_scale"].detach().numpy(), gold[pre + ".z_scale"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(o.last["z"].detach().numpy(), gold[pre + ".z"], rtol=1e-4, atol=1e-6)
        for key in o.p:
            check_digest(o.last_grads[key], gold, pre + ".grad." + key, rtol=2e-4, atol=1e-7, what=name)
            check_digest(o.p[key], gold, pre + ".param." + key, rtol=2e-5, atol=1e-7, what=name)
    # inference API
    z_loc, z_scale = o.encode(x)
    np.testing.assert_allclose(z_loc.numpy(), gold["enc.z_loc"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(z_scale.numpy(), gold["enc.z_scale"], rtol=1e-4, atol=1e-6)
    dec = o.decode(z_loc[:, -meta["latent_dim"]:])
    np.testing.assert_allclose(dec.numpy(), gold["dec.loc"], rtol=1e-4, atol=1e-6)
    if "dec.loc_ats" in gold:
        dec2 = o.decode(z_loc[:2, -meta["latent_dim"]:], angle=0.3, shift=torch.tensor([0.1, -0.2]), scale=1.2)
        np.testing.assert_allclose(dec2.numpy(), gold["dec.loc_ats"], rtol=1e-4, atol=1e-6)


def test_oracle_full_tensors():
    """The one fixture that keeps whole tensors: every gradient element of step 0 and every
    parameter element after 3 steps."""
    gold = load_golden("ivae_8x8_rts_b6")
    meta = meta_of(gold)
    model, cfg = build(meta)
    for k, p in model.state_dict().items():
        np.testing.assert_array_equal(p.numpy(), gold["full.init." + k])
    o = orc.SVIOracle(model.state_dict(), cfg)
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    for k in range(meta["steps"]):
        o.step(x, torch.from_numpy(gold["s%d.eps" % k]))
        for key in o.p:
            g = gold["full.s%d.grad.%s" % (k, key)]
            np.testing.assert_allclose(o.last_grads[key].numpy(), g, rtol=2e-4, atol=2e-6 * np.abs(g).max())
    for key in o.p:
        np.testing.assert_allclose(o.p[key].detach().numpy(), gold["full.s2.param." + key], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("name", FWD_CASES)
def test_oracle_forward_pieces(name):
    """Transformed grid, decoder output and ELBO terms of the forward pass (no update)."""
    gold = load_golden(name)
    meta = meta_of(gold)
    model, cfg = build(meta)
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    with torch.no_grad():
        out = orc.elbo(model.state_dict(), cfg, x, torch.from_numpy(gold["eps"]))
    np.testing.assert_allclose(out["loss"].item(), float(gold["loss"]), rtol=2e-6)
    np.testing.assert_allclose(out["z"].numpy(), gold["z"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(out["loc"].numpy(), gold["loc"], rtol=1e-5, atol=1e-7)
    if "x_coord_prime" in gold:
        np.testing.assert_allclose(out["x_coord_prime"].numpy(), gold["x_coord_prime"], rtol=1e-6, atol=5e-7)
        # the product's construction-time grid is the reference's grid
        g = pv.utils.generate_grid(meta["data_dim"])
        np.testing.assert_array_equal(g.numpy(), orc.generate_grid(meta["data_dim"]).numpy())


@pytest.mark.parametrize("name", STEP_CASES[:3])
def test_eps_stream_contract(name):
    """eps of step k == the k-th torch.empty(B, z).normal_() after torch.manual_seed(1)
    (SVItrainer re-seeds in its constructor: trainers/svi.py:76)."""
    gold = load_golden(name)
    meta = meta_of(gold)
    _, cfg = build(meta)
    torch.manual_seed(1)
    for k in range(meta["steps"]):
        eps = torch.empty(meta["batch"], cfg.z_dim).normal_()
        np.testing.assert_array_equal(eps.numpy(), gold["s%d.eps" % k])


@pytest.mark.parametrize("name", EPOCH_CASES)
def test_oracle_epochs_match_reference_trainer(name):
    """Whole SVItrainer.step(train[, test]) epochs: DataLoader shuffling order, eps stream,
    evaluate() semantics (svi.step under no_grad still runs the optimizer)."""
    gold = load_golden(name)
    inv = str(gold["meta.invariances"])
    data_dim = tuple(int(v) for v in gold["meta.data_dim"])
    train, test = torch.from_numpy(gold["train"]), torch.from_numpy(gold["test"])
    batch = int(gold["meta.batch"])
    train_loader = pv.utils.init_dataloader(train, batch_size=batch)
    test_loader = pv.utils.init_dataloader(test, batch_size=batch)
    model = pv.models.iVAE(data_dim, 2, list(inv) if inv else None, seed=1, device="cpu")
    cfg = orc.Config(data_dim=data_dim, latent_dim=2, invariances=list(inv) if inv else None)
    o = orc.SVIOracle(model.state_dict(), cfg)
    torch.manual_seed(1)                   # SVItrainer.__init__ (svi.py:76)
    tr, te = [], []
    for _ in range(int(gold["meta.epochs"])):
        tr.append(o.train_epoch(train_loader))
        if int(gold["meta.with_test"]):
            te.append(o.evaluate_epoch(test_loader))
    np.testing.assert_allclose(tr, gold["epochs.training_loss"], rtol=1e-5)
    np.testing.assert_allclose(te, gold["epochs.test_loss"], rtol=1e-5)
    for key in o.p:
        check_digest(o.p[key], gold, "final." + key, rtol=1e-4, atol=1e-7, what=name)


# ---------------------------------------------------------------- constructor variants / likelihoods
# (models/ivae.py:122-163, utils/prob.py:25-29): Gaussian and ContinuousBernoulli likelihoods, sigmoid_d, decoder_sig,
# c_dim on a plain iVAE, every activation, non-default hidden widths / depths, custom priors, other latent sizes —
# each branch of the oracle pinned to a fixture the reference's own code produced
VARIANT_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "ivaevar_*.npz")))


def test_variant_fixture_inventory():
    assert len(VARIANT_CASES) >= 19


@pytest.mark.parametrize("name", VARIANT_CASES)
def test_variant_oracle_steps_match_reference(name):
    from conftest import variant_of, variant_inputs
    gold = load_golden(name)
    meta, kw, cfg_kw = variant_of(gold)
    model = pv.models.iVAE(meta["data_dim"], meta["latent_dim"], meta["invariances"], c_dim=meta["c_dim"], seed=1,
                           device="cpu", **kw)
    keys = [k[len("init."):-len(".sum")] for k in gold if k.startswith("init.") and k.endswith(".sum")]
    assert sorted(keys) == sorted(model.state_dict().keys())
    for k, p in model.state_dict().items():
        check_digest(p, gold, "init." + k, rtol=0, atol=0, what=name)
    cfg = orc.Config(**cfg_kw)
    o = orc.SVIOracle(model.state_dict(), cfg)
    x, y = variant_inputs(meta)
    cb = cfg.sampler == "continuous_bernoulli"
    for k in range(meta["steps"]):
        pre = "s%d" % k
        loss = o.step(x, torch.from_numpy(gold[pre + ".eps"]), meta["beta"], y)
        # ContinuousBernoulli on a fresh model: the loss is a few units left over from B*N per-pixel terms of size
        # ~0.7 that nearly cancel -> compared on the scale of the sum of the terms
        np.testing.assert_allclose(loss, float(gold[pre + ".loss"]), rtol=3e-6,
                                   atol=(2e-7 * x.numel()) if cb else 0.0)
        np.testing.assert_allclose(o.last["ll"].item(), float(gold[pre + ".term.model.obs"]), rtol=3e-6,
                                   atol=(2e-7 * x.numel()) if cb else 0.0)
        np.testing.assert_allclose(o.last["logpz"].item(), float(gold[pre + ".term.model.latent"]), rtol=2e-5)
        np.testing.assert_allclose(o.last["logqz"].item(), float(gold[pre + ".term.guide.latent"]), rtol=2e-5)
        np.testing.assert_allclose(o.last["z"].detach().numpy(), gold[pre + ".z"], rtol=1e-4, atol=1e-6)
        for key in o.p:
            check_digest(o.last_grads[key], gold, pre + ".grad." + key, rtol=2e-4, atol=1e-6, what=name)
            check_digest(o.p[key], gold, pre + ".param." + key, rtol=2e-5, atol=1e-6, what=name)
    z_loc, z_scale = o.encode(x, y)
    np.testing.assert_allclose(z_loc.numpy(), gold["enc.z_loc"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(z_scale.numpy(), gold["enc.z_scale"], rtol=1e-4, atol=1e-6)
    dec = o.decode(z_loc[:, -meta["latent_dim"]:], y)
    np.testing.assert_allclose(dec.numpy(), gold["dec.loc"], rtol=1e-4, atol=1e-6)


# ---------------------------------------------------------------- jiVAE (models/jivae.py, TraceEnum_ELBO)
JSTEP_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "jivae_*.npz")))


def test_jivae_fixture_inventory():
    assert len(JSTEP_CASES) >= 5


@pytest.mark.parametrize("name", JSTEP_CASES)
def test_jivae_oracle_steps_match_reference(name):
    """The product's jiVAE constructor reproduces the reference's initial weights, and the oracle's enumerated ELBO
    (oracle.jelbo) reproduces the reference's jiVAE.model/guide run through SVItrainer(enumerate_parallel=True):
    loss, the five site terms, class probabilities, gradients, parameters after Adam, for every recorded step."""
    from conftest import jmeta_of, jivae_grad_tol
    gold = load_golden(name)
    meta = jmeta_of(gold)
    model = pv.models.jiVAE(meta["data_dim"], meta["latent_dim"], meta["discrete_dim"], meta["invariances"],
                            seed=1, device="cpu")
    for n, p in model.named_parameters():
        check_digest(p, gold, "init." + n, rtol=0, atol=0, what=name)
    cfg = orc.Config(data_dim=meta["data_dim"], latent_dim=meta["latent_dim"], invariances=meta["invariances"],
                     discrete_dim=meta["discrete_dim"])
    o = orc.SVIOracle(model.state_dict(), cfg)
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    for k in range(meta["steps"]):
        pre = "s%d" % k
        loss = o.step(x, torch.from_numpy(gold[pre + ".eps"]), meta["beta"])
        np.testing.assert_allclose(loss, float(gold[pre + ".loss"]), rtol=2e-6)
        for t, v in o.last["terms"].items():
            np.testing.assert_allclose(v.item(), float(gold[pre + ".term." + t]), rtol=2e-6, err_msg=t)
        np.testing.assert_allclose(o.last["alpha"].detach().numpy(), gold[pre + ".alpha"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(o.last["z"].detach().numpy(), gold[pre + ".z"], rtol=1e-6, atol=1e-7)
        for n in o.p:
            tol = jivae_grad_tol(n) or 1e-3
            check_digest(o.last_grads[n], gold, pre + ".grad." + n, rtol=2 * tol, atol=tol * float(gold[pre + ".grad." + n + ".l2"]) / 8,
                         what=name)
            check_digest(o.p[n], gold, pre + ".param." + n, rtol=1e-4, atol=2e-6, what=name)
    z_loc, z_scale, alpha = o.encode(x)
    np.testing.assert_allclose(z_loc.numpy(), gold["enc.z_loc"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(alpha.numpy(), gold["enc.alpha"], rtol=1e-4, atol=1e-6)
    assert (alpha.argmax(1).numpy() == gold["enc.classes"]).all()


# ---------------------------------------------------------------- jiVAE, sampled class (the trainer's default)
JSAMPLED_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "jsivae_*.npz")))


@pytest.mark.parametrize("name", JSAMPLED_CASES)
def test_jivae_sampled_class_oracle_matches_reference(name):
    """SVItrainer(jiVAE) with the reference's DEFAULT enumerate_parallel=False (trainers/svi.py:66, 83-91): Trace_ELBO on a
    class drawn by the guide, score-function gradient for the class logits (oracle.jelbo_sampled).  Loss, the five site
    terms, log_r, every gradient and parameter per recorded step — first with the recorded draws, then with the oracle
    drawing eps and the class itself from the trainer's seed (pins the generator contract: normal_ then multinomial)."""
    from conftest import jmeta_of
    gold = load_golden(name)
    meta = jmeta_of(gold)
    assert int(gold["meta.enumerate_parallel"]) == 0 and meta["invariances"] is None
    cfg = orc.Config(data_dim=meta["data_dim"], latent_dim=meta["latent_dim"], invariances=None,
                     discrete_dim=meta["discrete_dim"])
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    for own_draws in (False, True):
        model = pv.models.jiVAE(meta["data_dim"], meta["latent_dim"], meta["discrete_dim"], None, seed=1, device="cpu")
        o = orc.SVIOracle(model.state_dict(), cfg)
        o.sampled_class = True
        torch.manual_seed(1)                          # SVItrainer.__init__ (svi.py:76)
        for k in range(meta["steps"]):
            pre = "s%d" % k
            if own_draws:
                eps = torch.empty(meta["batch"], cfg.z_dim).normal_()
                np.testing.assert_array_equal(eps.numpy(), gold[pre + ".eps"])
                loss = o.step(x, eps, meta["beta"], None)
                np.testing.assert_array_equal(o.last["y"].numpy(), gold[pre + ".y"])
            else:
                loss = o.step(x, torch.from_numpy(gold[pre + ".eps"]), meta["beta"], torch.from_numpy(gold[pre + ".y"]))
            np.testing.assert_allclose(loss, float(gold[pre + ".loss"]), rtol=2e-6)
            for t, v in o.last["terms"].items():
                np.testing.assert_allclose(v.item(), float(gold[pre + ".term." + t]), rtol=2e-6, err_msg=t)
            np.testing.assert_allclose(o.last["log_r"].numpy(), gold[pre + ".log_r"], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(o.last["alpha"].detach().numpy(), gold[pre + ".alpha"], rtol=1e-5, atol=1e-7)
            for n in o.p:
                check_digest(o.last_grads[n], gold, pre + ".grad." + n, rtol=2e-4, atol=1e-6, what=name)
                check_digest(o.p[n], gold, pre + ".param." + n, rtol=2e-5, atol=1e-6, what=name)


def test_jivae_sampled_class_needs_the_vanilla_decoder():
    """With invariances the reference's own model cannot run without enumeration (models/jivae.py:181-189: z is repeated
    K times and then concatenated with the (B, K) drawn class -> RuntimeError from Concat's broadcast); mirrored."""
    cfg = orc.Config(data_dim=(8, 8), latent_dim=2, invariances=["r"], discrete_dim=3)
    model = pv.models.jiVAE((8, 8), 2, 3, ["r"], seed=1, device="cpu")
    with pytest.raises(RuntimeError):
        orc.jelbo_sampled(model.state_dict(), cfg, torch.rand(4, 8, 8), torch.randn(4, 3))


# ---------------------------------------------------------------- VED (models/ved.py, nets/conv.py)
VED_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "ved_*.npz")))


def ved_case(gold):
    import ast
    kw = {}
    for k in gold:
        if k.startswith("meta.model_kw."):
            v = str(gold[k])
            kw[k[len("meta.model_kw."):]] = ast.literal_eval(v) if (v[0] in "[(" or v in ("True", "False")) else v
    return dict(input_dim=tuple(int(v) for v in gold["meta.input_dim"]),
                output_dim=tuple(int(v) for v in gold["meta.output_dim"]),
                latent_dim=int(gold["meta.latent_dim"]), steps=int(gold["meta.steps"]),
                beta=float(gold["meta.scale_factor"]), kw=kw)


def test_ved_fixture_inventory():
    assert len(VED_CASES) >= 5


@pytest.mark.parametrize("name", VED_CASES)
def test_ved_oracle_steps_match_reference(name):
    """The product's VED constructor (conv nets mirror) reproduces the reference's initial weights under the same
    state_dict keys, and the oracle's conv restatement (oracle.ved_elbo) reproduces the reference's VED.model/guide
    through SVItrainer.svi.step(x, y): loss, gradients, parameters after Adam per step, then encode / decode."""
    gold = load_golden(name)
    c = ved_case(gold)
    model = pv.models.VED(c["input_dim"], c["output_dim"], latent_dim=c["latent_dim"], seed=1, device="cpu", **c["kw"])
    keys = sorted(k[len("init."):-len(".sum")] for k in gold if k.startswith("init.") and k.endswith(".sum"))
    assert keys == sorted(n for n, _ in model.named_parameters())
    for n, p in model.named_parameters():
        check_digest(p, gold, "init." + n, rtol=0, atol=0, what=name)
    cfg = orc.VedConfig(input_dim=c["input_dim"], output_dim=c["output_dim"], latent_dim=c["latent_dim"],
                        hidden_dim_e=c["kw"].get("hidden_dim_e"), hidden_dim_d=c["kw"].get("hidden_dim_d"),
                        activation=c["kw"].get("activation", "lrelu"))
    o = orc.VedOracle(model.state_dict(), cfg)
    x, y = torch.from_numpy(gold["x"]), torch.from_numpy(gold["y"])
    for k in range(c["steps"]):
        pre = "s%d" % k
        loss = o.step(x, y, torch.from_numpy(gold[pre + ".eps"]), c["beta"])
        np.testing.assert_allclose(loss, float(gold[pre + ".loss"]), rtol=3e-6)
        np.testing.assert_allclose(o.last["ll"].item(), float(gold[pre + ".term.model.obs"]), rtol=3e-6)
        np.testing.assert_allclose(o.last["z"].detach().numpy(), gold[pre + ".z"], rtol=1e-5, atol=1e-6)
        for n in o.p:
            check_digest(o.last_grads[n], gold, pre + ".grad." + n, rtol=2e-4, atol=1e-6, what=name)
            check_digest(o.p[n], gold, pre + ".param." + n, rtol=1e-5, atol=1e-6, what=name)
    z_loc, z_scale = o.encode(x)
    np.testing.assert_allclose(z_loc.numpy(), gold["enc.z_loc"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(o.decode(z_loc).numpy(), gold["dec.loc"], rtol=1e-5, atol=1e-6)


VEDBN_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "vedbn_*.npz")))


def vedbn_oracle(gold, device="cpu"):
    c = ved_case(gold)
    model = pv.models.VED(c["input_dim"], c["output_dim"], latent_dim=c["latent_dim"], seed=1, device=device, **c["kw"])
    cfg = orc.VedConfig(input_dim=c["input_dim"], output_dim=c["output_dim"], latent_dim=c["latent_dim"],
                        hidden_dim_e=c["kw"].get("hidden_dim_e"), hidden_dim_d=c["kw"].get("hidden_dim_d"),
                        activation=c["kw"].get("activation", "lrelu"), batchnorm=True)
    return c, model, cfg


@pytest.mark.parametrize("name", VEDBN_CASES)
def test_vedbn_oracle_steps_match_reference(name):
    """VED(batchnorm=True): batch statistics + running estimates while training, and — after encode() / decode() left the
    module in eval() mode, as the reference does — one more training step on the running statistics ("e0")."""
    gold = load_golden(name)
    c, model, cfg = vedbn_oracle(gold)
    assert c["kw"]["batchnorm"] is True
    for n, p in model.named_parameters():
        check_digest(p, gold, "init." + n, rtol=0, atol=0, what=name)
    o = orc.VedOracle(model.state_dict(), cfg)
    x, y = torch.from_numpy(gold["x"]), torch.from_numpy(gold["y"])
    for k in range(c["steps"]):
        pre = "s%d" % k
        loss = o.step(x, y, torch.from_numpy(gold[pre + ".eps"]), c["beta"])
        np.testing.assert_allclose(loss, float(gold[pre + ".loss"]), rtol=5e-6 if k == 0 else 1e-4)
        for n in o.p:
            check_digest(o.last_grads[n], gold, pre + ".grad." + n, rtol=3e-4 if k == 0 else 2e-2, atol=2e-6, what=name)
            check_digest(o.p[n], gold, pre + ".param." + n, rtol=1e-5, atol=2.2e-3, what=name, sum_slack=0.02 * o.p[n].numel() ** 0.5 * 1e-3)   # (gradients of the biases in front of a batch norm are noise-sized: Adam sign flips)
    z_loc, z_scale = o.encode(x)
    np.testing.assert_allclose(z_loc.numpy(), gold["enc.z_loc"], rtol=2e-3, atol=1e-4)
    np.testing.assert_allclose(o.decode(z_loc).numpy(), gold["dec.loc"], rtol=2e-3, atol=1e-4)
    for n, b_ in o.bufs.items():
        check_digest(b_.float(), gold, "buf." + n, rtol=1e-3, atol=1e-5, what=name)
    # the eval-mode step: by now the two runs stand on parameters that differ by Adam's noise-decided entries and the
    # batches are 4-5 samples behind (leaky) ReLU kinks, so only the loss and the gradients' norms are compared here; the
    # eval-mode arithmetic itself is pinned by the GPU tests against this oracle from identical parameters
    loss = o.step(x, y, torch.from_numpy(gold["e0.eps"]), c["beta"])
    np.testing.assert_allclose(loss, float(gold["e0.loss"]), rtol=2e-3)
    for n in o.p:
        np.testing.assert_allclose(o.last_grads[n].double().norm().item(), float(gold["e0.grad." + n + ".l2"]), rtol=0.1,
                                   atol=1e-5, err_msg=n)


# ---------------------------------------------------------------- iVAE with a convolutional encoder (set_encoder)
CONVENC_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "ivaeconv_*.npz")))


def convenc_model(gold, device):
    """iVAE + set_encoder(convEncoderNet(data_dim, latent_dim=z_dim, hidden_dim=...)) built like the fixture's model."""
    import ast
    meta = meta_of(gold)
    hid = ast.literal_eval(str(gold["meta.conv_encoder"]))
    model = pv.models.iVAE(meta["data_dim"], meta["latent_dim"], meta["invariances"], seed=1, device=device)
    model.set_encoder(pv.nets.convEncoderNet(meta["data_dim"], latent_dim=model.z_dim, hidden_dim=hid))
    cfg = orc.Config(data_dim=meta["data_dim"], latent_dim=meta["latent_dim"], invariances=meta["invariances"],
                     conv_encoder=hid)
    return meta, model, cfg


@pytest.mark.parametrize("name", CONVENC_CASES)
def test_convenc_oracle_steps_match_reference(name):
    gold = load_golden(name)
    meta, model, cfg = convenc_model(gold, "cpu")
    for n, p in model.named_parameters():
        check_digest(p, gold, "init." + n, rtol=0, atol=0, what=name)
    o = orc.SVIOracle(model.state_dict(), cfg)
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    for k in range(meta["steps"]):
        pre = "s%d" % k
        loss = o.step(x, torch.from_numpy(gold[pre + ".eps"]), meta["beta"])
        np.testing.assert_allclose(loss, float(gold[pre + ".loss"]), rtol=3e-6)
        for n in o.p:
            check_digest(o.last_grads[n], gold, pre + ".grad." + n, rtol=2e-4, atol=1e-6, what=name)
            check_digest(o.p[n], gold, pre + ".param." + n, rtol=1e-5, atol=1e-6, what=name)
    z_loc, z_scale = o.encode(x)
    np.testing.assert_allclose(z_loc.numpy(), gold["enc.z_loc"], rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------
# semi-supervised models (ssiVAE / ss_reg_iVAE through auxSVItrainer): fixtures produced by the reference's own
# models/ssivae.py, models/ss_reg_ivae.py and trainers/auxsvi.py
SS_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "ss*_*.npz")))


def test_ss_fixture_inventory():
    assert len(SS_CASES) >= 9


@pytest.mark.parametrize("name", SS_CASES)
def test_ss_oracle_matches_reference(name):
    from conftest import ssmeta_of, ss_build
    gold = load_golden(name)
    meta = ssmeta_of(gold)
    model = ss_build(meta, "cpu")
    keys = [k[len("init."):-len(".sum")] for k in gold if k.startswith("init.") and k.endswith(".sum")]
    assert sorted(keys) == sorted(model.state_dict().keys())
    for k, p in model.state_dict().items():
        check_digest(p, gold, "init." + k, rtol=0, atol=0, what=name)
    cfg = orc.Config(data_dim=meta["data_dim"], latent_dim=meta["latent_dim"], invariances=meta["invariances"],
                     c_dim=meta["dim"])
    o = orc.SSOracle(model.state_dict(), cfg, meta["task"])
    xu, xs, ys = (torch.from_numpy(gold[k]) for k in ("xu", "xs", "ys"))
    for c in range(meta["calls"]):
        pre = "c%d" % c
        unl = str(gold[pre + ".kind"]) == "u"
        x = xu[:meta["batch_u"]] if unl else xs[:meta["batch_s"]]
        y = None if unl else ys[:meta["batch_s"]]
        eps = torch.from_numpy(gold[pre + ".eps"])
        eps_y = torch.from_numpy(gold[pre + ".eps_y"]) if (pre + ".eps_y") in gold else None
        l1, l2 = o.compute_loss(x, y, eps, eps_y, meta["beta"], meta["mult"])
        np.testing.assert_allclose(l1, float(gold[pre + ".elbo.loss"]), rtol=3e-6 if c == 0 else 1e-4)
        np.testing.assert_allclose(l2, float(gold[pre + ".aux.loss"]), rtol=3e-6 if c < 2 else 2e-3, atol=1e-12)
        if (pre + ".alpha") in gold:
            np.testing.assert_allclose(o.last["alpha"].detach().numpy(), gold[pre + ".alpha"], rtol=1e-5, atol=1e-7)
        for which in ("elbo", "aux"):
            for key in o.p:
                gk = "%s.%s.grad.%s" % (pre, which, key)
                g = o.last_grads[which][key]
                if gk + ".sum" in gold:
                    # from the second call on the two runs stand on parameters that differ by Adam's noise-decided entries
                    check_digest(g, gold, gk, rtol=3e-4 if c == 0 else 5e-3, atol=2e-7, what=name)
                else:
                    assert g is None, "%s: oracle has a gradient the reference run did not" % gk
        for key in o.p:
            # (5e-4 = lr: up to ~10 noise-decided entries per 100k may land on the other side of Adam's first steps)
            check_digest(o.p[key], gold, pre + ".param." + key, rtol=2e-5, atol=1.1e-3, what=name,
                         sum_slack=1e-2 * 5e-4 * o.p[key].numel() ** 0.5)
    # inference API
    if meta["task"] == "classification":
        np.testing.assert_array_equal(o.predict(xs).numpy(), gold["cls.pred"])
        yq = pv.utils.to_onehot(torch.from_numpy(gold["enc.y_pred"]), meta["dim"])
    else:
        np.testing.assert_allclose(o.predict(xs).numpy(), gold["reg.pred"], rtol=1e-4, atol=1e-6)
        yq = torch.from_numpy(gold["enc.y"])
    z_loc, z_scale = o.encode(xu[:meta["batch_u"]], yq)
    np.testing.assert_allclose(z_loc.numpy(), gold["enc.z_loc"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(z_scale.numpy(), gold["enc.z_scale"], rtol=1e-4, atol=1e-6)
    dec = o.decode(torch.from_numpy(gold["enc.z_loc"])[:, -meta["latent_dim"]:], torch.from_numpy(gold["dec.y"]))
    np.testing.assert_allclose(dec.numpy().reshape(gold["dec.loc"].shape), gold["dec.loc"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", SS_CASES)
def test_ss_oracle_epochs_match_reference_trainer(name):
    """auxSVItrainer.step epochs (loader order, the interleaving of labeled batches, RNG stream, test metric)."""
    from conftest import ssmeta_of, ss_build
    gold = load_golden(name)
    meta = ssmeta_of(gold)
    model = ss_build(meta, "cpu")
    cfg = orc.Config(data_dim=meta["data_dim"], latent_dim=meta["latent_dim"], invariances=meta["invariances"],
                     c_dim=meta["dim"])
    xu, xs, ys = (torch.from_numpy(gold[k]) for k in ("xu", "xs", "ys"))
    pv.utils.set_deterministic_mode(1)                   # auxSVItrainer.__init__ re-seeds (auxsvi.py:60)
    o = orc.SSOracle(model.state_dict(), cfg, meta["task"])
    lu, ls, lv = pv.utils.init_ssvae_dataloaders(xu[:meta["n_u"]], (xs[:meta["n_s"]], ys[:meta["n_s"]]),
                                                 (xs[:meta["n_s"]], ys[:meta["n_s"]]), batch_size=meta["batch_s"])
    tr, te = [], []
    for _ in range(meta["epochs"]):
        tr.append(o.train_epoch(lu, ls, meta["beta"], meta["mult"]))
        te.append(o.evaluate(lv))
    np.testing.assert_allclose(tr, gold["epochs.training_loss"], rtol=1e-5)
    np.testing.assert_allclose(te, gold["epochs.test"], rtol=1e-4, atol=1e-7)
    for key in o.p:      # (after several Adam steps: noise-decided entries, see above)
        check_digest(o.p[key], gold, "final." + key, rtol=1e-4, atol=1.1e-3, what=name,
                     sum_slack=1e-2 * 5e-4 * o.p[key].numel() ** 0.5)


def test_conv_decisions_replay_reproduces_the_forward_and_counts_flips():
    """ConvDecisions (test infrastructure of the full-size conv parity tests): a float64 forward run under its OWN recorded
    leaky-ReLU signs and max-pool winners reproduces loss and gradients; so does the form a HIP step hands over (the sign of a
    conv + pool pair known at the pooled resolution only); the flip counter sees exactly the decisions that were changed."""
    cfg = orc.VedConfig(input_dim=(16, 16), output_dim=(32,), latent_dim=2)
    model = pv.models.VED((16, 16), (32,), seed=3, device="cpu")
    g = torch.Generator().manual_seed(0)
    x, y, eps = torch.rand(3, 1, 16, 16, generator=g), torch.rand(3, 1, 32, generator=g), torch.randn(3, 2, generator=g)

    def run(decisions):
        p_ = {k: v.detach().clone().double().requires_grad_(True) for k, v in model.state_dict().items()}
        out = orc.ved_elbo(p_, cfg, x.double(), y.double(), eps.double(), decisions=decisions)
        out["loss"].backward()
        return out["loss"].item(), {k: v.grad for k, v in p_.items()}
    rec = orc.ConvDecisions()
    loss0, g0 = run(rec)
    assert len(rec.sign) == 5 and len(rec.win) == 2 and rec.flips(rec)[:2] == (0, 0)
    loss_plain, g_plain = run(None)
    assert loss_plain == loss0
    # replay, full-resolution signs
    loss1, g1 = run(orc.ConvDecisions(sign=rec.sign, win=rec.win))
    np.testing.assert_allclose(loss1, loss0, rtol=1e-13)
    for k in g0:
        np.testing.assert_allclose(g1[k].numpy(), g0[k].numpy(), rtol=1e-10, atol=1e-13)
    # replay, the pooled pairs' signs at the pooled resolution (what a fused conv + pool epilogue leaves behind)
    pooled_after = {0: 0, 2: 1}                        # conv index -> pool index (blocks (32,), (64, 64), (128, 128))
    sign_p = [orc.ConvDecisions.pick(s_.to(torch.int8), rec.win[pooled_after[i]]).bool() if i in pooled_after else s_
              for i, s_ in enumerate(rec.sign)]
    hip_like = orc.ConvDecisions(sign=sign_p, win=rec.win)
    loss2, g2 = run(hip_like)
    np.testing.assert_allclose(loss2, loss0, rtol=1e-13)
    for k in g0:
        np.testing.assert_allclose(g2[k].numpy(), g0[k].numpy(), rtol=1e-10, atol=1e-13)
    assert rec.flips(hip_like)[:2] == (0, 0)
    # three flipped winners and two flipped signs are counted as such — and change the gradients
    win_f = [w.clone() for w in rec.win]
    win_f[1].view(-1)[:3] = (win_f[1].view(-1)[:3] + 1) % 4
    sign_f = [s_.clone() for s_ in rec.sign]
    sign_f[3].view(-1)[:2] = ~sign_f[3].view(-1)[:2]
    flipped = orc.ConvDecisions(sign=sign_f, win=win_f)
    ds, dw, ns, nw = rec.flips(flipped)
    assert dw == 3 and 2 <= ds <= 5 and ns > 0 and nw == rec.win[0].numel() + rec.win[1].numel()
    _, g3 = run(flipped)
    assert max(float((g3[k] - g0[k]).norm() / g0[k].norm()) for k in g0 if "feature_extractor" in k) > 1e-6
