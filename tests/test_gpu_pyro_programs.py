"""
model() / guide() as Pyro programs (pyroved_amd/models/_pyro_programs.py; reference: models/ivae.py:165-221,
models/jivae.py:152-220, models/ved.py:122-163) and SVItrainer given Pyro optimizer / loss OBJECTS (trainers/svi.py:66-91).
pyro-ppl is not installable here, so the programs run under the test suite's stand-in `pyro`
(tests/golden/_minipyro.py, registered in sys.modules for the duration of a test only) on the GPU, the networks as
differentiable operators over the library's GEMMs, and are compared with the fused HIP objective of the same model on
the same noise: loss, site terms and every parameter gradient.
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, make_x

import pyroved_amd as pv

pytestmark = pytest.mark.gpu


@pytest.fixture()
def minipyro():
    sys.path.insert(0, GOLDEN)
    import _minipyro
    saved = {k: v for k, v in sys.modules.items() if k == "pyro" or k.startswith("pyro.")}
    _minipyro.install()
    try:
        yield _minipyro
    finally:
        for k in [k for k in sys.modules if k == "pyro" or k.startswith("pyro.")]:
            del sys.modules[k]
        sys.modules.update(saved)
        sys.path.remove(GOLDEN)


def guide_eps(tap, site):
    """the standard-normal draw behind a reparameterised guide site: (z - loc) / scale of its recorded distribution"""
    fn = tap["guide_fns"][site].base_dist
    return ((tap["sites"]["guide." + site] - fn.loc.detach()) / fn.scale.detach()).contiguous()


def rel_l2(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("inv,c_dim,sampler", [(["r", "t", "s"], 0, "bernoulli"), (None, 0, "gaussian"), (["t"], 2, "bernoulli")])
def test_ivae_programs_match_the_fused_objective(gpu_device, minipyro, inv, c_dim, sampler):
    data_dim = (8, 8) if inv != ["t"] else (16,)
    b = 5
    model = pv.models.iVAE(data_dim, 2, inv, c_dim=c_dim, sampler_d=sampler, seed=2, device="cuda")
    x = make_x("rand", b, data_dim).cuda()
    y = None
    if c_dim:
        y = torch.zeros(b, c_dim, device="cuda")
        y[torch.arange(b), torch.arange(b) % c_dim] = 1.0
        x = x.flatten(1)
    elbo = minipyro.Trace_ELBO()
    args = (x,) if y is None else (x, y)
    torch.manual_seed(3)
    loss, params = elbo.loss_and_grads(model.model, model.guide, *args, scale_factor=1.5)
    tap = minipyro.tap()
    eps = guide_eps(tap, "latent")
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    for p_ in model.parameters():
        p_.grad = None
    eng = model.engine(fused=2)
    eng.loss_and_grads(x, eps, 1.5, y)
    s = eng.scalars.cpu().numpy()
    np.testing.assert_allclose(loss.item(), s[0], rtol=2e-5)
    np.testing.assert_allclose(tap["terms"]["model.obs"].item(), s[1], rtol=2e-5)
    np.testing.assert_allclose(tap["terms"]["model.latent"].item(), s[2], rtol=1e-4)
    np.testing.assert_allclose(tap["terms"]["guide.latent"].item(), s[3], rtol=1e-4)
    assert tap["sites"]["guide.latent"].shape == (b, model.z_dim)               # the reference's trace tests' shape checks
    assert tap["sites"]["model.obs"].shape == (b, int(np.prod(data_dim)))
    for n in grads:
        assert rel_l2(grads[n], eng.grad_of(n)) < 2e-4, n


def test_jivae_programs_enumerated_match_the_fused_objective(gpu_device, minipyro):
    b, K = 4, 3
    model = pv.models.jiVAE((8, 8), 2, K, ["r"], seed=2, device="cuda")
    x = make_x("rand", b, (8, 8)).cuda()
    elbo = minipyro.TraceEnum_ELBO(max_plate_nesting=1)
    torch.manual_seed(3)
    loss, _ = elbo.loss_and_grads(model.model, minipyro.config_enumerate(model.guide, "parallel", expand=True), x,
                                  scale_factor=[2.0, 0.5])
    tap = minipyro.tap()
    assert tap["sites"]["guide.latent_disc"].shape == (K, b, K)
    eps = guide_eps(tap, "latent_cont")
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    eng = model.engine(fused=2)
    eng.loss_and_grads(x, eps, [2.0, 0.5])
    np.testing.assert_allclose(loss.item(), eng.scalars[0].item(), rtol=2e-5)
    for n in grads:
        assert rel_l2(grads[n], eng.grad_of(n)) < (5e-3 if "fc13" in n or n.startswith("encoder_z.") else 3e-4), n


def test_ved_programs_match_the_fused_objective(gpu_device, minipyro):
    small = dict(hidden_dim_e=[(8,), (16, 16)], hidden_dim_d=[(16, 16), (8,)])
    model = pv.models.VED((16, 16), (32,), seed=2, device="cuda", **small)
    g = torch.Generator().manual_seed(0)
    x, y = torch.rand(4, 1, 16, 16, generator=g).cuda(), torch.rand(4, 1, 32, generator=g).cuda()
    eng = model.engine()                                   # (the conv nets now belong to an engine: the programs bypass it)
    torch.manual_seed(3)
    loss, _ = minipyro.Trace_ELBO().loss_and_grads(model.model, model.guide, x, y)
    eps = guide_eps(minipyro.tap(), "z")
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    eng.loss_and_grads(x, eps, 1.0, y)
    np.testing.assert_allclose(loss.item(), eng.scalars[0].item(), rtol=2e-5)
    for n in grads:
        assert rel_l2(grads[n], eng.grad_of(n)) < 1e-3, n        # (MIOpen convolutions on the torch side)


def test_trainer_with_pyro_objects_runs_the_pyro_route(gpu_device, minipyro):
    """SVItrainer(model, optimizer=<pyro.optim object>, loss=<ELBO object>) (svi.py:66-67): the generic Pyro route; its
    epoch losses equal the fused HIP trainer's from the same seeds (same data order, same eps stream)."""
    data = make_x("rand", 12, (8, 8))
    hist = []
    for route in ("pyro", "hip"):
        model = pv.models.iVAE((8, 8), 2, ["r", "t"], seed=1, device="cuda")
        loader = pv.utils.init_dataloader(data, batch_size=4)
        if route == "pyro":
            tr = pv.trainers.SVItrainer(model, optimizer=minipyro.Adam({"lr": 1e-3}), loss=minipyro.Trace_ELBO(), seed=1)
            assert tr.svi is not None
        else:
            tr = pv.trainers.SVItrainer(model, seed=1, rng="device")      # the Pyro route's Normal draws on the device
        for _ in range(2):
            tr.step(loader)
        hist.append(tr.loss_history["training_loss"])
    np.testing.assert_allclose(hist[0], hist[1], rtol=1e-4)


@pytest.mark.parametrize("task,inv", [("classification", ["r", "t"]), ("classification", None), ("regression", ["t", "s"])])
def test_semisupervised_programs_match_the_hip_objectives(gpu_device, minipyro, task, inv):
    """(round 3) model / guide / model_aux / guide_aux of ssiVAE and ss_reg_iVAE as Pyro programs (reference:
    models/ssivae.py:153-234, models/ss_reg_ivae.py:156-246) against the three objectives auxSVItrainer evaluates in the
    HIP library (engine_ss.SSEngine): the labeled ELBO step, the unlabeled ELBO step (classification: the guide's label
    enumerated in parallel under TraceEnum_ELBO; regression: a reparameterised label) and the auxiliary supervised loss —
    loss and every parameter gradient, on the noise the programs drew."""
    b, dim, data_dim = 5, 3, (8, 8)
    if task == "classification":
        model = pv.models.ssiVAE(data_dim, 2, dim, inv, seed=2, device="cuda")
        ys = torch.zeros(b, dim, device="cuda")
        ys[torch.arange(b), torch.arange(b) % dim] = 1.0
    else:
        model = pv.models.ss_reg_iVAE(data_dim, 2, dim, inv, seed=2, device="cuda")
        ys = torch.rand(b, dim, generator=torch.Generator().manual_seed(5)).cuda()
    x = make_x("rand", b, data_dim).cuda()
    eng = model.engine()

    def pyro_grads():
        g = {n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()}
        for p_ in model.parameters():
            p_.grad = None
        return g

    def check(grads, what, tol=3e-4):
        for n, gref in grads.items():
            if gref.abs().max() == 0:
                assert eng.grad_of(n).abs().max().item() == 0, "%s: %s" % (what, n)
            else:
                assert rel_l2(eng.grad_of(n), gref) < tol, "%s: %s" % (what, n)

    # (1) labeled batch: Trace_ELBO(model, guide)(xs, ys)
    torch.manual_seed(3)
    loss, _ = minipyro.Trace_ELBO().loss_and_grads(model.model, model.guide, x, ys, scale_factor=1.5)
    eps = guide_eps(minipyro.tap(), "z")
    grads = pyro_grads()
    eng.grad.zero_()
    got = eng.elbo_loss_and_grads(x, eps, ys, beta=1.5)
    np.testing.assert_allclose(got.item(), loss.item(), rtol=2e-5)
    check(grads, "labeled ELBO")
    # (2) unlabeled batch
    torch.manual_seed(4)
    if task == "classification":
        elbo = minipyro.TraceEnum_ELBO(max_plate_nesting=1)
        loss, _ = elbo.loss_and_grads(model.model, minipyro.config_enumerate(model.guide, "parallel", expand=True), x)
        tap = minipyro.tap()
        assert tap["sites"]["guide.y"].shape == (dim, b, dim)
        eps = guide_eps(tap, "z")                                     # (K, B, z)
        grads = pyro_grads()
        eng.grad.zero_()
        got = eng.elbo_loss_and_grads(x, eps)
        tol = 5e-3                                                    # (class-probability gradients: differences of per-class ELBOs)
    else:
        loss, _ = minipyro.Trace_ELBO().loss_and_grads(model.model, model.guide, x)
        tap = minipyro.tap()
        fy = tap["guide_fns"]["y"].base_dist
        eps_y = ((tap["sites"]["guide.y"] - fy.loc.detach()) / fy.scale).contiguous()
        eps = guide_eps(tap, "z")
        grads = pyro_grads()
        eng.grad.zero_()
        got = eng.elbo_loss_and_grads(x, eps, None, eps_y)
        tol = 3e-4
    np.testing.assert_allclose(got.item(), loss.item(), rtol=2e-5)
    check(grads, "unlabeled ELBO", tol)
    # (3) auxiliary loss
    loss, _ = minipyro.Trace_ELBO().loss_and_grads(model.model_aux, model.guide_aux, x, ys, aux_loss_multiplier=7.0)
    grads = pyro_grads()
    eng.grad.zero_()
    got = eng.aux_loss_and_grads(x, ys, 7.0)
    np.testing.assert_allclose(got.item(), loss.item(), rtol=2e-5)
    check(grads, "auxiliary loss")


def test_programs_need_pyro(gpu_device):
    if "pyro" in sys.modules:
        pytest.skip("a pyro module is importable here")
    model = pv.models.iVAE((8, 8), 2, ["r"], seed=1, device="cuda")
    with pytest.raises(NotImplementedError):
        model.guide(torch.rand(2, 8, 8).cuda())
    with pytest.raises(TypeError):
        pv.trainers.SVItrainer(model, optimizer=object())
