"""
GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C ABI
(libpyroved_amd.so via ctypes), against
  (1) the golden fixtures produced by the reference's own code, and
  (2) the CPU oracle on the same seeded inputs.
Tolerance: BASELINE.json's bar is 1e-4 relative in fp32 for the ELBO and reconstructions; the
asserts below use 2e-5 .. 1e-4 as noted per quantity.
"""
import ctypes as C
import dataclasses
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, make_x, check_digest, meta_of, jmeta_of, jivae_grad_tol

import pyroved_amd as pv
from pyroved_amd import _abi
from oracle import svi_oracle as orc

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

STEP_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "ivae_*.npz"))
                    if not p.endswith("_fwd.npz"))
FWD_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "ivae_*_fwd.npz")))
JSTEP_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "jivae_*.npz")))
EPOCH_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "epochs_*.npz")))

RTOL_ELBO = 2e-5      # ELBO terms (bar: 1e-4)
RTOL_GRAD = 1e-4      # per-tensor relative L2 error of gradients vs the fp32 oracle
FUSED = [0, 1, 2]       # 0 layer-by-layer kernels, 1 fused f32-MFMA decoder, 2 fused bf16 split-precision decoder


def rel_l2(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def build(meta, fused, **kw):
    model = pv.models.iVAE(meta["data_dim"], meta["latent_dim"], meta["invariances"], seed=1, device="cuda", **kw)
    cfg = orc.Config(data_dim=meta["data_dim"], latent_dim=meta["latent_dim"], invariances=meta["invariances"])
    eng = model.engine(fused=fused)
    return model, cfg, eng


def test_native_library_loaded(gpu_device):
    lib = _abi.lib()
    assert os.path.samefile(_abi.LIB_PATH, os.path.join(os.path.dirname(pv.__file__), "libpyroved_amd.so"))
    assert lib.pv_version() == _abi.PV_ABI_VERSION
    with open("/proc/self/maps") as f:
        assert "libpyroved_amd.so" in f.read()


@pytest.mark.parametrize("m,k,n,act", [(5, 3, 7, "tanh"), (200, 784, 128, "tanh"), (64, 128, 128, "relu"),
                                       (1000, 130, 70, "gelu"), (33, 4096, 128, "softplus"), (4096, 128, 128, "lrelu"),
                                       (256, 128, 10, None), (17, 2, 128, "sigmoid")])
def test_linear_fwd_bwd_blocks(gpu_device, m, k, n, act):
    """pv_linear_fwd / pv_linear_bwd vs a plain torch fp32 reference of the same op (computed in fp64)."""
    g = torch.Generator().manual_seed(m * 7 + k)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g)
    dy = torch.randn(m, n, generator=g)
    xd, wd_, bd = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    pre = torch.nn.functional.linear(xd, wd_, bd)
    y_ref = orc._ACT[act](pre) if act in orc._ACT else (torch.sigmoid(pre) if act == "sigmoid" else pre)
    from pyroved_amd import ops
    y = ops.linear_act(x.cuda(), w.cuda(), b.cuda(), act)
    assert rel_l2(y, y_ref.detach()) < 2e-6
    # backward: dpre given; dx (no act factor), dw, db
    pre.backward(dy.double())
    L = _abi.lib()
    xc, wc, dyc = x.cuda(), w.cuda(), dy.cuda()
    dx = torch.empty(m, k, device="cuda")
    dw = torch.empty(n, k, device="cuda")
    db = torch.empty(n, device="cuda")
    ws = torch.empty(max(int(L.pv_linear_workspace_bytes(m, k, n)), 256), dtype=torch.uint8, device="cuda")
    _abi.check(L.pv_linear_bwd(_abi.ptr(dyc), n, _abi.ptr(xc), k, _abi.ptr(wc), _abi.ptr(dx), k, None, None, 0, 0,
                               _abi.ptr(dw), _abi.ptr(db), m, k, n, _abi.ptr(ws), ws.numel(), _abi.current_stream()),
               "pv_linear_bwd")
    assert rel_l2(dx, xd.grad) < 2e-6
    assert rel_l2(dw, wd_.grad) < 2e-6
    assert rel_l2(db, bd.grad) < 2e-6


@pytest.mark.parametrize("act", ["tanh", "lrelu", "gelu", "softplus"])
def test_operator_boundary_nets_autograd(gpu_device, act):
    """SURVEY §8b operator boundary: nets.*.forward on GPU tensors = the library's Linear(+activation) operator,
    differentiable through torch.autograd (pv_linear_fwd / pv_linear_bwd).  Values and every parameter / input
    gradient vs the oracle's forward under torch autograd on the CPU.  Tolerance 1e-4 relative (fp32 GEMMs)."""
    torch.manual_seed(4)
    cfg = orc.Config((12, 12), 2, ['r', 't'], activation=act)
    m = pv.models.iVAE((12, 12), 2, ['r', 't'], activation=act, seed=3, device="cuda")
    P = {k: v.detach().cpu().clone().requires_grad_() for k, v in m.state_dict().items()}
    b = 6
    x = torch.rand(b, 12, 12)
    # encoder
    mu, sig = m.encoder_z(x.cuda())
    mu_r, sig_r = orc.encoder_forward(P, cfg, x)
    assert rel_l2(mu, mu_r.detach()) < 1e-5 and rel_l2(sig, sig_r.detach()) < 1e-5
    w1, w2 = torch.randn(b, cfg.z_dim), torch.randn(b, cfg.z_dim)
    ((mu * w1.cuda()).sum() + (sig * w2.cuda()).sum()).backward()
    ((mu_r * w1).sum() + (sig_r * w2).sum()).backward()
    for k, v in m.named_parameters():
        if k.startswith("encoder_z."):
            assert rel_l2(v.grad, P[k].grad) < 1e-4, k
    # spatial decoder on explicit coordinates, input gradients too
    grid = orc.generate_grid((12, 12))
    xc = orc.transform_coordinates(grid.expand(b, *grid.shape), torch.randn(b), 0.1 * torch.randn(b, 1, 2),
                                   1 + 0.1 * torch.randn(b))
    z = torch.randn(b, 2)
    xc_g, z_g = xc.cuda().requires_grad_(), z.cuda().requires_grad_()
    xc_r, z_r = xc.clone().requires_grad_(), z.clone().requires_grad_()
    out = m.decoder(xc_g, z_g)
    out_r = orc.sdecoder_forward(P, cfg, xc_r, z_r)
    assert out.shape == (b, 12, 12) and rel_l2(out, out_r.detach()) < 1e-5
    w = torch.randn(b, 12, 12)
    (out * w.cuda()).sum().backward()
    (out_r * w).sum().backward()
    for k, v in m.named_parameters():
        if k.startswith("decoder."):
            assert rel_l2(v.grad, P[k].grad) < 1e-4, k
    assert rel_l2(xc_g.grad, xc_r.grad) < 1e-4 and rel_l2(z_g.grad, z_r.grad) < 1e-4
    # the unflat=False form returns (B*N, 1) like the reference (fc.py:196-199)
    d2 = pv.nets.sDecoderNet((12, 12), 2, unflat=False).cuda()
    assert d2(xc.cuda(), z.cuda()).shape == (b * 144, 1)
    # vanilla decoder + classifier / regressor heads
    cfg0 = orc.Config((12, 12), 2, None, activation=act)
    m0 = pv.models.iVAE((12, 12), 2, None, activation=act, seed=5, device="cuda")
    P0 = {k: v.detach().cpu().clone().requires_grad_() for k, v in m0.state_dict().items()}
    o0 = m0.decoder(z.cuda())
    o0_r = orc.fcdecoder_forward(P0, cfg0, z)
    assert rel_l2(o0, o0_r.detach()) < 1e-5
    (o0 * w.cuda()).sum().backward()
    (o0_r * w).sum().backward()
    for k, v in m0.named_parameters():
        if k.startswith("decoder."):
            assert rel_l2(v.grad, P0[k].grad) < 1e-4, k
    for net, task in ((pv.nets.fcClassifierNet((12, 12), 3, activation=act), "classification"),
                      (pv.nets.fcRegressorNet((12, 12), 3, activation=act), "regression")):
        net = net.cuda()
        Pn = {"encoder_y." + k: v.detach().cpu().clone().requires_grad_() for k, v in net.state_dict().items()}
        o = net(x.cuda())
        o_r = orc.label_net_forward(Pn, cfg, x, task)
        assert rel_l2(o, o_r.detach()) < 1e-5
        wy = torch.randn(b, 3)
        (o * wy.cuda()).sum().backward()
        (o_r * wy).sum().backward()
        for k, v in net.named_parameters():
            assert rel_l2(v.grad, Pn["encoder_y." + k].grad) < 1e-4, k


def test_transform_coordinates(gpu_device):
    g = torch.Generator().manual_seed(3)
    for data_dim in [(8, 8), (28, 28), (7, 9), (16,)]:
        grid = orc.generate_grid(data_dim)
        b = 5
        phi = torch.randn(b, generator=g)
        sc = 1 + 0.1 * torch.randn(b, generator=g)
        dx = 0.1 * torch.randn(b, 1, grid.shape[1], generator=g)
        ref = orc.transform_coordinates(grid.expand(b, *grid.shape), phi, dx, sc)
        out = pv.utils.transform_coordinates(grid.cuda().expand(b, *grid.shape), phi.cuda(), dx.cuda(), sc.cuda())
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("fused", FUSED)
@pytest.mark.parametrize("name", FWD_CASES)
def test_forward_vs_golden(gpu_device, name, fused):
    """ELBO terms and decoder output (`loc` = reconstructions) of one forward pass."""
    gold = load_golden(name)
    meta = meta_of(gold)
    model, cfg, eng = build(meta, fused)
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"]).cuda()
    eps = torch.from_numpy(gold["eps"]).cuda()
    b = meta["batch"]
    loc = torch.empty(b, cfg.n_pix, device="cuda")
    zl, zs = torch.empty(b, cfg.z_dim, device="cuda"), torch.empty(b, cfg.z_dim, device="cuda")
    eng.loss_and_grads(x, eps, 1.0, want_grads=False, z_out=(zl, zs), loc_out=loc)
    s = eng.scalars.cpu().numpy()
    np.testing.assert_allclose(s[0], float(gold["loss"]), rtol=RTOL_ELBO)
    np.testing.assert_allclose(s[1], float(gold["term.model.obs"]), rtol=RTOL_ELBO)
    np.testing.assert_allclose(s[2], float(gold["term.model.latent"]), rtol=RTOL_ELBO)
    np.testing.assert_allclose(s[3], float(gold["term.guide.latent"]), rtol=RTOL_ELBO)
    np.testing.assert_allclose(loc.cpu().numpy().reshape(gold["loc"].shape), gold["loc"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("fused", FUSED)
@pytest.mark.parametrize("name", STEP_CASES)
def test_steps_vs_golden_and_oracle(gpu_device, name, fused):
    """k SVI steps: loss/terms/z per step vs the golden fixture, gradients per step vs both the
    fixture's digests and the fp32 CPU oracle, parameters after every Adam update vs the fixture;
    then encode()/decode() after training."""
    gold = load_golden(name)
    meta = meta_of(gold)
    if meta["batch"] > 64:
        torch.set_num_threads(8)
    model, cfg, eng = build(meta, fused)
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    xg = x.cuda()
    b = meta["batch"]
    zl, zs = torch.empty(b, cfg.z_dim, device="cuda"), torch.empty(b, cfg.z_dim, device="cuda")
    for k in range(meta["steps"]):
        pre = "s%d" % k
        eps = torch.from_numpy(gold[pre + ".eps"])
        eng.loss_and_grads(xg, eps.cuda(), meta["beta"], z_out=(zl, zs))
        s = eng.scalars.cpu().numpy()
        np.testing.assert_allclose(s[0], float(gold[pre + ".loss"]), rtol=RTOL_ELBO, err_msg="loss")
        np.testing.assert_allclose(s[1], float(gold[pre + ".term.model.obs"]), rtol=RTOL_ELBO)
        np.testing.assert_allclose(s[2], float(gold[pre + ".term.model.latent"]), rtol=1e-4)
        np.testing.assert_allclose(s[3], float(gold[pre + ".term.guide.latent"]), rtol=1e-4)
        np.testing.assert_allclose(zl.cpu().numpy(), gold[pre + ".z_loc"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(zs.cpu().numpy(), gold[pre + ".z_scale"], rtol=1e-4, atol=2e-6)
        o.step(x, eps, meta["beta"])
        for key in o.p:
            g = eng.grad_of(key)
            err = rel_l2(g, o.last_grads[key])
            assert err < RTOL_GRAD, "step %d grad %s: rel l2 error %.3e vs oracle" % (k, key, err)
            check_digest(g, gold, pre + ".grad." + key, rtol=5e-4, atol=1e-6, what=name)
        eng.adam_step()
        lr = 1e-3
        for key, p in model.state_dict().items():
            check_digest(p, gold, pre + ".param." + key, rtol=1e-4, atol=2e-6, what=name)
            # Adam divides by |g|: an entry whose gradient is within the summation-order noise of zero
            # (|g| < 1e-5 max|g|) takes a step of up to lr in a direction that noise decides.  Such entries
            # are held to Adam's own bound (|dp| <= 2 lr), all others to 1e-4 (split precision) / 5e-5.
            gref = o.last_grads[key]
            ill = (gref.abs() < 1e-5 * gref.abs().max()).reshape(p.shape)
            pc, pr = p.detach().cpu(), o.p[key].detach()
            assert ill.float().mean() < 0.01, "%s: %d near-zero gradient entries" % (key, int(ill.sum()))
            assert (pc - pr)[ill].abs().max().item() <= 2 * lr if ill.any() else True
            assert rel_l2(pc[~ill], pr[~ill]) < (1e-4 if fused == 2 else 5e-5), key
        # continue from the oracle's parameters, so that every step is compared from an identical state
        # (otherwise one noise-decided entry above shifts all later gradients by ~1e-4)
        model.load_state_dict({k_: v_.detach() for k_, v_ in o.p.items()})
    z_loc, z_scale = model.encode(x)
    np.testing.assert_allclose(z_loc.numpy(), gold["enc.z_loc"], rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(z_scale.numpy(), gold["enc.z_scale"], rtol=1e-4, atol=5e-6)
    dec = model.decode(torch.from_numpy(gold["enc.z_loc"])[:, -meta["latent_dim"]:])
    np.testing.assert_allclose(dec.numpy(), gold["dec.loc"], rtol=1e-4, atol=2e-6)
    if "dec.loc_ats" in gold:
        dec2 = model.decode(torch.from_numpy(gold["enc.z_loc"])[:2, -meta["latent_dim"]:], angle=torch.tensor(0.3),
                            shift=torch.tensor([0.1, -0.2]), scale=torch.tensor(1.2))
        np.testing.assert_allclose(dec2.numpy(), gold["dec.loc_ats"], rtol=1e-4, atol=2e-6)


# model variants the reference's constructor allows (models/ivae.py:122-135), each against the CPU oracle on the same
# seeded inputs: class-conditioning (c_dim), Gaussian likelihood with / without the output sigmoid, other activations,
# other hidden widths (these leave the specialised kernels for the generic layer path), prior scales, 1-D data
VARIANTS = {
    "cdim3_rt": dict(data_dim=(8, 8), invariances=["r", "t"], c_dim=3),
    "cdim2_none": dict(data_dim=(8, 8), invariances=None, c_dim=2),
    "gauss_rts": dict(data_dim=(8, 8), invariances=["r", "t", "s"], sampler_d="gaussian"),
    "gauss_nosig_r": dict(data_dim=(8, 8), invariances=["r"], sampler_d="gaussian", sigmoid_d=False),
    "gauss_sig02_t": dict(data_dim=(8, 8), invariances=["t"], sampler_d="gaussian", decoder_sig=0.2),
    "gauss_nosig_none": dict(data_dim=(8, 8), invariances=None, sampler_d="gaussian", sigmoid_d=False),
    "gelu_r": dict(data_dim=(8, 8), invariances=["r"], activation="gelu"),
    "cbern_rts": dict(data_dim=(8, 8), invariances=["r", "t", "s"], sampler_d="continuous_bernoulli"),
    "cbern_none": dict(data_dim=(8, 8), invariances=None, sampler_d="continuous_bernoulli"),
    "cbern_16x16_r": dict(data_dim=(16, 16), invariances=["r"], sampler_d="continuous_bernoulli"),
    "relu_rt": dict(data_dim=(8, 8), invariances=["r", "t"], activation="relu"),
    "softplus_s": dict(data_dim=(8, 8), invariances=["s"], activation="softplus"),
    "lrelu_none": dict(data_dim=(8, 8), invariances=None, activation="lrelu"),
    "hid64_rt": dict(data_dim=(8, 8), invariances=["r", "t"], hidden_dim_e=[64, 64], hidden_dim_d=[64, 64]),
    "hid3layers_r": dict(data_dim=(8, 8), invariances=["r"], hidden_dim_e=[128, 64, 32], hidden_dim_d=[32, 48, 16]),
    "priors_rts": dict(data_dim=(8, 8), invariances=["r", "t", "s"], dx_prior=0.3, dy_prior=0.05, sc_prior=0.25),
    "latent5_rt": dict(data_dim=(16, 16), invariances=["r", "t"], latent_dim=5),
    "1d32_t_cdim2": dict(data_dim=(32,), invariances=["t"], c_dim=2),
    "rect_12x20_rts": dict(data_dim=(12, 20), invariances=["r", "t", "s"]),
}


@pytest.mark.parametrize("fused", [0, 1, 2])
@pytest.mark.parametrize("vname", sorted(VARIANTS))
def test_model_variants_vs_oracle(gpu_device, vname, fused):
    kw = dict(VARIANTS[vname])
    data_dim = kw.pop("data_dim")
    inv = kw.pop("invariances")
    latent_dim = kw.pop("latent_dim", 2)
    model = pv.models.iVAE(data_dim, latent_dim, inv, seed=3, device="cuda", **kw)
    he, hd = kw.get("hidden_dim_e") or [128, 128], kw.get("hidden_dim_d") or [128, 128]
    cfg = orc.Config(data_dim=data_dim, latent_dim=latent_dim, invariances=inv, c_dim=kw.get("c_dim", 0),
                     n_hidden_e=len(he), n_hidden_d=len(hd), activation=kw.get("activation", "tanh"),
                     sampler=kw.get("sampler_d", "bernoulli"), sigmoid_d=kw.get("sigmoid_d", True),
                     dx_prior=kw.get("dx_prior", 0.1), dy_prior=kw.get("dy_prior"), sc_prior=kw.get("sc_prior", 0.1),
                     decoder_sig=kw.get("decoder_sig", 0.5))
    eng = model.engine(fused=fused)
    # ContinuousBernoulli: torch's closed form of d log C(p)/dp cancels catastrophically near p = 1/2 (where every
    # pixel of a fresh model sits), so the fp32 reference gradient itself carries ~1e-3 noise; the HIP path uses a
    # series there and is judged against the fp64 evaluation of the reference's formula
    odt = torch.float64 if kw.get("sampler_d") == "continuous_bernoulli" else torch.float32
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg, dtype=odt)
    b = 7
    g = torch.Generator().manual_seed(11)
    x = torch.rand(b, *data_dim, generator=g)
    y = None
    if cfg.c_dim:
        y = torch.zeros(b, cfg.c_dim)
        y[torch.arange(b), torch.randint(0, cfg.c_dim, (b,), generator=g)] = 1.0
    beta = 1.7
    for k in range(2):
        eps = torch.randn(b, cfg.z_dim, generator=g)
        eng.loss_and_grads(x.cuda(), eps.cuda(), beta, None if y is None else y.cuda())
        s = eng.scalars.cpu().numpy()
        o.step(x, eps, beta, y)
        # (ContinuousBernoulli: per-pixel log-densities are ~ -log 2 + log 2 near p = 1/2: the summed loss is a few
        #  units made of B*N terms of size 0.7 each good to an fp32 ulp -> absolute bar of 1e-6 per term)
        atol = 1e-6 * b * int(np.prod(data_dim)) if odt == torch.float64 else 0.0
        np.testing.assert_allclose(s[0], o.last["loss"].item(), rtol=RTOL_ELBO, atol=atol, err_msg="%s loss" % vname)
        for key in o.p:
            err = rel_l2(eng.grad_of(key), o.last_grads[key])
            assert err < RTOL_GRAD, "%s step %d grad %s: rel l2 error %.3e" % (vname, k, key, err)
        eng.adam_step()
        model.load_state_dict({k_: v_.detach().float() for k_, v_ in o.p.items()})     # identical state for the next step
    # inference API (models/ivae.py:230-275) with the conditioning vector
    args = (x,) if y is None else (x, y)
    z_loc, z_scale = model.encode(*args)
    zl, zs = o.encode(x, y)
    np.testing.assert_allclose(z_loc.numpy(), zl.numpy(), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(z_scale.numpy(), zs.numpy(), rtol=1e-4, atol=5e-6)
    zc = zl[:, -latent_dim:]
    dec = model.decode(zc) if y is None else model.decode(zc, y)
    np.testing.assert_allclose(dec.numpy(), o.decode(zc, y).numpy(), rtol=1e-4, atol=2e-6)


VARIANT_GOLD = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "ivaevar_*.npz")))


@pytest.mark.parametrize("fused", [0, 1, 2])
@pytest.mark.parametrize("name", VARIANT_GOLD)
def test_model_variants_vs_golden(gpu_device, name, fused):
    """The same constructor variants / likelihoods against fixtures produced by the REFERENCE's own code
    (tests/golden/make_golden.py variants; models/ivae.py:122-163, utils/prob.py:25-29): loss and the three ELBO terms,
    z, every gradient and every parameter after Adam for each recorded step, then encode / decode.
    ContinuousBernoulli variants (ivaevar_cbern_*): ONE recorded step only — the reference's own fp32 gradients carry ~1e-3
    of cancellation noise there, so Adam trajectories cannot be re-joined from digests; the likelihood itself is pinned
    against float64 in test_continuous_bernoulli_normaliser_vs_float64."""
    from conftest import variant_of, variant_inputs
    gold = load_golden(name)
    meta, kw, cfg_kw = variant_of(gold)
    model = pv.models.iVAE(meta["data_dim"], meta["latent_dim"], meta["invariances"], c_dim=meta["c_dim"], seed=1,
                           device="cuda", **kw)
    eng = model.engine(fused=fused)
    x, y = variant_inputs(meta)
    xg, yg = x.cuda(), (None if y is None else y.cuda())
    cb = cfg_kw["sampler"] == "continuous_bernoulli"
    b = meta["batch"]
    z_dim = model.z_dim
    zl, zs = torch.empty(b, z_dim, device="cuda"), torch.empty(b, z_dim, device="cuda")
    # ContinuousBernoulli: torch's closed form of log C(p) cancels catastrophically near p = 1/2 (where every pixel of a
    # fresh model sits), so the reference's OWN fp32 numbers carry ~1e-3 gradient noise and its loss is a small
    # remainder of B*N terms of size ~0.7: judged on the scale of the sum of terms / with a wider gradient bar; the
    # fp64 evaluation of the same formula is the tight check (test_model_variants_vs_oracle)
    atol = 1e-6 * x.numel() if cb else 0.0       # (1e-6 per term, as test_model_variants_vs_oracle)
    gbar = 5e-3 if cb else 5e-4
    # (ContinuousBernoulli: one step only — with ~1e-3 of noise in the reference's fp32 gradients Adam's first step,
    #  lr * sign(g), differs in the entries whose gradient is noise-sized, and step 1 starts from different parameters;
    #  the fixture holds digests, not the parameters, so the trajectories cannot be re-joined as the oracle tests do)
    for k in range(1 if cb else meta["steps"]):
        pre = "s%d" % k
        eps = torch.from_numpy(gold[pre + ".eps"]).cuda()
        eng.loss_and_grads(xg, eps, meta["beta"], yg, z_out=(zl, zs))
        s = eng.scalars.cpu().numpy()
        np.testing.assert_allclose(s[0], float(gold[pre + ".loss"]), rtol=RTOL_ELBO, atol=atol, err_msg="loss")
        np.testing.assert_allclose(s[1], float(gold[pre + ".term.model.obs"]), rtol=RTOL_ELBO, atol=atol)
        np.testing.assert_allclose(s[2], float(gold[pre + ".term.model.latent"]), rtol=1e-4)
        np.testing.assert_allclose(s[3], float(gold[pre + ".term.guide.latent"]), rtol=1e-4)
        np.testing.assert_allclose(zl.cpu().numpy(), gold[pre + ".z_loc"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(zs.cpu().numpy(), gold[pre + ".z_scale"], rtol=1e-4, atol=2e-6)
        for key in model.state_dict():
            check_digest(eng.grad_of(key), gold, pre + ".grad." + key, rtol=gbar, atol=1e-6, what=name)
        eng.adam_step()
        for key, p_ in model.state_dict().items():
            check_digest(p_, gold, pre + ".param." + key, rtol=1e-4, atol=2e-6, what=name, sum_slack=2e-3 * 8)
    if cb:
        return
    args = (xg,) if y is None else (xg, yg)
    z_loc, z_scale = model.encode(*[a.cpu() for a in args])
    np.testing.assert_allclose(z_loc.numpy(), gold["enc.z_loc"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(z_scale.numpy(), gold["enc.z_scale"], rtol=2e-4, atol=2e-5)
    zc = torch.from_numpy(gold["enc.z_loc"])[:, -meta["latent_dim"]:]
    dec = model.decode(zc) if y is None else model.decode(zc, y)
    np.testing.assert_allclose(dec.numpy().reshape(gold["dec.loc"].shape), gold["dec.loc"], rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("fused", [1, 2])
@pytest.mark.parametrize("name", JSTEP_CASES)
def test_jivae_steps_vs_golden_and_oracle(gpu_device, name, fused):
    """models.jiVAE through the HIP path (K*B decoder rows in the fused kernels, alpha-weighted) vs the reference's
    recorded SVItrainer(enumerate_parallel=True) steps and the CPU oracle: loss, site terms, class probabilities,
    gradients, parameters; then encode()/decode()."""
    gold = load_golden(name)
    meta = jmeta_of(gold)
    K = meta["discrete_dim"]
    model = pv.models.jiVAE(meta["data_dim"], meta["latent_dim"], K, meta["invariances"], seed=1, device="cuda")
    cfg = orc.Config(data_dim=meta["data_dim"], latent_dim=meta["latent_dim"], invariances=meta["invariances"],
                     discrete_dim=K)
    eng = model.engine(fused=fused)
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    xg = x.cuda()
    for k in range(meta["steps"]):
        pre = "s%d" % k
        eps = torch.from_numpy(gold[pre + ".eps"])
        eng.loss_and_grads(xg, eps.cuda(), meta["beta"])
        s = eng.scalars.cpu().numpy()
        np.testing.assert_allclose(s[0], float(gold[pre + ".loss"]), rtol=RTOL_ELBO, err_msg="loss")
        np.testing.assert_allclose(s[1], float(gold[pre + ".term.model.obs"]), rtol=RTOL_ELBO)
        np.testing.assert_allclose(s[2], float(gold[pre + ".term.model.latent_cont"]) +
                                   float(gold[pre + ".term.model.latent_disc"]), rtol=1e-4)
        np.testing.assert_allclose(s[3], float(gold[pre + ".term.guide.latent_cont"]) +
                                   float(gold[pre + ".term.guide.latent_disc"]), rtol=1e-4)
        o.step(x, eps, meta["beta"])
        for key in o.p:
            g = eng.grad_of(key)
            tol = jivae_grad_tol(key)
            if tol is None:
                # d(loss)/d(out.bias) = sum over the B*N pixels (and K classes, weights summing to 1) of alpha*(p - x):
                # |terms| sum to ~B*N/4 and cancel to ~1e-4 of that; fp32 summation is good to ~1e-6 of it
                bound = 1e-6 * meta["batch"] * int(np.prod(meta["data_dim"]))
                assert (g.cpu() - o.last_grads[key]).abs().max().item() < bound, key
                continue
            err = rel_l2(g, o.last_grads[key])
            assert err < tol, "step %d grad %s: rel l2 error %.3e vs oracle" % (k, key, err)
            check_digest(g, gold, pre + ".grad." + key, rtol=5 * tol, atol=tol * float(gold[pre + ".grad." + key + ".l2"]) / 4,
                         what=name)
        eng.adam_step()
        for key, p in model.state_dict().items():
            # Adam turns a noise-sized gradient entry into a +-lr step: tensors whose gradient carries the class-logit
            # cancellation noise (jivae_grad_tol) are held to a few such steps, the rest to the usual bar
            loose = (jivae_grad_tol(key) or 1.0) > 3e-4
            check_digest(p, gold, pre + ".param." + key, rtol=1e-4, atol=5e-3 if loose else 2e-6, what=name)
        model.load_state_dict({k_: v_.detach() for k_, v_ in o.p.items()})     # identical state for the next step
    # inference API: tight against the oracle on the same (its own) parameters; against the reference's recorded
    # outputs within the few-lr drift that the class-logit noise leaves in the parameters after Adam
    z_loc, z_scale, alpha = model.encode(x, logits=True)
    zl, zs, al = o.encode(x)
    np.testing.assert_allclose(z_loc.numpy(), zl.numpy(), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(z_scale.numpy(), zs.numpy(), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(alpha.numpy(), al.numpy(), rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(z_loc.numpy(), gold["enc.z_loc"], rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(alpha.numpy(), gold["enc.alpha"], rtol=5e-3, atol=1e-4)
    _, _, classes = model.encode(x)
    assert (classes.numpy() == al.argmax(1).numpy()).all()
    yy = pv.utils.to_onehot(torch.arange(meta["batch"]) % K, K)
    zc = torch.from_numpy(gold["enc.z_loc"])[:, -meta["latent_dim"]:]
    dec = model.decode(zc, yy)
    np.testing.assert_allclose(dec.numpy(), o.decode(zc, yy).numpy(), rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(dec.numpy(), gold["dec.loc"], rtol=2e-3, atol=2e-4)


JVARIANTS = {
    "lrelu_rt_k5": dict(data_dim=(8, 8), invariances=["r", "t"], K=5, activation="lrelu"),
    "hid32x48_rts_k3": dict(data_dim=(12, 20), invariances=["r", "t", "s"], K=3, hidden=[32, 48]),
    "hid3layers_t_k2_gauss": dict(data_dim=(16, 16), invariances=["t"], K=2, hidden=[128, 64, 32], activation="relu",
                                  sampler_d="gaussian"),
    "hid16_none_k3": dict(data_dim=(7, 9), invariances=None, K=3, hidden=[16]),
    "1d24_none_k5_softplus": dict(data_dim=(24,), invariances=None, K=5, activation="softplus"),
}


JSAMPLED_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "jsivae_*.npz")))


@pytest.mark.parametrize("name", JSAMPLED_CASES)
def test_jivae_sampled_class_steps_vs_golden(gpu_device, name):
    """jiVAE WITHOUT enumeration — SVItrainer's default enumerate_parallel=False (trainers/svi.py:66, 83-91;
    models/jivae.py:213-220) — through the C ABI (pv_ivae_plan.class_onehot) against the reference's recorded steps:
    loss, the five site terms folded into the 4 scalars, class probabilities, every gradient (incl. the score-function
    gradient of the class logits), parameters after Adam."""
    gold = load_golden(name)
    meta = jmeta_of(gold)
    model = pv.models.jiVAE(meta["data_dim"], meta["latent_dim"], meta["discrete_dim"], None, seed=1, device="cuda")
    eng = model.engine(fused=2)
    cfg = orc.Config(data_dim=meta["data_dim"], latent_dim=meta["latent_dim"], invariances=None,
                     discrete_dim=meta["discrete_dim"])
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    o.sampled_class = True
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    for k in range(meta["steps"]):
        pre = "s%d" % k
        eps, y = torch.from_numpy(gold[pre + ".eps"]), torch.from_numpy(gold[pre + ".y"])
        eng.loss_and_grads(x.cuda(), eps.cuda(), meta["beta"], class_onehot=y.cuda())
        s = eng.scalars.cpu().numpy()
        np.testing.assert_allclose(s[0], float(gold[pre + ".loss"]), rtol=RTOL_ELBO, err_msg="loss")
        np.testing.assert_allclose(s[1], float(gold[pre + ".term.model.obs"]), rtol=RTOL_ELBO)
        np.testing.assert_allclose(s[2], float(gold[pre + ".term.model.latent_cont"]) + float(gold[pre + ".term.model.latent_disc"]), rtol=1e-4)
        np.testing.assert_allclose(s[3], float(gold[pre + ".term.guide.latent_cont"]) + float(gold[pre + ".term.guide.latent_disc"]), rtol=1e-4)
        o.step(x, eps, meta["beta"], y)
        for key in o.p:
            tol = jivae_grad_tol(key) or 1e-3
            err = rel_l2(eng.grad_of(key), o.last_grads[key])
            assert err < tol, "step %d grad %s: rel l2 error %.3e vs oracle" % (k, key, err)
            check_digest(eng.grad_of(key), gold, pre + ".grad." + key, rtol=2 * tol, atol=1e-5, what=name)
        eng.adam_step()
        model.load_state_dict({k_: v_.detach() for k_, v_ in o.p.items()})


def test_jivae_default_trainer_trains_and_matches_oracle(gpu_device):
    """SVItrainer(jiVAE) with its defaults (the reference's own default path) runs: per-epoch losses against the CPU oracle
    drawing eps and the classes from the same seed (normal_ then multinomial per step, as the guide does); with
    invariances the constructor raises what the reference's first step would."""
    with pytest.raises(RuntimeError):
        pv.trainers.SVItrainer(pv.models.jiVAE((8, 8), 2, 3, ["r"], seed=1, device="cuda"), seed=1)
    data = make_x("rand", 24, (8, 8))
    loader = pv.utils.init_dataloader(data, batch_size=8, shuffle=False)
    model = pv.models.jiVAE((8, 8), 2, 3, None, seed=1, device="cuda")
    cfg = orc.Config(data_dim=(8, 8), latent_dim=2, invariances=None, discrete_dim=3)
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    o.sampled_class = True
    tr = pv.trainers.SVItrainer(model, seed=1)
    for _ in range(2):
        tr.step(loader)
    torch.manual_seed(1)
    ref = []
    for _ in range(2):
        iter(loader)                                    # (creating a DataLoader iterator draws its base seed)
        tot = 0.0
        for i in range(0, 24, 8):
            xb = data[i:i + 8]
            tot += o.step(xb, o.draw_eps(8), 1.0, None)
        ref.append(tot / 24)
    np.testing.assert_allclose(tr.loss_history["training_loss"], ref, rtol=1e-4)
    assert not np.isnan(tr.loss_history["training_loss"]).any()


@pytest.mark.parametrize("fused", [0, 2])
@pytest.mark.parametrize("vname", sorted(JVARIANTS))
def test_jivae_variants_vs_oracle(gpu_device, vname, fused):
    """jiVAE away from the default architecture (other widths / depths / activations / likelihoods): the enumerated
    ELBO on the layer-by-layer kernels with the generic encoder (K decoder passes [k][b], alpha-weighted) against the
    oracle: loss, every gradient tensor (class-logit cancellation: jivae_grad_tol)."""
    kw = dict(JVARIANTS[vname])
    dd, inv, K = kw.pop("data_dim"), kw.pop("invariances"), kw.pop("K")
    hid = kw.pop("hidden", None)
    model = pv.models.jiVAE(dd, 2, K, inv, hidden_dim_e=hid, hidden_dim_d=hid, seed=2, device="cuda", **kw)
    nh = len(hid) if hid else 2
    cfg = orc.Config(data_dim=dd, latent_dim=2, invariances=inv, discrete_dim=K, n_hidden_e=nh, n_hidden_d=nh,
                     activation=kw.get("activation", "tanh"), sampler=kw.get("sampler_d", "bernoulli"))
    eng = model.engine(fused=fused)
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    b = 6
    g = torch.Generator().manual_seed(4)
    x = torch.rand(b, *dd, generator=g)
    for k in range(2):
        eps = torch.randn(b, cfg.z_dim, generator=g)
        eng.loss_and_grads(x.cuda(), eps.cuda(), [1.5, 0.7])
        ref = o.step(x, eps, [1.5, 0.7])
        np.testing.assert_allclose(eng.scalars[0].item(), ref, rtol=RTOL_ELBO)
        for key in o.p:
            err = rel_l2(eng.grad_of(key), o.last_grads[key])
            tol = jivae_grad_tol(key)
            if tol is None:        # decoder.out.bias
                if o.last_grads[key].numel() == 1:      # (one scalar = a sum of K*B*N signed terms): absolute scale
                    assert abs(eng.grad_of(key).item() - o.last_grads[key].item()) < 1e-6 * K * b * cfg.n_pix + 1e-5, key
                    continue
                tol = 3e-4
            assert err < tol, "%s step %d grad %s: rel l2 %.3e" % (vname, k, key, err)
        eng.adam_step()
        model.load_state_dict({k_: v_.detach() for k_, v_ in o.p.items()})
    z_loc, z_scale, alpha = model.encode(x, logits=True)
    zl, zs, al = o.encode(x)
    np.testing.assert_allclose(z_loc.numpy(), zl.numpy(), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(alpha.numpy(), al.numpy(), rtol=1e-4, atol=5e-6)


def test_jivae_trainer_epochs_vs_oracle(gpu_device):
    """SVItrainer(jiVAE, enumerate_parallel=True).step(loader, scale_factor=[b0, b1]) against the oracle driven
    through the same DataLoader / eps stream (trainers/svi.py:139-162)."""
    data_dim, K = (8, 8), 3
    x = make_x("rand", 10, data_dim)
    losses = {}
    for which in ("gpu", "cpu"):
        model = pv.models.jiVAE(data_dim, 2, K, ["r", "t"], seed=1, device="cuda" if which == "gpu" else "cpu")
        loader = pv.utils.init_dataloader(x, batch_size=4)
        if which == "gpu":
            tr = pv.trainers.SVItrainer(model, enumerate_parallel=True, seed=1)
            for _ in range(2):
                tr.step(loader, scale_factor=[1.5, 2.5])
            losses[which] = tr.loss_history["training_loss"]
        else:
            cfg = orc.Config(data_dim=data_dim, latent_dim=2, invariances=["r", "t"], discrete_dim=K)
            pv.utils.set_deterministic_mode(1)
            o = orc.SVIOracle(model.state_dict(), cfg)
            losses[which] = [o.train_epoch(loader, [1.5, 2.5]) for _ in range(2)]
    np.testing.assert_allclose(losses["gpu"], losses["cpu"], rtol=1e-4)


VED_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "ved_*.npz")))


@pytest.mark.parametrize("name", VED_CASES)
def test_ved_steps_vs_golden_and_oracle(gpu_device, name):
    """models.VED through the HIP path (conv stacks as im2col + MFMA GEMMs, pooling / upsampling gathers) vs the
    reference's recorded SVI steps and the CPU oracle: loss, ELBO terms, z, gradients, parameters; encode / decode."""
    from test_oracle_golden import ved_case
    gold = load_golden(name)
    c = ved_case(gold)
    model = pv.models.VED(c["input_dim"], c["output_dim"], latent_dim=c["latent_dim"], seed=1, device="cuda", **c["kw"])
    cfg = orc.VedConfig(input_dim=c["input_dim"], output_dim=c["output_dim"], latent_dim=c["latent_dim"],
                        hidden_dim_e=c["kw"].get("hidden_dim_e"), hidden_dim_d=c["kw"].get("hidden_dim_d"),
                        activation=c["kw"].get("activation", "lrelu"))
    eng = model.engine()
    o = orc.VedOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    x, y = torch.from_numpy(gold["x"]), torch.from_numpy(gold["y"])
    b = x.shape[0]
    zl, zs = torch.empty(b, cfg.z_dim, device="cuda"), torch.empty(b, cfg.z_dim, device="cuda")
    for k in range(c["steps"]):
        pre = "s%d" % k
        eps = torch.from_numpy(gold[pre + ".eps"])
        eng.loss_and_grads(x.cuda(), eps.cuda(), c["beta"], y.cuda(), z_out=(zl, zs))
        s = eng.scalars.cpu().numpy()
        np.testing.assert_allclose(s[0], float(gold[pre + ".loss"]), rtol=RTOL_ELBO, err_msg="loss")
        np.testing.assert_allclose(s[1], float(gold[pre + ".term.model.obs"]), rtol=RTOL_ELBO)
        np.testing.assert_allclose(s[2], float(gold[pre + ".term.model.z"]), rtol=1e-4)
        np.testing.assert_allclose(s[3], float(gold[pre + ".term.guide.z"]), rtol=1e-4)
        np.testing.assert_allclose(zl.cpu().numpy(), gold[pre + ".z_loc"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(zs.cpu().numpy(), gold[pre + ".z_scale"], rtol=1e-4, atol=2e-6)
        o.step(x, y, eps, c["beta"])
        for key in o.p:
            g = eng.grad_of(key)
            err = rel_l2(g, o.last_grads[key])
            assert err < RTOL_GRAD, "step %d grad %s: rel l2 error %.3e vs oracle" % (k, key, err)
            check_digest(g, gold, pre + ".grad." + key, rtol=5e-4, atol=1e-6, what=name)
        eng.adam_step()
        for key, p in model.state_dict().items():
            gref = o.last_grads[key]
            ill = (gref.abs() < 1e-5 * gref.abs().max()).reshape(p.shape)        # see test_steps_vs_golden_and_oracle
            pc, pr = p.detach().cpu(), o.p[key].detach()
            # (dead relu / lrelu units give many exactly-zero gradients: no bound on how many entries are `ill`)
            assert not ill.any() or (pc - pr)[ill].abs().max().item() <= 2e-3, key
            assert rel_l2(pc[~ill], pr[~ill]) < 5e-5, key
        model.load_state_dict({k_: v_.detach() for k_, v_ in o.p.items()})
    z_loc, z_scale = model.encode(x)
    zlo, zso = o.encode(x)
    np.testing.assert_allclose(z_loc.numpy(), zlo.numpy(), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(z_scale.numpy(), zso.numpy(), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(z_loc.numpy(), gold["enc.z_loc"], rtol=2e-3, atol=2e-4)
    dec = model.decode(torch.from_numpy(gold["enc.z_loc"]))
    np.testing.assert_allclose(dec.numpy(), o.decode(torch.from_numpy(gold["enc.z_loc"])).numpy(), rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(dec.numpy(), gold["dec.loc"], rtol=2e-3, atol=2e-4)
    mu, sd = model.predict(x[:2])
    assert mu.shape == dec[:2].shape and sd.shape == mu.shape and torch.isfinite(mu).all()


def test_ved_spec2im_small_batch_keeps_the_three_product_backward(gpu_device):
    """conv mode 4 (the one-piece backward) is size-gated — it needs gradient sums of >= 16 384 pixels per
    convolution — and the gate must look at BOTH stacks.  spec2im (a 1-D encoder in front of a 2-D decoder, default channel
    counts) has no 2-D encoder convolution that could trip it, and at batch 4 the decoder's 4x4 / 8x8 / 16x16 layers sum a few
    hundred pixels: the default plan must run the three-product backward there, i.e. give the SAME BITS as the plan that asks
    for it (conv_x3), and hold the fp32-class gradient bar against the oracle."""
    def run(conv_x3):
        m = pv.models.VED((64,), (16, 16), latent_dim=2, seed=1, device="cuda")
        eng = m.engine(fused=2)
        eng.conv_x3 = conv_x3
        g = torch.Generator().manual_seed(11)
        x, y = torch.rand(4, 1, 64, generator=g), torch.rand(4, 1, 16, 16, generator=g)
        eps = torch.randn(4, m.z_dim, generator=g)
        eng.loss_and_grads(x.cuda(), eps.cuda(), 1.0, y.cuda())
        torch.cuda.synchronize()
        return m, eng, (x, y, eps)
    m0, e0, (x, y, eps) = run(False)
    m1, e1, _ = run(True)
    assert torch.equal(e0.grad[:e0.n_flat + 4], e1.grad[:e1.n_flat + 4]), "small-batch spec2im took the one-piece backward"
    cfg = orc.VedConfig(input_dim=(64,), output_dim=(16, 16), latent_dim=2)
    o = orc.VedOracle({k: v.cpu() for k, v in m0.state_dict().items()}, cfg)
    o.step(x, y, eps, 1.0)
    for key in o.p:
        err = rel_l2(e0.grad_of(key), o.last_grads[key])
        assert err < RTOL_GRAD, "grad %s: rel l2 error %.3e vs oracle" % (key, err)


def test_ved_bf16_mode_vs_oracle(gpu_device):
    """VED in the throughput precision (SVItrainer(precision="bf16") -> plan.conv_bf16 = 3, round 4: the 2-D kernel-3
    convolutions with a multiple of 32 input channels run forward, input gradient and weight gradient on the matrix cores
    with ONE fp16 piece per operand, one product): the full-size 64x64 -> 128 net at batch 4 against the fp32 oracle from
    identical parameters.  ELBO to 1e-4; gradients to 3e-2 — except the first encoder layer's weights, whose gradient is a sum
    with heavy cancellation (the three-product bf16 mode of rounds 2-3 left 7e-3 there) over only 4 samples: 3.1e-2 measured,
    bar 5e-2; at C5's batch 256 every tensor is inside 3e-2 (test_full_size_c5_ved[bf16])."""
    from test_oracle_golden import ved_case
    gold = load_golden("ved_64x64_to_128_b4")
    c = ved_case(gold)
    model = pv.models.VED(c["input_dim"], c["output_dim"], latent_dim=c["latent_dim"], seed=1, device="cuda", **c["kw"])
    cfg = orc.VedConfig(input_dim=c["input_dim"], output_dim=c["output_dim"], latent_dim=c["latent_dim"])
    eng = model.engine(fused=3)
    o = orc.VedOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    x, y = torch.from_numpy(gold["x"]), torch.from_numpy(gold["y"])
    worst = 0.0
    for k in range(c["steps"]):
        eps = torch.from_numpy(gold["s%d.eps" % k])
        eng.loss_and_grads(x.cuda(), eps.cuda(), c["beta"], y.cuda())
        ref = o.step(x, y, eps, c["beta"])
        np.testing.assert_allclose(eng.scalars[0].item(), ref, rtol=1e-4)
        for key in o.p:
            err = rel_l2(eng.grad_of(key), o.last_grads[key])
            worst = max(worst, err)
            bar = 5e-2 if key == "encoder_z.feature_extractor.layers.0.weight" else 3e-2
            assert err < bar, "step %d grad %s: rel l2 %.3e" % (k, key, err)
        eng.adam_step()
        model.load_state_dict({k_: v_.detach() for k_, v_ in o.p.items()})
    print("worst gradient rel l2 in the throughput conv mode: %.2e" % worst)


@pytest.mark.parametrize("in_dim, c0, b", [((12, 20), 3, 3), ((7, 9), 5, 1), ((16, 16), 64, 2), ((24,), 6, 5), ((40,), 32, 2),
                                           ((64, 64), 32, 9)])
def test_first_layer_wgrad_kernel_vs_oracle(gpu_device, in_dim, c0, b):
    """pv_conv3_wgrad_c1_kernel (the streaming weight gradient of a one-input-channel convolution) over odd image
    sizes, channel counts that are not powers of two, the 64-channel limit, 1-D data, fewer image lines than
    workgroups and the full 64x64 size: the first encoder layer's weight and bias gradients (and everything else)
    against the oracle.  Tolerance 1e-4 (fp32)."""
    import warnings
    warnings.filterwarnings("ignore")
    he, hd = [(c0,), (8, 8)], [(8, 8), (4,)]
    out_dim = (16,)
    torch.manual_seed(4)
    model = pv.models.VED(in_dim, out_dim, hidden_dim_e=he, hidden_dim_d=hd, activation="tanh", seed=2, device="cuda")
    cfg = orc.VedConfig(input_dim=in_dim, output_dim=out_dim, latent_dim=2, hidden_dim_e=he, hidden_dim_d=hd, activation="tanh")
    eng = model.engine()
    o = orc.VedOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    x, y, eps = torch.rand(b, 1, *in_dim), torch.rand(b, 1, *out_dim), torch.randn(b, 2)
    eng.loss_and_grads(x.cuda(), eps.cuda(), 1.0, y.cuda())
    ref = o.step(x, y, eps, 1.0)
    np.testing.assert_allclose(eng.scalars[0].item(), ref, rtol=2e-5)
    for key in o.p:
        assert rel_l2(eng.grad_of(key), o.last_grads[key]) < 1e-4, key


VEDBN_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "vedbn_*.npz")))


@pytest.mark.parametrize("name", VEDBN_CASES)
def test_vedbn_steps_vs_oracle(gpu_device, name):
    """VED(batchnorm=True) on the HIP path against the oracle from identical parameters AND running statistics at every
    step: training-mode steps (batch statistics, running estimates updated in the flat buffer), encode / decode in eval()
    mode, then a training step on the running statistics (the reference never switches back to train())."""
    from test_oracle_golden import vedbn_oracle
    gold = load_golden(name)
    c, model, cfg = vedbn_oracle(gold, "cuda")
    eng = model.engine()
    o = orc.VedOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    x, y = torch.from_numpy(gold["x"]), torch.from_numpy(gold["y"])

    def resync():
        sd = {k_: v_.detach() for k_, v_ in o.p.items()}
        sd.update(o.bufs)
        model.load_state_dict(sd)

    def one_step(eps, first):
        eng.loss_and_grads(x.cuda(), eps.cuda(), c["beta"], y.cuda())
        loss = eng.scalars[0].item()
        ref = o.step(x, y, eps, c["beta"])
        np.testing.assert_allclose(loss, ref, rtol=3e-5, err_msg="loss")
        if first:
            np.testing.assert_allclose(loss, float(gold["s0.loss"]), rtol=3e-5)
        for key in o.p:
            g, go = eng.grad_of(key), o.last_grads[key]
            if go.abs().max() < 1e-4 * max(v.abs().max() for v in o.last_grads.values()):
                assert (g.cpu() - go).abs().max() < 1e-5, key       # (the biases in front of a batch norm: ~0 gradient)
                continue
            assert rel_l2(g, go) < 2e-4, "grad %s: rel l2 %.3e" % (key, rel_l2(g, go))
        eng.adam_step()
        for k_, b_ in model.named_buffers():                        # running statistics, num_batches_tracked
            np.testing.assert_allclose(b_.detach().cpu().double().numpy(), o.bufs[k_].double().numpy(), rtol=2e-5, atol=1e-6,
                                       err_msg=k_)
        resync()

    for k in range(c["steps"]):
        one_step(torch.from_numpy(gold["s%d.eps" % k]), k == 0)
    assert model.training
    z_loc, z_scale = model.encode(x)
    assert not model.training                                        # VED.encode -> self.eval() (models/ved.py:178)
    zlo, zso = o.encode(x)
    np.testing.assert_allclose(z_loc.numpy(), zlo.numpy(), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(z_scale.numpy(), zso.numpy(), rtol=1e-4, atol=5e-6)
    dec = model.decode(zlo)
    np.testing.assert_allclose(dec.numpy(), o.decode(zlo).numpy(), rtol=1e-4, atol=2e-6)
    one_step(torch.from_numpy(gold["e0.eps"]), False)                # trains on the running statistics from here on


@pytest.mark.parametrize("fused", [0, 2])
def test_convenc_batchnorm_vs_oracle(gpu_device, fused):
    """iVAE.set_encoder(convEncoderNet(..., batchnorm=True)) against the oracle from identical parameters and running
    statistics: the batch-norm kernels inside the iVAE step (pv_plan.hip's conv-encoder branch); training mode
    throughout (the reference's iVAE never calls eval())."""
    dd, inv, hid, b = (16, 16), ["r", "t"], [(8,), (16, 16)], 6
    model = pv.models.iVAE(dd, 2, inv, seed=1, device="cuda")
    model.set_encoder(pv.nets.convEncoderNet(dd, latent_dim=model.z_dim, hidden_dim=hid, batchnorm=True))
    cfg = orc.Config(data_dim=dd, latent_dim=2, invariances=inv, conv_encoder=hid, conv_batchnorm=True)
    eng = model.engine(fused=fused)
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(b, *dd, generator=g)
    for k in range(3):
        eps = torch.randn(b, cfg.z_dim, generator=g)
        eng.loss_and_grads(x.cuda(), eps.cuda(), 1.0)
        ref = o.step(x, eps, 1.0)
        np.testing.assert_allclose(eng.scalars[0].item(), ref, rtol=3e-5)
        gmax = max(v.abs().max().item() for v in o.last_grads.values())
        for key in o.p:
            gq, go = eng.grad_of(key), o.last_grads[key]
            if go.abs().max().item() < 1e-4 * gmax:        # (conv biases in front of a batch norm: ~0)
                assert (gq.cpu() - go).abs().max().item() < 1e-5 * gmax, key
                continue
            if key == "decoder.out.bias":
                assert (gq.cpu() - go).abs().max().item() < 1e-6 * b * 256 + 1e-5
                continue
            assert rel_l2(gq, go) < 3e-4, "step %d grad %s: rel l2 %.3e" % (k, key, rel_l2(gq, go))
        eng.adam_step()
        for k_, b_ in model.named_buffers():
            if k_ in o.bufs:
                np.testing.assert_allclose(b_.detach().cpu().double().numpy(), o.bufs[k_].double().numpy(), rtol=2e-5, atol=1e-6,
                                           err_msg=k_)
        sd = {k_: v_.detach() for k_, v_ in o.p.items()}
        sd.update(o.bufs)
        model.load_state_dict(sd)


def test_convenc_batchnorm_runs(gpu_device):
    """iVAE.set_encoder(convEncoderNet(..., batchnorm=True)): the same batch-norm kernels inside the iVAE step (the
    arithmetic is pinned by test_vedbn_steps_vs_oracle): finite decreasing loss, running statistics move, state_dict
    keeps the reference's keys."""
    model = pv.models.iVAE((16, 16), 2, ["r", "t"], seed=1, device="cuda")
    model.set_encoder(pv.nets.convEncoderNet((16, 16), latent_dim=model.z_dim, hidden_dim=[(8,), (16, 16)], batchnorm=True))
    tr = pv.trainers.SVItrainer(model, seed=1)
    rm0 = model.encoder_z.feature_extractor.layers[2].running_mean.clone()
    loader = pv.utils.init_dataloader(make_x("rand", 64, (16, 16)), batch_size=16)
    for _ in range(3):
        tr.step(loader)
    h = tr.loss_history["training_loss"]
    assert all(np.isfinite(h)) and h[-1] < h[0]
    bn = model.encoder_z.feature_extractor.layers[2]
    assert not torch.equal(bn.running_mean, rm0) and int(bn.num_batches_tracked) == 12
    assert "encoder_z.feature_extractor.layers.2.running_var" in model.state_dict()
    z_loc, z_scale = model.encode(make_x("rand", 8, (16, 16)))
    assert torch.isfinite(z_loc).all() and (z_scale > 0).all()


def test_ved_trainer_epochs_vs_oracle(gpu_device):
    """SVItrainer(VED).step(loader of (x, y)) vs the oracle driven through the same DataLoader / eps stream."""
    small = dict(hidden_dim_e=[(8,), (16, 16)], hidden_dim_d=[(16, 16), (8,)])
    g = torch.Generator().manual_seed(3)
    x, y = torch.rand(9, 1, 16, 16, generator=g), torch.rand(9, 1, 32, generator=g)
    losses = {}
    for which in ("gpu", "cpu"):
        model = pv.models.VED((16, 16), (32,), seed=1, device="cuda" if which == "gpu" else "cpu", **small)
        loader = pv.utils.init_dataloader(x, y, batch_size=4)
        if which == "gpu":
            tr = pv.trainers.SVItrainer(model, seed=1)
            for _ in range(2):
                tr.step(loader, scale_factor=1.5)
            losses[which] = tr.loss_history["training_loss"]
        else:
            cfg = orc.VedConfig(input_dim=(16, 16), output_dim=(32,), **small)
            pv.utils.set_deterministic_mode(1)
            o = orc.VedOracle(model.state_dict(), cfg)
            hist = []
            for _ in range(2):
                tot = 0.0
                for xb, yb in loader:
                    tot += o.step(xb, yb, torch.empty(xb.shape[0], 2).normal_(), 1.5)
                hist.append(tot / len(loader.dataset))
            losses[which] = hist
    np.testing.assert_allclose(losses["gpu"], losses["cpu"], rtol=1e-4)


CONVENC_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "ivaeconv_*.npz")))


@pytest.mark.parametrize("fused", [0, 2])
@pytest.mark.parametrize("name", CONVENC_CASES)
def test_convenc_steps_vs_golden_and_oracle(gpu_device, name, fused):
    """iVAE with set_encoder(convEncoderNet) (BASELINE config 4 family): conv encoder ops + the spatial decoder
    kernels, vs the reference's recorded steps and the oracle."""
    from test_oracle_golden import convenc_model
    gold = load_golden(name)
    meta, model, cfg = convenc_model(gold, "cuda")
    eng = model.engine(fused=fused)
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    for k in range(meta["steps"]):
        pre = "s%d" % k
        eps = torch.from_numpy(gold[pre + ".eps"])
        eng.loss_and_grads(x.cuda(), eps.cuda(), meta["beta"])
        s = eng.scalars.cpu().numpy()
        np.testing.assert_allclose(s[0], float(gold[pre + ".loss"]), rtol=RTOL_ELBO, err_msg="loss")
        o.step(x, eps, meta["beta"])
        for key in o.p:
            if key == "decoder.out.bias":
                # one scalar = the sum of B*N signed terms (p - x): |terms| sum to ~B*N/4 and cancel; fp32 summation
                # is good to ~1e-6 of that (see test_jivae_steps_vs_golden_and_oracle)
                bound = 1e-6 * meta["batch"] * int(np.prod(meta["data_dim"]))
                assert (eng.grad_of(key).cpu() - o.last_grads[key]).abs().max().item() < bound
                continue
            err = rel_l2(eng.grad_of(key), o.last_grads[key])
            assert err < RTOL_GRAD, "step %d grad %s: rel l2 error %.3e vs oracle" % (k, key, err)
            check_digest(eng.grad_of(key), gold, pre + ".grad." + key, rtol=5e-4, atol=1e-6, what=name)
        eng.adam_step()
        model.load_state_dict({k_: v_.detach() for k_, v_ in o.p.items()})
    z_loc, z_scale = model.encode(x.unsqueeze(1))
    zlo, zso = o.encode(x)
    np.testing.assert_allclose(z_loc.numpy(), zlo.numpy(), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(z_scale.numpy(), zso.numpy(), rtol=1e-4, atol=5e-6)


class _UserEncoder(torch.nn.Module):
    """A user-defined encoder in the sense of iVAE.set_encoder (models/base.py:173-177): any module that maps the
    batch to (z_loc, z_scale).  Deliberately unlike the built-in ones (a 1x1-free conv + pooling + GELU MLP)."""
    def __init__(self, data_dim, z_dim):
        super().__init__()
        self.data_dim = data_dim
        self.conv = torch.nn.Conv2d(1, 3, 5, padding=2)
        self.fc = torch.nn.Linear(3 * (data_dim[0] // 2) * (data_dim[1] // 2), 24)
        self.mu = torch.nn.Linear(24, z_dim)
        self.sig = torch.nn.Linear(24, z_dim)

    def forward(self, x):
        h = torch.nn.functional.avg_pool2d(torch.nn.functional.gelu(self.conv(x.reshape(-1, 1, *self.data_dim))), 2)
        h = torch.tanh(self.fc(h.flatten(1)))
        return self.mu(h), torch.nn.functional.softplus(self.sig(h)) + 1e-3


@pytest.mark.parametrize("fused", [0, 2])
@pytest.mark.parametrize("inv", [["r", "t", "s"], None])
def test_user_defined_encoder(gpu_device, inv, fused):
    """iVAE.set_encoder(user module): the module runs in PyTorch on the device, the rest of the SVI step in the HIP
    library (plan.ext_head / ext_dhead); loss, decoder gradients, the module's own gradients and both Adam updates
    against the oracle with the same module on the CPU."""
    data_dim, b = (8, 8), 6
    torch.manual_seed(5)
    model = pv.models.iVAE(data_dim, 2, inv, seed=1, device="cuda")
    user = _UserEncoder(data_dim, model.z_dim)
    ref = _UserEncoder(data_dim, model.z_dim)
    ref.load_state_dict(user.state_dict())
    model.set_encoder(user)
    eng = model.engine(fused=fused)
    cfg = orc.Config(data_dim=data_dim, latent_dim=2, invariances=inv, custom_encoder=ref)
    dec_params = {k: v.cpu() for k, v in model.state_dict().items() if not k.startswith("encoder_z.")}
    o = orc.SVIOracle(dec_params, cfg)
    ref_opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(b, *data_dim, generator=g)
    for k in range(3):
        eps = torch.randn(b, model.z_dim, generator=g)
        eng.loss_and_grads(x.cuda(), eps.cuda(), 1.3)
        ref_opt.zero_grad()
        o.step(x, eps, 1.3)
        np.testing.assert_allclose(eng.scalars[0].item(), o.last["loss"].item(), rtol=RTOL_ELBO)
        for key in o.p:
            if key == "decoder.out.bias":
                continue
            assert rel_l2(eng.grad_of(key), o.last_grads[key]) < RTOL_GRAD, key
        for (n, pu), pr in zip(user.named_parameters(), ref.parameters()):
            assert rel_l2(pu.grad, pr.grad) < 2e-4, "encoder %s" % n
        eng.adam_step()
        ref_opt.step()
        for (n, pu), pr in zip(user.named_parameters(), ref.parameters()):
            assert rel_l2(pu.detach(), pr.detach()) < 1e-4, "encoder %s after Adam" % n
            assert pu.grad is not None and float(pu.grad.abs().sum()) == 0.0        # zero_grads semantics
        model.load_state_dict({**{k_: v_.detach() for k_, v_ in o.p.items()},
                               **{"encoder_z." + k_: v_ for k_, v_ in ref.state_dict().items()}})
    z_loc, z_scale = model.encode(x)
    with torch.no_grad():
        zl, zs = ref(x)
    np.testing.assert_allclose(z_loc.numpy(), zl.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(z_scale.numpy(), zs.numpy(), rtol=1e-4, atol=1e-5)
    # trainer epochs run through the same path
    tr = pv.trainers.SVItrainer(model, seed=1)
    tr.step(pv.utils.init_dataloader(x, batch_size=4))
    assert np.isfinite(tr.loss_history["training_loss"][0])


class _UserSpatialDecoder(torch.nn.Module):
    """A user-defined decoder in the sense of baseVAE.set_decoder (models/base.py:179-183) for an invariant model: any
    module mapping (x_coord_prime (B, N, 2), z (B, L)) to the image.  Deliberately unlike sDecoderNet (sinusoidal
    coordinate features, GELU, a multiplicative latent gate)."""
    def __init__(self, data_dim, latent_dim):
        super().__init__()
        self.data_dim = data_dim
        self.fx = torch.nn.Linear(4, 24)
        self.fz = torch.nn.Linear(latent_dim, 24)
        self.h = torch.nn.Linear(24, 16)
        self.o = torch.nn.Linear(16, 1)

    def forward(self, xc, z):
        feat = torch.cat([torch.sin(3.0 * xc), torch.cos(3.0 * xc)], -1)
        h = torch.nn.functional.gelu(self.fx(feat)) * torch.sigmoid(self.fz(z)).unsqueeze(1)
        return torch.sigmoid(self.o(torch.tanh(self.h(h)))).reshape(-1, *self.data_dim)


class _UserVanillaDecoder(torch.nn.Module):
    def __init__(self, data_dim, z_dim):
        super().__init__()
        self.data_dim = data_dim
        self.a = torch.nn.Linear(z_dim, 20)
        self.b = torch.nn.Linear(20, data_dim[0] * data_dim[1])

    def forward(self, z):
        return torch.sigmoid(self.b(torch.nn.functional.softplus(self.a(z)))).reshape(-1, *self.data_dim)


@pytest.mark.parametrize("inv", [["r", "t", "s"], ["t"], None])
def test_user_defined_decoder(gpu_device, inv):
    """iVAE.set_decoder(user module): the HIP library runs the guide half of the step (encoder, reparameterisation,
    sampled KL: pv_ivae_guide), the user's decoder + coordinate transform + likelihood run in PyTorch on the device, and
    pv_ivae_guide_backward carries d(-ll)/dz into the encoder.  Loss, encoder gradients, the module's gradients and both
    Adam updates against the oracle with the same module on the CPU; decode() with a fixed transform; a trainer epoch."""
    data_dim, b = (8, 8), 6
    torch.manual_seed(7)
    model = pv.models.iVAE(data_dim, 2, inv, seed=1, device="cuda")
    mk = (lambda: _UserSpatialDecoder(data_dim, 2)) if inv else (lambda: _UserVanillaDecoder(data_dim, 2))
    user, ref = mk(), mk()
    ref.load_state_dict(user.state_dict())
    model.set_decoder(user)
    eng = model.engine()
    cfg = orc.Config(data_dim=data_dim, latent_dim=2, invariances=inv, custom_decoder=ref)
    enc_params = {k: v.cpu() for k, v in model.state_dict().items() if k.startswith("encoder_z.")}
    o = orc.SVIOracle(enc_params, cfg)
    ref_opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(b, *data_dim, generator=g)
    for k in range(3):
        eps = torch.randn(b, model.z_dim, generator=g)
        eng.loss_and_grads(x.cuda(), eps.cuda(), 1.3)
        ref_opt.zero_grad()
        o.step(x, eps, 1.3)
        np.testing.assert_allclose(eng.scalars[0].item(), o.last["loss"].item(), rtol=RTOL_ELBO)
        np.testing.assert_allclose(eng.scalars[1].item(), o.last["ll"].item(), rtol=RTOL_ELBO)
        for key in o.p:
            assert rel_l2(eng.grad_of(key), o.last_grads[key]) < 2e-4, key
        for (n, pu), pr in zip(user.named_parameters(), ref.parameters()):
            assert rel_l2(pu.grad, pr.grad) < 2e-4, "decoder %s" % n
        eng.adam_step()
        ref_opt.step()
        for (n, pu), pr in zip(user.named_parameters(), ref.parameters()):
            assert rel_l2(pu.detach(), pr.detach()) < 1e-4, "decoder %s after Adam" % n
        model.load_state_dict({**{k_: v_.detach() for k_, v_ in o.p.items()},
                               **{"decoder." + k_: v_ for k_, v_ in ref.state_dict().items()}})
    z_loc, z_scale = model.encode(x)
    zl, zs = o.encode(x)
    np.testing.assert_allclose(z_loc.numpy(), zl.numpy(), rtol=1e-4, atol=5e-6)
    zc = zl[:, -2:]
    with torch.no_grad():
        if inv:
            grid = orc.generate_grid(data_dim)
            kw = dict(angle=0.3, shift=[0.1, -0.2], scale=1.2) if inv == ["r", "t", "s"] else dict(shift=[0.1, -0.2])
            gt = orc.transform_coordinates(grid.unsqueeze(0), torch.tensor([kw.get("angle", 0.0)]),
                                           torch.tensor(kw["shift"]).unsqueeze(0), torch.tensor([kw.get("scale", 1.0)]))
            want = ref(gt.expand(b, *grid.shape), zc)
            got = model.decode(zc, **{k_: torch.tensor(v_) for k_, v_ in kw.items()})
        else:
            want, got = ref(zc), model.decode(zc)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4, atol=2e-6)
    tr = pv.trainers.SVItrainer(model, seed=1)
    tr.step(pv.utils.init_dataloader(x, batch_size=4), pv.utils.init_dataloader(x, batch_size=3))
    assert np.isfinite(tr.loss_history["training_loss"][0]) and np.isfinite(tr.loss_history["test_loss"][0])


@pytest.mark.parametrize("inv", [["r", "t", "s"], None])
def test_user_defined_encoder_and_decoder(gpu_device, inv):
    """set_encoder AND set_decoder with user modules on the same model (models/base.py:173-183): both run in PyTorch on
    the device; the library keeps the middle of the step — reparameterisation, the sampled KL terms (pv_ivae_guide with
    plan.ext_head) and the gradient of the KL + likelihood w.r.t. (z_loc, z_scale) (pv_ivae_guide_backward ->
    plan.ext_dhead).  Loss terms, both modules' gradients and Adam updates vs the oracle with the same modules."""
    data_dim, b = (8, 8), 6
    torch.manual_seed(11)
    model = pv.models.iVAE(data_dim, 2, inv, seed=1, device="cuda")
    mk = (lambda: _UserSpatialDecoder(data_dim, 2)) if inv else (lambda: _UserVanillaDecoder(data_dim, 2))
    ue, re_ = _UserEncoder(data_dim, model.z_dim), _UserEncoder(data_dim, model.z_dim)
    ud, rd = mk(), mk()
    re_.load_state_dict(ue.state_dict())
    rd.load_state_dict(ud.state_dict())
    model.set_encoder(ue)
    model.set_decoder(ud)
    eng = model.engine()
    cfg = orc.Config(data_dim=data_dim, latent_dim=2, invariances=inv, custom_encoder=re_, custom_decoder=rd)
    o = orc.SVIOracle({}, cfg)
    ref_params = list(re_.parameters()) + list(rd.parameters())
    ref_opt = torch.optim.Adam(ref_params, lr=1e-3)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(b, *data_dim, generator=g)
    for k in range(3):
        eps = torch.randn(b, model.z_dim, generator=g)
        eng.loss_and_grads(x.cuda(), eps.cuda(), 0.8)
        ref_opt.zero_grad()
        o.step(x, eps, 0.8)
        for i, name in enumerate(("loss", "ll")):
            np.testing.assert_allclose(eng.scalars[i].item(), o.last[name].item(), rtol=RTOL_ELBO)
        for (n, pu), pr in zip(list(ue.named_parameters()) + list(ud.named_parameters()), ref_params):
            assert rel_l2(pu.grad, pr.grad) < 2e-4, n
        eng.adam_step()
        ref_opt.step()
        for (n, pu), pr in zip(list(ue.named_parameters()) + list(ud.named_parameters()), ref_params):
            assert rel_l2(pu.detach(), pr.detach()) < 1e-4, "%s after Adam" % n
        ue.load_state_dict(re_.state_dict())
        ud.load_state_dict(rd.state_dict())
    z_loc, _ = model.encode(x)
    with torch.no_grad():
        np.testing.assert_allclose(z_loc.numpy(), re_(x)[0].numpy(), rtol=1e-4, atol=1e-5)
    tr = pv.trainers.SVItrainer(model, seed=1)
    tr.step(pv.utils.init_dataloader(x, batch_size=4), pv.utils.init_dataloader(x, batch_size=3))
    assert np.isfinite(tr.loss_history["training_loss"][0]) and np.isfinite(tr.loss_history["test_loss"][0])


class _UserLabelNet(torch.nn.Module):
    """A user-defined label network (ssiVAE.set_classifier / ss_reg_iVAE.set_regressor, ssivae.py:236-240)."""
    def __init__(self, n_in, n_out, softmax):
        super().__init__()
        self.a, self.b, self.softmax = torch.nn.Linear(n_in, 12), torch.nn.Linear(12, n_out), softmax

    def forward(self, x):
        h = self.b(torch.nn.functional.silu(self.a(x.reshape(x.shape[0], -1))))
        return torch.softmax(h, -1) if self.softmax else h


@pytest.mark.parametrize("task", ["classification", "regression"])
def test_user_defined_label_network(gpu_device, task):
    """set_classifier / set_regressor(user module): the module runs in PyTorch (its output feeds the HIP steps, dloss/dout
    comes back through torch.autograd, its parameters take the same two Adam steps per call); everything else as before.
    Unlabeled and labeled calls against the oracle with the same module on the CPU."""
    dd, dim, b = (8, 8), 3 if task == "classification" else 2, 5
    cls = task == "classification"
    torch.manual_seed(9)
    ctor = pv.models.ssiVAE if cls else pv.models.ss_reg_iVAE
    model = ctor(dd, 2, dim, ["r", "t"], seed=1, device="cuda")
    user, ref = _UserLabelNet(64, dim, cls), _UserLabelNet(64, dim, cls)
    ref.load_state_dict(user.state_dict())
    (model.set_classifier if cls else model.set_regressor)(user)
    eng = model.engine(lr=5e-4)
    cfg = orc.Config(data_dim=dd, latent_dim=2, invariances=["r", "t"], c_dim=dim, custom_label_net=ref)
    o = orc.SSOracle({k: v.cpu() for k, v in model.state_dict().items() if not k.startswith("encoder_y.")}, cfg, task)
    ref_opt = torch.optim.Adam(ref.parameters(), lr=5e-4)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(b, 64, generator=g)
    ys = torch.eye(dim)[torch.randint(0, dim, (b,), generator=g)] if cls else torch.randn(b, dim, generator=g)
    for labeled in (False, True, False):
        y = ys if labeled else None
        eps = torch.randn((dim, b, model.z_dim) if (cls and not labeled) else (b, model.z_dim), generator=g)
        eps_y = torch.randn(b, dim, generator=g) if (not cls and not labeled) else None
        loss = eng.elbo_loss_and_grads(x.cuda(), eps.cuda(), None if y is None else y.cuda(),
                                       None if eps_y is None else eps_y.cuda(), 1.0)
        ref_opt.zero_grad(set_to_none=False)      # (pyro zero_grads: zero tensors, not None)
        out = orc.ss_elbo(o.p, cfg, task, x, eps, y, eps_y, 1.0, 0.5, o.grid)
        for v in o.p.values():
            v.grad = None
        out["loss"].backward()
        np.testing.assert_allclose(loss.item(), out["loss"].item(), rtol=2e-5)
        for key, v in o.p.items():
            assert rel_l2(eng.grad_of(key), v.grad) < 3e-4, key
        if not labeled:
            for (n, pu), pr in zip(user.named_parameters(), ref.parameters()):
                assert rel_l2(pu.grad, pr.grad) < (2e-3 if cls else 3e-4), "label net %s" % n
        eng.adam_step(); o.opt.step(); ref_opt.step()
        for v in o.p.values():
            v.grad = torch.zeros_like(v)
        if labeled:
            aux = eng.aux_loss_and_grads(x.cuda(), y.cuda(), 20.0)
            ref_opt.zero_grad(set_to_none=False)      # (pyro zero_grads: zero tensors, not None)
            la = orc.ss_aux_loss(o.p, cfg, task, x, y, 20.0, 0.5)
            la.backward()
            np.testing.assert_allclose(aux.item(), la.item(), rtol=2e-5)
            for (n, pu), pr in zip(user.named_parameters(), ref.parameters()):
                assert rel_l2(pu.grad, pr.grad) < 2e-4, "label net (aux) %s" % n
        else:
            for pr in ref.parameters():
                pr.grad = torch.zeros_like(pr)                    # zero_grads: a momentum-only step follows
        eng.adam_step(); o.opt.step(); ref_opt.step()
        for v in o.p.values():
            v.grad = torch.zeros_like(v)
        for (n, pu), pr in zip(user.named_parameters(), ref.parameters()):
            assert rel_l2(pu.detach(), pr.detach()) < 1e-4, "label net %s after Adam" % n
        model.load_state_dict({**{k_: v_.detach() for k_, v_ in o.p.items()},
                               **{"encoder_y." + k_: v_ for k_, v_ in ref.state_dict().items()}})
    pred = model.classifier(x) if cls else model.regressor(x)
    with torch.no_grad():
        want = ref(x).argmax(-1) if cls else ref(x)
    np.testing.assert_allclose(pred.numpy(), want.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", EPOCH_CASES)
def test_trainer_epochs_vs_golden(gpu_device, name):
    """The product's SVItrainer driven exactly like the reference's (same DataLoader, same seeds):
    loss_history of every epoch and the final weights vs the reference run."""
    gold = load_golden(name)
    inv = str(gold["meta.invariances"])
    data_dim = tuple(int(v) for v in gold["meta.data_dim"])
    train, test = torch.from_numpy(gold["train"]), torch.from_numpy(gold["test"])
    batch = int(gold["meta.batch"])
    train_loader = pv.utils.init_dataloader(train, batch_size=batch)
    test_loader = pv.utils.init_dataloader(test, batch_size=batch)
    model = pv.models.iVAE(data_dim, 2, list(inv) if inv else None, seed=1, device="cuda")
    trainer = pv.trainers.SVItrainer(model, seed=1)
    for _ in range(int(gold["meta.epochs"])):
        if int(gold["meta.with_test"]):
            trainer.step(train_loader, test_loader)
        else:
            trainer.step(train_loader)
    trainer.print_statistics()
    np.testing.assert_allclose(trainer.loss_history["training_loss"], gold["epochs.training_loss"], rtol=1e-4)
    np.testing.assert_allclose(trainer.loss_history["test_loss"], gold["epochs.test_loss"], rtol=1e-4)
    for key, p in model.state_dict().items():
        check_digest(p, gold, "final." + key, rtol=5e-4, atol=5e-6, what=name)
    assert trainer.current_epoch == int(gold["meta.epochs"])


@pytest.mark.parametrize("fused", FUSED + [3])
@pytest.mark.parametrize("cfgname", ["c2_28x28_rt_b256", "64x64_rts_b32"])
def test_full_size_properties(gpu_device, cfgname, fused):
    """Size-independent properties at BASELINE sizes: run-to-run bit reproducibility, and additivity
    of the ELBO and of its gradient over disjoint shards of the batch (the property data-parallel
    sharding relies on: loss and grads are plain sums over samples)."""
    if cfgname == "c2_28x28_rt_b256":
        data_dim, inv, b = (28, 28), ["r", "t"], 256
    else:
        data_dim, inv, b = (64, 64), ["r", "t", "s"], 32
    model = pv.models.iVAE(data_dim, 2, inv, seed=1, device="cuda")
    eng = model.engine(fused=fused)
    x = make_x("rand", b, data_dim).cuda()
    torch.manual_seed(1)
    eps = torch.empty(b, model.z_dim).normal_().cuda()
    eng.loss_and_grads(x, eps)
    g_full = eng.grad[:eng.n_flat].clone()
    s_full = eng.scalars.clone()
    eng.loss_and_grads(x, eps)
    assert torch.equal(g_full, eng.grad[:eng.n_flat]) and torch.equal(s_full, eng.scalars), "not reproducible"
    h = b // 2
    eng.loss_and_grads(x[:h], eps[:h])
    g0, s0 = eng.grad[:eng.n_flat].clone(), eng.scalars.clone()
    eng.loss_and_grads(x[h:], eps[h:])
    g1, s1 = eng.grad[:eng.n_flat].clone(), eng.scalars.clone()
    np.testing.assert_allclose((s0 + s1).cpu().numpy(), s_full.cpu().numpy(), rtol=2e-6)
    # (round 5: at batch == grid the full batch's guide rides in the decoder launch — fp32 matrix-vector products in another
    #  summation order than the encoder launch the half batches take; under bf16 decoder operands a last-bit change of z moves
    #  roundings: measured 2.9e-6 there, 2e-6 everywhere else)
    folds = fused == 3 and bool(_abi.lib().pv_ivae_guide_folds(C.byref(eng._plan(b))))
    # (round 6: the throughput kernel writes its per-workgroup partial sums of the two hidden matrices' gradients as bf16 pairs —
    #  half the record traffic, C2 0.1015 -> 0.0995 ms.  A partial is then rounded to 2^-9 before the cross-workgroup sum, so two
    #  shardings of a batch agree to ~1e-4 of the gradient instead of to fp32 summation order — inside the mode's own distance
    #  from the fp32 gradients (2e-4 .. 1.5e-2, test_bf16_mode_steps_vs_golden_and_oracle); the fp32-class paths keep 2e-6.)
    assert rel_l2(g0 + g1, g_full) < (BF16_SHARD_TOL if fused == 3 else 2e-6)
    assert torch.isfinite(g_full).all()


BF16_SHARD_TOL = 3e-4      # shard additivity of the throughput mode's gradients (packed bf16 partial records; see test_full_size_properties)


def _props(eng, call, n_samples, sl):
    """Bit reproducibility of loss / gradients and additivity over two batch shards, through `call(lo, hi)`."""
    call(0, n_samples)
    g_full, s_full = eng.grad[:eng.n_flat].clone(), eng.scalars.clone()
    call(0, n_samples)
    assert torch.equal(g_full, eng.grad[:eng.n_flat]) and torch.equal(s_full, eng.scalars), "not reproducible"
    h = n_samples // 2
    call(0, h)
    g0, s0 = eng.grad[:eng.n_flat].clone(), eng.scalars.clone()
    call(h, n_samples)
    g1, s1 = eng.grad[:eng.n_flat].clone(), eng.scalars.clone()
    np.testing.assert_allclose((s0 + s1).cpu().numpy(), s_full.cpu().numpy(), rtol=sl)
    assert rel_l2(g0 + g1, g_full) < sl
    assert torch.isfinite(g_full).all()
    return s_full.cpu().numpy()


def _grads_vs_oracle(eng, ref_grads, tol_of, what, abs_bound=None):
    """Every parameter tensor's gradient against the oracle's backward on the same inputs: relative L2 below tol_of(key)
    (None: judged on the absolute scale abs_bound).  Returns the worst (error, key) for the assertion messages."""
    worst = (0.0, None)
    for key, gref in ref_grads.items():
        if gref is None:
            continue
        g = eng.grad_of(key)
        tol = tol_of(key)
        if tol is None:
            assert (g.cpu() - gref).abs().max().item() < abs_bound, "%s: %s" % (what, key)
            continue
        err = rel_l2(g, gref)
        assert err < tol, "%s: grad %s rel l2 error %.3e (bar %.1e) vs the oracle's backward" % (what, key, err, tol)
        if err / tol > worst[0]:
            worst = (err / tol, key)
    return worst


@pytest.mark.parametrize("fused", [2, 3])
def test_full_size_c3_jivae(gpu_device, fused):
    """BASELINE config 3 at its own size: jiVAE K=10, 28x28 ['r'], batch 512 -> 4.0 M decoder rows (unit partition,
    slot arithmetic and workspace sizes differ from the toy fixtures'): reproducibility, shard additivity, and the
    enumerated ELBO and its five site terms against the CPU oracle on the same inputs (1e-4; the mixed-precision mode
    to its own ELBO bar)."""
    torch.set_num_threads(8)
    data_dim, inv, k_, b = (28, 28), ["r"], 10, 512
    model = pv.models.jiVAE(data_dim, 2, k_, inv, seed=1, device="cuda")
    eng = model.engine(fused=fused)
    x = make_x("rand", b, data_dim)
    torch.manual_seed(1)
    eps = torch.empty(b, model.z_dim).normal_()
    xg, eg = x.cuda(), eps.cuda()
    s = _props(eng, lambda lo, hi: eng.loss_and_grads(xg[lo:hi], eg[lo:hi]), b, BF16_SHARD_TOL if fused == 3 else 5e-6)
    cfg = orc.Config(data_dim=data_dim, latent_dim=2, invariances=inv, discrete_dim=k_)
    # (round 3) the oracle's BACKWARD at this size too (4.0 M decoder rows under autograd: ~12 GB, tens of seconds on 8
    # threads): every gradient tensor, not only the scalars — a size-dependent but deterministic and additive gradient bug
    # (split-K seams, slot arithmetic of the per-sample partial sums) would pass the properties above
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    out = o.loss_and_grads(x, eps)
    np.testing.assert_allclose(s[0], out["loss"].item(), rtol=1e-4 if fused == 3 else RTOL_ELBO)
    eng.loss_and_grads(xg, eg)
    ref = {k: v.grad for k, v in o.p.items()}
    tol = (lambda key: 3e-2) if fused == 3 else jivae_grad_tol
    _grads_vs_oracle(eng, ref, tol, "C3 jiVAE K=10 B=512 fused=%d" % fused, abs_bound=1e-6 * b * 784 * (3e4 if fused == 3 else 1))


#  the fp32-class conv paths pass `max(1e-4, 2 e32)` by 1.4x / 1.5x on the seeded draw
# (profiles/r04f_grad_margin.txt): a bar that thin is shown on three draws — the benchmark's own (data seed 0, noise seed 1),
# a second seed pair, and blob images (80 % exact zeros: other cancellation patterns in the first conv layer's sums)
FULL_SIZE_DRAWS = {"seed0": ("rand", 0, 1), "seed7": ("rand", 7, 11), "blobs": ("blobs", 3, 5)}
# What the three draws showed (profiles/r05b_grad_margin_{seed0,seed7,blobs}.txt): on the conv stack's tensors BOTH fp32-class
# implementations — the fp32 CPU oracle and the HIP path — sit 0.4e-4 .. 8e-4 from the float64 truth depending on the draw, in
# steps that appear at one layer and persist upstream: max-pool winners and leaky-ReLU signs that the two arithmetics decide
# differently for a handful of near-ties (the oracle's own e32 reaches 8.0e-4 on C4 / blobs; the HIP path's worst is 5.1e-4
# there, 4.8e-4 on C5 / blobs, 1.8e-4 where the oracle happens to draw 0.6e-4).  `2 e32` alone is therefore a lottery ticket,
# not a bar: the conv-stack tensors get the floor CONV_FLIP_FLOOR = the level of the reference precision's own deviations; every
# other tensor (heads, decoder) keeps max(1e-4, 2 e32).
CONV_FLIP_FLOOR = 5e-4


def full_size_tol(key, e32):
    floor = CONV_FLIP_FLOOR if ".feature_extractor." in key else RTOL_GRAD
    return max(floor, 2 * e32[key])


# ----  the conv bar that can see ARITHMETIC again: gradients under forced-equal decisions ----------------------
# The 5e-4 floor above is the end-to-end check; it cannot tell a 4e-4 arithmetic regression from a decision flip.  So the float64
# oracle is run a second time under the HIP forward's OWN decisions (oracle.ConvDecisions: the leaky-ReLU signs read off the
# activations the step's backward uses, the max-pool winners read off its winner bytes — pv_debug_*_conv_trace, a test hook outside
# include/): with equal decisions what is left between the two gradients is rounding, and every conv-stack tensor is held to
# max(1e-4, 2 e32m) — e32m = the fp32 CPU oracle under the same decisions against the float64 one.  The decisions that differ are
# counted as a quantity of their own: the HIP forward (three fp16 products per multiply-add, 3e-7 per convolution) must not flip
# more of them against float64 than the reference precision itself does on the same draw (a small allowance for counting noise).
def hip_conv_decisions(eng, batch, kind):
    """oracle.ConvDecisions of the encoder stack as the HIP step that just ran decided them."""
    import ctypes as C_
    from pyroved_amd import _abi
    fn = getattr(_abi.lib(), "pv_debug_%s_conv_trace" % kind)
    fn.restype, fn.argtypes = C_.c_int, [C_.c_void_p, C_.POINTER(C_.c_int64)]
    plan = eng._plan(batch)
    n_ops = plan.n_enc_ops
    ops = plan.enc_ops if kind == "ivae" else plan.enc
    out = (C_.c_int64 * (6 * 32))()
    assert fn(C_.byref(plan), out) == 0
    torch.cuda.synchronize()
    rows = [tuple(out[6 * i:6 * i + 6]) for i in range(n_ops)]

    def act(i):                                   # op i's output, (B, C, H, W) on the host
        off, h, w, c = rows[i][:4]
        n = batch * h * w * c
        return eng.ws[off:off + 4 * n].view(torch.float32).view(batch, h, w, c).permute(0, 3, 1, 2).cpu()
    sign, win, pool_of = [], [], {}
    li = 0
    for i in range(n_ops):
        kind_i = rows[i][5]
        if kind_i == 1:                           # PV_OP_CONV
            if rows[i][0] >= 0:
                sign.append(act(i) > 0)
            else:                                 # pooled in the convolution's epilogue: the winner's sign, from the pool's output
                assert i + 1 < n_ops and rows[i + 1][5] == 2 and rows[i + 1][0] >= 0
                sign.append(act(i + 1) > 0)
            li += 1
        elif kind_i == 2:                         # PV_OP_MAXPOOL2
            off, h, w, c = rows[i][4], rows[i][1], rows[i][2], rows[i][3]
            assert off >= 0, "max-pool %d keeps no winner bytes: not a stack the masked comparison covers" % i
            n = batch * h * w * c
            pool_of[li - 1] = len(win)
            win.append(eng.ws[off:off + n].view(batch, h, w, c).permute(0, 3, 1, 2).cpu().long())
            assert int(win[-1].max()) <= 3
    return orc.ConvDecisions(sign=sign, win=win, pool_of=pool_of)


def masked_conv_check(eng, kind, batch, grads_of, what, slack=None):
    """grads_of(dtype, decisions) -> {key: gradient} of the oracle.  Asserts the conv-stack gradients under the HIP forward's
    decisions and the flip counts; returns the numbers (also written to gpurun_out/ for profiles/)."""
    hip = hip_conv_decisions(eng, batch, kind)
    rec64, rec32 = orc.ConvDecisions(), orc.ConvDecisions()
    grads_of(torch.float64, rec64, False)           # (forward only: what each arithmetic decides for itself)
    grads_of(torch.float32, rec32, False)
    f_hip, f_32 = rec64.flips(hip), rec64.flips(rec32)
    ref = grads_of(torch.float64, orc.ConvDecisions(sign=hip.sign, win=hip.win, pool_of=hip.pool_of))
    g32 = grads_of(torch.float32, orc.ConvDecisions(sign=hip.sign, win=hip.win, pool_of=hip.pool_of))
    lines = ["%s: decisions that differ from the float64 oracle's — HIP forward: %d signs + %d winners; fp32 oracle: %d signs + %d winners"
             " (of %d / %d)" % (what, f_hip[0], f_hip[1], f_32[0], f_32[1], f_hip[2], f_hip[3])]
    worst, worst_key = 0.0, None
    for key in ref:
        if ".feature_extractor." not in key and "features2latent" not in key:
            continue
        e32m = rel_l2(g32[key], ref[key])
        err = rel_l2(eng.grad_of(key), ref[key].float())
        bar = max(RTOL_GRAD, 2 * e32m)
        lines.append("  %-52s HIP %.2e   fp32 oracle %.2e   bar %.1e" % (key, err, e32m, bar))
        if err / bar > worst:
            worst, worst_key = err / bar, key
    n_hip, n_32 = f_hip[0] + f_hip[1], f_32[0] + f_32[1]
    lines.append("  worst err / bar %.2f" % worst)
    os.makedirs(os.path.join(ROOT_DIR, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT_DIR, "gpurun_out", "grad_margin_masked.txt"), "a") as f:
        f.write("\n".join(lines) + "\n")
    assert worst < 1.0, "%s (equal decisions): grad %s at %.2f of its bar\n%s" % (what, worst_key, worst, "\n".join(lines))
    assert n_hip <= 2 * n_32 + 32, "%s: the HIP forward flips %d decisions against float64, the fp32 oracle %d" % (what, n_hip, n_32)
    return n_hip, n_32, worst


@pytest.mark.parametrize("fused,draw", [(0, "seed0"), (2, "seed0"), (2, "seed7"), (2, "blobs"), (20, "seed0"), (3, "seed0")])
def test_full_size_c4_conv_encoder(gpu_device, fused, draw):
    """BASELINE config 4 at its own shape and per-GPU batch: iVAE 64x64 ['r','t','s'] + set_encoder(convEncoderNet)
    with the default stack (nets/conv.py:24-64), batch 128 — properties + the ELBO terms vs the oracle."""
    torch.set_num_threads(8)
    data_dim, inv, b = (64, 64), ["r", "t", "s"], 128
    hid = [(32,), (64, 64), (128, 128)]
    xkind, xseed, eseed = FULL_SIZE_DRAWS[draw]
    model = pv.models.iVAE(data_dim, 2, inv, seed=1, device="cuda")
    model.set_encoder(pv.nets.convEncoderNet(data_dim, latent_dim=model.z_dim))
    # fused 20: the fp32-class path with the convolutions' BACKWARD at three products too (rounds 2-4's form, PV_PLAN_CONV_X3);
    # the default since round 5 keeps three products in the forward only (input gradient two, weight gradient one)
    conv_x3 = fused == 20
    fused = 2 if conv_x3 else fused
    eng = model.engine(fused=fused)
    eng.conv_x3 = conv_x3
    x = make_x(xkind, b, data_dim, seed=xseed)
    torch.manual_seed(eseed)
    eps = torch.empty(b, model.z_dim).normal_()
    xg, eg = x.cuda(), eps.cuda()
    # (shard additivity: the half batches are 262 k decoder rows — the H231 build of the fp32-class kernel — the full batch 524 k,
    #  from which the one-piece-activation H221 build runs since round 5: two roundings of the same sums, 8e-6 apart)
    s = _props(eng, lambda lo, hi: eng.loss_and_grads(xg[lo:hi], eg[lo:hi]), b, 2e-5 if fused == 2 else (BF16_SHARD_TOL if fused == 3 else 5e-6))
    cfg = orc.Config(data_dim=data_dim, latent_dim=2, invariances=inv, conv_encoder=hid)
    # (round 3) backward at size too.  At 0.5 M decoder rows / 0.5 M conv pixels per layer the fp32 CPU oracle is itself
    # 1e-4 .. 2e-4 off float64 on the encoder tensors (sums of ~1e6 cancelling terms): the float64 oracle is the truth here,
    # and a tensor's bar is 1e-4 or twice what the reference's own precision (the fp32 oracle) achieves, whichever is larger
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    o = orc.SVIOracle(sd, cfg)
    out = o.loss_and_grads(x, eps)
    o64 = orc.SVIOracle(sd, cfg, dtype=torch.float64)
    o64.loss_and_grads(x, eps)
    ref = {k: v.grad for k, v in o64.p.items()}
    e32 = {k: rel_l2(o.p[k].grad, ref[k]) for k in ref}
    eng.loss_and_grads(xg, eg)
    tol = (lambda key: 3e-2) if fused == 3 else (lambda key: full_size_tol(key, e32))
    _grads_vs_oracle(eng, {k: v.float() for k, v in ref.items()}, tol, "C4 conv-encoder iVAE 64x64 B=128 fused=%d %s" % (fused, draw))
    if fused == 2:
        # ... and under the HIP forward's own decisions, where the bar is max(1e-4, 2 e32m) again (both backward forms)
        def grads_of(dt, dec, grads=True):
            o_ = orc.SVIOracle(sd, dataclasses.replace(cfg, conv_decisions=dec), dtype=dt)
            with torch.set_grad_enabled(grads):
                o_.loss_and_grads(x, eps)
            return {k: v.grad for k, v in o_.p.items()}
        masked_conv_check(eng, "ivae", b, grads_of, "C4 conv-encoder iVAE 64x64 B=128 %s %s" % ("x3-3-3" if conv_x3 else "x3-3-1", draw))
    np.testing.assert_allclose(s[0], out["loss"].item(), rtol=1e-4 if fused == 3 else RTOL_ELBO)
    np.testing.assert_allclose(s[1], out["ll"].item(), rtol=1e-4 if fused == 3 else RTOL_ELBO)
    np.testing.assert_allclose(s[2], out["logpz"].item(), rtol=1e-4)
    np.testing.assert_allclose(s[3], out["logqz"].item(), rtol=1e-4)


@pytest.mark.parametrize("prec,draw", [("fp32", "seed0"), ("fp32", "seed7"), ("fp32", "blobs"), ("fp32x3", "seed0"), ("bf16", "seed0")])
def test_full_size_c5_ved(gpu_device, prec, draw):
    """BASELINE config 5 at its per-GPU size: VED 64x64 -> 128-point spectrum, batch 256 — properties + ELBO terms
    vs the oracle (three draws at the fp32-class precision: FULL_SIZE_DRAWS)."""
    torch.set_num_threads(8)
    b = 256
    xkind, xseed, eseed = FULL_SIZE_DRAWS[draw]
    model = pv.models.VED((64, 64), (128,), seed=1, device="cuda")
    eng = model.engine(fused=3 if prec == "bf16" else 2)
    eng.conv_x3 = prec == "fp32x3"       # (the convolutions' backward at three products too: rounds 2-4's fp32-class form)
    g = torch.Generator().manual_seed(xseed)
    x = torch.rand(b, 1, 64, 64, generator=g)
    if xkind == "blobs":
        x = (x > 0.8).float() * torch.rand(b, 1, 64, 64, generator=g)
    y = torch.rand(b, 1, 128, generator=g)
    torch.manual_seed(eseed)
    eps = torch.empty(b, model.z_dim).normal_()
    xg, yg, eg = x.cuda(), y.cuda(), eps.cuda()
    s = _props(eng, lambda lo, hi: eng.loss_and_grads(xg[lo:hi], eg[lo:hi], 1.0, yg[lo:hi]), b, 2e-5)
    cfg = orc.VedConfig(input_dim=(64, 64), output_dim=(128,), latent_dim=2)
    # (round 3) backward at size too, against the float64 oracle (see test_full_size_c4_conv_encoder)
    gr = {}
    for dt in (torch.float32, torch.float64):
        p_ = {k: v.detach().cpu().clone().to(dt).requires_grad_(True) for k, v in model.state_dict().items()}
        out = orc.ved_elbo(p_, cfg, x.to(dt), y.to(dt), eps.to(dt))
        out["loss"].backward()
        gr[dt] = {k: v.grad for k, v in p_.items()}
        if dt == torch.float32:
            np.testing.assert_allclose(s[0], out["loss"].item(), rtol=1e-4 if prec == "bf16" else RTOL_ELBO)
    ref = gr[torch.float64]
    e32 = {k: rel_l2(gr[torch.float32][k], ref[k]) for k in ref}
    eng.loss_and_grads(xg, eg, 1.0, yg)
    tol = (lambda key: 3e-2) if prec == "bf16" else (lambda key: full_size_tol(key, e32))
    _grads_vs_oracle(eng, {k: v.float() for k, v in ref.items()}, tol, "C5 VED 64x64->128 B=256 %s %s" % (prec, draw))
    if prec != "bf16":
        def grads_of(dt, dec, grads=True):
            p_ = {k: v.detach().cpu().clone().to(dt).requires_grad_(True) for k, v in model.state_dict().items()}
            with torch.set_grad_enabled(grads):
                out_ = orc.ved_elbo(p_, cfg, x.to(dt), y.to(dt), eps.to(dt), decisions=dec)
            if grads:
                out_["loss"].backward()
            return {k: v.grad for k, v in p_.items()}
        masked_conv_check(eng, "ved", b, grads_of, "C5 VED 64x64->128 B=256 %s %s" % (prec, draw))

BF16_CASES = ["ivae_28x28_rt_b256", "ivae_28x28_r_b128", "ivae_28x28_r_b32_blobs", "ivae_8x8_rts_b6", "ivae_8x8_r_b6",
              "ivae_1d16_t_b5", "ivae_8x8_rts_b6_randn", "ivae_8x8_rt_b6_beta4"]


@pytest.mark.parametrize("name", BF16_CASES)
def test_bf16_mode_steps_vs_golden_and_oracle(gpu_device, name):
    """The mixed-precision mode (fused=3 / SVItrainer(precision="bf16")): the two hidden-layer contractions of the
    spatial decoder take bf16 operands (fp32 accumulate), everything else is fp32.  Bars: the ELBO to BASELINE.json's
    1e-4 at the benchmark sizes (batch >= 128 at 28x28; rounding errors do not average out on a handful of 8x8
    samples: 5e-4 there), the encoder's outputs as in fp32, every gradient tensor to 3e-2 relative L2 of the fp32
    oracle's (measured: 2e-4 .. 1.5e-2) from identical parameters at every step."""
    gold = load_golden(name)
    meta = meta_of(gold)
    if meta["batch"] > 64:
        torch.set_num_threads(8)
    model, cfg, eng = build(meta, 3)
    assert eng.uses_fused(meta["batch"])
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    xg = x.cuda()
    b = meta["batch"]
    big = b >= 128
    zl, zs = torch.empty(b, cfg.z_dim, device="cuda"), torch.empty(b, cfg.z_dim, device="cuda")
    for k in range(meta["steps"]):
        pre = "s%d" % k
        eps = torch.from_numpy(gold[pre + ".eps"])
        eng.loss_and_grads(xg, eps.cuda(), meta["beta"], z_out=(zl, zs))
        s = eng.scalars.cpu().numpy()
        np.testing.assert_allclose(s[0], float(gold[pre + ".loss"]), rtol=1e-4 if big else 5e-4, err_msg="loss")
        np.testing.assert_allclose(s[1], float(gold[pre + ".term.model.obs"]), rtol=1e-4 if big else 5e-4)
        np.testing.assert_allclose(s[2], float(gold[pre + ".term.model.latent"]), rtol=1e-4)
        np.testing.assert_allclose(s[3], float(gold[pre + ".term.guide.latent"]), rtol=1e-4)
        np.testing.assert_allclose(zl.cpu().numpy(), gold[pre + ".z_loc"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(zs.cpu().numpy(), gold[pre + ".z_scale"], rtol=1e-4, atol=2e-6)
        o.step(x, eps, meta["beta"])
        for key in o.p:
            err = rel_l2(eng.grad_of(key), o.last_grads[key])
            assert err < 3e-2, "step %d grad %s: rel l2 error %.3e vs oracle" % (k, key, err)
        eng.adam_step()
        for key, p in model.state_dict().items():     # Adam's own bound: |dp| <= lr / (1 - beta1) early on
            assert (p.detach().cpu() - o.p[key].detach()).abs().max().item() <= 2.5e-3, key
        model.load_state_dict({k_: v_.detach() for k_, v_ in o.p.items()})


def test_bf16_mode_trainer_tracks_fp32(gpu_device):
    """SVItrainer(precision="bf16") against precision="fp32" from the same seeds: 6 epochs of 8 minibatches of 128
    28x28 images; the per-epoch ELBO stays within 1e-3 (it drifts as any mixed-precision run does: Adam turns a 1 %
    gradient difference into a slightly different trajectory)."""
    data = make_x("rand", 1024, (28, 28))
    hist = {}
    for prec in ("fp32", "bf16"):
        model = pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
        tr = pv.trainers.SVItrainer(model, seed=1, precision=prec)
        assert tr.engine.fused == (3 if prec == "bf16" else 2)
        loader = pv.utils.init_dataloader(data, batch_size=128)
        for _ in range(6):
            tr.step(loader)
        hist[prec] = np.array(tr.loss_history["training_loss"])
    assert np.all(np.diff(hist["bf16"]) < 0), hist["bf16"]
    np.testing.assert_allclose(hist["bf16"], hist["fp32"], rtol=1e-3)


# ---------------------------------------------------------------------------------------------------------------
# semi-supervised models through the C ABI: pv_ivae_loss_and_grads (row weights / per-row ELBO / dy), pv_mlp_*, pv_ss_*
SS_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "ss*_*.npz")))


def _ss_grad_tol(key, task):
    # the label network's gradient in the enumerated pass is a difference of K nearly equal per-class terms (as for
    # jiVAE's class logits): cancellation amplifies the summation-order noise of the decoder's ll sums
    if key.startswith("encoder_y."):
        return 2e-3 if task == "classification" else 3e-4
    return 3e-4


@pytest.mark.parametrize("fused", [0, 2])
@pytest.mark.parametrize("name", SS_CASES)
def test_ss_compute_loss_vs_golden_and_oracle(gpu_device, name, fused):
    """auxSVItrainer.compute_loss call by call (unlabeled, labeled, ...): both losses vs the fixture produced by the
    reference's own ssiVAE / ss_reg_iVAE / auxSVItrainer code, the gradients of both SVI steps vs the CPU oracle from
    identical parameters, the parameters after both Adam updates; then the inference API."""
    from conftest import ssmeta_of, ss_build
    gold = load_golden(name)
    meta = ssmeta_of(gold)
    model = ss_build(meta, "cuda")
    eng = model.engine(lr=5e-4, fused=fused)
    cfg = orc.Config(data_dim=meta["data_dim"], latent_dim=meta["latent_dim"], invariances=meta["invariances"],
                     c_dim=meta["dim"])
    o = orc.SSOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg, meta["task"])
    xu, xs, ys = (torch.from_numpy(gold[k]) for k in ("xu", "xs", "ys"))
    lr = 5e-4
    for c in range(meta["calls"]):
        pre = "c%d" % c
        unl = str(gold[pre + ".kind"]) == "u"
        x = xu[:meta["batch_u"]] if unl else xs[:meta["batch_s"]]
        y = None if unl else ys[:meta["batch_s"]]
        eps = torch.from_numpy(gold[pre + ".eps"])
        eps_y = torch.from_numpy(gold[pre + ".eps_y"]) if (pre + ".eps_y") in gold else None
        mid = {}
        l1o, l2o = o.compute_loss(x, y, eps, eps_y, meta["beta"], meta["mult"],
                                  after_elbo=lambda p_: mid.update({k_: v_.detach().clone() for k_, v_ in p_.items()}))
        # --- product: the ELBO step
        loss = eng.elbo_loss_and_grads(x.cuda(), eps.cuda(), None if y is None else y.cuda(),
                                       None if eps_y is None else eps_y.cuda(), meta["beta"])
        np.testing.assert_allclose(loss.item(), l1o, rtol=2e-5)
        if c == 0:
            np.testing.assert_allclose(loss.item(), float(gold[pre + ".elbo.loss"]), rtol=2e-5)
        for key in o.p:
            g, go = eng.grad_of(key), o.last_grads["elbo"][key]
            if go is None:
                assert float(g.abs().max()) == 0.0, key
                continue
            err = rel_l2(g, go)
            assert err < _ss_grad_tol(key, meta["task"]), "call %d elbo grad %s: rel l2 %.3e" % (c, key, err)
        eng.adam_step()
        # --- the auxiliary step, from the oracle's parameters after ITS ELBO update (Adam's early steps are +-lr whatever the
        # gradient's size, so entries whose ELBO gradient is rounding-noise-sized land on either side: held to Adam's own bound
        # here, and the auxiliary gradients are compared from identical parameters, as the ELBO step's are)
        if y is not None:
            for key, p in model.state_dict().items():
                d = (p.detach().cpu() - mid[key]).abs()
                assert d.max().item() <= 2.1 * lr, key
                assert (d > 1e-6 + 1e-4 * mid[key].abs()).float().mean().item() < 0.02, "%s: too many entries off" % key
            model.load_state_dict(mid)
            aux = eng.aux_loss_and_grads(x.cuda(), y.cuda(), meta["mult"])
            np.testing.assert_allclose(aux.item(), l2o, rtol=2e-5)
            if c == 1:
                np.testing.assert_allclose(aux.item(), float(gold[pre + ".aux.loss"]), rtol=1e-4)
            for key in o.p:
                g, go = eng.grad_of(key), o.last_grads["aux"][key]
                if key.startswith("encoder_y."):
                    assert rel_l2(g, go) < 1e-4, "call %d aux grad %s" % (c, key)
                else:
                    assert float(g.abs().max()) == 0.0, key
        eng.adam_step()
        for key, p in model.state_dict().items():
            pc, pr = p.detach().cpu(), o.p[key].detach()
            d = (pc - pr).abs()
            assert d.max().item() <= 4.2 * lr, key                     # two Adam steps: |dp| <= 2 * lr / (1 - beta1)... early on
            assert (d > 1e-6 + 1e-4 * pr.abs()).float().mean().item() < 0.02, "%s: too many entries off" % key
        model.load_state_dict({k_: v_.detach() for k_, v_ in o.p.items()})
    # inference API against the oracle on the synchronised parameters
    if meta["task"] == "classification":
        np.testing.assert_array_equal(model.classifier(xs).numpy(), o.predict(xs).numpy())
        z_loc, z_scale, y_pred = model.encode(xu[:meta["batch_u"]])
        yq = pv.utils.to_onehot(y_pred, meta["dim"])
    else:
        np.testing.assert_allclose(model.regressor(xs).numpy(), o.predict(xs).numpy(), rtol=1e-4, atol=2e-6)
        z_loc, z_scale, yq = model.encode(xu[:meta["batch_u"]])
    zo, so = o.encode(xu[:meta["batch_u"]], yq)
    np.testing.assert_allclose(z_loc.numpy(), zo.numpy(), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(z_scale.numpy(), so.numpy(), rtol=1e-4, atol=5e-6)
    yd = torch.from_numpy(gold["dec.y"])
    dec = model.decode(z_loc[:, -meta["latent_dim"]:], yd)
    deo = o.decode(z_loc[:, -meta["latent_dim"]:], yd)
    np.testing.assert_allclose(dec.numpy().reshape(deo.shape), deo.numpy(), rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("name", [n for n in SS_CASES if "28x28" not in n])
def test_ss_trainer_epochs_vs_golden(gpu_device, name):
    """auxSVItrainer.step epochs through the reference API (loaders, interleaved labeled batches, CPU RNG stream, test
    metric) against the reference trainer's recorded history."""
    from conftest import ssmeta_of, ss_build
    gold = load_golden(name)
    meta = ssmeta_of(gold)
    model = ss_build(meta, "cuda")
    trainer = pv.trainers.auxSVItrainer(model, task=meta["task"], seed=1)
    xu, xs, ys = (torch.from_numpy(gold[k]) for k in ("xu", "xs", "ys"))
    lu, ls, lv = pv.utils.init_ssvae_dataloaders(xu[:meta["n_u"]], (xs[:meta["n_s"]], ys[:meta["n_s"]]),
                                                 (xs[:meta["n_s"]], ys[:meta["n_s"]]), batch_size=meta["batch_s"])
    kw = {"scale_factor": meta["beta"], "aux_loss_multiplier": meta["mult"]}
    for _ in range(meta["epochs"]):
        trainer.step(lu, ls, lv, **kw)
    np.testing.assert_allclose(trainer.history["training_loss"], gold["epochs.training_loss"], rtol=2e-4)
    np.testing.assert_allclose([float(v) for v in trainer.history["test"]], gold["epochs.test"], rtol=2e-3, atol=1e-6)
    assert trainer.current_epoch == meta["epochs"]
    trainer.save_running_weights("encoder_y")
    trainer.average_weights("encoder_y")
    trainer.print_statistics()


@pytest.mark.parametrize("mode", ["weak", "strong"])
def test_bench_multirank_path_on_one_gpu(gpu_device, mode):
    """bench.py's N > 1 code path end to end (process group, replica broadcast, per-step all-reduce of [gradients | loss],
    barrier + max-over-ranks timing, one JSON line from rank 0) with two ranks sharing this box's single GPU over gloo
    (bench.py's test hook): the line must parse and both precision legs must report the same ELBO as a one-rank run
    would for a global batch of 512 (loss per image of step 0 within 1e-4 of the single-GPU value at batch 256 scale)."""
    import json
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PV_BENCH_BACKEND="gloo", PV_BENCH_ONE_DEVICE="1")
    cmd = [_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531" if mode == "weak" else "29533", os.path.join(root, "bench.py"), "--gpus", "2",
           "--steps", "6", "--warmup", "2", "--repeats", "2"] + (["--strong"] if mode == "strong" else [])
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["scaling"] == mode and d["value"] > 0
    assert "fp32_class" in d and abs(d["fp32_class"]["loss_per_image_step0"] - d["elbo"]["loss_per_image_step0"]) < 1e-2
    assert 500 < d["elbo"]["loss_per_image_step0"] < 600
    assert len(d["ms_per_step_all"]) == 2 and d["regions_discarded"] >= 1
    # (round 3) the collective is event-timed, and a default N > 1 run reports both scalings
    assert d["allreduce_ms"] > 0 and d["fp32_class"]["allreduce_ms"] > 0
    assert d["roofline"]["kernel"].startswith("void pv_sdec_")
    if mode == "weak":
        assert d["strong"]["global_batch"] == 256 and d["strong"]["batch_per_gpu"] == 128 and d["strong"]["value"] > 0
        # (round 4) the conv config's weak-scaling leg rides along, and the predictions the measured x are held against
        assert d["c5_weak"]["value"] > 0 and d["c5_weak"]["batch_per_gpu"] == 256 and d["c5_weak"]["allreduce_ms"] > 0
        assert "c5_weak" in d["expected_x8"]
    if mode == "strong":
        # the global batch stays 256 (128 per rank): the same data and noise as the single-GPU headline run, so the same
        # ELBO (544.5358 per image from the fp32 oracle; bench.py's own rel_err_step0 line at N = 1)
        assert "global 256" in d["config"]["workload"]
        assert abs(d["fp32_class"]["loss_per_image_step0"] - 544.5358) < 0.02


@pytest.mark.parametrize("kind", ["ivae", "ssivae"])
def test_trainer_data_parallel_two_ranks_one_gpu(gpu_device, kind):
    """The trainers' data-parallel path with the REAL engine: two ranks (gloo, sharing this box's GPU) shard every global
    minibatch, all-reduce [gradients | loss] once per SVI step and must reproduce the single-process loss history and
    weights (sums over samples: sharding is exact up to fp32 summation order)."""
    import json
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = os.path.join(root, "tests", "_dp_gpu_worker.py")

    def run(nproc, port):
        cmd = [_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
               "127.0.0.1", "--master-port", str(port), worker, kind]
        out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][0]
        return json.loads(line[len("RESULT "):])
    one, two = run(1, 29541), run(2, 29543)
    np.testing.assert_allclose(two["train"], one["train"], rtol=2e-5)
    np.testing.assert_allclose(two["test"], one["test"], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(two["wsum"], one["wsum"], rtol=1e-5)


def test_manifold2d_matches_decode_and_plots(gpu_device):
    """manifold2d / manifold_traversal with the reference's default plot=True (models/ivae.py:277-310,
    jivae.py:268-329, ved.py:218-243): the returned tensor is decode() of utils.generate_latent_grid's points, the
    same as the oracle's decoder on them, and the matplotlib tail runs (Agg backend)."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    for data_dim, inv in (((12, 12), ['r', 't']), ((16,), ['t']), ((10, 10), None)):
        m = pv.models.iVAE(data_dim, 2, inv, seed=2, device="cuda")
        cfg = orc.Config(data_dim, 2, inv)
        P = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        loc = m.manifold2d(4)
        z, _ = pv.utils.generate_latent_grid(4)
        assert loc.shape == (16, *data_dim)
        ref, _ = orc.decode_from_latent(P, cfg, torch.cat([torch.zeros(16, cfg.coord), z], -1))   # no transform
        np.testing.assert_allclose(loc.numpy(), ref.reshape(loc.shape).numpy(), rtol=1e-4, atol=2e-6)
        np.testing.assert_array_equal(m.manifold2d(4, plot=False).numpy(), loc.numpy())
        loc2 = m.manifold2d(3, z_coord=[-2., 2., -1., 1.], cmap="viridis", padding=1)
        assert loc2.shape == (9, *data_dim)
    mj = pv.models.jiVAE((12, 12), 2, 3, ['r'], seed=2, device="cuda")
    assert mj.manifold2d(3, disc_idx=1).shape == (9, 12, 12)
    assert mj.manifold_traversal(4, 0).shape == (16, 12, 12)
    mv = pv.models.VED((16, 16), (16, 16), hidden_dim_e=[(8,), (16, 16)], hidden_dim_d=[(16, 16), (8,)], seed=2,
                       device="cuda")
    assert mv.manifold2d(3).shape == (9, 1, 16, 16)
    mv1 = pv.models.VED((16, 16), (32,), hidden_dim_e=[(8,), (16, 16)], hidden_dim_d=[(16, 16), (8,)], seed=2,
                        device="cuda")
    assert mv1.manifold2d(2).shape == (4, 1, 32)
    ms = pv.models.ssiVAE((12, 12), 2, 3, ['r'], seed=2, device="cuda")
    assert ms.manifold2d(3, label=1).shape == (9, 12, 12)
    assert ms.manifold_traversal(3, 1).shape == (9, 12, 12)
    mr = pv.models.ss_reg_iVAE((12, 12), 2, 1, ['r'], seed=2, device="cuda")
    assert mr.manifold2d(3, torch.zeros(1, 1)).shape == (9, 12, 12)
    plt.close("all")


@pytest.mark.parametrize("kind", ["ivae_f2", "ivae_f3", "ivae_f0", "jivae", "cvae", "convenc", "b5000", "ivae_f2_b256", "ivae_f3_b256"])
def test_one_call_step_is_bit_identical(gpu_device, kind):
    """loss_and_grads(step=True) (pv_ivae_step: on the fused path Adam rides in the last gradient launch, every element
    updated by the workgroup that finalises its gradient + guest workgroups for the rest) against loss_and_grads() +
    adam_step(): parameters, both Adam moments, the zeroed gradients and the loss scalars must be bit-identical over
    several steps; with a conv encoder (round 4) Adam rides in the launch of fc_latent's weight gradient, the step's last; paths
    that cannot fuse (layered decoder, long batches) fall back to the same pair.
    (round 6, *_b256: at batch == number of CUs every decoder workgroup owns one image, runs its latent backward and encoder chain
    itself, and the step closes with ONE launch — record sums with Adam applied by the blocks that finalise them, the small
    weight gradients with Adam in their epilogues, no guest workgroups: pv_rec_wgrad_kernel.)"""
    torch.manual_seed(3)
    b = 5000 if kind == "b5000" else (256 if kind.endswith("_b256") else 37)
    def make():
        if kind == "jivae":
            m = pv.models.jiVAE((28, 28), 2, 3, ["r", "t"], seed=1, device="cuda")
        elif kind == "cvae":
            m = pv.models.iVAE((28, 28), 2, ["r", "t", "s"], c_dim=3, seed=1, device="cuda")
        else:
            m = pv.models.iVAE((28, 28) if kind != "convenc" else (16, 16), 2, ["r", "t"], seed=1, device="cuda")
            if kind == "convenc":
                m.set_encoder(pv.nets.convEncoderNet((16, 16), latent_dim=m.z_dim, hidden_dim=[(8,), (8, 8)]))
        return m, m.engine(fused={"ivae_f3": 3, "ivae_f0": 0, "ivae_f3_b256": 3}.get(kind, 2))
    (m1, e1), (m2, e2) = make(), make()
    dd = m1.data_dim
    x = torch.rand(b, *dd).cuda()
    y = pv.utils.to_onehot(torch.randint(0, 3, (b,)), 3).cuda() if kind == "cvae" else None
    for k in range(3):
        eps = torch.randn(b, m1.z_dim).cuda()
        e1.loss_and_grads(x, eps, 1.2, y)
        s1 = e1.scalars.clone()
        e1.adam_step()
        hist = torch.zeros(4, device="cuda")
        e2.loss_and_grads(x, eps, 1.2, y, scalars_out=hist, step=True)
        assert torch.equal(s1, hist), (kind, k)
        assert e1.adam_t == e2.adam_t == k + 1
        for name, a, b_ in (("params", e1.flat, e2.flat), ("m", e1.m, e2.m), ("v", e1.v, e2.v),
                            ("grad", e1.grad[:e1.n_flat], e2.grad[:e2.n_flat])):
            assert torch.equal(a, b_), (kind, k, name, (a - b_).abs().max().item())
    assert float(e2.grad[:e2.n_flat].abs().sum()) == 0.0


def test_fails_loudly_on_cpu_tensors(gpu_device):
    model = pv.models.iVAE((8, 8), 2, ["r"], seed=1, device="cuda")
    eng = model.engine()
    with pytest.raises(_abi.PvError):
        eng.loss_and_grads(torch.rand(4, 8, 8), torch.randn(4, 3))


def test_trainer_settings_reach_an_existing_engine(gpu_device):
    """model.engine(**kw) must not drop settings when the engine already exists (created by encode(), an earlier
    trainer ...), and every trainer starts a fresh Adam like the reference's (trainers/svi.py:75-81)."""
    x = make_x("rand", 64, (8, 8))
    loader = pv.utils.init_dataloader(x, batch_size=16)

    def run(pre_touch):
        model = pv.models.iVAE((8, 8), 2, ["r", "t"], seed=1, device="cuda")
        if pre_touch:
            model.encode(x[:4])                                     # creates the engine with default settings
            t0 = pv.trainers.SVItrainer(model, seed=1)              # an earlier trainer leaves Adam state behind
            t0.step(loader)
            model.load_state_dict(ref_state)
        tr = pv.trainers.SVItrainer(model, seed=1, precision="bf16", lr=5e-3)
        assert tr.engine.fused == 3 and tr.engine.lr == 5e-3 and tr.engine.adam_t == 0
        assert not tr.engine.m.any() and not tr.engine.v.any()
        tr.step(loader)
        return tr.loss_history["training_loss"][0], {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ref_state = {k: v.detach().clone() for k, v in pv.models.iVAE((8, 8), 2, ["r", "t"], seed=1, device="cuda").state_dict().items()}
    l0, p0 = run(False)
    l1, p1 = run(True)
    assert l0 == l1
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k


def test_model_on_a_non_current_device(gpu_device):
    """A model built on cuda:1 while cuda:0 is the current device: every library call runs on the model's device."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    torch.cuda.set_device(0)
    x = make_x("rand", 12, (8, 8))
    outs = []
    for dev in ("cuda:0", "cuda:1"):
        model = pv.models.iVAE((8, 8), 2, ["r", "t", "s"], seed=1, device=dev)
        tr = pv.trainers.SVItrainer(model, seed=1)
        tr.step(pv.utils.init_dataloader(x, batch_size=6))
        outs.append((tr.loss_history["training_loss"][0], model.encode(x)[0]))
        assert torch.cuda.current_device() == 0
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])


W8_SMALL = {
    "ivae_8x8_rts_b6": {}, "ivae_8x8_r_b6": {}, "ivae_1d16_t_b5": {}, "ivae_7x9_rts_b3": {}, "ivae_8x8_rt_b6_beta4": {},
    "ivae_8x8_rts_b6_randn": {}, "ivae_28x28_r_b32_blobs": {},
}


def _experiments_build() -> bool:
    """Is the loaded library the -DPV_EXPERIMENTS build (csrc/Makefile `experiments`, PV_LIB_PATH=.../libpyroved_amd_exp.so)?
    Only that build contains the dropped decoder-kernel variants."""
    return bool(_abi.lib().pv_experiments_build())


@pytest.fixture()
def force_w8():
    """pv_ivae_plan.dec_kernel = 2 (the 8-wave plain-bf16 kernel whatever the size) for every engine made meanwhile."""
    from pyroved_amd.engine import IVAEEngine
    IVAEEngine.dec_kernel = 2
    try:
        yield
    finally:
        IVAEEngine.dec_kernel = 0


@pytest.mark.parametrize("name", sorted(W8_SMALL))
def test_w8_kernel_on_small_and_odd_cases(gpu_device, force_w8, name):
    """The 8-wave plain-bf16 decoder kernel is chosen by problem size (>= 6 units per workgroup); here it is FORCED on the
    small / odd fixtures — partial tiles with most waves idle, 1-D data, ragged rows (7x9), non-unit KL scale, randn and
    saturated inputs — against the reference's recorded losses and the oracle's gradients (the mode's own bars)."""
    gold = load_golden(name)
    meta = meta_of(gold)
    model, cfg, eng = build(meta, 3)
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    for k in range(meta["steps"]):
        pre = "s%d" % k
        eps = torch.from_numpy(gold[pre + ".eps"])
        eng.loss_and_grads(x.cuda(), eps.cuda(), meta["beta"])
        s = eng.scalars.cpu().numpy()
        np.testing.assert_allclose(s[0], float(gold[pre + ".loss"]), rtol=5e-4, err_msg="loss")
        o.step(x, eps, meta["beta"])
        for key in o.p:
            err = rel_l2(eng.grad_of(key), o.last_grads[key])
            assert err < 3e-2, "step %d grad %s: rel l2 error %.3e vs oracle" % (k, key, err)
        # the forward-only launch (evaluate) of the same kernel
        eng.loss_and_grads(x.cuda(), eps.cuda(), meta["beta"], want_grads=False)
        np.testing.assert_allclose(eng.scalars[0].item(), s[0], rtol=1e-6)
        eng.adam_step()
        model.load_state_dict({k_: v_.detach() for k_, v_ in o.p.items()})


@pytest.mark.parametrize("fused", [1, 2, 3])
@pytest.mark.parametrize("case", ["28x28_rt", "12x20_rts_gauss", "1d32_t", "jivae_8x8_r"])
def test_fused_forward_only_decode(gpu_device, case, fused):
    """decode() runs the fused persistent decoder kernel forward-only (no (B N) x 128 activations; SURVEY 8f rank 1,
    reference models/base.py:145-171) — against the layer-by-layer kernels and the oracle's decode, with angle / shift /
    scale, at fp32-class precision whatever the training precision (fused = 3 decodes in split precision too), and its
    workspace stays small at a batch where the layered path would need tens of GB."""
    g = torch.Generator().manual_seed(5)
    kw, y = {}, None
    if case == "28x28_rt":
        dd, inv = (28, 28), ["r", "t"]
    elif case == "12x20_rts_gauss":
        dd, inv, kw = (12, 20), ["r", "t", "s"], dict(sampler_d="gaussian", sigmoid_d=False)
    elif case == "1d32_t":
        dd, inv = (32,), ["t"]
    else:
        dd, inv = (8, 8), ["r"]
    if case.startswith("jivae"):
        model = pv.models.jiVAE(dd, 2, 3, inv, seed=2, device="cuda")
        ref = pv.models.jiVAE(dd, 2, 3, inv, seed=2, device="cuda")
        y = torch.zeros(37, 3)
        y[torch.arange(37), torch.randint(0, 3, (37,), generator=g)] = 1.0
    else:
        model = pv.models.iVAE(dd, 2, inv, seed=2, device="cuda", **kw)
        ref = pv.models.iVAE(dd, 2, inv, seed=2, device="cuda", **kw)
    eng = model.engine(fused=fused)
    ref.engine(fused=0)
    z = torch.randn(37, 2, generator=g)
    tkw = {}
    if "r" in inv:
        tkw["angle"] = torch.tensor(0.4)
    if "t" in inv:
        tkw["shift"] = torch.tensor([0.1, -0.2][:len(dd)])
    if "s" in inv:
        tkw["scale"] = torch.tensor(1.2)
    for k in ({}, tkw):
        a = model.decode(z, y, **k) if y is not None else model.decode(z, **k)
        b = ref.decode(z, y, **k) if y is not None else ref.decode(z, **k)
        assert a.shape == (37, *dd) and not a.is_cuda
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-4, atol=2e-6)
    # several loader batches (batch_size 16): same values, one host copy
    a = model.decode(z, y, batch_size=16) if y is not None else model.decode(z, batch_size=16)
    np.testing.assert_allclose(a.numpy(), (ref.decode(z, y) if y is not None else ref.decode(z)).numpy(), rtol=1e-4, atol=2e-6)
    if case == "28x28_rt":
        p = eng._plan(32768, what=3)
        need = _abi.lib().pv_ivae_workspace_bytes_for(C.byref(p), 3)
        assert 0 < need < (1 << 26), need            # 64 MB (the layered decode of 32768 latents: 63 GB)
        zz = torch.randn(4096, 2, generator=g)
        big = model.decode(zz, batch_size=4096)
        small = ref.decode(zz[:64])
        np.testing.assert_allclose(big[:64].numpy(), small.numpy(), rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("fused", [0, 2])
def test_continuous_bernoulli_normaliser_vs_float64(gpu_device, fused):
    """ ContinuousBernoulli's log-normaliser log C(p) = log(2 atanh(1 - 2p) / (1 - 2p)) and its derivative
    cancel catastrophically at p -> 1/2 — where every pixel of a fresh model sits; the kernels use series in (1 - 2p)^2 there
    (pv_common.h: pv_cbern).  Pinned DIRECTLY against float64: a decoder whose output weights are zero emits the same logit
    a = out.bias for every pixel; sweep a through 0, the series / closed-form switch (|1 - 2p| = 0.3 at a = 0.619) and the
    saturated tails, and compare sum log p(x | z) and d loss / d out.bias with torch.distributions in float64
    (utils/prob.py:27: ContinuousBernoulli(probs=sigmoid(a)))."""
    inv = ["r"] if fused else None
    model = pv.models.iVAE((8, 8), 2, inv, seed=1, device="cuda", sampler_d="continuous_bernoulli")
    g = torch.Generator().manual_seed(4)
    x = torch.rand(9, 8, 8, generator=g)
    eps = torch.randn(9, model.z_dim, generator=g)
    # (|a| <= 4: further out fp32's resolution of 1 - p, not the normaliser, sets the error — in the reference's fp32 too)
    for a in (0.0, 1e-6, -1e-4, 3e-3, -2e-2, 0.3, -0.6, 0.619, 0.63, -1.5, 2.5, -4.0, 4.0):
        with torch.no_grad():
            model.decoder.out.weight.zero_()
            model.decoder.out.bias.fill_(a)
        eng = model.engine(fused=fused)
        eng.loss_and_grads(x.cuda(), eps.cuda())
        s = eng.scalars.cpu().numpy()
        ad = torch.full((9, 64), a, dtype=torch.float64, requires_grad=True)
        d = torch.distributions.ContinuousBernoulli(probs=torch.sigmoid(ad), validate_args=False)
        ll = d.log_prob(x.reshape(9, 64).double()).sum()
        (-ll).backward()
        # (a pixel's log-probability is -BCE + log C: two O(0.69) fp32 numbers that nearly cancel around p = 1/2, so one
        #  fp32 ulp of either — 6e-8 — is the floor per pixel: 576 pixels -> 7e-5 absolute)
        np.testing.assert_allclose(s[1], ll.item(), rtol=2e-6 if abs(a) < 1 else 2e-5, atol=576 * 1.2e-7, err_msg="sum log p(x|z) at logit %g" % a)
        gb = eng.grad_of("decoder.out.bias").cpu().double().sum().item()
        ref = ad.grad.sum().item()
        assert abs(gb - ref) <= 2e-5 * max(1.0, abs(ref)) + 2e-4, "d loss / d out.bias at logit %g: %.8g vs %.8g" % (a, gb, ref)


def test_conv_weight_range_switches_kernels(gpu_device):
    """ The default fp32-class 2-D convolution kernels carry the weights as fp16 pieces of w * 64 — exact only
    for max|w| < 1023.  A VED whose encoder weights are blown up beyond that: the engine notices at its first step, switches
    THIS model's plans to the three-piece bf16 kernels (no range limit; ABI v14: pv_ved_plan.conv_bf16 = 2) with a warning,
    and the step matches the oracle at the usual bars instead of returning inf / NaN — while a second model in the same
    process, with ordinary weights, keeps the fp16-piece kernels (round 3's process-wide switch moved both)."""
    other = pv.models.VED((32, 32), (32,), latent_dim=2, seed=2, device="cuda")
    eng_other = other.engine()
    model = pv.models.VED((32, 32), (32,), latent_dim=2, seed=1, device="cuda")
    cfg = orc.VedConfig(input_dim=(32, 32), output_dim=(32,), latent_dim=2, hidden_dim_e=None, hidden_dim_d=None,
                        activation="lrelu")
    with torch.no_grad():
        w = model.encoder_z.feature_extractor.layers[5].weight          # the 64 -> 64 kernel-3 convolution (split-operand kernel)
        assert w.dim() == 4 and w.shape[1] % 32 == 0
        w.mul_(2000.0 / w.abs().max())                                   # max|w| = 2000: fp16(w * 64) would be inf
        model.encoder_z.feature_extractor.layers[0].weight.mul_(1e-3)    # (keeps the activations in a sane range)
    eng = model.engine()
    o = orc.VedOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg, dtype=torch.float64)
    g = torch.Generator().manual_seed(3)
    x, y, eps = torch.rand(6, 1, 32, 32, generator=g), torch.rand(6, 1, 32, generator=g), torch.randn(6, 2, generator=g)
    if True:
        with pytest.warns(UserWarning, match="range-free bf16-piece"):
            eng.loss_and_grads(x.cuda(), eps.cuda(), 1.0, y.cuda())
        assert eng.wide_weights and eng._static.conv_bf16 == 2
        eng_other.loss_and_grads(x.cuda(), eps.cuda(), 1.0, y.cuda())
        assert not eng_other.wide_weights and eng_other._static.conv_bf16 == 4      # per plan: the other model is untouched (4: the default fp32-class mode)
        assert np.isfinite(eng_other.scalars.cpu().numpy()).all()
        s = eng.scalars.cpu().numpy()
        loss_ref = o.step(x, y, eps, 1.0)
        assert np.isfinite(s).all()
        np.testing.assert_allclose(s[0], loss_ref, rtol=1e-4)
        for key in o.p:
            # (a network with weights of 2000 next to weights of 1e-3 is ill-conditioned by construction: fp32 arithmetic
            #  itself is ~1e-3 off the float64 oracle in the first layers; what is checked is "finite and right", not 1e-4)
            err = rel_l2(eng.grad_of(key), o.last_grads[key].float())
            assert err < 1e-2, "grad %s: rel l2 error %.3e vs the float64 oracle" % (key, err)
        # encode() on freshly bound out-of-range weights is checked too (the check ran only in training steps)
        eng2 = model.engine()
        eng2.bind()
        with pytest.warns(UserWarning, match="range-free bf16-piece"):
            zl, _ = eng2.encode(x.cuda())
        assert eng2.wide_weights and torch.isfinite(zl).all()


def test_conv_weight_range_throughput_precision(gpu_device):
    """ The throughput precision (fused = 3 / precision="bf16") tiles kernel-3 weights as ONE fp16 piece of
    w * 64: |w| >= 1023 would become inf.  With a weight blown up beyond that the engine switches THIS model's plans to the
    range-free two-piece bf16 "mixed" kernels (pv_ved_plan.conv_bf16 = 1) and the step stays finite and right at the
    throughput precision's own bars (ELBO 1e-3 on this ill-conditioned network, gradients 5e-2)."""
    model = pv.models.VED((32, 32), (32,), latent_dim=2, seed=1, device="cuda")
    cfg = orc.VedConfig(input_dim=(32, 32), output_dim=(32,), latent_dim=2, hidden_dim_e=None, hidden_dim_d=None,
                        activation="lrelu")
    with torch.no_grad():
        w = model.encoder_z.feature_extractor.layers[5].weight
        w.mul_(2000.0 / w.abs().max())
        model.encoder_z.feature_extractor.layers[0].weight.mul_(1e-3)
    eng = model.engine(fused=3)
    o = orc.VedOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg, dtype=torch.float64)
    g = torch.Generator().manual_seed(3)
    x, y, eps = torch.rand(6, 1, 32, 32, generator=g), torch.rand(6, 1, 32, generator=g), torch.randn(6, 2, generator=g)
    with pytest.warns(UserWarning, match="range-free bf16-piece"):
        eng.loss_and_grads(x.cuda(), eps.cuda(), 1.0, y.cuda())
    assert eng.wide_weights and eng._static.conv_bf16 == 1
    s = eng.scalars.cpu().numpy()
    loss_ref = o.step(x, y, eps, 1.0)
    assert np.isfinite(s).all()
    np.testing.assert_allclose(s[0], loss_ref, rtol=1e-3)
    for key in o.p:
        gk = eng.grad_of(key)
        assert torch.isfinite(gk).all(), key
        err = rel_l2(gk, o.last_grads[key].float())
        assert err < 5e-2, "grad %s: rel l2 error %.3e vs the float64 oracle" % (key, err)
    # the conv-encoder iVAE plan takes the same route (pv_plan.hip: plan_conv_mode)
    m2 = pv.models.iVAE((32, 32), 2, ["r"], seed=1, device="cuda")
    m2.set_encoder(pv.nets.convEncoderNet((32, 32), latent_dim=m2.z_dim))
    with torch.no_grad():
        c4 = [p_ for n_, p_ in m2.encoder_z.named_parameters() if p_.dim() == 4]
        i = [k for k, p_ in enumerate(c4) if p_.shape[1] % 32 == 0][0]
        c4[i].mul_(2000.0 / c4[i].abs().max())          # out of the one-piece fp16 range
        c4[i + 1].mul_(3e-5)                             # (the next layer brings the features back to a sane range)
    e2 = m2.engine(fused=3)
    xx = torch.rand(4, 32, 32, generator=g)
    ee = torch.randn(4, m2.z_dim, generator=g)
    cfg2 = orc.Config(data_dim=(32, 32), latent_dim=2, invariances=["r"], conv_encoder=[(32,), (64, 64), (128, 128)])
    o2 = orc.SVIOracle({k: v.cpu() for k, v in m2.state_dict().items()}, cfg2, dtype=torch.float64)
    with pytest.warns(UserWarning, match="range-free bf16-piece"):
        e2.loss_and_grads(xx.cuda(), ee.cuda())
    assert e2.wide_weights and np.isfinite(e2.scalars.cpu().numpy()).all()
    assert torch.isfinite(e2.grad).all()
    np.testing.assert_allclose(e2.scalars[0].item(), o2.loss_and_grads(xx, ee)["loss"].item(), rtol=1e-3)


def test_engine_notices_moved_and_replaced_parameters(gpu_device):
    """Every call asks whether the model's parameters still live in the engine's flat buffer (engine.IVAEEngine._bound; the
    reference re-registers the modules with pyro.module at every model()/guide() call, models/ivae.py:145,171).  The
    per-call check is a snapshot walk (round 5: nn.Module.named_parameters() was a third of a 0.1 ms step's host time): it
    has to see a tensor whose storage moved, a replaced nn.Parameter, a swapped submodule and an added parameter — and
    the step after each must use the NEW values (compared with a fresh model holding them)."""
    def fresh(state):
        m = pv.models.iVAE((16, 16), latent_dim=2, invariances=["r"], hidden_dim_e=[64, 64], hidden_dim_d=[64, 64], seed=0, device="cuda")
        m.load_state_dict(state)
        return m
    g = torch.Generator().manual_seed(5)
    x, eps = torch.rand(32, 256, generator=g).cuda(), torch.randn(32, 3, generator=g).cuda()
    model = pv.models.iVAE((16, 16), latent_dim=2, invariances=["r"], hidden_dim_e=[64, 64], hidden_dim_d=[64, 64], seed=0, device="cuda")
    eng = model.engine()
    eng.loss_and_grads(x, eps)
    assert eng._bound() and eng._snap is not None          # the fast path is armed
    lin = next(m for m in model.decoder.modules() if isinstance(m, torch.nn.Linear) and m.bias is not None)

    def check(what):
        assert not eng._bound(), what + ": not noticed"
        eng.loss_and_grads(x, eps)
        assert eng._bound()
        ref = fresh({k: v.detach().clone() for k, v in model.state_dict().items()}).engine()
        ref.loss_and_grads(x, eps)
        assert torch.equal(eng.scalars, ref.scalars), what
        assert torch.equal(eng.grad[:eng.n_flat], ref.grad[:ref.n_flat]), what

    with torch.no_grad():                                  # 1. the storage moved (what .to() / .half().float() do)
        lin.weight.data = lin.weight.data.clone() * 1.25
    check("moved storage")
    lin.bias = torch.nn.Parameter(torch.full_like(lin.bias, 0.01))          # 2. a replaced nn.Parameter
    check("replaced parameter")
    owner = next(m for m in model.modules() if any(c is lin for c in m.children()))
    name = next(n for n, c in owner.named_children() if c is lin)
    new_lin = torch.nn.Linear(lin.in_features, lin.out_features).cuda()     # 3. a swapped submodule
    setattr(owner, name, new_lin)
    check("swapped submodule")


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_two_stream_conv_steps_are_bit_identical(gpu_device, precision):
    """Steps of models with a convolutional encoder put the encoder's kernel-3 weight gradients, the decoder's batched
    weight gradients and their reductions on the library's side stream (pv_side.hip; include/pyroved_amd.h
    PV_PLAN_NO_SIDE_STREAM, a plan flag since ABI v14).  Same kernels, same summation order: loss, every gradient and the parameters after three Adam
    steps are BIT-identical with the side stream off — for VED (models/ved.py:122-163) and for an iVAE with
    convEncoderNet (nets/conv.py:24-102), at a batch where every layer takes the split-operand kernels."""
    lib = _abi.lib()
    fused = 3 if precision == "bf16" else 2
    g = torch.Generator().manual_seed(5)
    xv, yv, ev = torch.rand(48, 1, 64, 64, generator=g), torch.rand(48, 1, 128, generator=g), torch.randn(48, 2, generator=g)
    xi, ei = torch.rand(24, 64, 64, generator=g), torch.randn(24, 6, generator=g)

    def run(side):
        out = []
        ved = pv.models.VED((64, 64), (128,), latent_dim=2, seed=1, device="cuda")
        eng = ved.engine(fused=fused)
        eng.side_stream = bool(side)
        for _ in range(3):
            eng.loss_and_grads(xv.cuda(), ev.cuda(), 1.0, yv.cuda())
            out.append(eng.scalars.clone())
            out.append(eng.grad.clone())
            eng.adam_step()
        out.append(eng.flat.clone())
        iv = pv.models.iVAE((64, 64), 2, ["r", "t", "s"], seed=1, device="cuda")
        iv.set_encoder(pv.nets.convEncoderNet((64, 64), latent_dim=6))
        eng = iv.engine(fused=fused)
        eng.side_stream = bool(side)
        for _ in range(3):
            eng.loss_and_grads(xi.cuda(), ei.cuda())
            out.append(eng.scalars.clone())
            out.append(eng.grad.clone())
            eng.adam_step()
        out.append(eng.flat.clone())
        torch.cuda.synchronize()
        return out

    one, two = run(0), run(1)
    assert len(one) == len(two)
    for k, (a, b) in enumerate(zip(one, two)):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b), "tensor %d differs between the one-stream and the two-stream step" % k


_CONV_TAIL_SCRIPT = r"""
import sys, torch
import pyroved_amd as pv
g = torch.Generator().manual_seed(7)
x, eps = torch.rand(40, 64, 64, generator=g), torch.randn(40, 6, generator=g)
iv = pv.models.iVAE((64, 64), 2, ["r", "t", "s"], seed=1, device="cuda")
iv.set_encoder(pv.nets.convEncoderNet((64, 64), latent_dim=6))
eng = iv.engine(fused=int(sys.argv[1]))
out = []
for _ in range(2):
    eng.loss_and_grads(x.cuda(), eps.cuda())
    out += [eng.scalars.clone().cpu(), eng.grad[:eng.n_flat].clone().cpu()]      # (eng.grad ends with the four scalars)
    eng.adam_step()
out.append(eng.flat.clone().cpu())
torch.save(out, sys.argv[2])
"""


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_conv_encoder_tail_launches_match_the_separate_ones(gpu_device, precision, tmp_path):
    """Round 4: in front of the spatial decoder a convEncoderNet's tail (nets/conv.py:77-102 features2latent, then ivae.py:170-196)
    is ONE launch — the conv head's partial sums, reparameterised sample + KL partials + transform parameters, fc_latent
    (pv_head_fwd_blocks) — its head weight gradient runs on the side stream next to the head's input gradient, and the loss scalars
    ride in the last weight-gradient launch.  Against the separate launches (PV_HEAD_MERGE=0 PV_HEAD_SIDE=0 PV_FIN_RIDE=0, a
    second process: the switches are read once): every gradient and the parameters after two Adam steps bit-identical, the loss
    scalars to 1e-6 (the KL sums now meet as per-16-sample partials)."""
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fused = "3" if precision == "bf16" else "2"
    outs = []
    # the A/B switches exist in the experiments build of the library only (csrc/Makefile `experiments`; the shipped library reads
    # no environment switch): the merged launches of the SHIPPED library against the separate ones of the experiments build
    if not os.path.exists(_abi.EXP_LIB_PATH):
        pytest.skip("libpyroved_amd_exp.so not built (make -C pyroved_amd/csrc experiments)")
    for k, extra in enumerate(({}, {"PV_LIB_PATH": _abi.EXP_LIB_PATH, "PV_HEAD_MERGE": "0", "PV_HEAD_SIDE": "0", "PV_FIN_RIDE": "0"})):
        f = str(tmp_path / ("o%d.pt" % k))
        r = subprocess.run([_sys.executable, "-c", _CONV_TAIL_SCRIPT, fused, f], env=dict(os.environ, **extra), cwd=root,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(f))
    new, old = outs
    assert len(new) == len(old) == 5
    for k in (0, 2):                                   # scalars: loss, ll, beta logp, beta logq
        assert torch.isfinite(new[k]).all()
        assert torch.allclose(new[k][:4], old[k][:4], rtol=1e-6, atol=1e-4), (new[k][:4], old[k][:4])
    for k in (1, 3, 4):
        assert torch.equal(new[k], old[k]), "tensor %d differs" % k


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_fused_1d_decoder_matches_layer_launches(gpu_device, precision):
    """VED's Conv1d decoder (nets/conv.py:190-262: kernel-3 blocks, UpsampleBlocks, the kernel-1 output layer) runs as one
    forward and one input-gradient launch (csrc/pv_dec1d.hip).  Against the layer-by-layer launches it replaces
    (plan flag PV_PLAN_NO_DEC1D): ELBO, reconstruction and every gradient agree to fp32 rounding — for the default im2spec shape, a
    shorter spectrum with another activation, and two output channels."""
    import ctypes as C
    dbg = C.CDLL(_abi.LIB_PATH)
    fused = 3 if precision == "bf16" else 2
    g = torch.Generator().manual_seed(11)
    cases = [((64, 64), (128,), 1, "lrelu", 20), ((32, 32), (64,), 1, "tanh", 7), ((32, 32), (32,), 2, "relu", 5)]
    try:
        for in_dim, out_dim, och, act, b in cases:
            x = torch.rand(b, 1, *in_dim, generator=g)
            y = torch.rand(b, och, *out_dim, generator=g)
            eps = torch.randn(b, 2, generator=g)
            res = []
            for on in (0, 1):
                m = pv.models.VED(in_dim, out_dim, input_channels=1, output_channels=och, latent_dim=2, activation=act,
                                  seed=1, device="cuda")
                eng = m.engine(fused=fused)
                eng.dec1d = bool(on)                     # pv_ved_plan.flags: PV_PLAN_NO_DEC1D (ABI v15)
                eng.loss_and_grads(x.cuda(), eps.cuda(), 1.0, y.cuda())
                zc = torch.randn(b, 2, generator=torch.Generator().manual_seed(3))
                res.append((eng.scalars.clone(), eng.grad.clone(), m.decode(zc)))
            (s0, g0, d0), (s1, g1, d1) = res
            assert torch.isfinite(g1).all()
            np.testing.assert_allclose(s1.cpu().numpy(), s0.cpu().numpy(), rtol=2e-6)
            err = rel_l2(g1, g0)
            # (mixed precision: the layer launches run the kernel-3 convolutions on two bf16 pieces, the fused launch in exact fp32)
            assert err < (2e-4 if precision == "bf16" else 2e-6), "%s -> %s: gradients differ by %.2e" % (in_dim, out_dim, err)
            assert torch.allclose(d1, d0, rtol=1e-5, atol=1e-6)
    finally:
        pass


@pytest.mark.parametrize("case", ["rt_b256", "r_b512", "rts_b256_16x16", "t1d_b256"])
def test_guide_folded_into_the_decoder_launch(gpu_device, case):
    """Round 5: where a decoder workgroup's rows are a whole number of images (batch a multiple of the grid: BASELINE's batch
    256 on 256 CUs) the plain-bf16 fused step runs the guide — fcEncoderNet.forward (nets/fc.py:51-61), the reparameterised sample
    and its KL terms (models/ivae.py:204-221), _split_latent (models/base.py:97-119), fc_latent — in the decoder launch's prologue
    (csrc/pv_sdec_fused_w8.hip, PvEncFold) as exact fp32 matrix-vector products.  Against the separate encoder launch
    (PV_PLAN_NO_ENC_FOLD) the encoder's outputs agree to fp32 rounding, the loss to 1e-5, every gradient to the mode's bars vs the
    ORACLE on both paths; the folded step is bit-reproducible and its one-call form (pv_ivae_step) bit-identical to the two calls."""
    torch.set_num_threads(8)
    data_dim, inv, b = {"rt_b256": ((28, 28), ["r", "t"], 256), "r_b512": ((28, 28), ["r"], 512),
                        "rts_b256_16x16": ((16, 16), ["r", "t", "s"], 256), "t1d_b256": ((64,), ["t"], 256)}[case]
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    g = torch.Generator().manual_seed(31)
    x = torch.rand(b, *data_dim, generator=g)
    res = {}
    for fold in (True, False):
        m = pv.models.iVAE(data_dim, 2, inv, seed=1, device="cuda")
        eng = m.engine(fused=3)
        eng.enc_fold = fold
        eps = torch.randn(b, m.z_dim, generator=torch.Generator().manual_seed(5))
        p = eng._plan(b)
        folds = bool(_abi.lib().pv_ivae_guide_folds(C.byref(p)))
        n_units = b * int(np.prod(data_dim)) // 16
        expect = fold and b == cus and n_units >= 6 * cus
        assert folds == expect, (case, fold, folds, expect)
        eng.loss_and_grads(x.cuda(), eps.cuda())
        torch.cuda.synchronize()
        rec = dict(scalars=eng.scalars.clone(), grad=eng.grad.clone(), folds=folds)
        eng.loss_and_grads(x.cuda(), eps.cuda())
        assert torch.equal(rec["grad"], eng.grad) and torch.equal(rec["scalars"], eng.scalars)      # bit-reproducible
        eng.loss_and_grads(x.cuda(), eps.cuda(), want_grads=False)
        np.testing.assert_allclose(eng.scalars.cpu().numpy(), rec["scalars"].cpu().numpy(), rtol=2e-6)
        cfg = orc.Config(data_dim=data_dim, latent_dim=2, invariances=inv)
        o = orc.SVIOracle({k: v.cpu() for k, v in m.state_dict().items()}, cfg)
        ref = o.step(x, eps)
        np.testing.assert_allclose(rec["scalars"][0].item(), ref, rtol=1e-4)
        for key in o.p:
            lo = eng._layout[key]
            err = rel_l2(rec["grad"][lo:lo + o.p[key].numel()].view_as(o.p[key]), o.last_grads[key])
            assert err < 3e-2, "%s fold=%s grad %s: rel l2 error %.3e vs oracle" % (case, fold, key, err)
        # the one-call step on the same state: parameters bit-identical to loss_and_grads + adam_step
        m2 = pv.models.iVAE(data_dim, 2, inv, seed=1, device="cuda")
        e2 = m2.engine(fused=3); e2.enc_fold = fold
        e2.loss_and_grads(x.cuda(), eps.cuda()); e2.adam_step()
        m3 = pv.models.iVAE(data_dim, 2, inv, seed=1, device="cuda")
        e3 = m3.engine(fused=3); e3.enc_fold = fold
        e3.loss_and_grads(x.cuda(), eps.cuda(), step=True)
        torch.cuda.synchronize()
        assert torch.equal(e2.flat, e3.flat)
        rec["z"] = m.encode(x)                          # (pv_ivae_encode: the encoder kernels — the same weights either way)
        res[fold] = rec
    a, c = res[True], res[False]
    np.testing.assert_allclose(a["scalars"].cpu().numpy(), c["scalars"].cpu().numpy(), rtol=1e-5)
    if a["folds"]:
        assert rel_l2(a["grad"][:eng.n_flat], c["grad"][:eng.n_flat]) < 2e-2       # (bf16 operands: a last-bit change of z moves roundings)


@pytest.mark.parametrize("fused", [2, 3])
@pytest.mark.parametrize("case", ["rt_b256", "r_b128", "rts_b6_8x8", "t1d_b5", "rts_b48_64x64", "none_t_b300"])
def test_guide_per_image_launch_vs_tiled_encoder_and_oracle(gpu_device, case, fused):
    """Round 6: a training step of up to 384 samples runs its guide — fcEncoderNet.forward (nets/fc.py:51-61), the reparameterised
    sample and its KL terms (models/ivae.py:204-221), _split_latent (models/base.py:97-119), fc_latent — as ONE launch with one
    workgroup per image (csrc/pv_guide_img.hip: fp32 matrix-vector products from the L2-resident weights, guest workgroups writing
    the decoder's weight images) instead of the tiled one-launch encoder (PV_PLAN_ENC_TILED keeps that).  Both forms against the
    ORACLE (loss 2e-5 / the throughput mode's 1e-4, every gradient at the path's bar), against each other (the encoder's outputs
    to fp32 rounding), bit-reproducible, and the one-call step bit-identical to the two calls."""
    torch.set_num_threads(8)
    data_dim, inv, b = {"rt_b256": ((28, 28), ["r", "t"], 256), "r_b128": ((28, 28), ["r"], 128), "rts_b6_8x8": ((8, 8), ["r", "t", "s"], 6),
                        "t1d_b5": ((16,), ["t"], 5), "rts_b48_64x64": ((64, 64), ["r", "t", "s"], 48),
                        "none_t_b300": ((28, 28), ["t"], 300)}[case]
    g = torch.Generator().manual_seed(17)
    x = torch.rand(b, *data_dim, generator=g)
    res = {}
    for per_image in (True, False):
        m = pv.models.iVAE(data_dim, 2, inv, seed=1, device="cuda")
        eng = m.engine(fused=fused)
        eng.enc_per_image = per_image
        eng.enc_fold = False                                         # (both arms launch a guide)
        eps = torch.randn(b, m.z_dim, generator=torch.Generator().manual_seed(5))
        zl, zs = torch.empty(b, m.z_dim, device="cuda"), torch.empty(b, m.z_dim, device="cuda")
        eng.loss_and_grads(x.cuda(), eps.cuda(), z_out=(zl, zs))
        torch.cuda.synchronize()
        rec = dict(scalars=eng.scalars.clone(), grad=eng.grad.clone(), zl=zl.clone(), zs=zs.clone())
        eng.loss_and_grads(x.cuda(), eps.cuda())
        assert torch.equal(rec["grad"], eng.grad) and torch.equal(rec["scalars"], eng.scalars)      # bit-reproducible
        eng.loss_and_grads(x.cuda(), eps.cuda(), want_grads=False)
        np.testing.assert_allclose(eng.scalars.cpu().numpy(), rec["scalars"].cpu().numpy(), rtol=2e-6)
        cfg = orc.Config(data_dim=data_dim, latent_dim=2, invariances=inv)
        o = orc.SVIOracle({k: v.cpu() for k, v in m.state_dict().items()}, cfg)
        ref = o.step(x, eps)
        small = b * int(np.prod(data_dim)) < 16384
        np.testing.assert_allclose(rec["scalars"][0].item(), ref, rtol=(5e-4 if small else 1e-4) if fused == 3 else RTOL_ELBO)
        np.testing.assert_allclose(zl.cpu().numpy(), o.last["z_loc"].detach().numpy(), rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(zs.cpu().numpy(), o.last["z_scale"].detach().numpy(), rtol=1e-4, atol=2e-6)
        for key in o.p:
            lo = eng._layout[key]
            err = rel_l2(rec["grad"][lo:lo + o.p[key].numel()].view_as(o.p[key]), o.last_grads[key])
            assert err < (5e-2 if fused == 3 else RTOL_GRAD), "%s per_image=%s grad %s: rel l2 error %.3e vs oracle" % (case, per_image, key, err)
        m2 = pv.models.iVAE(data_dim, 2, inv, seed=1, device="cuda")
        e2 = m2.engine(fused=fused); e2.enc_per_image = per_image; e2.enc_fold = False
        e2.loss_and_grads(x.cuda(), eps.cuda()); e2.adam_step()
        m3 = pv.models.iVAE(data_dim, 2, inv, seed=1, device="cuda")
        e3 = m3.engine(fused=fused); e3.enc_per_image = per_image; e3.enc_fold = False
        e3.loss_and_grads(x.cuda(), eps.cuda(), step=True)
        torch.cuda.synchronize()
        assert torch.equal(e2.flat, e3.flat)
        res[per_image] = rec
    a, c = res[True], res[False]
    np.testing.assert_allclose(a["zl"].cpu().numpy(), c["zl"].cpu().numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(a["zs"].cpu().numpy(), c["zs"].cpu().numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(a["scalars"].cpu().numpy(), c["scalars"].cpu().numpy(), rtol=1e-5 if fused == 2 else 2e-4)


def test_one_launch_encoder_at_large_batch(gpu_device):
    """The compact encoder's one-launch form (csrc/pv_encoder.hip pv_enc_kernel: first-layer tiles and the rest of the encoder in
    one grid, hand-off through per-tile flags) at a batch whose grid (4 600 workgroups) does not fit the device at once: the
    consumers are dispatched after every producer, so it must neither hang nor read a tile early.  Bit-identical to the
    two-launch form (the plan flag PV_PLAN_ENC_TWO_LAUNCH): encode and one training step, iVAE 28x28 (fcEncoderNet, nets/fc.py:51-61)."""
    g = torch.Generator().manual_seed(17)
    b = 8200                                            # (not a multiple of 16: the last row block is partial)
    x, eps = torch.rand(b, 28, 28, generator=g).cuda(), torch.randn(b, 5, generator=g).cuda()
    res = []
    for two in (1, 0):
        m = pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
        eng = m.engine(fused=3)
        eng.enc_two_launch = bool(two)                   # pv_ivae_plan.flags: PV_PLAN_ENC_TWO_LAUNCH (ABI v14)
        zl, zs = m.encode(x)
        eng.loss_and_grads(x, eps)
        torch.cuda.synchronize()
        res.append((torch.as_tensor(zl).clone(), torch.as_tensor(zs).clone(), eng.scalars.clone(), eng.grad.clone()))
    for a, c in zip(*res):
        assert torch.isfinite(a).all()
        assert torch.equal(a, c)


@pytest.mark.parametrize("b", [256, 8200])
def test_one_launch_encoder_fallback_is_exact(gpu_device, b):
    """HIP promises no dispatch order, so the one-launch encoder's consumers must not depend on their producers ever
    arriving: after a bounded number of polls a consumer computes its row block's first-layer tiles itself, with the
    producers' own K split (csrc/pv_encoder.hip enc_fwd_body).  PV_PLAN_ENC_NO_WAIT sends EVERY consumer down that
    path: encode, loss, every gradient and the parameters after a one-call step must be bit-identical to the normal run and
    to the two-launch form, and the fallback counter must show the path was taken."""
    import ctypes as C
    dbg = C.CDLL(_abi.LIB_PATH)
    dbg.pv_debug_enc_late_count.restype = C.c_longlong
    g = torch.Generator().manual_seed(23)
    x, eps = torch.rand(b, 28, 28, generator=g).cuda(), torch.randn(b, 5, generator=g).cuda()
    res, late = [], []
    if True:
        for two, spin in ((1, -1), (0, -1), (0, 0)):
            torch.cuda.synchronize()
            before = dbg.pv_debug_enc_late_count()
            m = pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
            eng = m.engine(fused=3)
            eng.enc_two_launch = bool(two)               # pv_ivae_plan.flags: PV_PLAN_ENC_TWO_LAUNCH
            eng.enc_no_wait = spin == 0                  # ... PV_PLAN_ENC_NO_WAIT (ABI v15; a process-wide debug setter before)
            eng.enc_fold = False                         # (round 5: at batch 256 the guide would ride in the decoder launch)
            eng.enc_per_image = False                    # (round 6: ... or run as one workgroup per image; this test is about the TILED encoder's hand-off)
            zl, zs = m.encode(x)
            eng.loss_and_grads(x, eps)
            torch.cuda.synchronize()
            rec = [torch.as_tensor(zl).clone(), torch.as_tensor(zs).clone(), eng.scalars.clone(), eng.grad.clone()]
            eng.loss_and_grads(x, eps, step=True)
            torch.cuda.synchronize()
            rec.append(torch.cat([p.detach().flatten() for p in m.parameters()]).clone())
            res.append(rec)
            late.append(dbg.pv_debug_enc_late_count() - before)
    assert late[0] == 0 and late[2] >= 2 * ((b + 15) // 16), late        # every consumer of both training launches fell back
    for a, c, d in zip(*res):
        assert torch.isfinite(a).all()
        assert torch.equal(a, c) and torch.equal(a, d)


def test_fused_1d_decoder_more_samples_than_workgroups(gpu_device):
    """The fused Conv1d decoder launches at most 2048 workgroups; beyond that a workgroup carries several samples one after
    the other through the same LDS buffers (csrc/pv_dec1d.hip: the grid-stride loop and its closing barrier).  A batch of
    2100 against the layer launches (plan flag PV_PLAN_NO_DEC1D): loss terms, every gradient, decode."""
    import ctypes as C
    dbg = C.CDLL(_abi.LIB_PATH)
    g = torch.Generator().manual_seed(13)
    b = 2100
    x, y, eps = torch.rand(b, 1, 16, 16, generator=g), torch.rand(b, 1, 16, generator=g), torch.randn(b, 2, generator=g)
    res = []
    try:
        for on in (0, 1):
            m = pv.models.VED((16, 16), (16,), latent_dim=2, hidden_dim_e=[(32,), (64, 64)], hidden_dim_d=[(64, 64), (32,)],
                              seed=1, device="cuda")
            eng = m.engine()
            eng.dec1d = bool(on)                         # pv_ved_plan.flags: PV_PLAN_NO_DEC1D (ABI v15)
            eng.loss_and_grads(x.cuda(), eps.cuda(), 1.0, y.cuda())
            res.append((eng.scalars.clone(), eng.grad.clone(), m.decode(torch.randn(b, 2, generator=torch.Generator().manual_seed(3)))))
    finally:
        pass
    (s0, g0, d0), (s1, g1, d1) = res
    assert torch.isfinite(g1).all()
    np.testing.assert_allclose(s1.cpu().numpy(), s0.cpu().numpy(), rtol=3e-6)
    assert rel_l2(g1, g0) < 3e-6
    assert torch.allclose(d1, d0, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("family", ["ivae", "ved", "ivae_folded_guide"])
def test_steps_replay_from_a_captured_graph(gpu_device, family):
    """A step captured into a graph (torch.cuda.graph on the stream the library launches on) replays correctly: under
    capture the library keeps to forms that can be replayed — the encoder as its two launches (the one-launch form hands its
    tiles over through a per-CALL flag value, csrc/pv_encoder.hip), no side stream (csrc/pv_side.hip).  Replays with new
    inputs match eager calls bit for bit (the default forms compute the same numbers in the same order)."""
    g = torch.Generator().manual_seed(21)
    if family == "ivae":
        model = pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
        eng = model.engine(fused=2)
        xs = [torch.rand(32, 28, 28, generator=g).cuda() for _ in range(3)]
        es = [torch.randn(32, model.z_dim, generator=g).cuda() for _ in range(3)]
        call = lambda x, e, y: eng.loss_and_grads(x, e)
        ys = [None] * 3
    elif family == "ivae_folded_guide":
        # (round 5) batch == decoder grid at the throughput precision: the decoder launch hosts the guide — no cross-workgroup
        # hand-off in it, so the SAME form runs under capture
        nb = torch.cuda.get_device_properties(0).multi_processor_count
        model = pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
        eng = model.engine(fused=3)
        xs = [torch.rand(nb, 28, 28, generator=g).cuda() for _ in range(3)]
        es = [torch.randn(nb, model.z_dim, generator=g).cuda() for _ in range(3)]
        call = lambda x, e, y: eng.loss_and_grads(x, e)
        ys = [None] * 3
        assert _abi.lib().pv_ivae_guide_folds(C.byref(eng._plan(nb))) == 1
    else:
        model = pv.models.VED((32, 32), (32,), latent_dim=2, seed=1, device="cuda")
        eng = model.engine()
        xs = [torch.rand(8, 1, 32, 32, generator=g).cuda() for _ in range(3)]
        es = [torch.randn(8, 2, generator=g).cuda() for _ in range(3)]
        ys = [torch.rand(8, 1, 32, generator=g).cuda() for _ in range(3)]
        call = lambda x, e, y: eng.loss_and_grads(x, e, 1.0, y)
    if True:
        # eager references (the default forms: same arithmetic, bit for bit, as the replayable ones)
        want = []
        for x, e, y in zip(xs, es, ys):
            call(x, e, y)
            want.append((eng.scalars.clone(), eng.grad.clone()))
        sx, se = xs[0].clone(), es[0].clone()
        sy = ys[0].clone() if ys[0] is not None else None
        call(sx, se, sy)                                   # (workspace allocated, library streams created)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph):
                call(sx, se, sy)
        except Exception as exc:                            # capture itself unsupported in this build: nothing to check
            pytest.skip("graph capture of the step failed: %s" % exc)
        for k in (1, 2, 0):
            sx.copy_(xs[k]); se.copy_(es[k])
            if sy is not None:
                sy.copy_(ys[k])
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(eng.scalars, want[k][0]), "replay %d: loss terms differ" % k
            assert torch.equal(eng.grad, want[k][1]), "replay %d: gradients differ" % k


def test_class_onehot_rejected_where_undefined(gpu_device):
    """ The sampled-class objective exists for the vanilla decoder only; a direct engine call with class_onehot
    on a jiVAE WITH invariances (fused or layered path) is refused instead of silently running the enumerated objective."""
    for fused in (0, 2):
        model = pv.models.jiVAE((8, 8), 2, 3, ["r"], seed=1, device="cuda")
        eng = model.engine(fused=fused)
        x, eps = torch.rand(5, 8, 8).cuda(), torch.randn(5, model.z_dim).cuda()
        y = torch.zeros(5, 3, device="cuda")
        y[:, 0] = 1.0
        with pytest.raises(_abi.PvError):
            eng.loss_and_grads(x, eps, class_onehot=y)
        eng.loss_and_grads(x, eps)                                       # the enumerated objective still runs


def test_conv_stack_with_other_pooling_falls_back_to_torch(gpu_device):
    """ A stand-alone conv stack whose pooling is not the reference's 2x / stride-2 window is NOT handed to the
    library (which would assume 2x pooling): the torch modules run it; the reference's own geometry still takes the library."""
    from pyroved_amd import ops
    fe = pv.nets.FeatureExtractor(2, 1, [(16,), (32,)]).cuda()
    x = torch.rand(3, 1, 16, 16, device="cuda")
    assert ops.conv_stack_supported(fe, x)
    want = fe(x)
    pools = [i for i, m in enumerate(fe.layers) if isinstance(m, torch.nn.MaxPool2d)]
    assert pools
    fe.layers[pools[0]] = torch.nn.MaxPool2d(3, stride=2, padding=1)
    assert not ops.conv_stack_supported(fe, x)
    got = fe(x)                                                          # torch composition
    assert got.shape[0] == 3 and torch.isfinite(got).all() and want.shape[0] == 3
    assert not ops.conv_stack_supported(pv.nets.FeatureExtractor(2, 1, [(16,), (32,)]).cuda(), torch.rand(3, 2, 16, 16, device="cuda"))


@pytest.fixture(params=[8, 4, 0], ids=["w8x3", "w4x3", "old4"])
def force_w8x3(request):
    """Forces one split-precision decoder kernel for every launch (training and forward-only): pv_sdec_fused_w8x3.hip with
    8 or 4 waves, or (0) the round-1/2 kernel of pv_sdec_fused_bf16.hip; by default the choice depends on launch kind and size."""
    from pyroved_amd.engine import IVAEEngine
    if request.param in (8, 4) and not _experiments_build():
        pytest.skip("the training forms of pv_sdec_fused_w8x3.hip are in the experiments build only (PV_LIB_PATH)")
    IVAEEngine.dec_kernel = request.param if request.param else 1         # (1: the bf16 three-product kernel)
    try:
        yield
    finally:
        IVAEEngine.dec_kernel = 0


@pytest.mark.parametrize("name", sorted(W8_SMALL) + ["ivae_28x28_r_b128", "ivae_28x28_rt_b256"])
def test_w8x3_kernel_forced_vs_golden_and_oracle(gpu_device, force_w8x3, name):
    """The 8-wave SPLIT-PRECISION decoder kernel (round 3: pv_sdec_fused_w8x3.hip, the default fused=2 path once a
    workgroup has >= 6 units) forced on the small / odd fixtures — partial tiles with most waves idle and half tiles never
    staged, 1-D data, ragged rows, non-unit KL scale, randn and saturated inputs — and on the two full-size fixtures,
    at the fp32-class bars: ELBO terms 2e-5 vs the reference's recorded numbers, every gradient 1e-4 vs the oracle."""
    gold = load_golden(name)
    meta = meta_of(gold)
    if meta["batch"] > 64:
        torch.set_num_threads(8)
    model, cfg, eng = build(meta, 2)
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    for k in range(meta["steps"]):
        pre = "s%d" % k
        eps = torch.from_numpy(gold[pre + ".eps"])
        eng.loss_and_grads(x.cuda(), eps.cuda(), meta["beta"])
        s = eng.scalars.cpu().numpy()
        np.testing.assert_allclose(s[0], float(gold[pre + ".loss"]), rtol=RTOL_ELBO, err_msg="loss")
        np.testing.assert_allclose(s[1], float(gold[pre + ".term.model.obs"]), rtol=RTOL_ELBO)
        o.step(x, eps, meta["beta"])
        for key in o.p:
            err = rel_l2(eng.grad_of(key), o.last_grads[key])
            assert err < RTOL_GRAD, "step %d grad %s: rel l2 error %.3e vs oracle" % (k, key, err)
        # bit-reproducible (fixed-order reductions), and the forward-only launch (evaluate) agrees
        g0 = {key: eng.grad_of(key).clone() for key in o.p}
        eng.loss_and_grads(x.cuda(), eps.cuda(), meta["beta"])
        for key in o.p:
            assert torch.equal(g0[key], eng.grad_of(key)), key
        eng.loss_and_grads(x.cuda(), eps.cuda(), meta["beta"], want_grads=False)
        np.testing.assert_allclose(eng.scalars[0].item(), s[0], rtol=1e-6)
        eng.loss_and_grads(x.cuda(), eps.cuda(), meta["beta"])
        eng.adam_step()
        model.load_state_dict({k_: v_.detach() for k_, v_ in o.p.items()})


@pytest.fixture(params=[28, 21, 41], ids=["h231", "h221", "w8h221"])
def force_h2(request):
    """Forces one fp16 build of the fp32-class decoder kernel (round 4: pv_sdec_fused_bf16_kernel<.., FB_P_H231 / H221>) for every
    training launch; by default H231 runs from 16 384 decoder rows up.  41: the H221 arithmetic in the 8-wave geometry
    (pv_sdec_fused_w8h.hip — a measured negative, profiles/r04e_w8h_experiments.txt — selectable for A/B runs)."""
    from pyroved_amd.engine import IVAEEngine
    if request.param == 41 and not _experiments_build():
        pytest.skip("pv_sdec_fused_w8h.hip is in the experiments build only (PV_LIB_PATH)")
    IVAEEngine.dec_kernel = request.param
    try:
        yield request.param
    finally:
        IVAEEngine.dec_kernel = 0


def h2_grad_tol(rows, kind):
    """The fp16 builds round activations and dL/dpre to ONE 16-bit piece where the rounding errors are independent from row to
    row: what is left in a gradient shrinks with the rows it sums.  At the sizes the launcher selects H231 for (>= 16 384 rows)
    the fp32-class bar holds; forced on toy problems the bar follows 1 / sqrt(rows)."""
    if kind == 28 and rows >= 16384:
        return RTOL_GRAD
    return max(RTOL_GRAD if kind == 28 else (2.5 if kind == 21 else 3.5) * RTOL_GRAD, 0.03 / rows ** 0.5)


@pytest.mark.parametrize("name", sorted(W8_SMALL) + ["ivae_28x28_r_b128", "ivae_28x28_rt_b256"])
def test_h2_kernel_forced_vs_golden_and_oracle(gpu_device, force_h2, name):
    """The fp16 builds (weights as two exact scaled pieces, activations one piece; H231: dL/dpre split in both dgrads, H221: one
    piece everywhere) forced on the small / odd fixtures — partial tiles, 1-D data, ragged rows, non-unit KL scale, randn and
    saturated inputs (rows of dL/dlogit ~ 1e-7: the per-row exponent path) — and on the two full-size fixtures.  ELBO terms at
    the fp32-class bar everywhere; gradients at 1e-4 where the launcher would choose the build, 1 / sqrt(rows) below."""
    gold = load_golden(name)
    meta = meta_of(gold)
    if meta["batch"] > 64:
        torch.set_num_threads(8)
    model, cfg, eng = build(meta, 2)
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    rows = meta["batch"] * int(np.prod(meta["data_dim"]))
    tol = h2_grad_tol(rows, force_h2)
    for k in range(meta["steps"]):
        pre = "s%d" % k
        eps = torch.from_numpy(gold[pre + ".eps"])
        eng.loss_and_grads(x.cuda(), eps.cuda(), meta["beta"])
        s = eng.scalars.cpu().numpy()
        np.testing.assert_allclose(s[0], float(gold[pre + ".loss"]), rtol=RTOL_ELBO, err_msg="loss")
        np.testing.assert_allclose(s[1], float(gold[pre + ".term.model.obs"]), rtol=RTOL_ELBO)
        o.step(x, eps, meta["beta"])
        for key in o.p:
            err = rel_l2(eng.grad_of(key), o.last_grads[key])
            assert err < tol, "step %d grad %s: rel l2 error %.3e vs oracle (bar %.1e)" % (k, key, err, tol)
        g0 = {key: eng.grad_of(key).clone() for key in o.p}
        eng.loss_and_grads(x.cuda(), eps.cuda(), meta["beta"])
        for key in o.p:
            assert torch.equal(g0[key], eng.grad_of(key)), key               # bit-reproducible
        eng.adam_step()
        model.load_state_dict({k_: v_.detach() for k_, v_ in o.p.items()})


@pytest.mark.parametrize("vname", ["gauss_rts", "gauss_nosig_r", "cbern_16x16_r", "cdim3_rt", "1d32_t_cdim2", "rect_12x20_rts", "priors_rts"])
def test_h2_kernel_model_variants(gpu_device, force_h2, vname):
    """Likelihoods (the Gaussian's exponent bias, ContinuousBernoulli), class conditioning, custom priors, 1-D + c_dim and
    rectangular data on the forced fp16 builds vs the oracle: ELBO 2e-5, gradients at the 1 / sqrt(rows) bar."""
    kw = dict(VARIANTS[vname])
    data_dim, inv, latent_dim = kw.pop("data_dim"), kw.pop("invariances"), kw.pop("latent_dim", 2)
    model = pv.models.iVAE(data_dim, latent_dim, inv, seed=3, device="cuda", **kw)
    cfg = orc.Config(data_dim=data_dim, latent_dim=latent_dim, invariances=inv, c_dim=kw.get("c_dim", 0),
                     sampler=kw.get("sampler_d", "bernoulli"), sigmoid_d=kw.get("sigmoid_d", True),
                     dx_prior=kw.get("dx_prior", 0.1), dy_prior=kw.get("dy_prior"), sc_prior=kw.get("sc_prior", 0.1),
                     decoder_sig=kw.get("decoder_sig", 0.5))
    eng = model.engine(fused=2)
    odt = torch.float64 if kw.get("sampler_d") == "continuous_bernoulli" else torch.float32
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg, dtype=odt)
    b = 7
    g = torch.Generator().manual_seed(11)
    x = torch.rand(b, *data_dim, generator=g)
    y = None
    if cfg.c_dim:
        y = torch.zeros(b, cfg.c_dim)
        y[torch.arange(b), torch.randint(0, cfg.c_dim, (b,), generator=g)] = 1.0
    eps = torch.randn(b, cfg.z_dim, generator=g)
    assert eng.uses_fused(b)
    eng.loss_and_grads(x.cuda(), eps.cuda(), 1.7, None if y is None else y.cuda())
    o.step(x, eps, 1.7, y)
    atol = 1e-6 * b * int(np.prod(data_dim)) if odt == torch.float64 else 0.0
    np.testing.assert_allclose(eng.scalars[0].item(), o.last["loss"].item(), rtol=RTOL_ELBO, atol=atol)
    tol = h2_grad_tol(b * int(np.prod(data_dim)), force_h2) * (3.0 if odt == torch.float64 else 1.0)
    for key in o.p:
        err = rel_l2(eng.grad_of(key), o.last_grads[key])
        assert err < tol, "%s grad %s: rel l2 error %.3e (bar %.1e)" % (vname, key, err, tol)


@pytest.mark.parametrize("scale", [1e-4, 1.0, 3e3])
def test_h2_kernel_gaussian_data_scale(gpu_device, force_h2, scale):
    """The fp16 builds must not care about the DATA's scale: with a Gaussian likelihood dL/dlogit ~ (x - loc) / sig^2 takes
    whatever magnitude the observations have.  The per-row exponent keeps every 16-bit operand of the dgrad chain O(1) and
    folds 2^e into the staged activations, so observations 1e-4 .. 3e3 times the usual [0, 1] (30 binades around the
    likelihood's own scale are exact) give the same relative errors."""
    data_dim, inv = (16, 16), ["r", "t", "s"]
    kw = dict(sampler_d="gaussian", sigmoid_d=False, decoder_sig=0.5)
    model = pv.models.iVAE(data_dim, 2, inv, seed=5, device="cuda", **kw)
    cfg = orc.Config(data_dim=data_dim, latent_dim=2, invariances=inv, sampler="gaussian", sigmoid_d=False, decoder_sig=0.5)
    eng = model.engine(fused=2)
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg, dtype=torch.float64)
    b = 64
    g = torch.Generator().manual_seed(3)
    x = torch.rand(b, *data_dim, generator=g) * scale
    eps = torch.randn(b, cfg.z_dim, generator=g)
    eng.loss_and_grads(x.cuda(), eps.cuda())
    o.step(x, eps)
    np.testing.assert_allclose(eng.scalars[0].item(), o.last["loss"].item(), rtol=RTOL_ELBO)
    tol = h2_grad_tol(b * 256, force_h2)
    for key in o.p:
        gp = eng.grad_of(key)
        assert torch.isfinite(gp).all(), key
        err = rel_l2(gp, o.last_grads[key])
        assert err < tol, "scale %g grad %s: rel l2 error %.3e (bar %.1e)" % (scale, key, err, tol)


def test_h2_kernel_jivae_and_row_weights(gpu_device, force_h2):
    """jiVAE (K enumerated passes, rows weighted by alpha — the per-row exponent carries the weight —, observations addressed
    modulo B*N) on the forced fp16 builds vs the oracle from the recorded noise; decoder tensors at the size bar, the class-logit
    path at jiVAE's own bars."""
    gold = load_golden("jivae_28x28_r_k10_b16")
    meta = jmeta_of(gold)
    model = pv.models.jiVAE(meta["data_dim"], meta["latent_dim"], meta["discrete_dim"], meta["invariances"], seed=1, device="cuda")
    eng = model.engine(fused=2)
    cfg = orc.Config(data_dim=meta["data_dim"], latent_dim=meta["latent_dim"], invariances=meta["invariances"],
                     discrete_dim=meta["discrete_dim"])
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    eps = torch.from_numpy(gold["s0.eps"])
    eng.loss_and_grads(x.cuda(), eps.cuda(), meta["beta"])
    np.testing.assert_allclose(eng.scalars[0].item(), float(gold["s0.loss"]), rtol=RTOL_ELBO)
    o.step(x, eps, meta["beta"])
    for key in o.p:
        tol = jivae_grad_tol(key)
        if tol is None:
            continue
        if force_h2 != 28:
            tol = max(tol, 1e-3)
        err = rel_l2(eng.grad_of(key), o.last_grads[key])
        assert err < tol, "grad %s: rel l2 error %.3e" % (key, err)


@pytest.mark.parametrize("vname", ["gauss_rts", "gauss_nosig_r", "cbern_16x16_r", "cdim3_rt", "1d32_t_cdim2", "rect_12x20_rts", "priors_rts"])
def test_w8_kernel_model_variants(gpu_device, force_w8, vname):
    """Likelihoods, class conditioning, custom priors, 1-D + c_dim and rectangular data on the forced 8-wave kernel vs the
    oracle (ELBO 5e-4 on these toy sizes, gradients 3e-2: the mixed-precision mode's bars)."""
    kw = dict(VARIANTS[vname])
    data_dim, inv, latent_dim = kw.pop("data_dim"), kw.pop("invariances"), kw.pop("latent_dim", 2)
    model = pv.models.iVAE(data_dim, latent_dim, inv, seed=3, device="cuda", **kw)
    cfg = orc.Config(data_dim=data_dim, latent_dim=latent_dim, invariances=inv, c_dim=kw.get("c_dim", 0),
                     sampler=kw.get("sampler_d", "bernoulli"), sigmoid_d=kw.get("sigmoid_d", True),
                     dx_prior=kw.get("dx_prior", 0.1), dy_prior=kw.get("dy_prior"), sc_prior=kw.get("sc_prior", 0.1),
                     decoder_sig=kw.get("decoder_sig", 0.5))
    eng = model.engine(fused=3)
    odt = torch.float64 if kw.get("sampler_d") == "continuous_bernoulli" else torch.float32
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg, dtype=odt)
    b = 7
    g = torch.Generator().manual_seed(11)
    x = torch.rand(b, *data_dim, generator=g)
    y = None
    if cfg.c_dim:
        y = torch.zeros(b, cfg.c_dim)
        y[torch.arange(b), torch.randint(0, cfg.c_dim, (b,), generator=g)] = 1.0
    eps = torch.randn(b, cfg.z_dim, generator=g)
    assert eng.uses_fused(b)
    eng.loss_and_grads(x.cuda(), eps.cuda(), 1.7, None if y is None else y.cuda())
    o.step(x, eps, 1.7, y)
    # (ContinuousBernoulli: the loss is a remainder of a few units of B*N terms of size ~0.7 — bf16 operands leave ~3e-6 of
    #  that sum; judged on its scale)
    atol = 1e-5 * b * int(np.prod(data_dim)) if odt == torch.float64 else 0.0
    np.testing.assert_allclose(eng.scalars[0].item(), o.last["loss"].item(), rtol=5e-4, atol=atol)
    for key in o.p:
        err = rel_l2(eng.grad_of(key), o.last_grads[key])
        assert err < (6e-2 if odt == torch.float64 else 3e-2), "%s grad %s: rel l2 error %.3e" % (vname, key, err)


def test_w8_kernel_jivae_and_row_weights(gpu_device, force_w8):
    """jiVAE (K enumerated passes, rows weighted by alpha, observations addressed modulo B*N) on the forced 8-wave kernel:
    gradients vs the oracle from the recorded noise."""
    gold = load_golden("jivae_8x8_rts_k4_b6")
    meta = jmeta_of(gold)
    model = pv.models.jiVAE(meta["data_dim"], meta["latent_dim"], meta["discrete_dim"], meta["invariances"], seed=1, device="cuda")
    eng = model.engine(fused=3)
    cfg = orc.Config(data_dim=meta["data_dim"], latent_dim=meta["latent_dim"], invariances=meta["invariances"],
                     discrete_dim=meta["discrete_dim"])
    o = orc.SVIOracle({k: v.cpu() for k, v in model.state_dict().items()}, cfg)
    x = make_x(meta["xkind"], meta["batch"], meta["data_dim"])
    eps = torch.from_numpy(gold["s0.eps"])
    eng.loss_and_grads(x.cuda(), eps.cuda(), meta["beta"])
    np.testing.assert_allclose(eng.scalars[0].item(), float(gold["s0.loss"]), rtol=5e-4)
    o.step(x, eps, meta["beta"])
    for key in o.p:
        err = rel_l2(eng.grad_of(key), o.last_grads[key])
        assert err < 5e-2, "grad %s: rel l2 error %.3e" % (key, err)
