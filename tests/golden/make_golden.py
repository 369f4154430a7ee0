#!/usr/bin/env python
"""
make_golden.py — generates the golden fixtures under tests/golden/*.npz by
running the REFERENCE's own code (/root/reference/pyroved) in this container.

The reference cannot be imported as-is (pyro-ppl / torchvision are not
installed and there is no network), so `_minipyro.install()` registers a
minimal restatement of the Pyro machinery the hot path runs through; every
line of pyroVED itself — `models.iVAE.model/guide` (models/ivae.py:165-221),
`baseVAE._split_latent` (models/base.py:97-119), `nets.fcEncoderNet /
sDecoderNet / fcDecoderNet` (nets/fc.py), `utils.transform_coordinates`
(utils/coord.py:47-88), `trainers.SVItrainer` (trainers/svi.py:64-175),
`utils.init_dataloader` (utils/data.py:6-38) — executes for real.

Run (only where /root/reference exists):
    python tests/golden/make_golden.py
The .npz files are data (inputs / expected outputs), committed; the reference
never travels to the GPU box.

What a fixture holds (per case):
  meta.*            the case definition (model kwargs, batch, seeds)
  init.<param>.*    digest of every initial parameter (sum, l2, strided sample)
  s<k>.loss, s<k>.term.*   loss and the three ELBO terms of SVI step k
  s<k>.eps / z_loc / z_scale / z       the guide's draw
  s0.loc            decoder output (probabilities) of step 0
  s<k>.grad.<param>.*   digest of dLoss/dparam at step k (before the update)
  s<k>.param.<param>.*  digest of the parameter after step k's Adam update
  full.*            (selected tiny cases) full tensors instead of digests
  epochs.*          loss_history of SVItrainer.step(train[, test]) epochs
  enc.* / dec.*     encode()/decode() outputs after training
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _minipyro  # noqa: E402

_minipyro.install()
sys.path.insert(0, "/root/reference")
from pyroved import models, trainers, utils  # noqa: E402

SAMPLE = 64


def digest(t):
    """sum / l2 in float64 + a fixed strided sample: pins a tensor in ~70 numbers."""
    a = t.detach().double().flatten()
    n = a.numel()
    idx = torch.linspace(0, n - 1, min(SAMPLE, n)).round().long()
    return {"shape": np.array(t.shape, dtype=np.int64),
            "sum": np.float64(a.sum().item()),
            "l2": np.float64(a.norm().item()),
            "sample": t.detach().flatten()[idx].numpy().copy()}


def put(out, prefix, d):
    for k, v in d.items():
        out[prefix + "." + k] = v


def make_x(kind, n, data_dim, seed=0):
    g = torch.Generator().manual_seed(seed)
    if kind == "rand":
        return torch.rand(n, *data_dim, generator=g)
    if kind == "randn":            # what the reference's own trainer tests feed (tests/test_trainers.py:29)
        return torch.randn(n, *data_dim, generator=g)
    if kind == "blobs":            # MNIST-like: sparse bright blobs, many saturated pixels
        x = torch.rand(n, *data_dim, generator=g)
        return (x > 0.8).float() * torch.rand(n, *data_dim, generator=g)
    raise KeyError(kind)


def run_steps(name, data_dim, invariances, batch, steps=3, latent_dim=2, xkind="rand",
              full=False, model_kw=None, step_kw=None, c_dim=0):
    model_kw = dict(model_kw or {})
    conv_enc = model_kw.pop("conv_encoder", None)
    step_kw = dict(step_kw or {})
    out = {}
    out["meta.data_dim"] = np.array(data_dim)
    out["meta.invariances"] = np.array("".join(invariances) if invariances else "")
    out["meta.batch"] = np.int64(batch)
    out["meta.latent_dim"] = np.int64(latent_dim)
    out["meta.c_dim"] = np.int64(c_dim)
    out["meta.xkind"] = np.array(xkind)
    out["meta.steps"] = np.int64(steps)
    out["meta.scale_factor"] = np.float64(step_kw.get("scale_factor", 1.0))
    for k, v in model_kw.items():
        out["meta.model_kw." + k] = np.array(v)
    model = models.iVAE(data_dim, latent_dim, invariances, c_dim=c_dim, seed=1, device="cpu", **model_kw)
    if conv_enc is not None:
        # a convolutional encoder installed the way the reference documents it (models/base.py:173-177); the
        # module is built right after the model, on the same RNG stream
        from pyroved.nets import convEncoderNet
        model.set_encoder(convEncoderNet(data_dim, latent_dim=model.z_dim, hidden_dim=conv_enc))
        out["meta.conv_encoder"] = np.array(str(conv_enc))
    names = {id(p): n for n, p in model.named_parameters()}
    for n, p in model.named_parameters():
        put(out, "init." + n, digest(p))
        if full:
            out["full.init." + n] = p.detach().numpy().copy()
    x = make_x(xkind, batch, data_dim)
    if conv_enc is not None:
        x = x.unsqueeze(1)                 # conv encoders take (B, C, H, W)
    y = None
    if c_dim:
        y = utils.to_onehot(torch.arange(batch) % c_dim, c_dim)
        # the reference's Concat broadcasts the LEADING axes of [x, y] (utils/nn.py:62-74): class-conditioned models
        # take flattened samples (B, H*W), as its semi-supervised trainers feed them (tests/test_trainers.py:57-75)
        x = x.flatten(1)
    trainer = trainers.SVItrainer(model, seed=1, device="cpu")
    for k in range(steps):
        grads = {}
        real_optim = trainer.svi.optim

        def spy(params, _real=real_optim, _g=grads):
            for p in params:
                _g[names[id(p)]] = p.grad.detach().clone()
            _real(params)
        trainer.svi.optim = spy
        args = (x,) if y is None else (x, y)
        loss = trainer.svi.step(*args, **step_kw)
        trainer.svi.optim = real_optim
        tap = _minipyro.tap()
        pre = "s%d" % k
        out[pre + ".loss"] = np.float64(loss)
        for tn, tv in tap["terms"].items():
            out[pre + ".term." + tn] = np.float64(tv.item())
        gfn = tap["guide_fns"]["latent"].base_dist
        out[pre + ".eps"] = tap["latent.eps"].numpy().copy()
        out[pre + ".z_loc"] = gfn.loc.detach().numpy().copy()
        out[pre + ".z_scale"] = gfn.scale.detach().numpy().copy()
        out[pre + ".z"] = tap["sites"]["guide.latent"].numpy().copy()
        if k == 0:
            # decoder output of step 0: recompute with the pre-update weights is not possible
            # after the step, so it is captured from the model trace's obs distribution
            pass
        for n, g in grads.items():
            put(out, pre + ".grad." + n, digest(g))
            if full:
                out["full." + pre + ".grad." + n] = g.numpy().copy()
        for n, p in model.named_parameters():
            put(out, pre + ".param." + n, digest(p))
            if full and k == steps - 1:
                out["full." + pre + ".param." + n] = p.detach().numpy().copy()
    # inference API after training (pins "reconstructions")
    enc_args = (x,) if y is None else (x, y)
    z_loc, z_scale = model.encode(*enc_args)
    out["enc.z_loc"] = z_loc.numpy().copy()
    out["enc.z_scale"] = z_scale.numpy().copy()
    zc = z_loc[:, -latent_dim:]
    dec = model.decode(zc) if y is None else model.decode(zc, y)
    out["dec.loc"] = dec.numpy().copy()
    if invariances and len(data_dim) == 2:
        dec2 = model.decode(zc[:2], angle=torch.tensor(0.3), shift=torch.tensor([0.1, -0.2]),
                            scale=torch.tensor(1.2)) if y is None else None
        if dec2 is not None:
            out["dec.loc_ats"] = dec2.numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "loss0=%.6f" % out["s0.loss"], "keys=%d" % len(out))


def run_jsteps(name, data_dim, invariances, discrete_dim, batch, steps=3, latent_dim=2, xkind="rand",
               scale_factor=None, full=False, enumerate_parallel=True):
    """jiVAE (models/jivae.py:109-220) through SVItrainer(enumerate_parallel=True) (trainers/svi.py:83-90):
    TraceEnum_ELBO with the guide's OneHotCategorical site enumerated in parallel."""
    out = {}
    out["meta.data_dim"] = np.array(data_dim)
    out["meta.invariances"] = np.array("".join(invariances) if invariances else "")
    out["meta.batch"] = np.int64(batch)
    out["meta.latent_dim"] = np.int64(latent_dim)
    out["meta.discrete_dim"] = np.int64(discrete_dim)
    out["meta.xkind"] = np.array(xkind)
    out["meta.steps"] = np.int64(steps)
    sf = [1.0, 1.0] if scale_factor is None else (list(scale_factor) if isinstance(scale_factor, (list, tuple))
                                                   else [scale_factor, scale_factor])
    out["meta.scale_factor"] = np.array(sf, dtype=np.float64)
    out["meta.enumerate_parallel"] = np.int64(enumerate_parallel)
    model = models.jiVAE(data_dim, latent_dim, discrete_dim, invariances, seed=1, device="cpu")
    names = {id(p): n for n, p in model.named_parameters()}
    for n, p in model.named_parameters():
        put(out, "init." + n, digest(p))
    x = make_x(xkind, batch, data_dim)
    # enumerate_parallel=False is the trainer's DEFAULT (trainers/svi.py:66): Trace_ELBO with the class drawn by the
    # guide and a score-function gradient for its logits
    trainer = trainers.SVItrainer(model, enumerate_parallel=enumerate_parallel, seed=1, device="cpu")
    step_kw = {} if scale_factor is None else {"scale_factor": scale_factor}
    for k in range(steps):
        grads = {}
        real_optim = trainer.svi.optim

        def spy(params, _real=real_optim, _g=grads):
            for p in params:
                _g[names[id(p)]] = p.grad.detach().clone()
            _real(params)
        trainer.svi.optim = spy
        loss = trainer.svi.step(x, **step_kw)
        trainer.svi.optim = real_optim
        tap = _minipyro.tap()
        pre = "s%d" % k
        out[pre + ".loss"] = np.float64(loss)
        for tn, tv in tap["enum_terms" if enumerate_parallel else "terms"].items():
            out[pre + ".term." + tn] = np.float64(tv.item())
        gfn = tap["guide_fns"]["latent_cont"].base_dist
        out[pre + ".eps"] = tap["latent_cont.eps"].numpy().copy()
        out[pre + ".z_loc"] = gfn.loc.detach().numpy().copy()
        out[pre + ".z_scale"] = gfn.scale.detach().numpy().copy()
        out[pre + ".z"] = tap["sites"]["guide.latent_cont"].numpy().copy()
        if enumerate_parallel:
            out[pre + ".alpha"] = tap["enum_weights"].t().numpy().copy()          # (B, K) = q(k | x_b)
        else:
            out[pre + ".alpha"] = tap["guide_fns"]["latent_disc"].probs.detach().numpy().copy()
            out[pre + ".y"] = tap["sites"]["guide.latent_disc"].numpy().copy()      # the drawn one-hot classes (B, K)
            out[pre + ".log_r"] = tap["log_r"].numpy().copy()
            # generator state right after the step (pins what the draws consumed: normal_ then multinomial)
            out[pre + ".rng_probe"] = torch.get_rng_state()[:64].numpy().copy()
        for n, g in grads.items():
            put(out, pre + ".grad." + n, digest(g))
            if full:
                out["full." + pre + ".grad." + n] = g.numpy().copy()
        for n, p in model.named_parameters():
            put(out, pre + ".param." + n, digest(p))
    z_loc, z_scale, logits = model.encode(x, logits=True)
    out["enc.z_loc"] = z_loc.numpy().copy()
    out["enc.z_scale"] = z_scale.numpy().copy()
    out["enc.alpha"] = logits.numpy().copy()
    _, _, classes = model.encode(x)
    out["enc.classes"] = classes.numpy().copy()
    zc = z_loc[:, -latent_dim:]
    yy = utils.to_onehot(torch.arange(batch) % discrete_dim, discrete_dim)
    out["dec.loc"] = model.decode(zc, yy).numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "loss0=%.6f" % out["s0.loss"], "keys=%d" % len(out))


def run_ved_steps(name, input_dim, output_dim, batch, steps=3, latent_dim=2, model_kw=None, scale_factor=None):
    """VED (models/ved.py:89-163: conv encoder -> z -> conv decoder, Bernoulli on the target) through
    SVItrainer.svi.step(x, y): loss, ELBO terms, eps, z, gradients and parameters per step, then encode/decode."""
    model_kw = dict(model_kw or {})
    out = {}
    out["meta.input_dim"] = np.array(input_dim)
    out["meta.output_dim"] = np.array(output_dim)
    out["meta.batch"] = np.int64(batch)
    out["meta.latent_dim"] = np.int64(latent_dim)
    out["meta.steps"] = np.int64(steps)
    out["meta.scale_factor"] = np.float64(1.0 if scale_factor is None else scale_factor)
    for k, v in model_kw.items():
        out["meta.model_kw." + k] = np.array(str(v))
    model = models.VED(input_dim, output_dim, latent_dim=latent_dim, seed=1, **model_kw)
    names = {id(p): n for n, p in model.named_parameters()}
    for n, p in model.named_parameters():
        put(out, "init." + n, digest(p))
    g = torch.Generator().manual_seed(0)
    in_ch, out_ch = model_kw.get("input_channels", 1), model_kw.get("output_channels", 1)
    x = torch.rand(batch, in_ch, *input_dim, generator=g)
    y = torch.rand(batch, out_ch, *output_dim, generator=g)
    out["x"], out["y"] = x.numpy().copy(), y.numpy().copy()
    trainer = trainers.SVItrainer(model, seed=1, device="cpu")
    step_kw = {} if scale_factor is None else {"scale_factor": scale_factor}
    for k in range(steps):
        grads = {}
        real_optim = trainer.svi.optim

        def spy(params, _real=real_optim, _g=grads):
            for p in params:
                _g[names[id(p)]] = p.grad.detach().clone()
            _real(params)
        trainer.svi.optim = spy
        loss = trainer.svi.step(x, y, **step_kw)
        trainer.svi.optim = real_optim
        tap = _minipyro.tap()
        pre = "s%d" % k
        out[pre + ".loss"] = np.float64(loss)
        for tn, tv in tap["terms"].items():
            out[pre + ".term." + tn] = np.float64(tv.item())
        gfn = tap["guide_fns"]["z"].base_dist
        out[pre + ".eps"] = tap["z.eps"].numpy().copy()
        out[pre + ".z_loc"] = gfn.loc.detach().numpy().copy()
        out[pre + ".z_scale"] = gfn.scale.detach().numpy().copy()
        out[pre + ".z"] = tap["sites"]["guide.z"].numpy().copy()
        for n, gr in grads.items():
            put(out, pre + ".grad." + n, digest(gr))
        for n, p in model.named_parameters():
            put(out, pre + ".param." + n, digest(p))
    z_loc, z_scale = model.encode(x)
    out["enc.z_loc"], out["enc.z_scale"] = z_loc.numpy().copy(), z_scale.numpy().copy()
    out["dec.loc"] = model.decode(z_loc).numpy().copy()
    if model_kw.get("batchnorm"):
        # encode() / decode() left the module in eval() mode (models/ved.py:178,193): the next training step runs with
        # the batch-norm layers on their running statistics — recorded as step "e0"
        for n, b_ in model.named_buffers():
            put(out, "buf." + n, digest(b_.float()))
        grads = {}
        real_optim = trainer.svi.optim

        def spy2(params, _real=real_optim, _g=grads):
            for p in params:
                _g[names[id(p)]] = p.grad.detach().clone()
            _real(params)
        trainer.svi.optim = spy2
        loss = trainer.svi.step(x, y, **step_kw)
        trainer.svi.optim = real_optim
        tap = _minipyro.tap()
        out["e0.loss"] = np.float64(loss)
        out["e0.eps"] = tap["z.eps"].numpy().copy()
        for n, gr in grads.items():
            put(out, "e0.grad." + n, digest(gr))
        for n, p in model.named_parameters():
            put(out, "e0.param." + n, digest(p))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "loss0=%.6f" % out["s0.loss"], "keys=%d" % len(out))


def _ss_data(task, n_u, n_s, data_dim, dim):
    """Flattened inputs as the reference's own trainer tests feed them (tests/test_trainers.py:57-75)."""
    g = torch.Generator().manual_seed(0)
    n_pix = int(np.prod(data_dim))
    xu = torch.rand(n_u, n_pix, generator=g)
    xs = torch.rand(n_s, n_pix, generator=g)
    if task == "classification":
        ys = utils.to_onehot(torch.arange(n_s) % dim, dim)
    else:
        ys = torch.randn(n_s, dim, generator=g)
    return xu, xs, ys


def run_ss(name, task, data_dim, invariances, dim, batch_u, batch_s, rounds=2, latent_dim=2, epochs=2, n_u=7, n_s=4,
           step_kw=None, model_kw=None):
    """ssiVAE (models/ssivae.py) / ss_reg_iVAE (models/ss_reg_ivae.py) through auxSVItrainer (trainers/auxsvi.py):
    `rounds` x [compute_loss(unlabeled batch), compute_loss(labeled batch)] with everything recorded, then
    auxSVItrainer.step epochs on small loaders, then the inference API."""
    step_kw = dict(step_kw or {})
    model_kw = dict(model_kw or {})
    cls = task == "classification"
    out = {"meta.task": np.array(task), "meta.data_dim": np.array(data_dim),
           "meta.invariances": np.array("".join(invariances) if invariances else ""),
           "meta.dim": np.int64(dim), "meta.latent_dim": np.int64(latent_dim), "meta.rounds": np.int64(rounds),
           "meta.batch_u": np.int64(batch_u), "meta.batch_s": np.int64(batch_s), "meta.epochs": np.int64(epochs),
           "meta.n_u": np.int64(n_u), "meta.n_s": np.int64(n_s),
           "meta.scale_factor": np.float64(step_kw.get("scale_factor", 1.0)),
           "meta.aux_loss_multiplier": np.float64(step_kw.get("aux_loss_multiplier", 20.0))}
    for k, v in model_kw.items():
        out["meta.model_kw." + k] = np.array(v)

    def build():
        ctor = models.ssiVAE if cls else models.ss_reg_iVAE
        m = ctor(data_dim, latent_dim, dim, invariances, seed=1, device="cpu", **model_kw)
        t = trainers.auxSVItrainer(m, task=task, seed=1, device="cpu")
        return m, t
    model, trainer = build()
    names = {id(p): n for n, p in model.named_parameters()}
    for n, p in model.named_parameters():
        put(out, "init." + n, digest(p))
    xu, xs, ys = _ss_data(task, max(n_u, batch_u), max(n_s, batch_s), data_dim, dim)
    out["xu"], out["xs"], out["ys"] = xu.numpy().copy(), xs.numpy().copy(), ys.numpy().copy()
    optim = trainer.loss_basic.optim
    assert trainer.loss_aux.optim is optim
    call = 0
    for r in range(rounds):
        for kind in ("u", "s"):
            args = (xu[:batch_u],) if kind == "u" else (xs[:batch_s], ys[:batch_s])
            pre = "c%d" % call
            out[pre + ".kind"] = np.array(kind)
            # the two SVI steps of compute_loss (auxsvi.py:88-99), with the optimizer spied on
            for which, svi in (("elbo", trainer.loss_basic), ("aux", trainer.loss_aux)):
                grads = {}

                def spy(params, _g=grads):
                    for q in params:
                        if q.grad is not None:
                            _g[names[id(q)]] = q.grad.detach().clone()
                    optim(params)
                svi.optim = spy
                loss = svi.step(*(args if len(args) == 2 else (args[0], None)), **step_kw)
                svi.optim = optim
                tap = _minipyro.tap()
                out["%s.%s.loss" % (pre, which)] = np.float64(loss)
                if which == "elbo":
                    out[pre + ".eps"] = tap["z.eps"].numpy().copy()
                    if not cls and kind == "u":
                        out[pre + ".eps_y"] = tap["y.eps"].numpy().copy()
                    if cls and kind == "u":
                        out[pre + ".alpha"] = tap["enum_weights"].t().numpy().copy()
                    terms = tap["enum_terms"] if (cls and kind == "u") else tap["terms"]
                    for tn, tv in terms.items():
                        out["%s.term.%s" % (pre, tn)] = np.float64(tv.item())
                for n, gr in grads.items():
                    put(out, "%s.%s.grad.%s" % (pre, which, n), digest(gr))
            for n, q in model.named_parameters():
                put(out, pre + ".param." + n, digest(q))
            call += 1
    out["meta.calls"] = np.int64(call)
    # inference API on the trained model
    if cls:
        z_loc, z_scale, y_pred = model.encode(xu[:batch_u])
        out["enc.y_pred"] = y_pred.numpy().copy()
        out["cls.pred"] = model.classifier(xs).numpy().copy()
        yy = utils.to_onehot(torch.arange(batch_u) % dim, dim)
    else:
        z_loc, z_scale, y_hat = model.encode(xu[:batch_u])
        out["enc.y"] = y_hat.numpy().copy()
        out["reg.pred"] = model.regressor(xs).numpy().copy()
        yy = ys[:1].expand(batch_u, dim).contiguous()
    out["enc.z_loc"], out["enc.z_scale"] = z_loc.numpy().copy(), z_scale.numpy().copy()
    out["dec.y"] = yy.numpy().copy()
    out["dec.loc"] = model.decode(z_loc[:, -latent_dim:], yy).numpy().copy()
    # epochs through auxSVItrainer.step on loaders (fresh model: pins loader / RNG order, auxsvi.py:101-127)
    model, trainer = build()
    lu, ls, lv = utils.init_ssvae_dataloaders(xu[:n_u], (xs[:n_s], ys[:n_s]), (xs[:n_s], ys[:n_s]), batch_size=batch_s)
    for _ in range(epochs):
        trainer.step(lu, ls, lv, **step_kw)
    out["epochs.training_loss"] = np.array(trainer.history["training_loss"], dtype=np.float64)
    out["epochs.test"] = np.array([float(v) for v in trainer.history["test"]], dtype=np.float64)
    for pn, q in model.named_parameters():
        put(out, "final." + pn, digest(q))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "c0 elbo=%.6f aux=%.6f" % (out["c0.elbo.loss"], out["c0.aux.loss"]),
          "epochs", out["epochs.training_loss"], out["epochs.test"], "keys=%d" % len(out))


def run_step0_loc(name, data_dim, invariances, batch, latent_dim=2, xkind="rand"):
    """Step-0 forward only, with the decoder's `loc` and the transformed grid."""
    out = {}
    out["meta.data_dim"] = np.array(data_dim)
    out["meta.invariances"] = np.array("".join(invariances) if invariances else "")
    out["meta.batch"] = np.int64(batch)
    out["meta.latent_dim"] = np.int64(latent_dim)
    out["meta.xkind"] = np.array(xkind)
    model = models.iVAE(data_dim, latent_dim, invariances, seed=1, device="cpu")
    x = make_x(xkind, batch, data_dim)
    trainer = trainers.SVItrainer(model, seed=1, device="cpu")
    # trace without stepping the optimizer: call the ELBO directly under no_grad
    with torch.no_grad():
        loss, _ = trainer.svi.loss.loss_and_grads(model.model, model.guide, x)
    tap = _minipyro.tap()
    out["loss"] = np.float64(loss.item())
    for tn, tv in tap["terms"].items():
        out["term." + tn] = np.float64(tv.item())
    out["eps"] = tap["latent.eps"].numpy().copy()
    z = tap["sites"]["guide.latent"]
    out["z"] = z.numpy().copy()
    with torch.no_grad():
        if model.coord > 0:
            phi, dx, sc, zc = model.split_latent(z)
            if 't' in model.invariances:
                dx = (dx * model.t_prior).unsqueeze(1)
            grid = model.grid.expand(batch, *model.grid.shape)
            xc = utils.transform_coordinates(grid, phi, dx, sc)
            out["x_coord_prime"] = xc.numpy().copy()
            loc = model.decoder(xc, zc)
        else:
            loc = model.decoder(z)
    out["loc"] = loc.numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "loss=%.6f" % out["loss"])


def run_epochs(name, data_dim, invariances, n, batch, epochs=2, with_test=True, xkind="randn"):
    """SVItrainer.step(train_loader[, test_loader]) epochs — pins the DataLoader / RNG
    consumption order and the evaluate() semantics (trainers/svi.py:117-137)."""
    out = {}
    out["meta.data_dim"] = np.array(data_dim)
    out["meta.invariances"] = np.array("".join(invariances) if invariances else "")
    out["meta.n"] = np.int64(n)
    out["meta.batch"] = np.int64(batch)
    out["meta.epochs"] = np.int64(epochs)
    out["meta.with_test"] = np.int64(with_test)
    out["meta.xkind"] = np.array(xkind)
    train = make_x(xkind, n, data_dim, seed=0)
    test = make_x(xkind, n, data_dim, seed=5)
    out["train"] = train.numpy().copy()
    out["test"] = test.numpy().copy()
    train_loader = utils.init_dataloader(train, batch_size=batch)
    test_loader = utils.init_dataloader(test, batch_size=batch)
    model = models.iVAE(data_dim, 2, invariances, seed=1, device="cpu")
    trainer = trainers.SVItrainer(model, seed=1, device="cpu")
    for _ in range(epochs):
        if with_test:
            trainer.step(train_loader, test_loader)
        else:
            trainer.step(train_loader)
    out["epochs.training_loss"] = np.array(trainer.loss_history["training_loss"], dtype=np.float64)
    out["epochs.test_loss"] = np.array(trainer.loss_history["test_loss"], dtype=np.float64)
    for pn, p in model.named_parameters():
        put(out, "final." + pn, digest(p))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, out["epochs.training_loss"], out["epochs.test_loss"])


# (tests/test_gpu_parity.py::VARIANTS mirrors this table)
VARIANT_CASES = {
    "cdim3_rt": dict(data_dim=(8, 8), invariances=["r", "t"], c_dim=3),
    "cdim2_none": dict(data_dim=(8, 8), invariances=None, c_dim=2),
    "gauss_rts": dict(data_dim=(8, 8), invariances=["r", "t", "s"], sampler_d="gaussian"),
    "gauss_nosig_r": dict(data_dim=(8, 8), invariances=["r"], sampler_d="gaussian", sigmoid_d=False),
    "gauss_sig02_t": dict(data_dim=(8, 8), invariances=["t"], sampler_d="gaussian", decoder_sig=0.2),
    "gauss_randn_none": dict(data_dim=(8, 8), invariances=None, sampler_d="gaussian", sigmoid_d=False, xkind="randn"),
    "cbern_rts": dict(data_dim=(8, 8), invariances=["r", "t", "s"], sampler_d="continuous_bernoulli"),
    "cbern_none": dict(data_dim=(8, 8), invariances=None, sampler_d="continuous_bernoulli"),
    "cbern_16x16_r": dict(data_dim=(16, 16), invariances=["r"], sampler_d="continuous_bernoulli"),
    "relu_rt": dict(data_dim=(8, 8), invariances=["r", "t"], activation="relu"),
    "softplus_s": dict(data_dim=(8, 8), invariances=["s"], activation="softplus"),
    "lrelu_none": dict(data_dim=(8, 8), invariances=None, activation="lrelu"),
    "gelu_r": dict(data_dim=(8, 8), invariances=["r"], activation="gelu"),
    "hid64_rt": dict(data_dim=(8, 8), invariances=["r", "t"], hidden_dim_e=[64, 64], hidden_dim_d=[64, 64]),
    "hid3layers_r": dict(data_dim=(8, 8), invariances=["r"], hidden_dim_e=[128, 64, 32], hidden_dim_d=[32, 48, 16]),
    "priors_rts": dict(data_dim=(8, 8), invariances=["r", "t", "s"], dx_prior=0.3, dy_prior=0.05, sc_prior=0.25),
    "latent5_rt": dict(data_dim=(16, 16), invariances=["r", "t"], latent_dim=5),
    "1d32_t_cdim2": dict(data_dim=(32,), invariances=["t"], c_dim=2),
    "rect_12x20_rts": dict(data_dim=(12, 20), invariances=["r", "t", "s"]),
}


if __name__ == "__main__":
    torch.set_num_threads(4)
    only = sys.argv[1:] or None          # e.g. `make_golden.py jivae`: only (re)generate that family
    _run_steps_real = run_steps
    if only is not None:
        run_steps = run_step0_loc = lambda *a, **k: None      # noqa: E731
    # tiny cases, every invariance set of the reference's own trainer tests
    # (tests/test_trainers.py:26-40) — full tensors kept for the richest one
    for inv in (None, ["r"], ["t"], ["s"], ["r", "t"], ["r", "t", "s"]):
        tag = "".join(inv) if inv else "none"
        run_steps("ivae_8x8_%s_b6" % tag, (8, 8), inv, batch=6, full=(tag == "rts"))
        run_step0_loc("ivae_8x8_%s_b6_fwd" % tag, (8, 8), inv, batch=6)
    run_steps("ivae_1d16_none_b5", (16,), None, batch=5)
    run_steps("ivae_1d16_t_b5", (16,), ["t"], batch=5)
    run_step0_loc("ivae_1d16_t_b5_fwd", (16,), ["t"], batch=5)
    # non-[0,1] inputs, as the reference's tests use (Bernoulli(validate_args=False))
    run_steps("ivae_8x8_rts_b6_randn", (8, 8), ["r", "t", "s"], batch=6, xkind="randn")
    # KL scale factor (trainers/svi.py:152-155; ivae.py:175,214)
    run_steps("ivae_8x8_rt_b6_beta4", (8, 8), ["r", "t"], batch=6, step_kw={"scale_factor": 4.0})
    # odd sizes: ragged tiles (N = 63 is not a multiple of any MFMA tile)
    run_steps("ivae_7x9_rts_b3", (7, 9), ["r", "t", "s"], batch=3)
    # BASELINE configs C1 / C2 (digests only)
    run_steps("ivae_28x28_r_b128", (28, 28), ["r"], batch=128)
    run_steps("ivae_28x28_rt_b256", (28, 28), ["r", "t"], batch=256)
    run_step0_loc("ivae_28x28_rt_b16_fwd", (28, 28), ["r", "t"], batch=16)
    # saturated pixels (blob images) after a few steps
    run_steps("ivae_28x28_r_b32_blobs", (28, 28), ["r"], batch=32, steps=4, xkind="blobs")
    # jiVAE: joint continuous + discrete latent, enumerated ELBO (BASELINE config 3 family)
    if only is None or "jivae" in only:
        run_jsteps("jivae_8x8_r_k3_b5", (8, 8), ["r"], 3, batch=5, full=True)
        run_jsteps("jivae_8x8_rts_k4_b6", (8, 8), ["r", "t", "s"], 4, batch=6)
        run_jsteps("jivae_8x8_rt_k3_b4_sf", (8, 8), ["r", "t"], 3, batch=4, scale_factor=[2.0, 3.0])
        run_jsteps("jivae_1d16_t_k2_b5", (16,), ["t"], 2, batch=5)
        run_jsteps("jivae_28x28_r_k10_b16", (28, 28), ["r"], 10, batch=16, steps=2)
    if only is None or "jsampled" in only:
        # the trainer's default (enumerate_parallel=False): defined for the vanilla decoder only — with invariances the
        # reference's model raises (z repeated K times cannot broadcast against the drawn class, models/jivae.py:181-189)
        run_jsteps("jsivae_8x8_none_k3_b5", (8, 8), None, 3, batch=5, enumerate_parallel=False, full=True)
        run_jsteps("jsivae_1d16_none_k4_b6_sf", (16,), None, 4, batch=6, scale_factor=[1.5, 0.5], enumerate_parallel=False)
        run_jsteps("jsivae_28x28_none_k10_b16", (28, 28), None, 10, batch=16, steps=2, enumerate_parallel=False)
    if only is None or "jvanilla" in only:
        run_jsteps("jivae_8x8_none_k3_b5", (8, 8), None, 3, batch=5)          # fcDecoderNet (constructor default)
        run_jsteps("jivae_1d16_none_k4_b6_sf", (16,), None, 4, batch=6, scale_factor=[1.5, 0.5])
    # iVAE with a convolutional encoder (BASELINE config 4 family)
    if only is None or "convenc" in only:
        _rs = globals()["_run_steps_real"]
        _rs("ivaeconv_8x8_rts_b5", (8, 8), ["r", "t", "s"], batch=5, model_kw={"conv_encoder": [(4,), (8, 8)]})
        _rs("ivaeconv_16x16_rt_b4", (16, 16), ["r", "t"], batch=4, model_kw={"conv_encoder": [(4,), (8, 8), (16, 16)]})
        _rs("ivaeconv_1d16_t_b5", (16,), ["t"], batch=5, model_kw={"conv_encoder": [(4,), (8, 8)]})
    if only is None or "convenc64" in only:
        # BASELINE config 4 at its own shape: 64x64, ['r','t','s'], the DEFAULT convEncoderNet stack (nets/conv.py:24-64)
        _rs = globals()["_run_steps_real"]
        _rs("ivaeconv_64x64_rts_b4", (64, 64), ["r", "t", "s"], batch=4, steps=2,
            model_kw={"conv_encoder": [(32,), (64, 64), (128, 128)]})
    if only is None or "midsize" in only:
        # (round 3) mid-size pins between the toy fixtures and the BASELINE sizes — one SVI step each through the
        # reference's own modules and trainer: C3's model at batch 64 (0.5 M decoder rows: several workgroup ranges per
        # sample, K-fold slot arithmetic), C4's at batch 16, C5's at batch 16 (multi-round conv launches, split-K seams)
        _rs = globals()["_run_steps_real"]
        run_jsteps("jivae_28x28_r_k10_b64", (28, 28), ["r"], 10, batch=64, steps=1)
        _rs("ivaeconv_64x64_rts_b16", (64, 64), ["r", "t", "s"], batch=16, steps=1,
            model_kw={"conv_encoder": [(32,), (64, 64), (128, 128)]})
        run_ved_steps("ved_64x64_to_128_b16", (64, 64), (128,), batch=16, steps=1)
    # constructor variants of models/ivae.py:122-163 and the likelihoods of utils/prob.py:25-29: every branch the
    # HIP-vs-oracle "variants" tests exercise gets a reference-generated pin
    if only is None or "variants" in only:
        _rs = globals()["_run_steps_real"]
        for vname, v in VARIANT_CASES.items():
            v = dict(v)
            _rs("ivaevar_" + vname, v.pop("data_dim"), v.pop("invariances"), batch=7, steps=2,
                latent_dim=v.pop("latent_dim", 2), c_dim=v.pop("c_dim", 0), xkind=v.pop("xkind", "rand"),
                step_kw=v.pop("step_kw", {"scale_factor": 1.7}), model_kw=v)
    # VED: conv encoder / conv decoder (BASELINE config 5 family: 2-D image -> 1-D spectrum)
    if only is None or "ved" in only:
        small = dict(hidden_dim_e=[(8,), (16, 16)], hidden_dim_d=[(16, 16), (8,)])
        run_ved_steps("ved_16x16_to_32_small_b5", (16, 16), (32,), batch=5, model_kw=small)
        run_ved_steps("ved_16x16_to_32_small_b5_relu_sf", (16, 16), (32,), batch=5, scale_factor=2.5,
                      model_kw=dict(small, activation="relu"))
        run_ved_steps("ved_12x20_to_24_small_b3_l3", (12, 20), (24,), batch=3, latent_dim=3, model_kw=small)
        run_ved_steps("ved_1d32_to_1d32_small_b4", (32,), (32,), batch=4, model_kw=small)
        run_ved_steps("ved_64x64_to_128_b4", (64, 64), (128,), batch=4, steps=2)
    if only is None or "vedbn" in only:
        small = dict(hidden_dim_e=[(8,), (16, 16)], hidden_dim_d=[(16, 16), (8,)], batchnorm=True)
        run_ved_steps("vedbn_16x16_to_32_small_b5", (16, 16), (32,), batch=5, model_kw=small)
        run_ved_steps("vedbn_16x16_to_8x12_small_b4", (16, 16), (8, 12), batch=4, model_kw=small)
        run_ved_steps("vedbn_1d32_to_1d32_small_b4_relu", (32,), (32,), batch=4, model_kw=dict(small, activation="relu"))
    if only is None or "ved2d" in only:
        small = dict(hidden_dim_e=[(8,), (16, 16)], hidden_dim_d=[(16, 16), (8,)])
        run_ved_steps("ved_16x16_to_8x12_small_b4", (16, 16), (8, 12), batch=4, model_kw=small)     # bilinear upsampling
        run_ved_steps("ved_1d32_to_16x16_small_b3", (32,), (16, 16), batch=3, model_kw=small)
    # semi-supervised models through auxSVItrainer
    if only is None or "ss" in only:
        run_ss("sscls_8x8_rt_k3", "classification", (8, 8), ["r", "t"], 3, batch_u=4, batch_s=3)
        run_ss("sscls_8x8_none_k3", "classification", (8, 8), None, 3, batch_u=4, batch_s=3)
        run_ss("sscls_8x8_rts_k4_sf", "classification", (8, 8), ["r", "t", "s"], 4, batch_u=3, batch_s=2,
               step_kw={"scale_factor": 2.0, "aux_loss_multiplier": 50})
        run_ss("sscls_1d16_t_k2", "classification", (16,), ["t"], 2, batch_u=5, batch_s=3)
        run_ss("sscls_28x28_r_k10", "classification", (28, 28), ["r"], 10, batch_u=16, batch_s=8, rounds=1, epochs=1,
               n_u=16, n_s=8)
        run_ss("ssreg_8x8_rt_c2", "regression", (8, 8), ["r", "t"], 2, batch_u=4, batch_s=3)
        run_ss("ssreg_8x8_none_c1", "regression", (8, 8), None, 1, batch_u=4, batch_s=3)
        run_ss("ssreg_8x8_rts_c2_sf", "regression", (8, 8), ["r", "t", "s"], 2, batch_u=3, batch_s=2,
               step_kw={"scale_factor": 3.0, "aux_loss_multiplier": 5})
        run_ss("ssreg_28x28_r_c2", "regression", (28, 28), ["r"], 2, batch_u=16, batch_s=8, rounds=1, epochs=1,
               n_u=16, n_s=8)
    if only is not None:
        sys.exit(0)
    # epoch loops through the reference SVItrainer + DataLoader
    run_epochs("epochs_8x8_rts", (8, 8), ["r", "t", "s"], n=5, batch=2)
    run_epochs("epochs_8x8_r_notest", (8, 8), ["r"], n=7, batch=3, with_test=False, xkind="rand")
    run_epochs("epochs_8x8_none", (8, 8), None, n=5, batch=2)
