"""
_pyro_programs.py — `model()` / `guide()` of iVAE, jiVAE, VED and (round 3) the two semi-supervised models as real Pyro programs, for users who have pyro-ppl
("an optional `import pyro` path").  Pyro is NOT a dependency of this build and nothing on the HIP training
path imports this module's `pyro`: `trainers.SVItrainer` runs the same objectives in the library's fused kernels.  The
programs exist so that a Pyro user can hand `model.model` / `model.guide` to `pyro.infer.SVI`, `poutine.trace`, custom
ELBOs or `pyro.optim` objects (SVItrainer does that itself when it is given Pyro `optimizer` / `loss` objects) — the
networks then run as differentiable operators over the library's HIP GEMMs (`pyroved_amd.ops`), the coordinate
transform and the likelihood in PyTorch on the device.

Site names, plates, scales and tensor shapes follow the reference's programs (models/ivae.py:165-221,
models/jivae.py:152-220, models/ved.py:122-163, models/ssivae.py:153-248, models/ss_reg_ivae.py:156-246) so that traces are interchangeable.  Checked on the GPU under the test
suite's stand-in `pyro` (tests/golden/_minipyro.py) against the fused HIP objective (tests/test_gpu_pyro_programs.py).
"""
import torch


def _pyro():
    try:
        import pyro
        import pyro.distributions as dist
    except ImportError as e:
        raise NotImplementedError(
            "model() / guide() are Pyro programs and need pyro-ppl, which is not installed; training "
            "(trainers.SVItrainer) and evaluation (elbo_terms / encode / decode) do not need it") from e
    return pyro, dist


def _beta2(kwargs):
    beta = kwargs.get("scale_factor", [1., 1.])
    beta = torch.as_tensor(beta, dtype=torch.float32)
    return beta.expand(2) if beta.ndim == 0 else beta


def _likelihood(m, dist, loc):
    name = m.sampler_d.name
    if name == "bernoulli":
        return dist.Bernoulli(loc, validate_args=False)
    if name == "continuous_bernoulli":
        return dist.ContinuousBernoulli(loc)
    return dist.Normal(loc, m.sampler_d.decoder_sig)


def _transformed_grid(m, z):
    """split the latent, scale by the priors, transform the grid (ivae.py:184-192) — differentiable torch ops."""
    b = z.shape[0]
    phi, dx, sc, zc = m._split_latent(z)
    if 't' in m.invariances:
        dx = (dx * m.t_prior.to(z.device)).unsqueeze(1)
    grid = m.grid.to(z.device).expand(b, *m.grid.shape)
    if m.ndim == 1:
        return grid + dx, zc
    phi = phi if phi.ndim else phi.expand(b)
    sc = sc if sc.ndim else sc.expand(b)
    rot = torch.stack([torch.stack([torch.cos(phi), torch.sin(phi)], 1),
                       torch.stack([-torch.sin(phi), torch.cos(phi)], 1)], 1)
    return torch.bmm(grid, rot) * sc.reshape(b, 1, 1) + dx, zc


# ------------------------------------------------------------------------------------------------ iVAE
def ivae_guide(m, x, y=None, **kwargs):
    """q(z|x[,y]) (models/ivae.py:204-221)."""
    pyro, dist = _pyro()
    pyro.module("encoder_z", m.encoder_z)
    beta = kwargs.get("scale_factor", 1.)
    with pyro.plate("data", x.shape[0]):
        z_loc, z_scale = m.encoder_z(x if y is None else [x, y])
        with pyro.poutine.scale(scale=beta):
            pyro.sample("latent", dist.Normal(z_loc, z_scale).to_event(1))


def ivae_model(m, x, y=None, **kwargs):
    """p(x|z[,y]) p(z) (models/ivae.py:165-202)."""
    pyro, dist = _pyro()
    pyro.module("decoder", m.decoder)
    beta = kwargs.get("scale_factor", 1.)
    b = x.shape[0]
    n = x[0].numel()
    with pyro.plate("data", b):
        with pyro.poutine.scale(scale=beta):
            z = pyro.sample("latent", dist.Normal(x.new_zeros(b, m.z_dim), x.new_ones(b, m.z_dim)).to_event(1))
        if m.coord > 0:
            xc, z = _transformed_grid(m, z)
        if y is not None:
            z = torch.cat([z, y], dim=-1)
        loc = m.decoder(xc, z) if m.coord > 0 else m.decoder(z)
        pyro.sample("obs", _likelihood(m, dist, loc.reshape(-1, n)).to_event(1), obs=x.reshape(-1, n))


# ------------------------------------------------------------------------------------------------ jiVAE
def jivae_guide(m, x, **kwargs):
    """q(z, c|x) (models/jivae.py:199-220)."""
    pyro, dist = _pyro()
    pyro.module("encoder_z", m.encoder_z)
    beta = _beta2(kwargs)
    with pyro.plate("data"):
        z_loc, z_scale, alpha = m.encoder_z(x)
        with pyro.poutine.scale(scale=beta[0]):
            pyro.sample("latent_cont", dist.Normal(z_loc, z_scale).to_event(1))
        with pyro.poutine.scale(scale=beta[1]):
            pyro.sample("latent_disc", dist.OneHotCategorical(alpha))


def jivae_model(m, x, **kwargs):
    """p(x|z,c) p(z) p(c) (models/jivae.py:152-197); with the class enumerated in parallel its value is (K, B, K)."""
    pyro, dist = _pyro()
    pyro.module("decoder", m.decoder)
    beta = _beta2(kwargs)
    b, K = x.shape[0], m.discrete_dim
    n = x[0].numel()
    with pyro.plate("data"):
        with pyro.poutine.scale(scale=beta[0]):
            z = pyro.sample("latent_cont", dist.Normal(x.new_zeros(b, m.z_dim), x.new_ones(b, m.z_dim)).to_event(1))
        with pyro.poutine.scale(scale=beta[1]):
            z_disc = pyro.sample("latent_disc", dist.OneHotCategorical(x.new_ones(b, K) / K))
        if m.coord > 0:
            xc, zc = _transformed_grid(m, z.repeat(K, 1))
            loc = m.decoder(xc, [zc, z_disc.reshape(-1, K)])
        else:
            loc = m.decoder([z, z_disc])
        loc = loc.reshape(*z_disc.shape[:-1], n)
        pyro.sample("obs", _likelihood(m, dist, loc).to_event(1), obs=x.reshape(-1, n))


# ------------------------------------------------------------------------------------------------ VED
class _torch_conv_nets:
    """Inside a VED the conv nets normally run on the engine's HIP conv stack (no autograd); a Pyro program needs
    gradients through them, so their stand-alone (PyTorch / MIOpen) composition is used while this context is active."""
    def __init__(self, m):
        self.nets = (m.encoder_z, m.decoder)

    def __enter__(self):
        self.saved = [getattr(n, "_pv_engine", None) for n in self.nets]
        for n in self.nets:
            n._pv_engine = None

    def __exit__(self, *exc):
        for n, e in zip(self.nets, self.saved):
            n._pv_engine = e


def ved_guide(m, x=None, y=None, **kwargs):
    """q(z|x) (models/ved.py:147-163)."""
    pyro, dist = _pyro()
    pyro.module("encoder_z", m.encoder_z)
    beta = kwargs.get("scale_factor", 1.)
    with pyro.plate("data", x.shape[0]), _torch_conv_nets(m):
        z_loc, z_scale = m.encoder_z(x)
        with pyro.poutine.scale(scale=beta):
            pyro.sample("z", dist.Normal(z_loc, z_scale).to_event(1))


def ved_model(m, x=None, y=None, **kwargs):
    """p(y|z) p(z) (models/ved.py:122-145)."""
    pyro, dist = _pyro()
    pyro.module("decoder", m.decoder)
    beta = kwargs.get("scale_factor", 1.)
    b = x.shape[0]
    with pyro.plate("data", b), _torch_conv_nets(m):
        with pyro.poutine.scale(scale=beta):
            z = pyro.sample("z", dist.Normal(x.new_zeros(b, m.z_dim), x.new_ones(b, m.z_dim)).to_event(1))
        loc = m.decoder(z)
        pyro.sample("obs", _likelihood(m, dist, loc.flatten(1)).to_event(1), obs=y.flatten(1))


# ------------------------------------------------------------------------------------------------ ssiVAE / ss_reg_iVAE
def _ss_label_prior(m, dist, xs):
    """p(y): uniform over the classes (ssivae.py:183-185) or N(0, reg_sig) (ss_reg_ivae.py:183-186)."""
    b = xs.shape[0]
    if hasattr(m, "num_classes"):
        return dist.OneHotCategorical(xs.new_ones(b, m.num_classes) / m.num_classes)
    return dist.Normal(xs.new_zeros(b, m.reg_dim), m.reg_sig).to_event(1)


def _ss_label_guide(m, dist, out):
    """q(y|x) from the label network's output: OneHotCategorical(alpha) / N(c, reg_sig)."""
    if hasattr(m, "num_classes"):
        return dist.OneHotCategorical(out)
    return dist.Normal(out, m.reg_sig).to_event(1)


def ss_model(m, xs, ys=None, **kwargs):
    """p(x|z,y) p(y) p(z) (models/ssivae.py:153-190, models/ss_reg_ivae.py:156-193).  With the label enumerated in parallel
    by the guide (classification, unlabeled batch) ys is (K, B, K) and the decoder runs on K * B rows."""
    pyro, dist = _pyro()
    pyro.module("ss_vae", m)
    beta = kwargs.get("scale_factor", 1.)
    b = xs.shape[0]
    with pyro.plate("data"):
        with pyro.poutine.scale(scale=beta):
            zs = pyro.sample("z", dist.Normal(xs.new_zeros(b, m.z_dim), xs.new_ones(b, m.z_dim)).to_event(1))
        ys = pyro.sample("y", _ss_label_prior(m, dist, xs), obs=ys)
        c = ys.shape[-1]
        lead = ys.shape[:-1]                               # (B,) or, enumerated, (K, B)
        yr = ys.reshape(-1, c)
        zr = zs.reshape(-1, zs.shape[-1])                  # (enumerated: the guide's z is (K, B, z) — split_latent's view(-1, z))
        if zr.shape[0] != yr.shape[0]:
            zr = zr.repeat(yr.shape[0] // zr.shape[0], 1)
        if m.coord > 0:
            xc, zc = _transformed_grid(m, zr)
            loc = m.decoder(xc, [zc, yr])
        else:
            loc = m.decoder([zr, yr])
        loc = loc.reshape(*lead, -1)
        pyro.sample("x", _likelihood(m, dist, loc).to_event(1), obs=xs.flatten(1))


def ss_guide(m, xs, ys=None, **kwargs):
    """q(z|y,x) q(y|x) (models/ssivae.py:192-211, models/ss_reg_ivae.py:195-212)."""
    pyro, dist = _pyro()
    beta = kwargs.get("scale_factor", 1.)
    with pyro.plate("data"):
        if ys is None:
            ys = pyro.sample("y", _ss_label_guide(m, dist, m.encoder_y(xs)))
        if ys.dim() > 2:                                   # enumerated label (K, B, K): the z-encoder sees K * B rows
            k = ys.shape[0]
            loc, scale = m.encoder_z([xs.flatten(1).repeat(k, 1), ys.reshape(-1, ys.shape[-1])])
            loc, scale = loc.reshape(k, xs.shape[0], -1), scale.reshape(k, xs.shape[0], -1)
        else:
            loc, scale = m.encoder_z([xs.flatten(1), ys])
        with pyro.poutine.scale(scale=beta):
            pyro.sample("z", dist.Normal(loc, scale).to_event(1))


def ss_model_aux(m, xs, ys=None, **kwargs):
    """The auxiliary (supervised) objective (models/ssivae.py:215-228, models/ss_reg_ivae.py:226-240)."""
    pyro, dist = _pyro()
    pyro.module("ss_vae", m)
    with pyro.plate("data"):
        mult = kwargs.get("aux_loss_multiplier", 20)
        if ys is not None:
            out = m.encoder_y(xs)
            with pyro.poutine.scale(scale=mult):
                pyro.sample("y_aux", _ss_label_guide(m, dist, out), obs=ys)


def ss_guide_aux(m, xs, ys=None, **kwargs):
    """Dummy guide of the auxiliary objective (models/ssivae.py:230-234)."""
    return None
