"""
oracle/svi_oracle.py — CPU restatement of pyroVED's SVI training hot path.

TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg may import this module; the product package
(pyroved_amd/) never does, and fails loudly when its HIP library is missing.

What it restates (citations are /root/reference/pyroved/...):
  generate_grid / imcoordgrid / grid2xy      utils/coord.py:7-44
  transform_coordinates (+rotate, scale)     utils/coord.py:47-88
  split_latent                               models/base.py:97-119
  encoder_forward  (fcEncoderNet)            nets/fc.py:51-61, 307-324
  sdecoder_forward (sDecoderNet+coord_latent) nets/fc.py:189-199, 220-237
  fcdecoder_forward (fcDecoderNet)           nets/fc.py:143-152
  elbo            (iVAE.guide + iVAE.model under Trace_ELBO)
                                             models/ivae.py:165-221, utils/prob.py:25-29,
                                             trainers/svi.py:79-91
  SVIOracle.step  (SVI.step: loss, backward, per-parameter Adam, zero grads)
                                             trainers/svi.py:104-113 + pyro-ppl (see below)
  SVIOracle.train_epoch / evaluate_epoch     trainers/svi.py:95-137

It is an eager, op-for-op torch-CPU restatement: the same torch ops the
reference executes (F.linear, torch.bmm, tanh, torch.distributions.Normal /
Bernoulli log_prob, torch.optim.Adam), so on CPU in fp32 it is bit-identical
to the reference for everything that lives under /root/reference.  It is
functional: parameters come in as a state_dict-keyed dict of tensors (key
names of SURVEY §3.1), so it shares no code with the product's nn.Modules.

Third-party arithmetic that is NOT under /root/reference: `pyro-ppl`, pinned
by the reference only as `>=1.6.0` (setup.py:29, requirements.txt:3).  Its
published algorithm on this path (Trace_ELBO with one particle: sampled
log p(x|z) + beta*log p(z) - beta*log q(z|x), summed over the batch; SVI.step =
backward + one torch.optim.Adam per parameter + grads zeroed to zero tensors)
is restated here.

Pinning status: the oracle is checked in tests/test_oracle_golden.py against
fixtures produced by running the reference's OWN modules and its OWN
SVItrainer in this container (tests/golden/make_golden.py).  The reference's
nets / coordinate transform / split-latent / model()/guide() bodies /
SVItrainer loop ran for real; the Pyro runtime underneath them was a minimal
restatement (tests/golden/_minipyro.py) because pyro-ppl cannot be installed
here.  So: PINNED for every function under /root/reference; PARITY UNPINNED at
the Pyro boundary (Trace_ELBO / SVI / optim.Adam glue), where neither the
reference's tests nor a runnable Pyro hold a number.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import math

import torch
import torch.nn.functional as F
import torch.distributions as td

Params = Dict[str, torch.Tensor]


@dataclass
class Config:
    data_dim: Tuple[int, ...]
    latent_dim: int = 2
    invariances: Optional[Sequence[str]] = None
    c_dim: int = 0
    discrete_dim: int = 0          # > 0: jiVAE (models/jivae.py), K classes enumerated in the ELBO
    conv_encoder: Optional[Sequence[Sequence[int]]] = None   # iVAE.set_encoder(convEncoderNet(data_dim, z_dim, hidden_dim=...)): conv filters per block
    conv_activation: str = "lrelu"
    conv_batchnorm: bool = False   # convEncoderNet(..., batchnorm=True); the running statistics travel in Config.bufs
    bufs: Optional[dict] = None    # batch-norm buffers of the conv encoder (set by SVIOracle), always training mode: the
                                   # reference's iVAE never calls eval() (models/base.py:121-143)
    conv_decisions: Optional[object] = None     # test infrastructure (ConvDecisions): record / apply the conv encoder's sign and winner decisions
    custom_encoder: Optional[object] = None     # iVAE.set_encoder(user module): a callable x -> (z_loc, z_scale) in torch
    custom_decoder: Optional[object] = None     # iVAE.set_decoder(user module): (x_coord_prime, z) -> loc, or z -> loc (vanilla)
    custom_label_net: Optional[object] = None   # ssiVAE.set_classifier / ss_reg_iVAE.set_regressor(user module): x -> probabilities / means
    n_hidden_e: int = 2            # number of hidden Linear layers in encoder_z.fc_layers
    n_hidden_d: int = 2
    activation: str = "tanh"
    sampler: str = "bernoulli"
    sigmoid_d: bool = True
    dx_prior: float = 0.1
    dy_prior: Optional[float] = None
    sc_prior: float = 0.1
    decoder_sig: float = 0.5

    @property
    def ndim(self):
        return len(self.data_dim)

    @property
    def coord(self):
        # models/base.py:56-67
        if self.invariances is None:
            return 0
        c = len(self.invariances)
        if "t" in self.invariances and self.ndim == 2:
            c += 1
        return c

    @property
    def z_dim(self):
        return self.latent_dim + self.coord

    @property
    def n_pix(self):
        n = 1
        for d in self.data_dim:
            n *= d
        return n


_ACT = {
    "tanh": torch.tanh,
    "relu": torch.relu,
    "lrelu": lambda t: F.leaky_relu(t, 0.01),
    "softplus": F.softplus,
    "gelu": F.gelu,
}


# ---------------------------------------------------------------- coordinates
def generate_grid(data_dim, dtype=torch.float32):
    """utils/coord.py:21-44 (2-D rows n=i*W+j -> (xx[i], yy[j]); 1-D: (L,1))."""
    if len(data_dim) == 1:
        return torch.linspace(1, -1, data_dim[0])[:, None].to(dtype)
    xx = torch.linspace(-1, 1, data_dim[0])
    yy = torch.linspace(1, -1, data_dim[1])
    x0, x1 = torch.meshgrid(xx, yy, indexing="ij")
    return torch.stack((x0.reshape(-1), x1.reshape(-1)), 1).to(dtype)


def transform_coordinates(coord, phi=0, coord_dx=0, scale=1.0):
    """utils/coord.py:47-88: rotate -> scale -> translate; 1-D: translate only."""
    if coord.shape[-1] == 1:
        return coord + coord_dx
    if not torch.is_tensor(phi) or torch.sum(phi) == 0:
        phi = coord.new_zeros(coord.shape[0])
    r1 = torch.stack([torch.cos(phi), torch.sin(phi)], 1)
    r2 = torch.stack([-torch.sin(phi), torch.cos(phi)], 1)
    rot = torch.stack([r1, r2], 1)
    coord = torch.bmm(coord, rot)
    sm = coord.new_zeros(coord.shape[0], 2, 2)
    sm[:, 0, 0] = scale
    sm[:, 1, 1] = scale
    coord = torch.bmm(coord, sm)
    return coord + coord_dx


def split_latent(cfg: Config, z):
    """models/base.py:97-119: [phi | dx,dy | s | content] regardless of list order."""
    if cfg.ndim == 1:
        return None, z[:, 0:1], None, z[:, 1:]
    phi = z.new_tensor(0.0)
    dx = z.new_tensor(0.0)
    sc = z.new_tensor(1.0)
    if "r" in cfg.invariances:
        phi, z = z[:, 0], z[:, 1:]
    if "t" in cfg.invariances:
        dx, z = z[:, :2], z[:, 2:]
    if "s" in cfg.invariances:
        sc = sc + cfg.sc_prior * z[:, 0]
        z = z[:, 1:]
    return phi, dx, sc, z


def t_prior(cfg: Config, like):
    """models/base.py:73-77."""
    dy = cfg.dx_prior if cfg.dy_prior is None else cfg.dy_prior
    if cfg.ndim == 2:
        return like.new_tensor([cfg.dx_prior, dy])
    return like.new_tensor(cfg.dx_prior)


# ---------------------------------------------------------------------- nets
def _fc_stack(p: Params, prefix: str, n_layers: int, act, h):
    # make_fc_layers: Sequential(Linear, act, Linear, act, ...) -> indices 0, 2, 4 (nets/fc.py:307-324)
    for i in range(n_layers):
        h = act(F.linear(h, p["%s.%d.weight" % (prefix, 2 * i)], p["%s.%d.bias" % (prefix, 2 * i)]))
    return h


def encoder_forward(p: Params, cfg: Config, x, y=None):
    """fcEncoderNet.forward (nets/fc.py:51-61); Concat of [x, y] (utils/nn.py:62-74)."""
    act = _ACT[cfg.activation]
    h = x.reshape(x.shape[0], -1)
    if y is not None:
        h = torch.cat([h, y], -1)
    h = _fc_stack(p, "encoder_z.fc_layers", cfg.n_hidden_e, act, h)
    mu = F.linear(h, p["encoder_z.fc11.weight"], p["encoder_z.fc11.bias"])
    sigma = F.softplus(F.linear(h, p["encoder_z.fc12.weight"], p["encoder_z.fc12.bias"]))
    return mu, sigma


def jencoder_forward(p: Params, cfg: Config, x):
    """jfcEncoderNet.forward (nets/fc.py:97-108): (mu, softplus sigma, softmax alpha)."""
    act = _ACT[cfg.activation]
    h = _fc_stack(p, "encoder_z.fc_layers", cfg.n_hidden_e, act, x.reshape(x.shape[0], -1))
    mu = F.linear(h, p["encoder_z.fc11.weight"], p["encoder_z.fc11.bias"])
    sigma = F.softplus(F.linear(h, p["encoder_z.fc12.weight"], p["encoder_z.fc12.bias"]))
    alpha = torch.softmax(F.linear(h, p["encoder_z.fc13.weight"], p["encoder_z.fc13.bias"]), dim=-1)
    return mu, sigma, alpha


def sdecoder_forward(p: Params, cfg: Config, x_coord, z):
    """sDecoderNet.forward + coord_latent.forward (nets/fc.py:189-199, 220-237)."""
    act = _ACT[cfg.activation]
    b, n = x_coord.shape[:2]
    h_x = F.linear(x_coord.reshape(b * n, -1), p["decoder.coord_latent.fc_coord.weight"],
                   p["decoder.coord_latent.fc_coord.bias"]).reshape(b, n, -1)
    h_z = F.linear(z, p["decoder.coord_latent.fc_latent.weight"])
    h = torch.tanh((h_x + h_z.unsqueeze(1)).reshape(b * n, -1))     # coord_latent's tanh is hard-wired (fc.py:218)
    h = _fc_stack(p, "decoder.fc_layers", cfg.n_hidden_d, act, h)
    out = F.linear(h, p["decoder.out.weight"], p["decoder.out.bias"])
    if cfg.sigmoid_d:
        out = torch.sigmoid(out)
    return out.view(-1, *cfg.data_dim)


def fcdecoder_forward(p: Params, cfg: Config, z):
    """fcDecoderNet.forward (nets/fc.py:143-152)."""
    act = _ACT[cfg.activation]
    h = _fc_stack(p, "decoder.fc_layers", cfg.n_hidden_d, act, z)
    out = F.linear(h, p["decoder.out.weight"], p["decoder.out.bias"])
    if cfg.sigmoid_d:
        out = torch.sigmoid(out)
    return out.view(-1, *cfg.data_dim)


def decode_from_latent(p: Params, cfg: Config, z, y=None, grid=None):
    """The decoder half of iVAE.model (models/ivae.py:184-198): split, transform, decode."""
    b = z.shape[0]
    if cfg.coord > 0:
        if grid is None:
            grid = generate_grid(cfg.data_dim, z.dtype)
        phi, dx, sc, zc = split_latent(cfg, z)
        if "t" in cfg.invariances:
            dx = (dx * t_prior(cfg, z)).unsqueeze(1)
        xc = transform_coordinates(grid.expand(b, *grid.shape), phi, dx, sc)
        if y is not None:
            zc = torch.cat([zc, y], -1)
        if cfg.custom_decoder is not None:
            return cfg.custom_decoder(xc, zc), xc
        return sdecoder_forward(p, cfg, xc, zc), xc
    if y is not None:
        z = torch.cat([z, y], -1)
    if cfg.custom_decoder is not None:
        return cfg.custom_decoder(z), None
    return fcdecoder_forward(p, cfg, z), None


def likelihood(cfg: Config, loc):
    """utils/prob.py:25-29."""
    if cfg.sampler == "bernoulli":
        return td.Bernoulli(loc, validate_args=False)
    if cfg.sampler == "continuous_bernoulli":
        return td.ContinuousBernoulli(loc)
    if cfg.sampler == "gaussian":
        return td.Normal(loc, cfg.decoder_sig)
    raise KeyError(cfg.sampler)


# ---------------------------------------------------------------------- ELBO
def elbo(p: Params, cfg: Config, x, eps, beta=1.0, y=None, grid=None):
    """One-particle Trace_ELBO of iVAE.guide/model (models/ivae.py:165-221).

    loss = -( sum_b log p(x_b|z_b) + beta*sum_b log N(z_b;0,1) - beta*sum_b log N(z_b;mu_b,sigma_b) ),
    z = mu + sigma*eps (Normal.rsample), summed — not averaged — over the batch.
    """
    b = x.shape[0]
    z_loc, z_scale = _encode_any(p, cfg, x, y)
    z = z_loc + z_scale * eps
    logq = td.Normal(z_loc, z_scale).log_prob(z).sum(-1)                      # guide site "latent"
    logp = td.Normal(torch.zeros_like(z), torch.ones_like(z)).log_prob(z).sum(-1)   # model site "latent"
    loc, xc = decode_from_latent(p, cfg, z, y, grid)
    ll = likelihood(cfg, loc.reshape(b, -1)).log_prob(x.reshape(b, -1)).sum(-1)     # model site "obs"
    t_ll, t_lp, t_lq = ll.sum(), (beta * logp).sum(), (beta * logq).sum()
    loss = -(t_ll + t_lp - t_lq)
    return dict(loss=loss, ll=t_ll, logpz=t_lp, logqz=t_lq, z_loc=z_loc, z_scale=z_scale, z=z,
                loc=loc, x_coord_prime=xc, ll_per_sample=ll)


def jelbo(p: Params, cfg: Config, x, eps, beta=1.0, grid=None):
    """TraceEnum_ELBO of jiVAE.guide/model (models/jivae.py:152-220) with the guide's OneHotCategorical site
    enumerated in parallel (trainers/svi.py:83-90):

    loss = -sum_b [ b0*(log N(z_b;0,1) - log N(z_b;mu_b,sigma_b))
                    + sum_k alpha_bk * ( log p(x_b | z_b, k) + b1*log(1/K) - b1*log alpha_bk ) ]
    z = mu + sigma*eps;  the decoder runs on K*B rows ordered [k][b] (z.repeat(K, 1), jivae.py:181) with the one-hot
    class appended to the content part of z (jivae.py:189-192; without invariances Concat broadcasts z over the K
    enumerated classes, utils/nn.py:62-74, which is the same row set); scale_factor: scalar -> both, [cont, disc].
    """
    b0, b1 = (beta, beta) if not isinstance(beta, (list, tuple)) else beta
    bsz, K = x.shape[0], cfg.discrete_dim
    z_loc, z_scale, alpha = jencoder_forward(p, cfg, x)
    z = z_loc + z_scale * eps
    logq = td.Normal(z_loc, z_scale).log_prob(z).sum(-1)
    logp = td.Normal(torch.zeros_like(z), torch.ones_like(z)).log_prob(z).sum(-1)
    z_disc = torch.eye(K, dtype=z.dtype).repeat_interleave(bsz, 0)              # (K*B, K): rows [k][b] = onehot(k)
    loc, xc = decode_from_latent(p, cfg, z.repeat(K, 1), z_disc, grid)
    ll = likelihood(cfg, loc.reshape(K, bsz, -1)).log_prob(x.reshape(bsz, -1)).sum(-1)       # (K, B)
    w = alpha.t()                                                                           # q(k | x_b), (K, B)
    logq_d = td.OneHotCategorical(probs=alpha).log_prob(torch.eye(K, dtype=z.dtype).unsqueeze(1))   # (K, B)
    logp_d = torch.full_like(logq_d, -math.log(K))
    t_ll = (w * ll).sum()
    t_lp = (b0 * logp).sum() + (w * b1 * logp_d).sum()
    t_lq = (b0 * logq).sum() + (w * b1 * logq_d).sum()
    loss = -(t_ll + t_lp - t_lq)
    return dict(loss=loss, ll=t_ll, logpz=t_lp, logqz=t_lq, z_loc=z_loc, z_scale=z_scale, z=z, alpha=alpha,
                loc=loc, x_coord_prime=xc, ll_per_sample=ll,
                terms={"model.latent_cont": (b0 * logp).sum(), "model.latent_disc": (w * b1 * logp_d).sum(),
                       "model.obs": t_ll, "guide.latent_cont": (b0 * logq).sum(),
                       "guide.latent_disc": (w * b1 * logq_d).sum()})


def jelbo_sampled(p: Params, cfg: Config, x, eps, y_onehot=None, beta=1.0):
    """Trace_ELBO of jiVAE.guide/model WITHOUT enumeration — SVItrainer's default enumerate_parallel=False
    (trainers/svi.py:83-91) — for the vanilla decoder (invariances=None; with invariances the reference's own model
    cannot broadcast its K-times repeated z against the sampled class, models/jivae.py:181-189, and raises).

    The class y_b ~ OneHotCategorical(alpha_b) is a guide site without rsample: Pyro's Trace_ELBO (trace_elbo.py:
    _compute_log_r, score_parts) differentiates the surrogate
        sum(model log-probs) - sum(log q of the reparameterised site) + sum_b detach(log_r_b) * log q(y_b | x_b),
        log_r_b = ll_b + b0 (log p(z_b) - log q(z_b)) + b1 (log(1/K) - log alpha_b[y_b])        (score function unscaled)
    while the reported loss is -(sum model - sum guide) on the drawn (z, y).  y_onehot None: drawn here, on the global
    CPU generator, exactly where the guide draws it (after eps).  Returns `loss` carrying the surrogate's gradient."""
    b0, b1 = (beta, beta) if not isinstance(beta, (list, tuple)) else beta
    bsz, K = x.shape[0], cfg.discrete_dim
    if cfg.coord > 0:
        raise RuntimeError("jiVAE without enumeration is defined for the vanilla decoder only (invariances=None)")
    z_loc, z_scale, alpha = jencoder_forward(p, cfg, x)
    z = z_loc + z_scale * eps
    if y_onehot is None:
        with torch.no_grad():
            y_onehot = td.OneHotCategorical(probs=alpha.detach().float()).sample().to(z.dtype)
    logq = td.Normal(z_loc, z_scale).log_prob(z).sum(-1)
    logp = td.Normal(torch.zeros_like(z), torch.ones_like(z)).log_prob(z).sum(-1)
    logq_d = td.OneHotCategorical(probs=alpha).log_prob(y_onehot)                        # (B,)
    logp_d = torch.full_like(logq_d, -math.log(K))
    loc, xc = decode_from_latent(p, cfg, z, y_onehot, None)
    ll = likelihood(cfg, loc.reshape(bsz, -1)).log_prob(x.reshape(bsz, -1)).sum(-1)
    t_ll = ll.sum()
    t_lp = (b0 * logp).sum() + (b1 * logp_d).sum()
    t_lq = (b0 * logq).sum() + (b1 * logq_d).sum()
    elbo_v = t_ll + t_lp - t_lq
    log_r = (ll + b0 * logp - b0 * logq + b1 * logp_d - b1 * logq_d).detach()
    surrogate = t_ll + t_lp - (b0 * logq).sum() + (log_r * logq_d).sum()
    loss = -(surrogate - surrogate.detach() + elbo_v.detach())
    return dict(loss=loss, ll=t_ll, logpz=t_lp, logqz=t_lq, z_loc=z_loc, z_scale=z_scale, z=z, alpha=alpha,
                y=y_onehot, loc=loc, log_r=log_r, ll_per_sample=ll,
                terms={"model.latent_cont": (b0 * logp).sum(), "model.latent_disc": (b1 * logp_d).sum(),
                       "model.obs": t_ll, "guide.latent_cont": (b0 * logq).sum(),
                       "guide.latent_disc": (b1 * logq_d).sum()})


# ======================================================================= VED (models/ved.py, nets/conv.py)
@dataclass
class VedConfig:
    """models.VED constructor arguments that shape the computation (models/ved.py:89-118)."""
    input_dim: Tuple[int, ...]
    output_dim: Tuple[int, ...]
    input_channels: int = 1
    output_channels: int = 1
    latent_dim: int = 2
    hidden_dim_e: Optional[Sequence[Sequence[int]]] = None      # default [(32,), (64, 64), (128, 128)]
    hidden_dim_d: Optional[Sequence[Sequence[int]]] = None      # default [(128, 128), (64, 64), (32,)]
    activation: str = "lrelu"
    sampler: str = "bernoulli"
    sigmoid_d: bool = True
    decoder_sig: float = 0.5
    batchnorm: bool = False        # nn.BatchNormNd after every conv + activation (nets/conv.py:185-186, 239-240)

    @property
    def z_dim(self):
        return self.latent_dim

    @property
    def he(self):
        return [tuple(b) for b in (self.hidden_dim_e or [(32,), (64, 64), (128, 128)])]

    @property
    def hd(self):
        return [tuple(b) for b in (self.hidden_dim_d or [(128, 128), (64, 64), (32,)])]


def _conv(ndim):
    return {1: F.conv1d, 2: F.conv2d}[ndim]


def _bn(p: Params, bufs, pre: str, h, training: bool):
    """nn.BatchNormNd.forward: batch statistics + in-place update of the running estimates in training mode, the running
    estimates in eval mode (F.batch_norm, momentum 0.1, eps 1e-5).  `bufs`: dict of the running_mean / running_var /
    num_batches_tracked tensors (plain tensors, no gradient)."""
    if training:
        bufs[pre + ".num_batches_tracked"] += 1
    return F.batch_norm(h, bufs[pre + ".running_mean"], bufs[pre + ".running_var"], p[pre + ".weight"], p[pre + ".bias"],
                        training, 0.1, 1e-5)


class ConvDecisions:
    """The data-dependent DECISIONS of a conv encoder's forward — leaky-ReLU signs and 2x2 max-pool winners — as test infrastructure
    (VERDICT r5 item 3).  Two implementations of nets/conv.py:150-213 in different arithmetics decide a handful of near-ties
    differently, and a flipped decision changes a gradient by far more than either arithmetic's rounding: to see ARITHMETIC error the
    float64 oracle is run under another implementation's decisions, and the flips are counted as a quantity of their own.
      record()            -> the forward fills `sign` / `win` with its own decisions (one entry per conv layer / pool, in order);
      given decisions     -> the forward applies them instead of deciding: y = x where sign else 0.01 x; pooled = x[winner].
    sign[i]: bool (B, C, H, W) after conv i — at the POOLED resolution when that conv's block ends in a max-pool whose input the
    other implementation never stores (the sign of the window's winner; losers carry no gradient); win[j]: int64 (B, C, H/2, W/2)
    with k = 2 dy + dx of the window's winner.  2-D stacks with activation 'lrelu' or 'relu', no batch norm."""

    def __init__(self, sign=None, win=None, pool_of=None):
        self.sign = [] if sign is None else sign
        self.win = [] if win is None else win
        self.recording = sign is None
        self.pool_of = {} if pool_of is None else dict(pool_of)   # conv index -> index of the max-pool that follows it directly

    @staticmethod
    def pick(h, k):
        """h (B, C, H, W), k (B, C, H/2, W/2) in 0..3 -> the window element 2 dy + dx = k (differentiable gather)."""
        b, c, hh, ww = h.shape
        win = h.reshape(b, c, hh // 2, 2, ww // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(b, c, hh // 2, ww // 2, 4)
        return win.gather(-1, k.unsqueeze(-1)).squeeze(-1)

    @staticmethod
    def first_max(h):
        """torch's max-pool winner (first maximum in scan order) of every 2x2 window as k = 2 dy + dx."""
        b, c, hh, ww = h.shape
        win = h.reshape(b, c, hh // 2, 2, ww // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(b, c, hh // 2, ww // 2, 4)
        m = win.max(-1, keepdim=True).values
        return (win == m).to(torch.int8).argmax(-1)      # argmax of a 0/1 tensor: the FIRST maximum

    def flips(self, other):
        """(differing signs, differing winners, signs compared, winners compared) against another decision set of the same net;
        a sign recorded at full resolution is compared at the winners of `self` where `other` holds the pooled one."""
        ds = dw = ns = nw = 0
        j = 0
        for i, (a, b) in enumerate(zip(self.sign, other.sign)):
            if a.shape != b.shape:                    # one side knows this sign at the pooled resolution only: compare at ITS winners
                pooled_first = a.shape[-1] < b.shape[-1]
                full, pooled = (b, a) if pooled_first else (a, b)
                owner = self if pooled_first else other
                k = owner.win[(owner.pool_of or self.pool_of or other.pool_of)[i]]
                full = self.pick(full.to(torch.int8), k).bool()
                a, b = (pooled, full) if pooled_first else (full, pooled)
            ds += int((a != b).sum()); ns += a.numel()
        for a, b in zip(self.win, other.win):
            dw += int((a != b).sum()); nw += a.numel()
        return ds, dw, ns, nw


def conv_encoder_forward(p: Params, cfg: VedConfig, x, bufs=None, training=True, decisions: "ConvDecisions" = None):
    """convEncoderNet.forward (nets/conv.py:24-64): FeatureExtractor (conv k3 s1 p1 + activation per filter, a 2x
    max-pool after every block but the last; conv.py:150-213) -> flatten (C, spatial) -> Linear -> (mu, softplus).
    decisions (test infrastructure, see ConvDecisions): record this forward's leaky-ReLU signs and max-pool winners, or apply
    given ones instead of deciding."""
    act, nd = _ACT[cfg.activation], len(cfg.input_dim)
    h, idx = x, 0
    blocks = cfg.he
    dec = decisions
    if dec is not None:
        assert nd == 2 and cfg.activation in ("lrelu", "relu") and not cfg.batchnorm, "ConvDecisions: 2-D lrelu / relu stacks"
        slope = 0.01 if cfg.activation == "lrelu" else 0.0
        li = pi = 0
    for bi, block in enumerate(blocks):
        pool_done = False
        for ci, _ in enumerate(block):
            pre = "encoder_z.feature_extractor.layers.%d" % idx
            h = _conv(nd)(h, p[pre + ".weight"], p[pre + ".bias"], stride=1, padding=1)
            pooled_next = dec is not None and ci + 1 == len(block) and bi + 1 < len(blocks)
            if dec is None:
                h = act(h)
            elif dec.recording:
                dec.sign.append((h > 0).detach())
                if pooled_next:
                    k = ConvDecisions.first_max(h.detach())      # (the activation is monotone: the winner before = after it)
                    dec.pool_of[li] = len(dec.win); dec.win.append(k)
                h = act(h)
                li += 1
            else:
                sg = dec.sign[li]
                if pooled_next and sg.shape[-1] * 2 == h.shape[-1]:
                    # the other implementation pooled in the convolution's epilogue: winner first, then the winner's sign
                    h = ConvDecisions.pick(h, dec.win[pi])
                    h = torch.where(sg, h, slope * h)
                    pi += 1; li += 1; idx += 2         # conv, activation (the pool's own index is counted below)
                    pool_done = True
                    continue
                h = torch.where(sg, h, slope * h)
                li += 1
            idx += 2                                   # conv, activation
            if cfg.batchnorm:
                h = _bn(p, bufs, "encoder_z.feature_extractor.layers.%d" % idx, h, training)
                idx += 1
        if bi + 1 < len(blocks):
            if pool_done:
                pass
            elif dec is not None and not dec.recording:
                h = ConvDecisions.pick(h, dec.win[pi]); pi += 1
            else:
                h = (F.max_pool1d if nd == 1 else F.max_pool2d)(h, 2, 2)
            idx += 1
    enc = F.linear(h.reshape(h.shape[0], -1), p["encoder_z.features2latent.fc_latent.weight"],
                   p["encoder_z.features2latent.fc_latent.bias"])
    mu, sigma = enc.split(cfg.latent_dim, 1)
    return mu, F.softplus(sigma)


def conv_decoder_forward(p: Params, cfg: VedConfig, z, bufs=None, training=True):
    """convDecoderNet.forward (nets/conv.py:67-102): Linear -> (C0, *out_dim / 2^blocks) -> per block [conv k3 +
    activation per filter, then UpsampleBlock = 2x interpolate (nearest in 1-D, bilinear in 2-D; conv.py:105-147) +
    conv k1] -> conv k1 to the output channels -> sigmoid (conv.py:216-262)."""
    act, nd = _ACT[cfg.activation], len(cfg.output_dim)
    blocks = cfg.hd
    d0 = [int(v) // 2 ** len(blocks) for v in cfg.output_dim]
    h = F.linear(z, p["decoder.latent2features.fc.weight"], p["decoder.latent2features.fc.bias"])
    h = h.view(-1, blocks[0][0], *d0)
    idx = 0
    for block in blocks:
        for _ in block:
            pre = "decoder.upsampler.layers.%d" % idx
            h = act(_conv(nd)(h, p[pre + ".weight"], p[pre + ".bias"], stride=1, padding=1))
            idx += 2
            if cfg.batchnorm:
                h = _bn(p, bufs, "decoder.upsampler.layers.%d" % idx, h, training)
                idx += 1
        pre = "decoder.upsampler.layers.%d.conv" % idx
        h = F.interpolate(h, scale_factor=2, mode="nearest" if nd == 1 else "bilinear")
        h = _conv(nd)(h, p[pre + ".weight"], p[pre + ".bias"])
        idx += 1
    pre = "decoder.upsampler.layers.%d" % idx
    h = _conv(nd)(h, p[pre + ".weight"], p[pre + ".bias"])
    return torch.sigmoid(h) if cfg.sigmoid_d else h


def ved_elbo(p: Params, cfg: VedConfig, x, y, eps, beta=1.0, bufs=None, training=True, decisions=None):
    """Trace_ELBO of VED.guide/model (models/ved.py:122-163): z = mu + sigma*eps,
    loss = -( sum_b log p(y_b | z_b) + beta*sum_b log N(z_b;0,1) - beta*sum_b log N(z_b;mu_b,sigma_b) )."""
    b = x.shape[0]
    z_loc, z_scale = conv_encoder_forward(p, cfg, x, bufs, training, decisions)
    z = z_loc + z_scale * eps
    logq = td.Normal(z_loc, z_scale).log_prob(z).sum(-1)
    logp = td.Normal(torch.zeros_like(z), torch.ones_like(z)).log_prob(z).sum(-1)
    loc = conv_decoder_forward(p, cfg, z, bufs, training)
    ll = likelihood(cfg, loc.flatten(1)).log_prob(y.reshape(b, -1)).sum(-1)
    t_ll, t_lp, t_lq = ll.sum(), (beta * logp).sum(), (beta * logq).sum()
    return dict(loss=-(t_ll + t_lp - t_lq), ll=t_ll, logpz=t_lp, logqz=t_lq, z_loc=z_loc, z_scale=z_scale, z=z, loc=loc)


class VedOracle:
    """SVI.step for VED restated (same conventions as SVIOracle)."""

    def __init__(self, params: Params, cfg: VedConfig, lr: float = 1e-3, dtype=torch.float32):
        self.cfg = cfg
        # `params`: a state_dict; batch-norm buffers (running_mean / running_var / num_batches_tracked) are split off
        is_buf = lambda k: k.rsplit(".", 1)[-1] in ("running_mean", "running_var", "num_batches_tracked")
        self.p = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in params.items() if not is_buf(k)}
        self.bufs = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.detach().clone())
                     for k, v in params.items() if is_buf(k)}
        self.training = True            # nn.Module.training: VED.encode / decode switch to eval() and nothing switches back
        self.opt = torch.optim.Adam(list(self.p.values()), lr=lr)
        self.dtype = dtype
        self.last = self.last_grads = None

    def step(self, x, y, eps, beta=1.0) -> float:
        out = ved_elbo(self.p, self.cfg, x.to(self.dtype), y.to(self.dtype), eps.to(self.dtype), beta, self.bufs,
                       self.training)
        if out["loss"].requires_grad:
            out["loss"].backward()
        self.last = out
        self.last_grads = {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in self.p.items()}
        if any(v.grad is not None for v in self.p.values()):
            self.opt.step()
        for v in self.p.values():
            if v.grad is not None:
                v.grad = torch.zeros_like(v.grad)
        return out["loss"].item()

    def encode(self, x):
        self.training = False                              # models/ved.py:178
        with torch.no_grad():
            return conv_encoder_forward(self.p, self.cfg, x.to(self.dtype), self.bufs, False)

    def decode(self, z):
        self.training = False                              # models/ved.py:193
        with torch.no_grad():
            return conv_decoder_forward(self.p, self.cfg, z.to(self.dtype), self.bufs, False)


def _encode_any(p: Params, cfg: Config, x, y=None):
    """encoder_z(x): fcEncoderNet, or the convEncoderNet a user installed with set_encoder (models/base.py:173-177),
    which sees x as (B, 1, *data_dim)."""
    if cfg.custom_encoder is not None:
        return cfg.custom_encoder(x)
    if cfg.conv_encoder is not None:
        vc = VedConfig(input_dim=cfg.data_dim, output_dim=cfg.data_dim, latent_dim=cfg.z_dim,
                       hidden_dim_e=cfg.conv_encoder, activation=cfg.conv_activation, batchnorm=cfg.conv_batchnorm)
        return conv_encoder_forward(p, vc, x.reshape(x.shape[0], 1, *cfg.data_dim), cfg.bufs, True, cfg.conv_decisions)
    return encoder_forward(p, cfg, x, y)


def param_order(p: Params, cfg: Config) -> List[str]:
    """state_dict order == construction order."""
    return list(p.keys())


class SVIOracle:
    """SVI.step restated: Trace_ELBO loss -> backward -> Adam(lr, betas=(0.9,0.999), eps=1e-8)
    -> grads set to zero tensors.  `params` is updated in place; tensors are leaf
    tensors owned by this object (cloned from the dict passed in)."""

    def __init__(self, params: Params, cfg: Config, lr: float = 1e-3, dtype=torch.float32):
        self.cfg = cfg
        is_buf = lambda k: k.rsplit(".", 1)[-1] in ("running_mean", "running_var", "num_batches_tracked")
        self.p = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in params.items() if not is_buf(k)}
        self.bufs = {k: v.detach().clone() for k, v in params.items() if is_buf(k)}     # (batch-norm conv encoder)
        if self.bufs:
            import dataclasses
            self.cfg = cfg = dataclasses.replace(cfg, bufs=self.bufs)
        self.grid = generate_grid(cfg.data_dim, dtype) if cfg.coord > 0 else None
        # one Adam over all tensors is elementwise-identical to Pyro's one-Adam-per-tensor
        # (no tensor of its own when encoder AND decoder are custom modules: their owner steps them)
        self.opt = torch.optim.Adam(list(self.p.values()), lr=lr) if self.p else None
        self.dtype = dtype
        self.last = None
        self.last_grads = None

    def loss_and_grads(self, x, eps, beta=1.0, y=None):
        if self.cfg.discrete_dim > 0 and getattr(self, "sampled_class", False):
            # SVItrainer(jiVAE, enumerate_parallel=False): y = the drawn class (one-hot) or None (drawn inside)
            out = jelbo_sampled(self.p, self.cfg, x.to(self.dtype), eps.to(self.dtype),
                                None if y is None else y.to(self.dtype), beta)
        elif self.cfg.discrete_dim > 0:
            out = jelbo(self.p, self.cfg, x.to(self.dtype), eps.to(self.dtype), beta, self.grid)
        else:
            out = elbo(self.p, self.cfg, x.to(self.dtype), eps.to(self.dtype), beta,
                       None if y is None else y.to(self.dtype), self.grid)
        if out["loss"].requires_grad:
            out["loss"].backward()
        self.last = out
        return out

    def step(self, x, eps, beta=1.0, y=None) -> float:
        out = self.loss_and_grads(x, eps, beta, y)
        self.last_grads = {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in self.p.items()}
        # Under torch.no_grad() (SVItrainer.evaluate, trainers/svi.py:126-135) there is no backward,
        # but the optimizer still runs on whatever .grad holds (zero tensors after any training step).
        if any(v.grad is not None for v in self.p.values()):
            self.opt.step()
        for v in self.p.values():
            if v.grad is not None:
                v.grad = torch.zeros_like(v.grad)
        return out["loss"].item()

    def draw_eps(self, b):
        """Normal.rsample's draw: torch.empty(shape).normal_() on the global CPU generator."""
        return torch.empty(b, self.cfg.z_dim).normal_()

    def train_epoch(self, loader, beta=1.0) -> float:
        """SVItrainer.train (trainers/svi.py:95-115)."""
        total = 0.0
        for data in loader:
            x = data[0]
            y = data[1] if len(data) > 1 else None
            total += self.step(x, self.draw_eps(x.shape[0]), beta, y)
        return total / len(loader.dataset)

    def evaluate_epoch(self, loader, beta=1.0) -> float:
        """SVItrainer.evaluate (trainers/svi.py:117-137): svi.step under no_grad."""
        total = 0.0
        with torch.no_grad():
            for data in loader:
                x = data[0]
                y = data[1] if len(data) > 1 else None
                total += self.step(x, self.draw_eps(x.shape[0]), beta, y)
        return total / len(loader.dataset)

    # inference API (models/base.py:121-171, models/ivae.py:230-275)
    def encode(self, x, y=None):
        with torch.no_grad():
            if self.cfg.discrete_dim > 0:
                return jencoder_forward(self.p, self.cfg, x.to(self.dtype))      # (mu, sigma, alpha)
            return _encode_any(self.p, self.cfg, x.to(self.dtype), y)

    def decode(self, z, y=None, angle=0.0, shift=0.0, scale=1.0):
        cfg = self.cfg
        with torch.no_grad():
            z = z.to(self.dtype)
            if y is not None:
                z = torch.cat([z, y.to(self.dtype)], -1)
            if cfg.coord > 0:
                g = self.grid.unsqueeze(0)
                a = torch.as_tensor(angle, dtype=self.dtype).reshape(1)
                t = torch.as_tensor(shift, dtype=self.dtype).unsqueeze(0)
                s = torch.as_tensor(scale, dtype=self.dtype).reshape(1)
                g = transform_coordinates(g, a, t, s).squeeze(0)
                return sdecoder_forward(self.p, cfg, g.expand(z.shape[0], *g.shape), z)
            return fcdecoder_forward(self.p, cfg, z)


# ======================================================================================
# Semi-supervised models: ssiVAE (models/ssivae.py), ss_reg_iVAE (models/ss_reg_ivae.py) trained by
# auxSVItrainer (trainers/auxsvi.py).  `cfg.c_dim` is num_classes / reg_dim; the label network encoder_y is
# fcClassifierNet (nets/fc.py:240-271) or fcRegressorNet (nets/fc.py:274-304).
# ======================================================================================
def label_net_forward(p: Params, cfg: Config, x, task: str, n_hidden: int = 2):
    if cfg.custom_label_net is not None:
        return cfg.custom_label_net(x.reshape(x.shape[0], -1))
    act = _ACT[cfg.activation]
    h = _fc_stack(p, "encoder_y.fc_layers", n_hidden, act, x.reshape(x.shape[0], -1))
    out = F.linear(h, p["encoder_y.out.weight"], p["encoder_y.out.bias"])
    return torch.softmax(out, dim=-1) if task == "classification" else out


def ss_elbo(p: Params, cfg: Config, task: str, x, eps, ys=None, eps_y=None, beta=1.0, reg_sig=0.5, grid=None):
    """The ELBO step of auxSVItrainer.compute_loss (auxsvi.py:88-99) = SVI(model.model, guide, ...) with
      classification: TraceEnum_ELBO, the guide's label site enumerated in parallel when ys is None
                      (ssivae.py:150-211; auxsvi.py:73-77): rows [k][b], eps (K, B, z_dim) in one draw;
      regression:     Trace_ELBO, the guide's label a reparameterised Normal(encoder_y(x), reg_sig) sample
                      (ss_reg_ivae.py:158-199), drawn BEFORE z.
    With ys given the label is observed: iVAE's ELBO with y = ys plus the constant -sum log p(ys)."""
    b = x.shape[0]
    xf = x.reshape(b, -1)
    K = cfg.c_dim
    if task == "classification":
        if ys is None:
            alpha = label_net_forward(p, cfg, xf, task)                      # (B, K)
            y_rep = torch.eye(K, dtype=xf.dtype).repeat_interleave(b, 0)     # rows [k][b] = onehot(k)
            out = elbo(p, cfg, xf.repeat(K, 1), eps.reshape(K * b, -1), beta, y_rep, grid)
            # per-row terms: ll + beta (log p(z) - log q(z))
            z, z_loc, z_scale = out["z"], out["z_loc"], out["z_scale"]
            logq = td.Normal(z_loc, z_scale).log_prob(z).sum(-1)
            logp = td.Normal(torch.zeros_like(z), torch.ones_like(z)).log_prob(z).sum(-1)
            e = (out["ll_per_sample"] + beta * (logp - logq)).reshape(K, b)
            w = alpha.t()
            logq_y = td.OneHotCategorical(probs=alpha).log_prob(torch.eye(K, dtype=xf.dtype).unsqueeze(1))   # (K, B)
            logp_y = torch.full_like(logq_y, -math.log(K))
            loss = -((w * e).sum() + (w * (logp_y - logq_y)).sum())
            return dict(loss=loss, alpha=alpha, row_elbo=e, inner=out)
        out = elbo(p, cfg, xf, eps, beta, ys, grid)
        logp_y = td.OneHotCategorical(probs=torch.full_like(ys, 1.0 / K)).log_prob(ys).sum()
        return dict(loss=out["loss"] - logp_y, inner=out)
    if ys is None:
        c = label_net_forward(p, cfg, xf, task)
        y = c + reg_sig * eps_y
        logq_y = td.Normal(c, reg_sig).log_prob(y).sum()
        logp_y = td.Normal(torch.zeros_like(y), reg_sig).log_prob(y).sum()
        out = elbo(p, cfg, xf, eps, beta, y, grid)
        return dict(loss=out["loss"] - (logp_y - logq_y), c=c, ys=y, inner=out)
    out = elbo(p, cfg, xf, eps, beta, ys, grid)
    logp_y = td.Normal(torch.zeros_like(ys), reg_sig).log_prob(ys).sum()
    return dict(loss=out["loss"] - logp_y, inner=out)


def ss_aux_loss(p: Params, cfg: Config, task: str, x, ys, mult=20.0, reg_sig=0.5):
    """model_aux under Trace_ELBO with the empty guide_aux (ssivae.py:215-234 / ss_reg_ivae.py:221-240)."""
    out = label_net_forward(p, cfg, x.reshape(x.shape[0], -1), task)
    if task == "classification":
        return -(mult * td.OneHotCategorical(probs=out).log_prob(ys)).sum()
    return -(mult * td.Normal(out, reg_sig).log_prob(ys).sum(-1)).sum()


class SSOracle:
    """auxSVItrainer restated: compute_loss = SVI step on the ELBO + SVI step on the auxiliary loss, both through the
    SAME per-parameter Adam (auxsvi.py:60-84, 88-99): with ys = None the auxiliary step has no loss and no backward,
    yet the optimizer is still called on every parameter of the module (pyro.module registers them all), so every
    parameter that has ever had a gradient takes a momentum-only Adam step."""

    def __init__(self, params: Params, cfg: Config, task: str = "classification", lr: float = 5e-4, reg_sig: float = 0.5,
                 dtype=torch.float32):
        self.cfg, self.task, self.reg_sig, self.dtype = cfg, task, reg_sig, dtype
        self.p = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in params.items()}
        self.grid = generate_grid(cfg.data_dim, dtype) if cfg.coord > 0 else None
        self.opt = torch.optim.Adam(list(self.p.values()), lr=lr)
        self.last_grads = {}

    def _svi_step(self, loss, which):
        if torch.is_tensor(loss) and loss.requires_grad:
            loss.backward()
        self.last_grads[which] = {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in self.p.items()}
        if any(v.grad is not None for v in self.p.values()):
            self.opt.step()
        for v in self.p.values():
            if v.grad is not None:
                v.grad = torch.zeros_like(v.grad)
        return loss.item() if torch.is_tensor(loss) else float(loss)

    def draw(self, b, unlabeled):
        """The guide's draws on the global CPU generator, in the guide's order."""
        cfg, eps_y = self.cfg, None
        if self.task == "classification":
            shape = (cfg.c_dim, b, cfg.z_dim) if unlabeled else (b, cfg.z_dim)
        else:
            if unlabeled:
                eps_y = torch.empty(b, cfg.c_dim).normal_()
            shape = (b, cfg.z_dim)
        return torch.empty(shape).normal_(), eps_y

    def compute_loss(self, x, ys=None, eps=None, eps_y=None, beta=1.0, mult=20.0, after_elbo=None):
        """after_elbo (tests): called with the parameter dict between the ELBO step's Adam update and the auxiliary step."""
        x = x.to(self.dtype)
        if eps is None:
            eps, eps_y = self.draw(x.shape[0], ys is None)
        out = ss_elbo(self.p, self.cfg, self.task, x, eps.to(self.dtype), None if ys is None else ys.to(self.dtype),
                      None if eps_y is None else eps_y.to(self.dtype), beta, self.reg_sig, self.grid)
        self.last = out
        l1 = self._svi_step(out["loss"], "elbo")
        if after_elbo is not None:
            after_elbo(self.p)
        aux = ss_aux_loss(self.p, self.cfg, self.task, x, ys.to(self.dtype), mult, self.reg_sig) if ys is not None else 0.0
        l2 = self._svi_step(aux, "aux")
        return l1, l2

    def train_epoch(self, loader_unsup, loader_sup, beta=1.0, mult=20.0) -> float:
        """auxSVItrainer.train (auxsvi.py:101-127)."""
        p = (len(loader_sup) + len(loader_unsup)) // len(loader_sup)
        it_sup = iter(loader_sup)
        total, count = 0.0, 0
        for i, (xs,) in enumerate(loader_unsup):
            total += sum(self.compute_loss(xs, beta=beta, mult=mult))
            count += xs.shape[0]
            if i % p == 1:
                xs, ys = next(it_sup)
                self.compute_loss(xs, ys, beta=beta, mult=mult)
        return total / count

    def predict(self, x):
        with torch.no_grad():
            out = label_net_forward(self.p, self.cfg, x.to(self.dtype), self.task)
        return out.argmax(-1) if self.task == "classification" else out

    def evaluate(self, loader_val) -> float:
        """auxSVItrainer.evaluate_cls / evaluate_reg (auxsvi.py:139-163)."""
        correct, total = 0.0, 0
        for data, labels in loader_val:
            # model.classifier / model.regressor wrap the batch in their own DataLoader (ssivae.py:262-266): creating its
            # iterator draws a base seed from the global CPU generator, which is part of the run's RNG stream
            inner = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(data), batch_size=100, shuffle=False)
            pred = torch.cat([self.predict(x_i) for (x_i,) in inner])
            if self.task == "classification":
                correct += (pred == labels.argmax(-1)).sum().item()
                total += data.shape[0]
            else:
                correct += F.mse_loss(pred, labels.to(self.dtype)).item()
                total += 1
        return correct / total

    def encode(self, x, y):
        with torch.no_grad():
            return encoder_forward(self.p, self.cfg, x.to(self.dtype), y.to(self.dtype))

    def decode(self, z, y):
        return SVIOracle.decode(self, z, y)
