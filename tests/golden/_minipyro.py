"""
_minipyro.py — a minimal restatement of the slice of Pyro (pyro-ppl >= 1.6,
unpinned by the reference: setup.py:29) that pyroVED's SVI hot path runs
through.  TEST TOOLING ONLY: it exists so that `make_golden.py` can import and
*execute the reference's own code* (/root/reference/pyroved: models/ivae.py
model()/guide(), trainers/svi.py SVItrainer, nets/fc.py, utils/coord.py) in a
container where pyro-ppl is not installable, and record golden vectors.
It never ships to the GPU box as a dependency of anything but that script and
is never imported by the product package.

Semantics restated (from knowledge of Pyro 1.8/1.9 — "PYRO-RECALL" in
SURVEY.md; none of this arithmetic lives under /root/reference):

* ``pyro.sample(name, fn, obs=None)`` under a guide trace draws
  ``fn.rsample()`` (reparameterised Normal == loc + scale * empty().normal_());
  under a replayed model trace it returns the guide's value for the same site
  name, or ``obs`` for observed sites.  Each site records
  ``log_prob_sum = (scale * fn.log_prob(value)).sum()`` where ``scale`` is the
  product of the enclosing ``poutine.scale`` contexts.
* ``pyro.plate`` with no subsampling is a no-op for log-prob arithmetic
  (the batch dim is already dim -1 of every site; nothing is rescaled).
* ``Trace_ELBO`` (num_particles=1): ``elbo = sum(model log_prob_sum) -
  sum(guide log_prob_sum)``; with fully reparameterised latent sites
  surrogate == elbo; ``loss = -elbo``; ``loss.backward()`` unless nothing
  requires grad (e.g. under ``torch.no_grad()``).
* A guide site WITHOUT ``rsample`` (jiVAE's ``OneHotCategorical`` when the
  trainer runs with its default ``enumerate_parallel=False``) is drawn with
  ``fn.sample()`` and enters the surrogate through the score-function term
  (pyro/infer/trace_elbo.py ``_compute_log_r`` + ``score_parts``):
  ``surrogate = sum(model log_prob_sum) - sum(reparameterised guide
  log_prob_sum) + sum_b detach(log_r_b) * log q(value_b)``, with
  ``log_r_b = sum over sites of (scaled model log_prob - scaled guide log_prob)``
  of plate element b and the score function ``log q`` UNscaled
  (``ScoreParts.scale_and_mask`` leaves ``score_function`` unscaled).  The
  reported loss stays ``-(sum model - sum guide)``.
* ``SVI.step``: loss_and_grads → optimizer over every parameter registered by
  ``pyro.module`` during the step → ``zero_grads`` (grads become *zero tensors*,
  not None) → python float of the loss.
* ``pyro.optim.Adam({"lr": ...})``: one ``torch.optim.Adam`` per parameter
  tensor, created lazily the first time the parameter is seen.
* ``TraceEnum_ELBO(max_plate_nesting=1)`` + ``config_enumerate(guide,
  "parallel", expand=True)`` for a guide-side enumerated OneHotCategorical
  site: the site's value is the full support with a new leftmost dim
  (K, B, K); downstream sites broadcast against it; the ELBO is the exact
  expectation over the enumerated site:
  ``sum_b [ cont terms ] + sum_b sum_k q(k|x_b) [ log p(x_b|z,k) + s*(log p(k)
  - log q(k|x_b)) ]``.
"""
import contextlib
import sys
import types

import torch
import torch.distributions as td


# --------------------------------------------------------------------------
# distributions with .to_event()
# --------------------------------------------------------------------------
class _ToEvent:
    def to_event(self, n=None):
        if n is None:
            n = len(self.batch_shape)
        if n == 0:
            return self
        return Independent(self, n)


class Independent(td.Independent, _ToEvent):
    pass


class Normal(td.Normal, _ToEvent):
    pass


class Bernoulli(td.Bernoulli, _ToEvent):
    pass


class ContinuousBernoulli(td.ContinuousBernoulli, _ToEvent):
    pass


class OneHotCategorical(td.OneHotCategorical, _ToEvent):
    pass


class Categorical(td.Categorical, _ToEvent):
    pass


# --------------------------------------------------------------------------
# runtime state
# --------------------------------------------------------------------------
class _Ctx:
    def __init__(self, mode, replay=None, enumerate_sites=False):
        self.mode = mode                # "guide" | "model"
        self.replay = replay or {}      # name -> value (from the guide)
        self.enumerate_sites = enumerate_sites
        self.sites = {}                 # name -> dict
        self.params = {}                # id -> tensor
        self.scales = []


_CTX = []          # stack of active traces
_TAP = {}          # name -> captured tensors (eps etc.) of the last guide trace


def _scale():
    s = 1.0
    for c in _CTX[-1].scales:
        s = s * c
    return s


def sample(name, fn, obs=None, infer=None):
    if not _CTX:
        return obs if obs is not None else fn.rsample() if fn.has_rsample else fn.sample()
    ctx = _CTX[-1]
    scale = _scale()
    enumerated = False
    if obs is not None:
        value = obs
    elif ctx.mode == "model":
        value = ctx.replay[name]["value"]
    else:
        base = fn.base_dist if isinstance(fn, td.Independent) else fn
        if ctx.enumerate_sites and isinstance(base, td.OneHotCategorical):
            # parallel enumeration, expand=True: (K, *batch_shape, K)
            value = fn.enumerate_support(expand=True)
            enumerated = True
        else:
            if fn.has_rsample:
                state = torch.get_rng_state()
                value = fn.rsample()
                after = torch.get_rng_state()
                torch.set_rng_state(state)
                _TAP[name + ".eps"] = torch.empty(value.shape).normal_()
                torch.set_rng_state(after)
            else:
                with torch.no_grad():
                    value = fn.sample()              # score-function site (no reparameterisation)
    ctx.sites[name] = dict(name=name, fn=fn, value=value, scale=scale,
                           is_observed=obs is not None, enumerated=enumerated,
                           reparam=bool(fn.has_rsample) or obs is not None or ctx.mode == "model",
                           log_prob=fn.log_prob(value))
    return value


@contextlib.contextmanager
def plate(name, size=None, subsample_size=None, dim=None):
    if subsample_size is not None:
        raise NotImplementedError("minipyro: no subsampling")
    yield


def module(name, nn_module, update_module_params=False):
    if _CTX:
        for p in nn_module.parameters():
            _CTX[-1].params[id(p)] = p
    return nn_module


def clear_param_store():
    pass


@contextlib.contextmanager
def _scale_ctx(scale=1.0):
    if _CTX:
        _CTX[-1].scales.append(scale)
        try:
            yield
        finally:
            _CTX[-1].scales.pop()
    else:
        yield


def _run(fn, ctx, *args, **kwargs):
    _CTX.append(ctx)
    try:
        fn(*args, **kwargs)
    finally:
        _CTX.pop()
    return ctx


# --------------------------------------------------------------------------
# ELBOs
# --------------------------------------------------------------------------
class ELBO:
    pass


class Trace_ELBO(ELBO):
    def __init__(self, num_particles=1, **kw):
        assert num_particles == 1

    enumerate_sites = False

    def _terms(self, model, guide, *args, **kwargs):
        g = _run(guide, _Ctx("guide", enumerate_sites=self.enumerate_sites), *args, **kwargs)
        m = _run(model, _Ctx("model", replay=g.sites), *args, **kwargs)
        return g, m

    def _elbo(self, g, m):
        elbo = 0.0
        for s in m.sites.values():
            elbo = elbo + (s["scale"] * s["log_prob"]).sum()
        for s in g.sites.values():
            elbo = elbo - (s["scale"] * s["log_prob"]).sum()
        return elbo

    def _surrogate(self, g, m, elbo):
        """elbo + the score-function terms of non-reparameterised guide sites (none on the default paths)."""
        score = [s for s in g.sites.values() if not s["reparam"] and not s["enumerated"]]
        if not score:
            return elbo
        log_r = 0.0                                  # per element of the data plate (dim -1 of every site's log_prob)
        for s in m.sites.values():
            log_r = log_r + s["scale"] * s["log_prob"]
        for s in g.sites.values():
            log_r = log_r - s["scale"] * s["log_prob"]
        log_r = log_r.detach()
        sur = 0.0
        for s in m.sites.values():
            sur = sur + (s["scale"] * s["log_prob"]).sum()
        for s in g.sites.values():
            if s["reparam"]:
                sur = sur - (s["scale"] * s["log_prob"]).sum()          # entropy term
            else:
                sur = sur + (log_r * s["log_prob"]).sum()               # score function (unscaled log q)
        _TAP["log_r"] = log_r.clone()
        return sur

    def loss_and_grads(self, model, guide, *args, **kwargs):
        g, m = self._terms(model, guide, *args, **kwargs)
        params = dict(g.params)
        params.update(m.params)
        elbo = self._elbo(g, m)
        loss = -elbo
        if not self.enumerate_sites:
            sur = self._surrogate(g, m, elbo)
            if sur is not elbo:
                # the value reported is the ELBO's; gradients come from the surrogate
                loss = -(sur - sur.detach() + elbo.detach())
        _TAP["terms"] = {k: (s["scale"] * s["log_prob"]).sum().detach()
                         for tr, pre in ((g, "guide."), (m, "model."))
                         for k, s in ((pre + n, s) for n, s in tr.sites.items())}
        _TAP["sites"] = {pre + n: s["value"].detach().clone() if torch.is_tensor(s["value"]) else s["value"]
                         for tr, pre in ((g, "guide."), (m, "model."))
                         for n, s in tr.sites.items()}
        _TAP["guide_fns"] = {n: s["fn"] for n, s in g.sites.items()}
        if torch.is_tensor(loss) and loss.requires_grad:
            loss.backward()
        return loss, list(params.values())


class TraceEnum_ELBO(Trace_ELBO):
    """Exact expectation over guide-side parallel-enumerated discrete sites
    (only the structure jiVAE uses: one enumerated OneHotCategorical site
    inside one plate; max_plate_nesting=1)."""
    enumerate_sites = True

    def __init__(self, max_plate_nesting=1, strict_enumeration_warning=False, **kw):
        assert max_plate_nesting == 1

    def _elbo(self, g, m):
        enum = [s for s in g.sites.values() if s["enumerated"]]
        if not enum:
            return super()._elbo(g, m)
        assert len(enum) == 1
        e = enum[0]
        logq = e["log_prob"]                      # (K, B)
        w = logq.exp()                            # q(k | x_b), (K, B)
        elbo = 0.0
        wterms = {}
        for pre, tr, sign in (("model.", m, 1.0), ("guide.", g, -1.0)):
            for n, s in tr.sites.items():
                lp = s["scale"] * s["log_prob"]
                t = (w * lp).sum() if lp.dim() == 2 else lp.sum()   # (K, B): depends on the enumerated value
                wterms[pre + n] = t.detach()
                elbo = elbo + sign * t
        _TAP["enum_terms"] = wterms
        _TAP["enum_weights"] = w.detach().clone()
        return elbo


def config_enumerate(guide=None, default="parallel", expand=False, num_samples=None, tmc="diagonal"):
    assert default == "parallel"
    return guide


class SVI:
    def __init__(self, model, guide, optim, loss, **kw):
        self.model, self.guide, self.optim, self.loss = model, guide, optim, loss

    def step(self, *args, **kwargs):
        loss, params = self.loss.loss_and_grads(self.model, self.guide, *args, **kwargs)
        self.optim(params)
        for p in params:                          # pyro.infer.util.zero_grads
            if p.grad is not None:
                p.grad = torch.zeros_like(p.grad)
        return loss.item() if torch.is_tensor(loss) else float(loss)


class PyroOptim:
    def __init__(self, ctor, optim_args):
        self.ctor, self.optim_args, self.optim_objs = ctor, optim_args, {}

    def __call__(self, params):
        for p in params:
            if id(p) not in self.optim_objs:
                self.optim_objs[id(p)] = self.ctor([p], **self.optim_args)
            self.optim_objs[id(p)].step()


def Adam(optim_args):
    return PyroOptim(torch.optim.Adam, optim_args)


# --------------------------------------------------------------------------
# install into sys.modules
# --------------------------------------------------------------------------
def install():
    """Registers the stub as `pyro` (+ submodules) and a stub `torchvision`."""
    if "pyro" in sys.modules and not getattr(sys.modules["pyro"], "_minipyro", False):
        raise RuntimeError("a real pyro is installed; use it instead of the stub")
    pyro = types.ModuleType("pyro")
    pyro._minipyro = True
    pyro.sample, pyro.plate, pyro.module = sample, plate, module
    pyro.clear_param_store = clear_param_store

    dist = types.ModuleType("pyro.distributions")
    for k, v in dict(Normal=Normal, Bernoulli=Bernoulli, Independent=Independent,
                     ContinuousBernoulli=ContinuousBernoulli, Categorical=Categorical,
                     OneHotCategorical=OneHotCategorical, Distribution=td.Distribution).items():
        setattr(dist, k, v)
    dutil = types.ModuleType("pyro.distributions.util")
    dutil.broadcast_shape = lambda *s, **kw: torch.broadcast_shapes(*s)
    dist.util = dutil

    poutine = types.ModuleType("pyro.poutine")
    poutine.scale = _scale_ctx

    infer = types.ModuleType("pyro.infer")
    infer.SVI, infer.ELBO = SVI, ELBO
    infer.Trace_ELBO, infer.TraceEnum_ELBO = Trace_ELBO, TraceEnum_ELBO
    infer.config_enumerate = config_enumerate

    optim = types.ModuleType("pyro.optim")
    optim.Adam, optim.PyroOptim = Adam, PyroOptim

    contrib = types.ModuleType("pyro.contrib")
    gp = types.ModuleType("pyro.contrib.gp")
    contrib.gp = gp

    pyro.distributions, pyro.poutine, pyro.infer, pyro.optim, pyro.contrib = dist, poutine, infer, optim, contrib
    mods = {"pyro": pyro, "pyro.distributions": dist, "pyro.distributions.util": dutil,
            "pyro.poutine": poutine, "pyro.infer": infer, "pyro.optim": optim,
            "pyro.contrib": contrib, "pyro.contrib.gp": gp}
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvu = types.ModuleType("torchvision.utils")
        tvu.make_grid = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("stub"))
        tv.utils = tvu
        mods.update({"torchvision": tv, "torchvision.utils": tvu})
    sys.modules.update(mods)
    return pyro


def tap():
    """Tensors captured during the last SVI step (eps, per-site terms, values)."""
    return _TAP
