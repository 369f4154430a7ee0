"""
A CPU stand-in for pyroved_amd.engine.IVAEEngine, used ONLY by the CPU tests of the
trainer's host logic (data order, eps stream, evaluate semantics, data-parallel sharding
with gloo).  Compute is the oracle (test infrastructure); the product never selects it.
It mimics the engine's device-memory contract: a flat gradient buffer with the 4 ELBO
scalars in its last slots, `flat` parameters, `adam_step()` that also zeroes the grads.
"""
import torch

from oracle import svi_oracle as orc


class OracleEngine:
    def __init__(self, model, cfg, lr=1e-3):
        self.model = model
        self.cfg = cfg
        self.o = orc.SVIOracle(model.state_dict(), cfg, lr=lr)
        self.keys = list(self.o.p.keys())
        self.sizes = [self.o.p[k].numel() for k in self.keys]
        self.n_flat = sum(self.sizes)
        self.grad = torch.zeros(self.n_flat + 4)
        self.scalars = self.grad[self.n_flat:]
        self.flat = torch.cat([self.o.p[k].detach().reshape(-1) for k in self.keys])
        self.device = torch.device("cpu")
        self.grads_live = False
        self.adam_t = 0
        self.lr, self.betas, self.adam_eps = lr, (0.9, 0.999), 1e-8

    def _sync_params_from_flat(self):
        off = 0
        with torch.no_grad():
            for k, n in zip(self.keys, self.sizes):
                self.o.p[k].copy_(self.flat[off:off + n].view_as(self.o.p[k]))
                off += n

    def loss_and_grads(self, x, eps, beta=1.0, y=None, want_grads=True, **kw):
        self._sync_params_from_flat()
        for p in self.o.p.values():
            p.grad = None
        if want_grads:
            out = self.o.loss_and_grads(x, eps, beta, y)
        else:
            with torch.no_grad():
                out = self.o.loss_and_grads(x, eps, beta, y)
        self.scalars.copy_(torch.stack([out["loss"], out["ll"], out["logpz"], out["logqz"]]).detach())
        if want_grads:
            off = 0
            for k, n in zip(self.keys, self.sizes):
                self.grad[off:off + n].copy_(self.o.p[k].grad.reshape(-1))
                off += n
            self.grads_live = True

    def adam_step(self):
        self.adam_t += 1
        off = 0
        for k, n in zip(self.keys, self.sizes):
            self.o.p[k].grad = self.grad[off:off + n].view_as(self.o.p[k]).clone()
            off += n
        self.o.opt.step()
        self.grad[:self.n_flat].zero_()
        with torch.no_grad():
            self.flat.copy_(torch.cat([self.o.p[k].detach().reshape(-1) for k in self.keys]))

    def adam_step_hist(self, hist_slot):
        """The data-parallel step's post-all-reduce half (engine.IVAEEngine.adam_step_hist)."""
        hist_slot.copy_(self.scalars)
        self.adam_step()
