"""
A CPU stand-in for pyroved_amd.engine_ss.SSEngine, used ONLY by the CPU tests of auxSVItrainer's host logic (loader
interleaving, noise stream, the two optimizer steps per call, data-parallel sharding with gloo).  Compute is the oracle
(test infrastructure); the product never selects it.  Same memory contract as the real engine: flat parameters, a flat
gradient buffer with 4 trailing scalar slots, adam_step() that also zeroes the gradients.
"""
import torch

from oracle import svi_oracle as orc


class OracleSSEngine:
    def __init__(self, model, cfg, task, lr=5e-4):
        self.model, self.cfg, self.task = model, cfg, task
        self.o = orc.SSOracle(model.state_dict(), cfg, task, lr=lr)
        self.keys = list(self.o.p.keys())
        self.sizes = [self.o.p[k].numel() for k in self.keys]
        self.n_flat = sum(self.sizes)
        self.grad = torch.zeros(self.n_flat + 4)
        self.scalars = self.grad[self.n_flat:]
        self.flat = torch.cat([self.o.p[k].detach().reshape(-1) for k in self.keys])
        self.device = torch.device("cpu")
        self.grads_live = False
        self.lr, self.betas, self.adam_eps = lr, (0.9, 0.999), 1e-8

    def _sync_params_from_flat(self):
        off = 0
        with torch.no_grad():
            for k, n in zip(self.keys, self.sizes):
                self.o.p[k].copy_(self.flat[off:off + n].view_as(self.o.p[k]))
                off += n

    def _collect(self, loss):
        for p in self.o.p.values():
            p.grad = None
        if torch.is_tensor(loss) and loss.requires_grad:
            loss.backward()
        off = 0
        for k, n in zip(self.keys, self.sizes):
            g = self.o.p[k].grad
            self.grad[off:off + n].copy_(g.reshape(-1) if g is not None else torch.zeros(n))
            off += n
        self.grads_live = True
        return loss.detach() if torch.is_tensor(loss) else torch.tensor(float(loss))

    def elbo_loss_and_grads(self, x, eps, ys=None, eps_y=None, beta=1.0):
        self._sync_params_from_flat()
        out = orc.ss_elbo(self.o.p, self.cfg, self.task, x, eps, ys, eps_y, beta, self.o.reg_sig, self.o.grid)
        return self._collect(out["loss"])

    def aux_loss_and_grads(self, x, ys, multiplier=20.0):
        self._sync_params_from_flat()
        return self._collect(orc.ss_aux_loss(self.o.p, self.cfg, self.task, x, ys, multiplier, self.o.reg_sig))

    def label_forward(self, x):
        self._sync_params_from_flat()
        with torch.no_grad():
            return orc.label_net_forward(self.o.p, self.cfg, x, self.task)

    def adam_step(self):
        off = 0
        for k, n in zip(self.keys, self.sizes):
            self.o.p[k].grad = self.grad[off:off + n].view_as(self.o.p[k]).clone()
            off += n
        self.o.opt.step()
        self.grad[:self.n_flat].zero_()
        with torch.no_grad():
            self.flat.copy_(torch.cat([self.o.p[k].detach().reshape(-1) for k in self.keys]))
