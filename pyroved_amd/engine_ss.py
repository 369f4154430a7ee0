"""
engine_ss.py — host-side driver of the semi-supervised models' steps (models.ssiVAE, models.ss_reg_iVAE trained by
trainers.auxSVItrainer; reference: pyroved/models/ssivae.py, ss_reg_ivae.py, trainers/auxsvi.py).

Extends IVAEEngine: one flat parameter / gradient / Adam buffer holds encoder_z, encoder_y and decoder.  Every step is
  * pv_ivae_loss_and_grads on encoder_z / decoder with the label vector as the conditioning input y — observed;
    enumerated over the K classes (the K*B rows (x_b, onehot(k)) laid out [k][b], per-row weights q(k | x_b), per-row
    ELBO terms handed back); or a reparameterised sample (dloss/dy handed back);
  * the label network encoder_y through pv_mlp_forward / pv_mlp_backward;
  * the objective's small kernels (pv_ss_*);
  * pv_adam_step over the whole flat buffer.
Replicating / transposing the inputs of the enumerated pass is tensor plumbing in torch; all arithmetic is HIP.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _abi
from .engine import IVAEEngine
from .nets.fc import fcClassifierNet, fcRegressorNet
from ._convplan import UnsupportedModel


class SSEngine(IVAEEngine):
    supports_step = False            # (pv_ivae_step covers the iVAE / jiVAE plan only)
    supports_scalars_out = False

    def _check_model(self):
        m = self.model
        net = m.encoder_y
        # a user-defined label network (set_classifier / set_regressor, ssivae.py:236-240) runs in PyTorch on the device:
        # its output enters the HIP steps, dloss/d(output) comes back and torch.autograd carries it into the module
        self.ext_y = not isinstance(net, (fcClassifierNet, fcRegressorNet))
        super()._check_model()
        if self.ext_enc or self.ext_dec:
            raise UnsupportedModel("user-defined encoder_z / decoder are not combined with the semi-supervised models here")
        self.task = "classification" if hasattr(m, "num_classes") else "regression"
        if not self.ext_y and len([q for q in net.fc_layers if isinstance(q, nn.Linear)]) > _abi.PV_MAX_LAYERS:
            raise UnsupportedModel("more than %d hidden layers" % _abi.PV_MAX_LAYERS)
        self.reg_sig = float(getattr(m, "reg_sig", 0.5))
        self.ws_y = None

    def bind(self):
        super().bind()
        self._scal = torch.zeros(8, device=self.device, dtype=torch.float32)      # loss_add, aux loss
        self._mlp = None if self.ext_y else self._mlp_plan()
        self._in_dim = self.model.encoder_z.in_dim - self.model.c_dim
        self._y_graph = None
        self.ws_y = None

    def _mlp_plan(self) -> _abi.pv_mlp_plan:
        net = self.model.encoder_y
        q = _abi.pv_mlp_plan()
        q.in_dim = net.in_dim
        idx = [i for i, mod in enumerate(net.fc_layers) if isinstance(mod, nn.Linear)]
        q.n_layers = len(idx)
        for j, i in enumerate(idx):
            q.layers[j] = self._layer("encoder_y.fc_layers.%d" % i, net.fc_layers[i], net.activation)
        q.out = self._layer("encoder_y.out", net.out, None)
        q.out_kind = _abi.MLP_OUT["softmax" if self.task == "classification" else "linear"]
        q.params, q.grads = self.flat.data_ptr(), self.grad.data_ptr()
        return q

    def _mlp_for(self, x):
        self.ensure_bound()
        q = self._mlp
        q.batch = x.shape[0]
        need = _abi.lib().pv_mlp_workspace_bytes(C.byref(q))
        if need < 0:
            raise _abi.PvError("pyroved_amd: unsupported label network (pv_mlp_workspace_bytes -> %d)" % need)
        if self.ws_y is None or self.ws_y.numel() < need:
            self.ws_y = torch.empty(int(need), device=self.device, dtype=torch.uint8)
        q.ws, q.ws_bytes = self.ws_y.data_ptr(), self.ws_y.numel()
        q.x = x.data_ptr()
        return q

    # ------------------------------------------------------------------ label network
    @_abi.on_device
    def label_forward(self, x: torch.Tensor) -> torch.Tensor:
        """encoder_y(x): class probabilities (B, K) or regression means (B, c)."""
        x = self._prep(x, "x", (x.shape[0], self._in_dim))
        if self.ext_y:
            with torch.set_grad_enabled(torch.is_grad_enabled()):
                out = self.model.encoder_y(x)
            self._y_graph = out if out.requires_grad else None
            return out.detach().to(torch.float32).contiguous()
        q = self._mlp_for(x)
        out = torch.empty(x.shape[0], q.out.out_dim, device=self.device, dtype=torch.float32)
        _abi.check(_abi.lib().pv_mlp_forward(C.byref(q), _abi.ptr(out), _abi.current_stream()), "pv_mlp_forward")
        self._keep_y = (x, out)
        return out

    @_abi.on_device
    def label_backward(self, x: torch.Tensor, out: torch.Tensor, dout: torch.Tensor) -> None:
        """Gradients of encoder_y's parameters from dloss/d(out), after label_forward(x)."""
        if self.ext_y:
            if self._y_graph is None:
                raise RuntimeError("label_backward needs a preceding label_forward with gradients enabled")
            for q in self._enc_params:
                q.grad = None
            torch.autograd.backward([self._y_graph], [dout.to(self._y_graph.dtype)])
            self._y_graph = None
            self.grads_live = True
            return
        x = self._prep(x, "x", (x.shape[0], self._mlp.in_dim))
        q = self._mlp_for(x)
        _abi.check(_abi.lib().pv_mlp_backward(C.byref(q), _abi.ptr(out), _abi.ptr(dout), _abi.current_stream()),
                   "pv_mlp_backward")
        self.grads_live = True
        self._keep_y = (x, out, dout)

    # ------------------------------------------------------------------ the two SVI steps of auxSVItrainer.compute_loss
    @_abi.on_device
    def elbo_loss_and_grads(self, x, eps, ys=None, eps_y=None, beta: float = 1.0) -> torch.Tensor:
        """The ELBO step (SVI(model.model, guide): auxsvi.py:73-81).  Returns the loss as a 0-dim device tensor;
        gradients land in self.grad."""
        lib, st = _abi.lib(), _abi.current_stream
        b = x.shape[0]
        x = self._prep(x, "x", (b, self._in_dim))
        c = self.model.c_dim
        add = self._scal[0:1]
        if ys is not None:                                    # observed label: iVAE's ELBO with y = ys + a constant
            ys = self._prep(ys, "ys", (b, c))
            self.loss_and_grads(x, eps, beta, y=ys)
            if self.task == "classification":
                add.fill_(b * float(torch.log(torch.tensor(float(c)))))      # -sum log(1/K)
            else:
                _abi.check(lib.pv_ss_reg_terms(None, None, _abi.ptr(ys), None, b, c, self.reg_sig, None, _abi.ptr(add),
                                               st()), "pv_ss_reg_terms")
            return self.scalars[0] + add[0]
        if self.task == "classification":
            # TraceEnum_ELBO over the guide's enumerated label (ssivae.py:197-211): K passes laid out [k][b]
            alpha = self.label_forward(x)
            s = c * b
            x_rep = x.repeat(c, 1)
            y_rep = torch.eye(c, device=self.device, dtype=torch.float32).repeat_interleave(b, 0)
            w = alpha.t().contiguous().reshape(s)
            e = torch.empty(s, device=self.device, dtype=torch.float32)
            self.loss_and_grads(x_rep, eps.reshape(s, -1), beta, y=y_rep, row_w=w, row_elbo=e)
            dalpha = torch.empty_like(alpha)
            _abi.check(lib.pv_ss_enum_terms(_abi.ptr(alpha), _abi.ptr(e), b, c, _abi.ptr(dalpha), _abi.ptr(add), st()),
                       "pv_ss_enum_terms")
            self.label_backward(x, alpha, dalpha)
            self._keep_ss = (x_rep, y_rep, w, e, dalpha, alpha)
            return self.scalars[0] + add[0]
        # regression: the guide's label is a reparameterised sample (ss_reg_ivae.py:196-199)
        cm = self.label_forward(x)
        eps_y = self._prep(eps_y, "eps_y", (b, c))
        ysamp = torch.empty_like(cm)
        _abi.check(lib.pv_ss_reg_sample(_abi.ptr(cm), _abi.ptr(eps_y), b, c, self.reg_sig, _abi.ptr(ysamp), st()),
                   "pv_ss_reg_sample")
        dy = torch.empty_like(cm)
        self.loss_and_grads(x, eps, beta, y=ysamp, dy=dy)
        dc = torch.empty_like(cm)
        _abi.check(lib.pv_ss_reg_terms(_abi.ptr(cm), _abi.ptr(eps_y), _abi.ptr(ysamp), _abi.ptr(dy), b, c, self.reg_sig,
                                       _abi.ptr(dc), _abi.ptr(add), st()), "pv_ss_reg_terms")
        self.label_backward(x, cm, dc)
        self._keep_ss = (cm, eps_y, ysamp, dy, dc)
        return self.scalars[0] + add[0]

    @_abi.on_device
    def aux_loss_and_grads(self, x, ys, multiplier: float = 20.0) -> torch.Tensor:
        """The auxiliary (supervised) step (SVI(model.model_aux, guide_aux): auxsvi.py:82-84) for a labeled batch."""
        b = x.shape[0]
        c = self.model.c_dim
        ys = self._prep(ys, "ys", (b, c))
        out = self.label_forward(x)
        dout = torch.empty_like(out)
        loss = self._scal[1:2]
        _abi.check(_abi.lib().pv_ss_aux_loss(_abi.SS_TASK[self.task], _abi.ptr(out), _abi.ptr(ys), b, c, float(multiplier),
                                             self.reg_sig, _abi.ptr(dout), _abi.ptr(loss), _abi.current_stream()),
                   "pv_ss_aux_loss")
        self.label_backward(x, out, dout)
        self._keep_aux = (ys, out, dout)
        return loss[0]
