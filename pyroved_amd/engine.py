"""
engine.py — host-side driver of the HIP SVI step for models.iVAE.

Owns the device memory layout the C ABI works on:
  * ONE flat fp32 parameter buffer; every nn.Parameter of the model's encoder_z
    and decoder is re-pointed to a view of it (state_dict keys and values are
    unchanged, save/load keep working), laid out so that fc11/fc12 are adjacent
    (the kernels treat them as one Linear of width 2*z_dim);
  * flat gradient buffer with 4 trailing slots for the ELBO scalars, so that the
    data-parallel path needs exactly one all-reduce (grads + loss);
  * flat Adam moment buffers and the workspace.
and fills a `pv_ivae_plan` (include/pyroved_amd.h) per call.

Replaces, for the product, what Pyro's SVI/Trace_ELBO/PyroOptim objects hold in
the reference (pyroved/trainers/svi.py:79-91).
"""
import ctypes as C
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _abi
from .nets.fc import fcEncoderNet, jfcEncoderNet, fcDecoderNet, sDecoderNet
from .nets.conv import convEncoderNet
from ._convplan import UnsupportedModel, conv_ops, fill_ops, bn_modules

ALIGN = 64      # floats: every tensor starts on a 256-byte boundary of the flat buffer
N_SCALARS = 4   # loss, ll, beta*log p(z), beta*log q(z|x)


def _linears(seq: nn.Sequential) -> List[nn.Linear]:
    return [m for m in seq if isinstance(m, nn.Linear)]


class IVAEEngine:
    """Binds an iVAE-like model (encoder_z: fcEncoderNet, decoder: sDecoderNet | fcDecoderNet)
    to the HIP library."""
    supports_dp_step = True          # loss_and_grads(step=True, comm=NativeComm): the data-parallel step as one library call
    supports_scalars_out = True      # loss_and_grads can write the 4 loss scalars to a caller-given device slot
    supports_step = True             # loss_and_grads(step=True) = SVI.step in one library call
    # per-plan switches (ABI v14 / v15 plan fields; the library keeps no process-wide switch).  Set on an engine — or, in tests,
    # on the class — before the next call:
    dec_kernel = 0                   # pv_ivae_plan.dec_kernel: 0 = the library picks the decoder-kernel build by problem size
    enc_two_launch = False           # PV_PLAN_ENC_TWO_LAUNCH
    enc_no_wait = False              # PV_PLAN_ENC_NO_WAIT: the one-launch encoder's consumers compute their tiles themselves
    side_stream = True               # PV_PLAN_NO_SIDE_STREAM when False
    dec1d = True                     # PV_PLAN_NO_DEC1D when False (VED's Conv1d decoder layer by layer)
    enc_per_image = True             # the guide of a training step as one workgroup per image (False: PV_PLAN_ENC_TILED, the tiled one-launch encoder)
    conv_x3 = False                  # PV_PLAN_CONV_X3 / conv_bf16 = 0: fp32-class kernel-3 convolutions with both operands as two fp16 pieces
    enc_fold = True                  # PV_PLAN_NO_ENC_FOLD when False (the guide as its own launch even where the decoder launch could host it)

    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, fused: int = 2):
        self.model = model
        self.lr, self.betas, self.adam_eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.fused = int(fused)         # 0 layered kernels, 1 fused f32-MFMA decoder, 2 fused bf16x3 decoder, 3 fused plain-bf16 decoder
        self.adam_t = 0                 # number of optimizer steps taken (incl. evaluate()'s, see SVItrainer)
        self.grads_live = False         # reference: .grad is None until the first backward
        self.device = None
        self.flat = self.grad = self.m = self.v = None
        self.ws = None
        self.events = (None, None)      # optional raw hipEvent_t pair recorded around the dominant kernel
        self._bn_enc, self._bn_dec = [], []   # batch-norm modules of the convolutional stacks
        self._enc_opt = None            # torch.optim.Adam of a user-defined encoder's parameters (ext_enc)
        self._enc_params = []
        self._layout: Dict[str, int] = {}
        self._views: Dict[str, torch.Tensor] = {}
        self._check_model()
        self.bind()

    # ------------------------------------------------------------------ structure
    def _check_model(self):
        m = self.model
        enc, dec = m.encoder_z, m.decoder
        self.conv_enc = isinstance(enc, convEncoderNet)          # iVAE.set_encoder(convEncoderNet(...))
        # any other user module (models/base.py:173-177: "Sets a user-defined encoder neural network") runs in PyTorch
        # on the device: its (z_loc, z_scale) outputs enter the HIP step through plan.ext_head, the gradients come back
        # through plan.ext_dhead and are back-propagated with torch.autograd; its parameters get their own Adam
        self.ext_enc = not isinstance(enc, (fcEncoderNet, jfcEncoderNet, convEncoderNet))
        # a user-defined decoder (models/base.py:179-183) likewise: the library runs the guide half of the step
        # (pv_ivae_guide / pv_ivae_guide_backward), the decoder, the coordinate transform and the likelihood run in PyTorch
        self.ext_dec = not isinstance(dec, (sDecoderNet, fcDecoderNet))
        self.K = 0
        if self.ext_dec:
            if getattr(m, "discrete_dim", 0):
                raise UnsupportedModel("a user-defined decoder cannot be combined with discrete latents here")
            if m.sampler_d.name not in _abi.LIK:
                raise UnsupportedModel("decoder sampler %r is not implemented" % m.sampler_d.name)
        if self.ext_enc:
            if getattr(m, "c_dim", 0) != 0 or getattr(m, "discrete_dim", 0):
                raise UnsupportedModel("a user-defined encoder cannot be combined with c_dim / discrete latents here")
        elif self.conv_enc:
            if tuple(enc.input_dim) != tuple(int(d) for d in m.data_dim) or enc.input_channels != 1:
                raise UnsupportedModel("conv encoder: input_dim must equal the model's data_dim, one input channel")
            if enc.latent_dim != m.z_dim or getattr(m, "c_dim", 0) != 0:
                raise UnsupportedModel("conv encoder: latent_dim must equal the model's z_dim (latent + coord); no c_dim")
            conv_ops(enc.feature_extractor.layers, enc.feature_extractor.activation)      # validates
        elif not isinstance(enc, (fcEncoderNet, jfcEncoderNet)):
            raise UnsupportedModel("the HIP SVI path needs encoder_z to be pyroved_amd.nets.fcEncoderNet / "
                                   "jfcEncoderNet / convEncoderNet (got %s)" % type(enc).__name__)
        self.K = int(getattr(m, "discrete_dim", 0)) if isinstance(enc, jfcEncoderNet) else 0
        if self.ext_enc and self.ext_dec:
            return                                   # both user modules: the library keeps the reparameterisation + KL
        if self.ext_enc and isinstance(dec, (sDecoderNet, fcDecoderNet)):
            if len(_linears(dec.fc_layers)) > _abi.PV_MAX_LAYERS:
                raise UnsupportedModel("more than %d hidden layers" % _abi.PV_MAX_LAYERS)
            name = m.sampler_d.name
            if name not in _abi.LIK or (name != "gaussian" and not dec.sigmoid_out):
                raise UnsupportedModel("decoder sampler %r / sigmoid_d combination is not implemented" % name)
            if (m.coord > 0) != isinstance(dec, sDecoderNet):
                raise UnsupportedModel("invariant models need the spatial decoder, vanilla models fcDecoderNet")
            return
        if not self.ext_enc and not enc.softplus_out:
            raise UnsupportedModel("encoder without softplus_out is not supported")
        if not self.conv_enc and len(_linears(enc.fc_layers)) > _abi.PV_MAX_LAYERS:
            raise UnsupportedModel("more than %d hidden layers" % _abi.PV_MAX_LAYERS)
        if self.ext_dec:
            return
        if m.coord > 0 and not isinstance(dec, sDecoderNet):
            raise UnsupportedModel("invariant models need the spatial decoder")
        if m.coord == 0 and not isinstance(dec, fcDecoderNet):
            raise UnsupportedModel("vanilla models need fcDecoderNet")
        name = m.sampler_d.name
        if name not in _abi.LIK:
            raise UnsupportedModel("decoder sampler %r is not implemented in the HIP path yet" % name)
        if name in ("bernoulli", "continuous_bernoulli") and not dec.sigmoid_out:
            raise UnsupportedModel("%s likelihood needs sigmoid_d=True" % name)
        if len(_linears(dec.fc_layers)) > _abi.PV_MAX_LAYERS:
            raise UnsupportedModel("more than %d hidden layers" % _abi.PV_MAX_LAYERS)

    def _param_order(self):
        """(key, tensor) in flat-buffer order: state_dict order, except that the heads are merged:
        fc11.weight, fc12.weight[, fc13.weight], then fc11.bias, fc12.bias[, fc13.bias]."""
        named = dict(self.model.named_parameters())
        if self.ext_enc:                         # only the decoder lives in the flat buffers (nothing if it is a user's too)
            return [(k, v) for k, v in named.items()
                    if not k.startswith("encoder_z.") and not (self.ext_dec and k.startswith("decoder."))]
        if self.ext_dec:                         # only the encoder does
            named = {k: v for k, v in named.items() if not k.startswith("decoder.")}
        if getattr(self, "ext_y", False):        # a user-defined label network (semi-supervised models) stays in torch
            named = {k: v for k, v in named.items() if not k.startswith("encoder_y.")}
        if self.conv_enc:
            return list(named.items())           # features2latent.fc_latent already is the merged [mu | sigma] head
        heads = ["fc11", "fc12"] + (["fc13"] if self.K > 0 else [])
        merged = ["encoder_z.%s.weight" % h for h in heads] + ["encoder_z.%s.bias" % h for h in heads]
        order = []
        for k in named:
            if k == merged[0]:
                order.extend(merged)
            elif k not in merged:
                order.append(k)
        return [(k, named[k]) for k in order]

    def _stat_buffers(self):
        """(key, tensor) of the batch-norm running statistics: they live in the flat buffer next to the parameters (the
        kernels update them in place; they never get a gradient, so Adam leaves them alone)."""
        return [(k, b) for k, b in self.model.named_buffers()
                if k.endswith(".running_mean") or k.endswith(".running_var")]

    def bind(self):
        """(Re)builds the flat buffers from the model's current parameters and re-points the
        parameters at them.  Called at construction and whenever the parameters were moved."""
        self._conv_slices = None                # (re-derived, and the weight-range check runs at the next call)
        self.wide_weights = False               # this model's kernel-3 weights fit the fp16-piece kernels until a check says otherwise
        items = self._param_order()
        n_par = len(items)
        items = items + self._stat_buffers()
        dev = items[0][1].device if items else next(self.model.parameters()).device
        if dev.type != "cuda":
            raise _abi.PvError(
                "pyroved_amd: the model lives on %s; the SVI path runs only on a HIP device "
                "(no CPU fallback). Construct the model with device='cuda'." % dev)
        _abi.lib()
        off = 0
        layout = {}
        for k, p in items:
            # tensors that the kernels address as one matrix must be packed back to back
            packed = k in ("encoder_z.fc12.weight", "encoder_z.fc12.bias", "encoder_z.fc13.weight", "encoder_z.fc13.bias")
            if not packed:
                off = (off + ALIGN - 1) // ALIGN * ALIGN
            layout[k] = off
            off += p.numel()
        total = max((off + ALIGN - 1) // ALIGN * ALIGN, ALIGN)      # (never empty: the ABI wants non-NULL buffers)
        old_m, old_v, old_layout = self.m, self.v, self._layout
        flat = torch.zeros(total, device=dev, dtype=torch.float32)
        for k, p in items:
            flat[layout[k]:layout[k] + p.numel()].copy_(p.detach().reshape(-1))
        self.flat = flat
        self.grad = torch.zeros(total + N_SCALARS, device=dev, dtype=torch.float32)
        self.m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.v = torch.zeros(total, device=dev, dtype=torch.float32)
        if old_m is not None and old_layout == layout and old_m.numel() == total:
            self.m.copy_(old_m)
            self.v.copy_(old_v)
        self._views = {}
        for j, (k, p) in enumerate(items):
            view = flat[layout[k]:layout[k] + p.numel()].view(p.shape)
            if j < n_par:
                p.data = view
            else:                                    # a registered buffer: re-point the module's entry
                owner, _, leaf = k.rpartition(".")
                self.model.get_submodule(owner)._buffers[leaf] = view
            self._views[k] = view
        self._layout = layout
        self.n_flat = total
        self.device = dev
        self.scalars = self.grad[total:total + N_SCALARS]
        self.grid = self.model.grid.to(dev).contiguous() if self.model.coord > 0 else None
        self.ws = None
        if self.ext_enc or self.ext_dec or getattr(self, "ext_y", False):
            # the user module's parameters: their own torch Adam (zero_grads semantics)
            owners = ([self.model.encoder_z] if self.ext_enc else []) + ([self.model.decoder] if self.ext_dec else [])
            if not owners:
                owners = [self.model.encoder_y]
            self._enc_params = [q for o_ in owners for q in o_.parameters() if q.requires_grad]
            if self._enc_opt is None or [id(q) for q in self._enc_opt.param_groups[0]["params"]] != [id(q) for q in self._enc_params]:
                self._enc_opt = torch.optim.Adam(self._enc_params, lr=self.lr, betas=self.betas, eps=self.adam_eps)
        self._static = self._static_plan()

    def configure(self, lr=None, betas=None, eps=None, fused=None):
        """Applies trainer-level settings to an engine that already exists (model.engine(**kw) on a model whose engine
        was created earlier — by encode(), manifold2d(), a previous trainer — must not silently drop them)."""
        if lr is not None:
            self.lr = float(lr)
        if betas is not None:
            self.betas = (float(betas[0]), float(betas[1]))
        if eps is not None:
            self.adam_eps = float(eps)
        if fused is not None and int(fused) != self.fused:
            self.fused = int(fused)
            self.ws = None
        if self.flat is not None:
            self._static = self._static_plan()
        return self

    def reset_optimizer(self):
        """A fresh optimizer, as every trainer constructor of the reference makes one (trainers/svi.py:75-81:
        pyro.clear_param_store() + a new optim.Adam): zero moments, step count 0, no live gradients."""
        self.adam_t = 0
        self.grads_live = False
        if self.m is not None:
            self.m.zero_()
            self.v.zero_()
            self.grad.zero_()
        if self._enc_opt is not None:
            self._enc_opt = torch.optim.Adam(self._enc_params, lr=self.lr, betas=self.betas, eps=self.adam_eps)
            for q in self._enc_params:
                q.grad = None

    def _bound_snapshot(self):
        """What _bound's fast path compares against: the module tree (every submodule by identity, the number of parameters
        and buffers each one holds) and, per bound tensor, its owner's dict, leaf name, address and shape."""
        mods = [(mod, len(mod._parameters), len(mod._buffers), tuple(mod._modules.items())) for mod in self.model.modules()]
        tens = []
        for k, v in self._views.items():
            owner, _, leaf = k.rpartition(".")
            mod = self.model.get_submodule(owner) if owner else self.model
            tens.append((mod._parameters if leaf in mod._parameters else mod._buffers, leaf, v.data_ptr(), v.shape))
        return mods, tens

    def _bound(self) -> bool:
        """Do the model's parameters (and batch-norm statistics) still live in the flat buffer?  Asked at every call: the
        fast path walks the snapshot taken at the last full check (a few microseconds; nn.Module.named_parameters() of a
        small model costs ~40 — a third of a 0.1 ms step's host time); anything that differs from it — a tensor moved,
        replaced, added or removed, a submodule swapped — falls through to the full comparison."""
        snap = getattr(self, "_snap", None)
        if snap is not None and snap[2] is self._views:
            ok = True
            for mod, npar, nbuf, kids in snap[0]:
                md = mod._modules
                if len(mod._parameters) != npar or len(mod._buffers) != nbuf or len(md) != len(kids):
                    ok = False
                    break
                for name, child in kids:
                    if md.get(name) is not child:
                        ok = False
                        break
                if not ok:
                    break
            if ok:
                for d, leaf, ptr, shape in snap[1]:
                    t = d.get(leaf)
                    if t is None or t.data_ptr() != ptr or t.shape != shape:
                        ok = False
                        break
            if ok:
                return True
        self._snap = None
        if self._views is None or not self._bound_full():
            return False
        self._snap = self._bound_snapshot() + (self._views,)
        return True

    def _bound_full(self) -> bool:
        named = dict(self.model.named_parameters())
        if self.ext_enc:
            named = {k: v for k, v in named.items() if not k.startswith("encoder_z.")}
        if self.ext_dec:
            named = {k: v for k, v in named.items() if not k.startswith("decoder.")}
        if getattr(self, "ext_y", False):
            named = {k: v for k, v in named.items() if not k.startswith("encoder_y.")}
        named.update(dict(self._stat_buffers()))
        if len(named) != len(self._views):
            return False
        for k, v in self._views.items():
            p = named.get(k)
            if p is None or p.data_ptr() != v.data_ptr() or p.shape != v.shape:
                return False
        return True

    # ---- numeric range of the fp16-piece convolution kernels (include/pyroved_amd.h: pv_ivae_plan.conv_wide) ----
    _CONV_W_HI, _CONV_W_LO, _CONV_W_EVERY = 500.0, 2.0 ** -16, 64

    def _conv3_weight_slices(self):
        """(offset, numel) in the flat buffer of every kernel-3 convolution weight the model's conv stacks hold."""
        out = []
        for key, par in self._views.items():
            if key.endswith(".weight") and par.dim() >= 3 and par.shape[-1] == 3:
                out.append((self._layout[key], par.numel()))
        return out

    def _check_conv_weight_range(self, force: bool = False):
        """At the first call after a bind (training step, encode or decode alike) and every _CONV_W_EVERY-th call: if a
        kernel-3 convolution weight of THIS model left the range the fp16-piece kernels are exact in, its plans ask for the
        unbounded three-piece bf16 kernels from then on (plan field, ABI v14; other models in the process are unaffected).
        One scalar read-back per 64 calls; Adam moves a weight by at most lr per step, so the margin to fp16's limit (1023)
        cannot be crossed between two checks.  The read-back synchronises: while the current stream is being captured the
        check is skipped (captured steps are never range-checked — check before capturing)."""
        if getattr(self, "_conv_slices", None) is None:
            self._conv_slices = self._conv3_weight_slices()
            self._conv_tick = 0
        if not self._conv_slices or self.wide_weights:
            return
        self._conv_tick += 1
        if not force and self._conv_tick % self._CONV_W_EVERY != 1:
            return
        if torch.cuda.is_current_stream_capturing():
            self._conv_tick -= 1                # (the next uncaptured call checks)
            return
        mx = float(torch.stack([self.flat[o:o + n].abs().max() for o, n in self._conv_slices]).max())
        if mx >= self._CONV_W_HI or 0.0 < mx < self._CONV_W_LO or mx != mx:
            import warnings
            warnings.warn("pyroved_amd: a convolution weight reached |w| = %.3g, outside the range of the fp16-piece "
                          "kernels; this model switches to the range-free bf16-piece convolution kernels (three pieces at "
                          "fp32-class precision, the two-piece mixed form at the throughput precision)" % mx)
            self.wide_weights = True

    def _plan_flags(self) -> int:
        """pv_ivae_plan.flags / pv_ved_plan.flags from the engine's switches (ABI v14; process-wide setters before)."""
        return ((_abi.PV_PLAN_ENC_TWO_LAUNCH if getattr(self, "enc_two_launch", False) else 0) |
                (0 if getattr(self, "side_stream", True) else _abi.PV_PLAN_NO_SIDE_STREAM) |
                (_abi.PV_PLAN_ENC_NO_WAIT if getattr(self, "enc_no_wait", False) else 0) |
                (0 if getattr(self, "dec1d", True) else _abi.PV_PLAN_NO_DEC1D) |
                (0 if getattr(self, "enc_fold", True) else _abi.PV_PLAN_NO_ENC_FOLD) |
                (0 if getattr(self, "enc_per_image", True) else _abi.PV_PLAN_ENC_TILED) |
                (_abi.PV_PLAN_CONV_X3 if getattr(self, "conv_x3", False) else 0))

    def ensure_bound(self):
        if not self._bound():
            self.bind()

    # ------------------------------------------------------------------ plan
    def _layer(self, prefix: str, lin: nn.Linear, act) -> _abi.pv_layer:
        l = _abi.pv_layer()
        l.in_dim, l.out_dim, l.act = lin.in_features, lin.out_features, _abi.ACT[act]
        l.w_off = self._layout[prefix + ".weight"]
        l.b_off = self._layout[prefix + ".bias"] if lin.bias is not None else -1
        return l

    def _fc_encoder_plan(self, p, enc, m):
        idx = [i for i, mod in enumerate(enc.fc_layers) if isinstance(mod, nn.Linear)]
        p.n_enc = len(idx)
        for j, i in enumerate(idx):
            p.enc[j] = self._layer("encoder_z.fc_layers.%d" % i, enc.fc_layers[i], enc.activation)
        h = _abi.pv_layer()
        h.in_dim, h.out_dim, h.act = enc.fc11.in_features, 2 * m.z_dim + self.K, 0
        h.w_off, h.b_off = self._layout["encoder_z.fc11.weight"], self._layout["encoder_z.fc11.bias"]
        assert self._layout["encoder_z.fc12.weight"] == h.w_off + enc.fc11.weight.numel()
        assert self._layout["encoder_z.fc12.bias"] == h.b_off + enc.fc11.bias.numel()
        p.discrete_dim = self.K
        if self.K > 0:
            assert self._layout["encoder_z.fc13.weight"] == h.w_off + 2 * enc.fc11.weight.numel()
            assert self._layout["encoder_z.fc13.bias"] == h.b_off + 2 * enc.fc11.bias.numel()
        p.head = h

    def _static_plan(self) -> _abi.pv_ivae_plan:
        m = self.model
        enc, dec = m.encoder_z, m.decoder
        p = _abi.pv_ivae_plan()
        p.n_pix = 1
        for d in m.data_dim:
            p.n_pix *= int(d)
        p.coord_dim = 0 if m.coord == 0 else (1 if m.ndim == 1 else 2)
        p.z_dim, p.c_dim = m.z_dim, m.c_dim
        p.latent_dim = m.z_dim - m.coord
        inv = m.invariances or []
        p.has_r, p.has_t, p.has_s = int('r' in inv), int('t' in inv), int('s' in inv)
        tp = getattr(m, "t_prior", None)
        if tp is not None:
            tpl = tp.detach().cpu().reshape(-1).tolist()
            p.t_prior[0] = tpl[0]
            p.t_prior[1] = tpl[1] if len(tpl) > 1 else tpl[0]
        sp = getattr(m, "sc_prior", None)
        p.sc_prior = float(sp) if sp is not None else 0.0
        p.lik = _abi.LIK[m.sampler_d.name]
        p.sigmoid_out = int(getattr(dec, "sigmoid_out", True))
        p.decoder_sig = m.sampler_d.decoder_sig
        p.ext_decoder = int(self.ext_dec)
        p.fused = int(self.fused)
        if self.ext_enc:
            p.n_enc = 0
            p.discrete_dim = 0
            p.ext_encoder = 1
            p.head.in_dim, p.head.out_dim = 1, 2 * m.z_dim
        elif self.conv_enc:
            p.n_enc = 0
            p.enc_ndim = len(enc.input_dim)
            for i, d in enumerate(enc.input_dim):
                p.enc_in_dim[i] = d
            eops = conv_ops(enc.feature_extractor.layers, enc.feature_extractor.activation,
                            "encoder_z.feature_extractor.layers")
            p.n_enc_ops = fill_ops(p.enc_ops, eops, self._layout)
            self._bn_enc = bn_modules(eops)
            p.head = self._layer("encoder_z.features2latent.fc_latent", enc.features2latent.fc_latent, None)
            p.discrete_dim = 0
        else:
            self._fc_encoder_plan(p, enc, m)
        if not self.ext_dec:
            if p.coord_dim > 0:
                p.fc_coord = self._layer("decoder.coord_latent.fc_coord", dec.coord_latent.fc_coord, "tanh")
                p.fc_latent = self._layer("decoder.coord_latent.fc_latent", dec.coord_latent.fc_latent, None)
            idx = [i for i, mod in enumerate(dec.fc_layers) if isinstance(mod, nn.Linear)]
            p.n_dec = len(idx)
            for j, i in enumerate(idx):
                p.dec[j] = self._layer("decoder.fc_layers.%d" % i, dec.fc_layers[i], dec.activation)
            p.out = self._layer("decoder.out", dec.out, None)
        p.params = self.flat.data_ptr()
        p.grads = self.grad.data_ptr()
        p.adam_m = self.m.data_ptr()
        p.adam_v = self.v.data_ptr()
        p.n_params = self.n_flat
        p.grid = self.grid.data_ptr() if self.grid is not None else None
        p.scalars = self.scalars.data_ptr()
        p.lr, p.adam_beta1, p.adam_beta2, p.adam_eps = self.lr, self.betas[0], self.betas[1], self.adam_eps
        return p

    def _plan(self, batch: int, beta: float = 1.0, what: int = 1) -> _abi.pv_ivae_plan:
        """what: 1 = training step, 2 = encode, 3 = decode (PV_WS_*): the workspace grows to what that call needs."""
        p = self._static
        p.batch = batch
        if isinstance(beta, (list, tuple)) or (torch.is_tensor(beta) and beta.ndim > 0):
            b0, b1 = (float(v) for v in beta)            # jiVAE: [continuous, discrete] KL scale factors
        else:
            b0 = b1 = float(beta)
        p.beta, p.beta_disc = b0, b1
        p.bn_eval = int(not self.model.training)
        p.conv_wide = int(self.wide_weights)
        p.flags = self._plan_flags()
        p.dec_kernel = int(getattr(self, "dec_kernel", 0))     # 0: the library picks the decoder-kernel build by size (ABI v15)
        p.x = p.y = p.eps = p.z_loc = p.z_scale = p.loc = p.alpha = p.ext_head = p.ext_dhead = None
        p.row_w = p.row_elbo = p.dy = None
        p.ext_z = p.ext_dz = p.ext_ll = None
        p.class_onehot = None
        p.ev_start, p.ev_stop = self.events
        ce = getattr(self, "conv_events", None)          # (start, stop, ctypes double for the launch's FLOPs) or None
        p.conv_ev_start, p.conv_ev_stop, p.conv_ev_flops = (ce[0], ce[1], C.addressof(ce[2])) if ce else (None, None, None)
        need = _abi.lib().pv_ivae_workspace_bytes_for(C.byref(p), what)
        if need < 0:
            raise _abi.PvError("pyroved_amd: unsupported plan (pv_ivae_workspace_bytes_for -> %d)" % need)
        if self.ws is None or self.ws.numel() < need:
            self.ws = torch.empty(int(need), device=self.device, dtype=torch.uint8)
        p.ws = self.ws.data_ptr()
        p.ws_bytes = self.ws.numel()
        return p

    def _prep(self, t: Optional[torch.Tensor], what: str, shape=None) -> Optional[torch.Tensor]:
        if t is None:
            return None
        _abi.require_device(t, what)
        t = t.contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            t = t.reshape(shape)
        return t

    # ------------------------------------------------------------------ calls
    @_abi.on_device
    def loss_and_grads(self, x, eps, beta: float = 1.0, y=None, want_grads: bool = True,
                       scalars_out: Optional[torch.Tensor] = None, z_out=None, loc_out=None,
                       row_w: Optional[torch.Tensor] = None, row_elbo: Optional[torch.Tensor] = None,
                       dy: Optional[torch.Tensor] = None, step: bool = False,
                       class_onehot: Optional[torch.Tensor] = None, comm=None, hist_out: Optional[torch.Tensor] = None):
        """Enqueues Trace_ELBO.loss_and_grads on the current stream.  Results land in
        self.scalars (device, 4 floats) and self.grad[:n_flat]; nothing is synchronised.
        row_w (B): per-sample weights of the ELBO terms; row_elbo (B) / dy (B, c_dim): extra outputs
        (include/pyroved_amd.h: pv_ivae_plan.row_w / row_elbo / dy).
        step=True: the whole SVI.step — the Adam update follows in the same library call (pv_ivae_step; on the fused
        decoder path it rides in the last gradient launch), identical in effect to loss_and_grads() + adam_step().
        class_onehot (B, discrete_dim): jiVAE WITHOUT enumeration — the class the guide drew for every sample
        (pv_ivae_plan.class_onehot; the trainer's default enumerate_parallel=False).
        step=True with comm (a dist.NativeComm): the DATA-PARALLEL step as one library call on this stream (pv_ivae_dp_step):
        this rank's shard's loss and gradients -> ncclAllReduce(SUM) of [gradients | loss scalars] -> Adam, with the reduced
        scalars written to hist_out (4 floats) in the optimizer's launch."""
        self.ensure_bound()
        if self.conv_enc:
            self._check_conv_weight_range()
        if step and (self.ext_enc or self.ext_dec or getattr(self, "ext_y", False) or not want_grads):
            raise ValueError("step=True needs every parameter in the library (no user-defined modules) and want_grads")
        if comm is not None and (not step or scalars_out is not None):
            raise ValueError("comm= goes with step=True and the engine's own scalars (no scalars_out)")
        if self.ext_dec:
            return self._loss_and_grads_ext_decoder(x, eps, beta, y, want_grads, scalars_out, z_out, loc_out)
        b = x.shape[0]
        p = self._plan(b, beta)
        head = dhead = None
        if self.ext_enc:
            # the user's encoder in PyTorch on the same stream; its outputs are the step's (z_loc, z_scale)
            with torch.set_grad_enabled(want_grads):
                z_loc, z_scale = self.model.encoder_z(x)
            head = torch.cat([z_loc, z_scale], -1).detach().to(torch.float32).contiguous()
            if tuple(head.shape) != (b, 2 * p.z_dim):
                raise ValueError("encoder_z must return (z_loc, z_scale) of shape (batch, %d) each" % p.z_dim)
            dhead = torch.empty_like(head)
            p.ext_head, p.ext_dhead = head.data_ptr(), dhead.data_ptr()
        x = self._prep(x, "x", (b, p.n_pix))
        eps = self._prep(eps, "eps", (b, p.z_dim))
        y = self._prep(y, "y", (b, p.c_dim)) if p.c_dim > 0 else None
        if p.c_dim > 0 and y is None:
            raise ValueError("class-conditioned model (c_dim=%d) needs y" % p.c_dim)
        p.x, p.eps = x.data_ptr(), eps.data_ptr()
        p.y = y.data_ptr() if y is not None else None
        if z_out is not None:
            p.z_loc, p.z_scale = z_out[0].data_ptr(), z_out[1].data_ptr()
        if loc_out is not None:
            p.loc = loc_out.data_ptr()
        if scalars_out is not None:
            p.scalars = scalars_out.data_ptr()
        row_w = self._prep(row_w, "row_w", (b,))
        for t_, name_, shape_ in ((row_elbo, "row_elbo", (b,)), (dy, "dy", (b, p.c_dim))):
            if t_ is not None:
                _abi.require_device(t_, name_)
                if not t_.is_contiguous() or tuple(t_.shape) != shape_:
                    raise ValueError("%s must be a contiguous float32 tensor of shape %s" % (name_, shape_))
        p.row_w = row_w.data_ptr() if row_w is not None else None
        p.row_elbo = row_elbo.data_ptr() if row_elbo is not None else None
        p.dy = dy.data_ptr() if dy is not None else None
        class_onehot = self._prep(class_onehot, "class_onehot", (b, self.K)) if class_onehot is not None else None
        if class_onehot is not None and self.K <= 0:
            raise ValueError("class_onehot needs a model with a discrete latent (models.jiVAE)")
        p.class_onehot = class_onehot.data_ptr() if class_onehot is not None else None
        try:
            if step:
                p.lr, p.adam_beta1, p.adam_beta2, p.adam_eps = self.lr, self.betas[0], self.betas[1], self.adam_eps
                p.adam_step = self.adam_t + 1
                if comm is not None:
                    if hist_out is not None:
                        _abi.require_device(hist_out, "hist_out")
                    _abi.check(_abi.lib().pv_ivae_dp_step(C.byref(p), comm.handle, _abi.ptr(hist_out), _abi.current_stream()),
                               "pv_ivae_dp_step")
                else:
                    _abi.check(_abi.lib().pv_ivae_step(C.byref(p), _abi.current_stream()), "pv_ivae_step")
                self.adam_t += 1
            else:
                _abi.check(_abi.lib().pv_ivae_loss_and_grads(C.byref(p), int(want_grads), _abi.current_stream()),
                           "pv_ivae_loss_and_grads")
        finally:
            p.scalars = self.scalars.data_ptr()
            p.ext_head = p.ext_dhead = None
            p.row_w = p.row_elbo = p.dy = None
            p.class_onehot = None
        if self.ext_enc and want_grads:
            zd = p.z_dim
            for q in self._enc_params:
                q.grad = None
            torch.autograd.backward([z_loc, z_scale], [dhead[:, :zd], dhead[:, zd:]])
        if want_grads:
            self.grads_live = True
        self._count_bn(self._bn_enc)
        self._keep = (x, eps, y, head, dhead, row_w, class_onehot)   # keep inputs alive until the stream has consumed them

    # ------------------------------------------------------------------ user-defined decoder (torch) around the HIP guide
    def _torch_decode(self, z, y=None, angle=None, shift=None, scale=None):
        """iVAE.model's decoder half in PyTorch (models/ivae.py:184-198): split, scale by the priors, transform the grid,
        decode.  Differentiable w.r.t. z and the decoder's parameters.  angle / shift / scale: the fixed transform of
        baseVAE._decode (models/base.py:153-170) instead of the one read off z (z is then the content part)."""
        m = self.model
        b = z.shape[0]
        if m.coord == 0:
            return m.decoder(z if y is None else torch.cat([z, y], -1))
        grid = m.grid.to(z.device).expand(b, *m.grid.shape)
        if angle is None:
            phi, dx, sc, zc = m._split_latent(z)
            if 't' in m.invariances:
                dx = (dx * m.t_prior.to(z.device)).unsqueeze(1)
        else:
            phi = torch.full((b,), float(angle), device=z.device)
            dx = torch.tensor([float(shift[0]), float(shift[1])][:m.ndim], device=z.device).expand(b, 1, m.ndim)
            sc = torch.full((b,), float(scale), device=z.device)
            zc = z
        if m.ndim == 1:
            xc = grid + dx
        else:
            phi = phi if phi.ndim else phi.expand(b)
            sc = sc if sc.ndim else sc.expand(b)
            rot = torch.stack([torch.stack([torch.cos(phi), torch.sin(phi)], 1),
                               torch.stack([-torch.sin(phi), torch.cos(phi)], 1)], 1)
            xc = torch.bmm(grid, rot) * sc.reshape(b, 1, 1) + dx                # utils/coord.py:47-88
        if y is not None:
            zc = torch.cat([zc, y], -1)
        return m.decoder(xc, zc)

    def _torch_likelihood(self, loc):
        import torch.distributions as td
        s = self.model.sampler_d
        if s.name == "bernoulli":
            return td.Bernoulli(loc, validate_args=False)
        if s.name == "continuous_bernoulli":
            return td.ContinuousBernoulli(loc)
        return td.Normal(loc, s.decoder_sig)

    def _loss_and_grads_ext_decoder(self, x, eps, beta, y, want_grads, scalars_out, z_out, loc_out):
        b = x.shape[0]
        p = self._plan(b, beta)
        x = self._prep(x, "x", (b, p.n_pix))
        eps = self._prep(eps, "eps", (b, p.z_dim))
        y = self._prep(y, "y", (b, p.c_dim)) if p.c_dim > 0 else None
        if p.c_dim > 0 and y is None:
            raise ValueError("class-conditioned model (c_dim=%d) needs y" % p.c_dim)
        z = torch.empty(b, p.z_dim, device=self.device, dtype=torch.float32)
        p.x, p.eps, p.y, p.ext_z = x.data_ptr(), eps.data_ptr(), (y.data_ptr() if y is not None else None), z.data_ptr()
        head = dhead = None
        if self.ext_enc:                             # a user-defined encoder as well: its outputs enter through ext_head
            with torch.set_grad_enabled(want_grads):
                z_loc, z_scale = self.model.encoder_z(x.reshape(b, *self.model.data_dim))
            head = torch.cat([z_loc, z_scale], -1).detach().to(torch.float32).contiguous()
            if tuple(head.shape) != (b, 2 * p.z_dim):
                raise ValueError("encoder_z must return (z_loc, z_scale) of shape (batch, %d) each" % p.z_dim)
            dhead = torch.empty_like(head)
            p.ext_head, p.ext_dhead = head.data_ptr(), dhead.data_ptr()
        if z_out is not None:
            p.z_loc, p.z_scale = z_out[0].data_ptr(), z_out[1].data_ptr()
        if scalars_out is not None:
            p.scalars = scalars_out.data_ptr()
        try:
            _abi.check(_abi.lib().pv_ivae_guide(C.byref(p), _abi.current_stream()), "pv_ivae_guide")
            zt = z.detach().requires_grad_(want_grads)
            with torch.set_grad_enabled(want_grads):
                loc = self._torch_decode(zt, y)
                ll = self._torch_likelihood(loc.reshape(b, -1)).log_prob(x).sum()
            if loc_out is not None:
                loc_out.copy_(loc.detach().reshape(loc_out.shape))
            dz = None
            if want_grads:
                for q in self._enc_params:
                    q.grad = None
                (-ll).backward()
                dz = zt.grad.contiguous()
                p.ext_dz = dz.data_ptr()
            ll1 = ll.detach().reshape(1).to(torch.float32)
            p.ext_ll = ll1.data_ptr()
            _abi.check(_abi.lib().pv_ivae_guide_backward(C.byref(p), int(want_grads), _abi.current_stream()),
                       "pv_ivae_guide_backward")
            if self.ext_enc and want_grads:
                zd = p.z_dim
                torch.autograd.backward([z_loc, z_scale], [dhead[:, :zd], dhead[:, zd:]])
        finally:
            p.scalars = self.scalars.data_ptr()
            p.ext_z = p.ext_dz = p.ext_ll = None
            p.ext_head = p.ext_dhead = None
        if want_grads:
            self.grads_live = True
        self._count_bn(self._bn_enc)
        self._keep = (x, eps, y, z, dz, ll1, head, dhead)

    def _count_bn(self, mods):
        """nn.BatchNorm's num_batches_tracked (a forward in training mode counts; the statistics themselves are updated
        by the kernels)."""
        if self.model.training:
            for b_ in mods:
                b_.num_batches_tracked += 1

    @_abi.on_device
    def adam_step(self):
        """pyro.optim.Adam over every parameter + zero_grads (one fused kernel)."""
        self.adam_t += 1
        _abi.check(_abi.lib().pv_adam_step(
            _abi.ptr(self.flat), _abi.ptr(self.grad), _abi.ptr(self.m), _abi.ptr(self.v), self.n_flat,
            self.lr, self.betas[0], self.betas[1], self.adam_eps, self.adam_t, _abi.current_stream()),
            "pv_adam_step")
        if (self.ext_enc or self.ext_dec or getattr(self, "ext_y", False)) and any(q.grad is not None for q in self._enc_params):
            for g_ in self._enc_opt.param_groups:
                g_["lr"], g_["betas"], g_["eps"] = self.lr, self.betas, self.adam_eps
            self._enc_opt.step()
            for q in self._enc_params:               # pyro.infer.util.zero_grads: zero tensors, not None
                if q.grad is not None:
                    q.grad = torch.zeros_like(q.grad)

    @_abi.on_device
    def adam_step_hist(self, hist_slot: torch.Tensor):
        """adam_step() plus, in the same launch, the copy of the 4 (all-reduced) loss scalars into `hist_slot` — the
        data-parallel step after its one all-reduce (pv_adam_step_hist)."""
        if self.ext_enc or self.ext_dec or getattr(self, "ext_y", False):
            hist_slot.copy_(self.scalars)
            return self.adam_step()
        self.adam_t += 1
        _abi.check(_abi.lib().pv_adam_step_hist(
            _abi.ptr(self.flat), _abi.ptr(self.grad), _abi.ptr(self.m), _abi.ptr(self.v), self.n_flat,
            self.lr, self.betas[0], self.betas[1], self.adam_eps, self.adam_t,
            _abi.ptr(self.scalars), _abi.ptr(hist_slot), N_SCALARS, _abi.current_stream()), "pv_adam_step_hist")

    def extra_grads(self):
        """Gradient tensors that live outside the flat buffer (a user-defined encoder's): reduced separately in
        data-parallel runs."""
        ext = self.ext_enc or self.ext_dec or getattr(self, "ext_y", False)
        return [q.grad for q in self._enc_params if q.grad is not None] if ext else []

    @_abi.on_device
    def encode(self, x, y=None):
        self.ensure_bound()
        if self.ext_enc:
            with torch.no_grad():
                z_loc, z_scale = self.model.encoder_z(x)
            return z_loc.to(torch.float32), z_scale.to(torch.float32)
        if self.conv_enc:
            self._check_conv_weight_range()
        b = x.shape[0]
        p = self._plan(b, what=2)
        x = self._prep(x, "x", (b, p.n_pix))
        y = self._prep(y, "y", (b, p.c_dim)) if p.c_dim > 0 else None
        if p.c_dim > 0 and y is None:
            raise ValueError("class-conditioned model (c_dim=%d) needs y" % p.c_dim)
        p.x = x.data_ptr()
        p.y = y.data_ptr() if y is not None else None
        z_loc = torch.empty(b, p.z_dim, device=self.device, dtype=torch.float32)
        z_scale = torch.empty_like(z_loc)
        alpha = torch.empty(b, self.K, device=self.device, dtype=torch.float32) if self.K > 0 else None
        if alpha is not None:
            p.alpha = alpha.data_ptr()
        _abi.check(_abi.lib().pv_ivae_encode(C.byref(p), _abi.ptr(z_loc), _abi.ptr(z_scale), _abi.current_stream()),
                   "pv_ivae_encode")
        self._count_bn(self._bn_enc)
        self._keep = (x, y)
        if alpha is not None:
            return z_loc, z_scale, alpha
        return z_loc, z_scale

    @_abi.on_device
    def decode(self, z, angle: float = 0.0, shift=(0.0, 0.0), scale: float = 1.0):
        """z: (B, latent_dim + c_dim) content latents [+ class vector]."""
        self.ensure_bound()
        if self.ext_dec:
            with torch.no_grad():
                loc = self._torch_decode(z.to(self.device, torch.float32), None, angle, shift, scale)
            return loc.reshape(z.shape[0], *self.model.data_dim)
        b = z.shape[0]
        p = self._plan(b, what=3)
        lat_in = (p.latent_dim if p.coord_dim > 0 else p.z_dim) + p.c_dim + self.K
        z = self._prep(z, "z", (b, lat_in))
        loc = torch.empty(b, p.n_pix, device=self.device, dtype=torch.float32)
        _abi.check(_abi.lib().pv_ivae_decode(C.byref(p), _abi.ptr(z), float(angle), float(shift[0]), float(shift[1]),
                                             float(scale), _abi.ptr(loc), _abi.current_stream()), "pv_ivae_decode")
        self._keep = (z,)
        return loc.view(b, *self.model.data_dim)

    @_abi.on_device
    def uses_fused(self, batch: int) -> bool:
        """Whether loss_and_grads runs the fused persistent decoder kernel for this batch size."""
        return bool(_abi.lib().pv_ivae_uses_fused(C.byref(self._plan(batch))))

    # views for tests / data-parallel reduction
    def grad_of(self, key: str) -> torch.Tensor:
        n = self._views[key].numel()
        return self.grad[self._layout[key]:self._layout[key] + n].view(self._views[key].shape)
