"""
engine_ved.py — host-side driver of the HIP SVI step for models.VED (conv encoder -> z -> conv decoder).

Same device-memory contract as engine.IVAEEngine (one flat fp32 parameter buffer the nn.Parameters are views of,
flat gradient buffer with 4 trailing ELBO scalars, flat Adam moments, caller-owned workspace), driving
pv_ved_loss_and_grads / pv_ved_encode / pv_ved_decode (include/pyroved_amd.h) with a pv_ved_plan read off the
modules' layer sequences (nets/conv.py: FeatureExtractor.layers, Upsampler.layers).

Replaces what Pyro's SVI / Trace_ELBO hold for VED in the reference (models/ved.py:122-163, trainers/svi.py:109).
"""
import ctypes as C
from typing import Optional

import torch
import torch.nn as nn

from . import _abi
from .engine import IVAEEngine, UnsupportedModel
from ._convplan import conv_ops, fill_ops, bn_modules
from .nets.conv import convEncoderNet, convDecoderNet


class VEDEngine(IVAEEngine):
    """Binds a VED model (encoder_z: convEncoderNet, decoder: convDecoderNet) to the HIP library."""
    supports_step = False            # (pv_ivae_step covers the iVAE / jiVAE plan only)

    # ------------------------------------------------------------------ structure
    def _check_model(self):
        m = self.model
        enc, dec = m.encoder_z, m.decoder
        self.K, self.ext_enc, self.ext_dec, self.conv_enc = 0, False, False, False
        if not isinstance(enc, convEncoderNet) or not isinstance(dec, convDecoderNet):
            raise UnsupportedModel("the HIP VED path needs convEncoderNet / convDecoderNet (got %s / %s)"
                                   % (type(enc).__name__, type(dec).__name__))
        if not enc.softplus_out:
            raise UnsupportedModel("encoder without softplus_out is not supported")
        if len(enc.input_dim) not in (1, 2) or len(dec.output_dim) not in (1, 2):
            raise UnsupportedModel("the HIP conv path covers 1-D and 2-D data")
        name = m.sampler_d.name
        if name not in _abi.LIK:
            raise UnsupportedModel("decoder sampler %r is not implemented in the HIP path yet" % name)
        if name in ("bernoulli", "continuous_bernoulli") and not dec.sigmoid_out:
            raise UnsupportedModel("%s likelihood needs sigmoid_d=True" % name)
        self._ops(enc.feature_extractor.layers, enc.feature_extractor.activation)      # validates
        self._ops(dec.upsampler.layers, dec.upsampler.activation)

    def _param_order(self):
        return list(self.model.named_parameters())          # state_dict order, nothing merged

    def _ops(self, layers, activation, prefix=None):
        return conv_ops(layers, activation, prefix)

    def _fill_ops(self, arr, ops):
        return fill_ops(arr, ops, self._layout)

    def _static_plan(self):
        m = self.model
        enc, dec = m.encoder_z, m.decoder
        p = _abi.pv_ved_plan()
        p.ndim_in, p.ndim_out = len(enc.input_dim), len(dec.output_dim)
        for i, d in enumerate(enc.input_dim):
            p.in_dim[i] = d
        for i, d in enumerate(dec.output_dim):
            p.out_dim[i] = d
        p.in_ch, p.out_ch, p.z_dim = enc.input_channels, dec.output_channels, m.z_dim
        p.lik = _abi.LIK[m.sampler_d.name]
        p.sigmoid_out = int(dec.sigmoid_out)
        p.decoder_sig = m.sampler_d.decoder_sig
        eops = self._ops(enc.feature_extractor.layers, enc.feature_extractor.activation, "encoder_z.feature_extractor.layers")
        dops = self._ops(dec.upsampler.layers, dec.upsampler.activation, "decoder.upsampler.layers")
        p.n_enc_ops = self._fill_ops(p.enc, eops)
        p.n_dec_ops = self._fill_ops(p.dec, dops)
        self._bn_enc, self._bn_dec = bn_modules(eops), bn_modules(dops)
        p.head = self._layer("encoder_z.features2latent.fc_latent", enc.features2latent.fc_latent, None)
        p.l2f = self._layer("decoder.latent2features.fc", dec.latent2features.fc, None)
        shape0 = [int(v) for v in dec.latent2features.reshape_]
        p.dec_c0 = shape0[0]
        for i, d in enumerate(shape0[1:]):
            p.dec_dim0[i] = d
        p.params, p.grads = self.flat.data_ptr(), self.grad.data_ptr()
        p.adam_m, p.adam_v = self.m.data_ptr(), self.v.data_ptr()
        p.n_params = self.n_flat
        p.scalars = self.scalars.data_ptr()
        return p

    def _plan(self, batch: int, beta: float = 1.0):
        p = self._static
        p.batch = batch
        p.beta = float(beta)
        p.bn_eval = int(not self.model.training)
        # SVItrainer(precision="bf16"): the throughput precision (one fp16 piece per operand of the 2-D kernel-3 convolutions, one
        # matrix product); otherwise fp32-class.  Once a weight left fp16's range (engine._check_conv_weight_range) the throughput
        # precision falls back to the range-free two-piece bf16 "mixed" kernels and fp32-class to the three-piece bf16 form
        # (round 5: fp32-class = "f16w2" — weights two fp16 pieces, activations one — unless conv_x3 asks for both operands split)
        p.conv_bf16 = ((1 if self.wide_weights else 3) if self.fused == 3 else
                       (2 if self.wide_weights else (0 if self.conv_x3 else 4)))
        p.flags = self._plan_flags()
        p.x = p.y = p.eps = p.z_loc = p.z_scale = p.loc = None
        ce = getattr(self, "conv_events", None)          # (start, stop, ctypes double for the launch's FLOPs) or None
        p.conv_ev_start, p.conv_ev_stop, p.conv_ev_flops = (ce[0], ce[1], C.addressof(ce[2])) if ce else (None, None, None)
        need = _abi.lib().pv_ved_workspace_bytes(C.byref(p))
        if need < 0:
            raise _abi.PvError("pyroved_amd: unsupported VED plan (pv_ved_workspace_bytes -> %d)" % need)
        if self.ws is None or self.ws.numel() < need:
            self.ws = torch.empty(int(need), device=self.device, dtype=torch.uint8)
        p.ws, p.ws_bytes = self.ws.data_ptr(), self.ws.numel()
        return p

    def _in_shape(self, b):
        e = self.model.encoder_z
        return (b, e.input_channels) + tuple(e.input_dim)

    def _out_shape(self, b):
        d = self.model.decoder
        return (b, d.output_channels) + tuple(d.output_dim)

    # ------------------------------------------------------------------ calls
    @_abi.on_device
    def loss_and_grads(self, x, eps, beta: float = 1.0, y=None, want_grads: bool = True,
                       scalars_out: Optional[torch.Tensor] = None, z_out=None, loc_out=None, step: bool = False, comm=None,
                       hist_out: Optional[torch.Tensor] = None):
        """Enqueues Trace_ELBO.loss_and_grads(VED.model, VED.guide) on the current stream: x (B, C, *input_dim),
        target y (B, C', *output_dim), eps (B, z_dim).
        step=True with comm (a dist.NativeComm): the data-parallel SVI.step as one library call (pv_ved_dp_step): this shard's
        gradients -> ncclAllReduce(SUM) of [gradients | loss scalars] -> Adam, the reduced scalars into hist_out."""
        self.ensure_bound()
        self._check_conv_weight_range()
        if y is None:
            raise ValueError("VED needs the target y")
        b = x.shape[0]
        p = self._plan(b, beta)
        x = self._prep(x, "x", self._in_shape(b))
        y = self._prep(y, "y", self._out_shape(b))
        eps = self._prep(eps, "eps", (b, p.z_dim))
        p.x, p.y, p.eps = x.data_ptr(), y.data_ptr(), eps.data_ptr()
        if z_out is not None:
            p.z_loc, p.z_scale = z_out[0].data_ptr(), z_out[1].data_ptr()
        if loc_out is not None:
            p.loc = loc_out.data_ptr()
        if scalars_out is not None:
            p.scalars = scalars_out.data_ptr()
        if step and (comm is None or not want_grads or scalars_out is not None):
            raise ValueError("step=True is the data-parallel step: it needs comm=, want_grads and the engine's own scalars")
        try:
            if step:
                if hist_out is not None:
                    _abi.require_device(hist_out, "hist_out")
                _abi.check(_abi.lib().pv_ved_dp_step(C.byref(p), comm.handle, self.lr, self.betas[0], self.betas[1],
                                                     self.adam_eps, self.adam_t + 1, _abi.ptr(hist_out),
                                                     _abi.current_stream()), "pv_ved_dp_step")
                self.adam_t += 1
            else:
                _abi.check(_abi.lib().pv_ved_loss_and_grads(C.byref(p), int(want_grads), _abi.current_stream()),
                           "pv_ved_loss_and_grads")
        finally:
            p.scalars = self.scalars.data_ptr()
        if want_grads:
            self.grads_live = True
        self._count_bn(self._bn_enc + self._bn_dec)
        self._keep = (x, y, eps)

    @_abi.on_device
    def encode(self, x, y=None):
        self.ensure_bound()
        self._check_conv_weight_range()
        b = x.shape[0]
        p = self._plan(b)
        x = self._prep(x, "x", self._in_shape(b))
        p.x = x.data_ptr()
        z_loc = torch.empty(b, p.z_dim, device=self.device, dtype=torch.float32)
        z_scale = torch.empty_like(z_loc)
        _abi.check(_abi.lib().pv_ved_encode(C.byref(p), _abi.ptr(z_loc), _abi.ptr(z_scale), _abi.current_stream()),
                   "pv_ved_encode")
        self._count_bn(self._bn_enc)
        self._keep = (x,)
        return z_loc, z_scale

    @_abi.on_device
    def decode(self, z, *unused, **unused_kw):
        self.ensure_bound()
        self._check_conv_weight_range()
        b = z.shape[0]
        p = self._plan(b)
        z = self._prep(z, "z", (b, p.z_dim))
        loc = torch.empty(self._out_shape(b), device=self.device, dtype=torch.float32)
        _abi.check(_abi.lib().pv_ved_decode(C.byref(p), _abi.ptr(z), _abi.ptr(loc), _abi.current_stream()),
                   "pv_ved_decode")
        self._count_bn(self._bn_dec)
        self._keep = (z,)
        return loc

    def uses_fused(self, batch: int) -> bool:
        return False


def _owner_engine(net):
    eng = getattr(net, "_pv_engine", None)
    if eng is None:
        raise _abi.PvError("pyroved_amd: conv nets run through the model's HIP engine (models.VED); a stand-alone "
                           "convEncoderNet / convDecoderNet has no forward of its own in this build")
    return eng


def conv_encoder_forward(net, x):
    return _owner_engine(net).encode(x)


def conv_decoder_forward(net, z):
    return _owner_engine(net).decode(z)
