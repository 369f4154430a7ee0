"""
_abi.py — ctypes binding of libpyroved_amd.so (the C ABI declared in include/pyroved_amd.h).

The library is the product's only compute path.  There is no CPU or pure-PyTorch
fallback: `lib()` raises if the shared object is missing or does not export every
symbol of the header, and every wrapper raises on a non-zero return code.
"""
import contextlib
import ctypes as C
import functools
import os
import threading

import torch  # noqa: F401  (must be imported first: it loads the HIP runtime libamdhip64.so.7 the library binds to)

PV_ABI_VERSION = 16
# pv_ivae_plan.flags / pv_ved_plan.flags / pv_convnet_plan.flags
PV_PLAN_ENC_TWO_LAUNCH, PV_PLAN_NO_SIDE_STREAM, PV_PLAN_ENC_NO_WAIT, PV_PLAN_NO_DEC1D, PV_PLAN_NO_ENC_FOLD = 1, 2, 4, 8, 16
PV_PLAN_CONV_X3 = 64
PV_PLAN_ENC_TILED = 32
PV_MAX_LAYERS = 8

# enum pv_act / pv_lik (include/pyroved_amd.h)
ACT = {None: 0, "none": 0, "tanh": 1, "relu": 2, "lrelu": 3, "softplus": 4, "gelu": 5, "sigmoid": 6}
LIK = {"bernoulli": 0, "gaussian": 1, "continuous_bernoulli": 2}

_HERE = os.path.dirname(os.path.abspath(__file__))
# PV_LIB_PATH: the experiments build (csrc/Makefile `experiments`: libpyroved_amd_exp.so) or a profiling build; the shipped
# library itself reads no environment variable besides PV_ROCTX
LIB_PATH = os.environ.get("PV_LIB_PATH") or os.path.join(_HERE, "libpyroved_amd.so")
EXP_LIB_PATH = os.path.join(_HERE, "libpyroved_amd_exp.so")

c_float_p = C.POINTER(C.c_float)


class pv_layer(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("out_dim", C.c_int32), ("act", C.c_int32), ("_pad", C.c_int32),
                ("w_off", C.c_int64), ("b_off", C.c_int64)]


PV_MAX_OPS = 32
OP = {"conv": 1, "maxpool2": 2, "upsample2": 3, "upsample2_bilinear": 4, "batchnorm": 5}


class pv_op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("ksize", C.c_int32),
                ("act", C.c_int32), ("_pad", C.c_int32), ("w_off", C.c_int64), ("b_off", C.c_int64),
                ("aux0_off", C.c_int64), ("aux1_off", C.c_int64)]


class pv_ivae_plan(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("n_pix", C.c_int32), ("coord_dim", C.c_int32), ("z_dim", C.c_int32),
        ("latent_dim", C.c_int32), ("c_dim", C.c_int32),
        ("has_r", C.c_int32), ("has_t", C.c_int32), ("has_s", C.c_int32),
        ("t_prior", C.c_float * 2), ("sc_prior", C.c_float), ("beta", C.c_float),
        ("lik", C.c_int32), ("sigmoid_out", C.c_int32), ("decoder_sig", C.c_float), ("fused", C.c_int32),
        ("discrete_dim", C.c_int32), ("beta_disc", C.c_float),
        ("n_enc", C.c_int32), ("n_dec", C.c_int32),
        ("enc", pv_layer * PV_MAX_LAYERS), ("head", pv_layer),
        ("fc_coord", pv_layer), ("fc_latent", pv_layer),
        ("dec", pv_layer * PV_MAX_LAYERS), ("out", pv_layer),
        ("n_enc_ops", C.c_int32), ("enc_ndim", C.c_int32), ("enc_in_dim", C.c_int32 * 2),
        ("enc_ops", pv_op * PV_MAX_OPS),
        ("params", C.c_void_p), ("grads", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p),
        ("n_params", C.c_int64),
        ("x", C.c_void_p), ("y", C.c_void_p), ("eps", C.c_void_p), ("grid", C.c_void_p),
        ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
        ("scalars", C.c_void_p), ("z_loc", C.c_void_p), ("z_scale", C.c_void_p), ("loc", C.c_void_p),
        ("alpha", C.c_void_p), ("ext_head", C.c_void_p), ("ext_dhead", C.c_void_p),
        ("ext_encoder", C.c_int32), ("bn_eval", C.c_int32),
        ("row_w", C.c_void_p), ("row_elbo", C.c_void_p), ("dy", C.c_void_p),
        ("ext_z", C.c_void_p), ("ext_dz", C.c_void_p), ("ext_ll", C.c_void_p),
        ("ext_decoder", C.c_int32), ("conv_wide", C.c_int32),
        ("lr", C.c_float), ("adam_beta1", C.c_float), ("adam_beta2", C.c_float), ("adam_eps", C.c_float),
        ("adam_step", C.c_int32), ("flags", C.c_int32),
        ("ev_start", C.c_void_p), ("ev_stop", C.c_void_p),
        ("class_onehot", C.c_void_p),
        ("conv_ev_start", C.c_void_p), ("conv_ev_stop", C.c_void_p), ("conv_ev_flops", C.c_void_p),
        ("dec_kernel", C.c_int32), ("reserved0", C.c_int32),
    ]


class pv_ved_plan(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("ndim_in", C.c_int32), ("ndim_out", C.c_int32),
        ("in_dim", C.c_int32 * 2), ("out_dim", C.c_int32 * 2),
        ("in_ch", C.c_int32), ("out_ch", C.c_int32), ("z_dim", C.c_int32),
        ("beta", C.c_float), ("lik", C.c_int32), ("sigmoid_out", C.c_int32), ("decoder_sig", C.c_float),
        ("n_enc_ops", C.c_int32), ("n_dec_ops", C.c_int32),
        ("enc", pv_op * PV_MAX_OPS), ("dec", pv_op * PV_MAX_OPS),
        ("head", pv_layer), ("l2f", pv_layer),
        ("dec_c0", C.c_int32), ("dec_dim0", C.c_int32 * 2), ("bn_eval", C.c_int32),
        ("conv_bf16", C.c_int32), ("flags", C.c_int32),
        ("params", C.c_void_p), ("grads", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p),
        ("n_params", C.c_int64),
        ("x", C.c_void_p), ("y", C.c_void_p), ("eps", C.c_void_p),
        ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
        ("scalars", C.c_void_p), ("z_loc", C.c_void_p), ("z_scale", C.c_void_p), ("loc", C.c_void_p),
        ("conv_ev_start", C.c_void_p), ("conv_ev_stop", C.c_void_p), ("conv_ev_flops", C.c_void_p),
    ]


class pv_convnet_plan(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("ndim", C.c_int32), ("in_ch", C.c_int32), ("in_dim", C.c_int32 * 2),
        ("n_ops", C.c_int32), ("bn_eval", C.c_int32), ("conv_bf16", C.c_int32), ("need_dx", C.c_int32),
        ("flags", C.c_int32),
        ("ops", pv_op * PV_MAX_OPS),
        ("params", C.c_void_p), ("grads", C.c_void_p),
        ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
    ]


MLP_OUT = {"linear": 0, "softmax": 1}
SS_TASK = {"classification": 0, "regression": 1}


class pv_mlp_plan(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("in_dim", C.c_int32), ("n_layers", C.c_int32), ("out_kind", C.c_int32),
        ("layers", pv_layer * PV_MAX_LAYERS), ("out", pv_layer),
        ("params", C.c_void_p), ("grads", C.c_void_p), ("x", C.c_void_p),
        ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
    ]


# name -> (restype, argtypes); must list every function include/pyroved_amd.h declares
SIGNATURES = {
    "pv_version": (C.c_int, []),
    "pv_experiments_build": (C.c_int, []),
    "pv_ivae_workspace_bytes": (C.c_int64, [C.POINTER(pv_ivae_plan)]),
    "pv_ivae_workspace_bytes_for": (C.c_int64, [C.POINTER(pv_ivae_plan), C.c_int]),
    "pv_ved_workspace_bytes": (C.c_int64, [C.POINTER(pv_ved_plan)]),
    "pv_ved_loss_and_grads": (C.c_int, [C.POINTER(pv_ved_plan), C.c_int, C.c_void_p]),
    "pv_ved_encode": (C.c_int, [C.POINTER(pv_ved_plan), C.c_void_p, C.c_void_p, C.c_void_p]),
    "pv_ved_decode": (C.c_int, [C.POINTER(pv_ved_plan), C.c_void_p, C.c_void_p, C.c_void_p]),
    "pv_convnet_workspace_bytes": (C.c_int64, [C.POINTER(pv_convnet_plan)]),
    "pv_convnet_out_shape": (C.c_int, [C.POINTER(pv_convnet_plan), C.POINTER(C.c_int32)]),
    "pv_convnet_forward": (C.c_int, [C.POINTER(pv_convnet_plan), C.c_void_p, C.c_void_p, C.c_void_p]),
    "pv_convnet_backward": (C.c_int, [C.POINTER(pv_convnet_plan), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pv_ivae_uses_fused": (C.c_int, [C.POINTER(pv_ivae_plan)]),
    "pv_ivae_guide_folds": (C.c_int, [C.POINTER(pv_ivae_plan)]),
    "pv_ivae_loss_and_grads": (C.c_int, [C.POINTER(pv_ivae_plan), C.c_int, C.c_void_p]),
    "pv_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p]),
    "pv_adam_step_hist": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "pv_ivae_step": (C.c_int, [C.POINTER(pv_ivae_plan), C.c_void_p]),
    "pv_dist_load": (C.c_int, [C.c_char_p]),
    "pv_dist_library": (C.c_char_p, []),
    "pv_dist_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pv_dist_allreduce_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "pv_ivae_dp_step": (C.c_int, [C.POINTER(pv_ivae_plan), C.c_void_p, C.c_void_p, C.c_void_p]),
    "pv_ved_dp_step": (C.c_int, [C.POINTER(pv_ved_plan), C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_int32, C.c_void_p, C.c_void_p]),
    "pv_ivae_guide": (C.c_int, [C.POINTER(pv_ivae_plan), C.c_void_p]),
    "pv_ivae_guide_backward": (C.c_int, [C.POINTER(pv_ivae_plan), C.c_int, C.c_void_p]),
    "pv_ivae_encode": (C.c_int, [C.POINTER(pv_ivae_plan), C.c_void_p, C.c_void_p, C.c_void_p]),
    "pv_ivae_decode": (C.c_int, [C.POINTER(pv_ivae_plan), C.c_void_p, C.c_float, C.c_float, C.c_float,
                                 C.c_float, C.c_void_p, C.c_void_p]),
    "pv_linear_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int64, C.c_int64]),
    "pv_linear_fwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                C.c_void_p]),
    "pv_linear_bwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "pv_mlp_workspace_bytes": (C.c_int64, [C.POINTER(pv_mlp_plan)]),
    "pv_mlp_forward": (C.c_int, [C.POINTER(pv_mlp_plan), C.c_void_p, C.c_void_p]),
    "pv_mlp_backward": (C.c_int, [C.POINTER(pv_mlp_plan), C.c_void_p, C.c_void_p, C.c_void_p]),
    "pv_ss_enum_terms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pv_ss_aux_loss": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_float,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "pv_ss_reg_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "pv_ss_reg_terms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "pv_transform_coordinates": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int64, C.c_void_p, C.c_void_p]),
}

_lib = None
_lock = threading.Lock()


class PvError(RuntimeError):
    pass


def lib():
    """Loads libpyroved_amd.so once.  Raises if it is missing or incomplete — never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise PvError(
                "pyroved_amd: HIP library %s not found. Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C pyroved_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise PvError("pyroved_amd: %s does not export %s" % (LIB_PATH, name)) from e
            fn.restype = res
            fn.argtypes = args
        v = handle.pv_version()
        if v != PV_ABI_VERSION:
            raise PvError("pyroved_amd: ABI version mismatch (library %d, binding %d)" % (v, PV_ABI_VERSION))
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        kind = "hipError_t" if code > 0 else {-1: "PV_EINVAL", -2: "PV_EWS", -3: "PV_ECOLL"}.get(code, "PV error")
        raise PvError("pyroved_amd: %s failed with %s (%d)" % (what, kind, code))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def require_device(t, what):
    if not t.is_cuda:
        raise PvError("pyroved_amd: %s must live on a HIP device (got %s); the compute path is GPU-only "
                      "and has no CPU fallback" % (what, t.device))
    if t.dtype != torch.float32:
        raise PvError("pyroved_amd: %s must be float32 (got %s)" % (what, t.dtype))


def current_stream():
    """The current HIP stream of the CURRENT device: library calls are made under `on_device` / `device_of`, which make
    the tensors' device current first (kernels launched on device A's stream against device B's pointers fault)."""
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)      # (the handle without a torch.cuda.Stream object around it)
    if raw is not None:
        return C.c_void_p(raw(torch._C._cuda_getDevice()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def device_of(dev):
    """Context manager: `dev` (torch.device / tensor device) is the current HIP device inside; free when it already is."""
    dev = torch.device(dev) if not isinstance(dev, torch.device) else dev
    if dev.type != "cuda" or dev.index is None or torch.cuda.current_device() == dev.index:
        return contextlib.nullcontext()
    return torch.cuda.device(dev)


def on_device(method):
    """Decorator for engine methods that enqueue library work: runs them with `self.device` current, so a model built
    with device='cuda:1' works whatever the caller's current device is (the reference accepts any device string,
    models/base.py:51-52)."""
    @functools.wraps(method)
    def call(self, *args, **kwargs):
        dev = getattr(self, "device", None)
        if dev is None:
            return method(self, *args, **kwargs)
        with device_of(dev):
            return method(self, *args, **kwargs)
    return call
