"""
base.py — variational encoder-decoder base class; host-side mirror of
pyroved/models/base.py:21-192 (same constructor logic, attributes, state_dict keys,
`_split_latent`, `_encode`, `_decode`, `set_encoder/decoder`, `save/load_weights`).

The batched `_encode` / `_decode` run the encoder / decoder through the HIP library
(pv_ivae_encode / pv_ivae_decode) instead of eager torch modules.
"""
from typing import Tuple, Type, Union, List
from abc import abstractmethod

import torch
import torch.nn as nn

from ..utils import iter_batches, generate_grid

tt = torch.tensor


def _to_host(t: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """Device -> host copy of a result tensor into `out` (an ordinary pageable CPU tensor of t's shape; allocated when None).
    Large results go through a page-locked STAGING buffer (torch's caching host allocator keeps the block for the next chunk
    or call), 2-3x the rate of a pageable `.cpu()`; what is returned is never pinned."""
    if out is None:
        out = torch.empty(t.shape, dtype=t.dtype, device="cpu")
    if not t.is_cuda or t.numel() * t.element_size() < (1 << 22):
        out.copy_(t)
        return out
    try:
        stage = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
    except RuntimeError:
        out.copy_(t)
        return out
    stage.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    out.copy_(stage)
    return out


_FLUSH_BYTES = 1 << 28     # results kept on the device between two device -> host copies of _encode / _decode (256 MB)


def _gather_batches(batches, total_rows: int = None) -> torch.Tensor:
    """cat() of per-batch device results on the host with BOUNDED residency on both sides: the batches are kept on the device
    and copied out in chunks of ~_FLUSH_BYTES (one synchronising copy per chunk, not per batch as the reference's loop pays,
    and not one for the whole result either — 1e6 decoded 64x64 images are 16 GB: ADVICE r3), each chunk through ONE page-locked
    staging block into its slice of the pageable result, which is allocated once when `total_rows` is known (no
    second copy of the result, nothing pinned is returned)."""
    result, filled, parts = None, 0, []

    def flush(pend):
        nonlocal result, filled
        t = torch.cat(pend) if len(pend) > 1 else pend[0]
        if total_rows is not None and result is None:
            result = torch.empty((total_rows,) + tuple(t.shape[1:]), dtype=t.dtype, device="cpu")
        if result is not None and filled + t.shape[0] <= result.shape[0]:
            _to_host(t, result[filled:filled + t.shape[0]])
            filled += t.shape[0]
        else:                                         # (row count unknown or exceeded: fall back to parts + cat)
            parts.append(_to_host(t))

    pend, nbytes = [], 0
    for t in batches:
        pend.append(t)
        nbytes += t.numel() * t.element_size()
        if nbytes >= _FLUSH_BYTES:
            flush(pend)
            pend, nbytes = [], 0
    if pend:
        flush(pend)
    if result is not None:
        head = result if filled == result.shape[0] else result[:filled]
        return head if not parts else torch.cat([head] + parts)
    if not parts:
        return torch.empty(0)
    return parts[0] if len(parts) == 1 else torch.cat(parts)


class baseVAE(nn.Module):
    """Base class for regular and invariant variational encoder-decoder models.

    Args:
        data_dim: (height, width) for images or (length,) for spectra.
        invariances: list with invariances to enforce: 'r' (rotation), 't'
            (translation), 's' (scale) for 2D; 't' for 1D; None = vanilla VAE.

    Keyword Args:
        device: defaults to 'cuda' if a GPU is available, else 'cpu' (on which the
            model can be built and inspected, but not run: the compute path is HIP only).
        dx_prior, dy_prior: translational priors; sc_prior: scale prior.
    """
    def __init__(self, *args, **kwargs: str):
        super(baseVAE, self).__init__()
        data_dim, invariances = args
        self.device = kwargs.get(
            "device", 'cuda' if torch.cuda.is_available() else 'cpu')
        self.data_dim = tuple(int(d) for d in data_dim)
        self.ndim = len(data_dim)
        # latent coordinates spent on the enforced invariances (pyroved/models/base.py:56-67): one per symmetry, two
        # for a 2-D translation; 1-D data only knows 't'
        inv = list(invariances) if invariances is not None else []
        if self.ndim == 1 and inv and inv != ['t']:
            raise ValueError("For 1D data, the only invariance to enforce is translation ('t')")
        self.coord = len(inv) + (1 if ('t' in inv and self.ndim == 2) else 0)
        self.invariances = invariances
        if self.coord > 0:
            self.grid = generate_grid(data_dim).to(self.device)
        # prior widths of the translational / scale disorder (base.py:73-80), defaults 0.1
        if 't' in inv:
            dx_pri = tt(kwargs.get("dx_prior", 0.1))
            dy_pri = kwargs.get("dy_prior", dx_pri.clone())
            self.t_prior = (tt([dx_pri, dy_pri]) if self.ndim == 2 else dx_pri).to(self.device)
        if 's' in inv:
            self.sc_prior = tt(kwargs.get("sc_prior", 0.1)).to(self.device)
        self.encoder_z = None
        self.decoder = None
        self._engine = None

    @abstractmethod
    def model(self, *args, **kwargs):
        raise NotImplementedError

    @abstractmethod
    def guide(self, *args, **kwargs):
        raise NotImplementedError

    def _split_latent(self, z: torch.Tensor) -> Tuple[torch.Tensor]:
        """Splits a latent vector into the parts associated with coordinate transformations
        and image content: always [phi | dx, dy | scale | content] (base.py:97-119).
        Pure slicing of the caller's tensor (any device)."""
        if self.ndim == 1:
            return None, z[:, 0:1], None, z[:, 1:]
        inv = self.invariances or []
        dev, k = z.device, 0
        phi, dx, sc = tt(0).to(dev), tt(0).to(dev), tt(1).to(dev)      # the "no transform" scalars of base.py:104-106
        if 'r' in inv:
            phi, k = z[:, k], k + 1
        if 't' in inv:
            dx, k = z[:, k:k + 2], k + 2
        if 's' in inv:
            sc, k = sc + self.sc_prior.to(dev) * z[:, k], k + 1
        return phi, dx, sc, z[:, k:]

    # ------------------------------------------------------------------ HIP engine
    def _engine_ready(self) -> bool:
        """Whether the model is a complete VAE (encoder, decoder, likelihood) the plan-based engine can drive."""
        return self.encoder_z is not None and self.decoder is not None and hasattr(self, "sampler_d")

    def engine(self, **kw):
        """The HIP driver bound to this model's parameters (created on first use)."""
        from ..engine import IVAEEngine
        if self._engine is None:
            self._engine = IVAEEngine(self, **kw)
        elif kw:
            self._engine.configure(**kw)      # an engine made earlier (encode, a previous trainer) takes the new settings
        return self._engine

    def _encode(self, *input_args, device: str = None, **kwargs: int) -> torch.Tensor:
        """Encodes data batch-by-batch with the trained encoder (base.py:121-143);
        returns cat([z_loc, z_scale], -1) on the CPU."""
        loader = iter_batches(*input_args, batch_size=kwargs.get("batch_size", 100))
        z_encoded = []
        if not self._engine_ready():
            # a bare baseVAE with only set_encoder() called (the reference's own tests use it so): the network's
            # operator-level forward (nets.*: HIP Linear(+activation) kernels), as base.py:132-136 calls it
            dev = torch.device(self.device if device is None else device)
            for data in loader:
                data = [d.to(dev, torch.float32) for d in data]
                with torch.no_grad():
                    encoded = self.encoder_z(data if len(data) > 1 else data[0])
                z_encoded.append(torch.cat(encoded, -1).cpu())
            return torch.cat(z_encoded)
        eng = self.engine()
        # results stay on the device between copies: one device -> host copy per ~256 MB instead of a synchronising .cpu()
        # per batch (the reference's loop, base.py:137-142, pays one per batch)
        def run():
            for data in loader:
                x = data[0].to(eng.device, torch.float32, non_blocking=True)
                y = data[1].to(eng.device, torch.float32, non_blocking=True) if len(data) > 1 else None
                yield torch.cat(eng.encode(x, y), -1)                    # (z_loc, z_scale[, class probabilities])
        return _gather_batches(run(), total_rows=int(input_args[0].shape[0]))

    def _decode(self, z_new: torch.Tensor, device: str = None, **kwargs: int) -> torch.Tensor:
        """Decodes latent coordinates batch-by-batch (base.py:145-171).  kwargs: batch_size,
        and for invariant models angle / shift / scale of the coordinate grid."""
        loader = iter_batches(z_new, batch_size=kwargs.get("batch_size", 100))
        if not self._engine_ready():
            # bare baseVAE with only set_decoder() called: transform the grid once, then the decoder's own forward
            from ..utils import transform_coordinates
            dev = torch.device(self.device if device is None else device)
            grid = None
            if self.invariances:
                g0 = self.grid.to(dev, torch.float32)
                a = torch.as_tensor(kwargs.get("angle", 0.), dtype=torch.float32).reshape(1).to(dev)
                t = torch.as_tensor(kwargs.get("shift", 0.), dtype=torch.float32).reshape(-1).to(dev)
                t = (t if t.numel() == g0.shape[-1] else t[:1].expand(g0.shape[-1])).reshape(1, 1, -1)
                sc = torch.as_tensor(kwargs.get("scale", 1.), dtype=torch.float32).reshape(1).to(dev)
                grid = transform_coordinates(g0.unsqueeze(0), a, t, sc).squeeze(0)
            x_decoded = []
            for (z,) in loader:
                z = z.to(dev, torch.float32)
                with torch.no_grad():
                    loc = self.decoder(grid.expand(z.shape[0], *grid.shape), z) if grid is not None else self.decoder(z)
                x_decoded.append(loc.cpu())
            return torch.cat(x_decoded)
        eng = self.engine()
        angle, shift, scale = 0.0, (0.0, 0.0), 1.0
        if self.invariances:
            angle = float(kwargs.get("angle", 0.0))
            t = kwargs.get("shift", 0.0)
            t = torch.as_tensor(t, dtype=torch.float32).reshape(-1).tolist()
            shift = (t[0], t[1] if len(t) > 1 else t[0])
            scale = float(kwargs.get("scale", 1.0))
        # decoded batches stay on the device between copies (see _encode)
        return _gather_batches((eng.decode(z.to(eng.device, torch.float32, non_blocking=True), angle, shift, scale)
                                for (z,) in loader), total_rows=int(z_new.shape[0]))

    def set_encoder(self, encoder_net: Type[torch.nn.Module]) -> None:
        """Sets a user-defined encoder neural network."""
        self.encoder_z = encoder_net.to(self.device)
        self._engine = None

    def set_decoder(self, decoder_net: Type[torch.nn.Module]) -> None:
        """Sets a user-defined decoder neural network."""
        self.decoder = decoder_net.to(self.device)
        self._engine = None

    def save_weights(self, filepath: str) -> None:
        """Saves trained weights of encoder(s) and decoder (same keys as the reference)."""
        torch.save(self.state_dict(), filepath + '.pt')

    def load_weights(self, filepath: str) -> None:
        """Loads saved weights of encoder(s) and decoder."""
        weights = torch.load(filepath, map_location=self.device)
        self.load_state_dict(weights)
