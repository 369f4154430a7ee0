"""
Coordinate grids and the rotate -> scale -> translate transform.

Mirrors pyroved/utils/coord.py:7-88.  `generate_grid` is construction-time host
logic (a constant of the model); `transform_coordinates` runs on the GPU through
the C ABI (pv_transform_coordinates) — inside training it is fused into the
spatial decoder's first layer and never materialised.
"""
from typing import Tuple, Union

import torch

from .. import _abi


def generate_grid(data_dim: Tuple[int]) -> torch.Tensor:
    """(N, 2) grid for images — row n = i*W + j holds (xx[i], yy[j]) with
    xx = linspace(-1, 1, H), yy = linspace(1, -1, W) — or (L, 1) for 1-D data
    (pyroved/utils/coord.py:7-44)."""
    if len(data_dim) not in [1, 2]:
        raise NotImplementedError("Currently supports only 1D and 2D data")
    if len(data_dim) == 1:
        return torch.linspace(1, -1, data_dim[0])[:, None]
    xx = torch.linspace(-1, 1, data_dim[0])
    yy = torch.linspace(1, -1, data_dim[1])
    x0, x1 = torch.meshgrid(xx, yy, indexing="ij")
    return torch.stack((x0.reshape(-1), x1.reshape(-1)), 1).contiguous()


def _per_sample(v, b, n, width, dev):
    """Broadcasts a reference-style argument (python number, 0-d, (B,), (B,1,w), (1,w) ...)
    to a contiguous (B, width) device tensor."""
    v = torch.as_tensor(v, dtype=torch.float32).to(dev)
    if width == 1 and v.dim() <= 1:           # phi / scale: one number per sample
        return v.reshape(-1).expand(b).contiguous()
    if v.dim() == 3 and v.shape[1] != 1:
        raise NotImplementedError("per-point shifts are not supported")
    return torch.broadcast_to(v, (b, n, width))[:, 0, :].contiguous()


def transform_coordinates(coord: torch.Tensor,
                          phi: Union[torch.Tensor, float] = 0,
                          coord_dx: Union[torch.Tensor, float] = 0,
                          scale: Union[torch.Tensor, float] = 1.,
                          ) -> torch.Tensor:
    """Rotation of 2D coordinates followed by scaling and translation; 1D grids are
    only translated.  Operates on batches: coord is (B, N, 2) or (B, N, 1), one grid
    expanded over the batch as the reference's callers pass it
    (pyroved/utils/coord.py:47-60; callers models/ivae.py:191-192, models/base.py:160-162).
    GPU only."""
    _abi.require_device(coord, "coord")
    if coord.dim() != 3 or coord.shape[-1] not in (1, 2):
        raise ValueError("coord must be (batch, n_points, 1|2)")
    b, n, cd = coord.shape
    dev = coord.device
    base = coord[0].contiguous()
    phi_t = _per_sample(phi, b, n, 1, dev) if cd == 2 else None
    sc_t = _per_sample(scale, b, n, 1, dev) if cd == 2 else None
    dx_t = _per_sample(coord_dx, b, n, cd, dev)
    out = torch.empty(b, n, cd, device=dev, dtype=torch.float32)
    with _abi.device_of(dev):
        _abi.check(_abi.lib().pv_transform_coordinates(
            _abi.ptr(base), n, cd, _abi.ptr(phi_t), _abi.ptr(dx_t), _abi.ptr(sc_t), b, _abi.ptr(out),
            _abi.current_stream()), "pv_transform_coordinates")
    return out


def generate_latent_grid(d, **kwargs):
    """Grid of 2-D latent coordinates (pyroved/utils/coord.py:91-109): inverse normal CDF of linspace(0.05, 0.95)
    unless z_coord=[z1, z2, z3, z4] is given.  Returns (z (d0*d1, 2), (grid_x, grid_y))."""
    import torch.distributions as td
    if isinstance(d, int):
        d = [d, d]
    z_coord = kwargs.get("z_coord")
    if z_coord:
        z1, z2, z3, z4 = z_coord
        grid_x = torch.linspace(z2, z1, d[0])
        grid_y = torch.linspace(z3, z4, d[1])
    else:
        grid_x = td.Normal(0, 1).icdf(torch.linspace(0.95, 0.05, d[0]))
        grid_y = td.Normal(0, 1).icdf(torch.linspace(0.05, 0.95, d[1]))
    z = [torch.tensor([xi, yi]).float().unsqueeze(0) for xi in grid_x for yi in grid_y]
    return torch.cat(z), (grid_x, grid_y)


def generate_latent_grid_traversal(d: int, cont_dim: int, disc_dim: int, cont_idx: int, cont_idx_fixed: int,
                                   num_samples: int):
    """Continuous and discrete grids for a latent-space traversal (pyroved/utils/coord.py:112-133)."""
    import torch.distributions as td
    samples_cont = torch.zeros(size=(num_samples, cont_dim)) + cont_idx_fixed
    cont_traversal = td.Normal(0, 1).icdf(torch.linspace(0.95, 0.05, d))
    for i in range(d):
        for j in range(d):
            samples_cont[i * d + j, cont_idx] = cont_traversal[j]
    n = torch.arange(0, disc_dim).tile(d // disc_dim + 1)[:d]
    samples_disc = []
    for i in range(d):
        block = torch.zeros((d, disc_dim))
        block[:, n[i]] = 1
        samples_disc.append(block)
    return samples_cont, torch.cat(samples_disc)
