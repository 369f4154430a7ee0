"""
viz.py — the three plotting helpers the models' manifold2d / manifold_traversal call when plot=True
(pyroved/utils/viz.py:7-83).  Host-side matplotlib only; the image mosaic the reference gets from
torchvision.utils.make_grid (not a dependency here) is laid out by `tile_images`.
"""
from typing import List, Tuple, Union

import torch


def tile_images(imgdata: torch.Tensor, nrow: int, padding: int = 2, pad_value: float = 0) -> torch.Tensor:
    """2-D mosaic of a (n, h, w) or (n, 1, h, w) stack, `nrow` images per mosaic row, each cell preceded by
    `padding` pixels of `pad_value` above / left and one closing border below / right — the first channel of
    torchvision's make_grid(imgdata, nrow, padding, pad_value=...) (viz.py:16-18)."""
    if imgdata.ndim == 4:
        imgdata = imgdata[:, 0]
    if imgdata.ndim != 3:
        raise AssertionError("Images must be passed as a 3D or 4D tensor")
    imgdata = imgdata.detach().cpu().to(torch.float32)
    n, h, w = imgdata.shape
    cols = min(int(nrow), n)
    rows = (n + cols - 1) // cols
    ch, cw = h + padding, w + padding
    out = torch.full((rows * ch + padding, cols * cw + padding), float(pad_value))
    for i in range(n):
        r, c = divmod(i, cols)
        out[r * ch + padding:r * ch + padding + h, c * cw + padding:c * cw + padding + w] = imgdata[i]
    return out


def _plt():
    try:
        import matplotlib.pyplot as plt
    except ImportError as e:                                   # pragma: no cover
        raise ImportError("plot=True needs matplotlib; call with plot=False to get the decoded tensor only") from e
    return plt


def _extent(extent):
    if not extent:
        return None
    return [float(e) for e in extent]


def plot_img_grid(imgdata: torch.Tensor, d: int, **kwargs: Union[str, int, List[float]]) -> None:
    """d-by-d mosaic of decoded 2-D images over the latent grid (viz.py:7-30)."""
    if imgdata.ndim < 3:
        raise AssertionError("Images must be passed as a 3D or 4D tensor")
    plt = _plt()
    mosaic = tile_images(imgdata, d, kwargs.get("padding", 2), kwargs.get("pad_value", 0))
    plt.figure(figsize=(8, 8))
    plt.imshow(mosaic.numpy(), cmap=kwargs.get("cmap", "gnuplot"), origin=kwargs.get("origin", "upper"),
               extent=_extent(kwargs.get("extent")))
    plt.xticks(fontsize=14)
    plt.yticks(fontsize=14)
    plt.xlabel("$z_1$", fontsize=18)
    plt.ylabel("$z_2$", fontsize=18)
    plt.show()


def plot_spect_grid(spectra: torch.Tensor, d: int, **kwargs: List[float]) -> None:
    """d-by-d panel of decoded 1-D spectra (viz.py:33-46)."""
    plt = _plt()
    _, axes = plt.subplots(d, d, figsize=(8, 8), subplot_kw={"xticks": [], "yticks": []},
                           gridspec_kw=dict(hspace=0.1, wspace=0.1), squeeze=False)
    ylim = kwargs.get("ylim")
    for ax, y in zip(axes.flat, spectra.detach().cpu()):
        ax.plot(y.squeeze().numpy())
        if ylim:
            ax.set_ylim(*ylim)
    plt.show()


def plot_grid_traversal(imgdata: torch.Tensor, d: int, data_dim: Tuple[int], disc_dim: int,
                        **kwargs: Union[str, int, List[float]]) -> None:
    """disc_dim-by-d mosaic: one row per class, one column per value of the swept latent (viz.py:49-83)."""
    if imgdata.ndim < 3:
        raise AssertionError("Images must be passed as a 3D or 4D tensor")
    plt = _plt()
    padding = kwargs.get("padding", 2)
    mosaic = tile_images(imgdata, d, padding, kwargs.get("pad_value", 0))[:(data_dim[0] + padding) * disc_dim]
    plt.figure(figsize=(8, 8))
    plt.imshow(mosaic.numpy(), cmap=kwargs.get("cmap", "gnuplot"), origin=kwargs.get("origin", "upper"),
               extent=_extent(kwargs.get("extent")))
    plt.xlabel("$z_{cont}$", fontsize=18)
    plt.ylabel("$z_{disc}$", fontsize=18)
    plt.xticks([])
    plt.yticks([])
    plt.show()
