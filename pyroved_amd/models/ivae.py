"""
ivae.py — variational autoencoder that enforces invariance to rotation, translation
and scale; host-side mirror of pyroved/models/ivae.py:27-310.

Same constructor signature, attributes (encoder_z, decoder, sampler_d, z_dim, c_dim,
grid, t_prior, sc_prior, coord, ndim, invariances, device), parameter initialisation
order (so `seed` reproduces the reference's initial weights bit-for-bit) and
inference API (encode / decode / manifold2d / split_latent).

`model()` / `guide()` are Pyro programs in the reference; here the SVI objective they
define is evaluated by the HIP library (trainers.SVItrainer -> engine.loss_and_grads ->
pv_ivae_loss_and_grads), so they are kept only as thin hooks that run the same
computation and return the ELBO terms.
"""
from typing import Optional, Tuple, Union, List

import torch

from .base import baseVAE
from ..nets import fcDecoderNet, fcEncoderNet, sDecoderNet
from ..utils import get_sampler, set_deterministic_mode


def _plot_manifold(ndim, loc, d, grid_x, grid_y, kwargs):
    """The plot=True tail shared by every manifold2d (models/ivae.py:302-309)."""
    from ..utils.viz import plot_img_grid, plot_spect_grid
    dd = d[0] if isinstance(d, (list, tuple)) else d
    if ndim == 2:
        plot_img_grid(loc, dd, extent=[grid_x.min(), grid_x.max(), grid_y.min(), grid_y.max()], **kwargs)
    elif ndim == 1:
        plot_spect_grid(loc, dd, **kwargs)


class iVAE(baseVAE):
    """
    Variational autoencoder that enforces rotational, translational,
    and scale invariances.

    Args:
        data_dim: (height, width) or (length,)
        latent_dim: number of latent dimensions (content)
        invariances: e.g. ['r'], ['r', 't'], ['r', 't', 's'], ['t'] (1D), None (vanilla VAE)
        c_dim: "feature dimension" of the class vector for class-conditioned VAEs
        hidden_dim_e / hidden_dim_d: hidden layer widths of encoder / decoder (default [128, 128])
        activation: 'tanh' (default), 'lrelu', 'softplus', 'relu', 'gelu'
        sampler_d: 'bernoulli' (default), 'continuous_bernoulli', 'gaussian'
        sigmoid_d: sigmoid at the decoder output (default True)
        seed: seed used in torch.manual_seed(seed)

    Keyword Args:
        device, dx_prior, dy_prior, sc_prior, decoder_sig — as in the reference.
    """

    def __init__(
        self,
        data_dim: Tuple[int],
        latent_dim: int = 2,
        invariances: List[str] = None,
        c_dim: int = 0,
        hidden_dim_e: List[int] = None,
        hidden_dim_d: List[int] = None,
        activation: str = "tanh",
        sampler_d: str = "bernoulli",
        sigmoid_d: bool = True,
        seed: int = 1,
        **kwargs: Union[str, float]
         ) -> None:
        args = (data_dim, invariances)
        super(iVAE, self).__init__(*args, **kwargs)

        # same RNG consumption as the reference (models/ivae.py:140-154): seed, then the
        # encoder's Linear layers, then the decoder's, all on the CPU generator
        set_deterministic_mode(seed)

        self.encoder_z = fcEncoderNet(
            data_dim, latent_dim + self.coord, c_dim, hidden_dim_e,
            activation, softplus_out=True
        )
        dnet = sDecoderNet if 0 < self.coord < 5 else fcDecoderNet
        self.decoder = dnet(
            data_dim, latent_dim, c_dim, hidden_dim_d,
            activation, sigmoid_out=sigmoid_d
        )
        self.sampler_d = get_sampler(sampler_d, **kwargs)

        self.z_dim = latent_dim + self.coord
        self.c_dim = c_dim

        self.to(self.device)

    # ---------------------------------------------------------------- ELBO hooks
    def elbo_terms(self, x: torch.Tensor, y: Optional[torch.Tensor] = None,
                   eps: Optional[torch.Tensor] = None, **kwargs: float):
        """Evaluates guide + model once (no gradients, no update) through the HIP library and
        returns dict(loss, ll, logpz, logqz) as python floats.  `eps` (B, z_dim) is the
        standard-normal draw of the guide; drawn on the CPU generator when omitted.
        Replaces tracing `model`/`guide` with Pyro (models/ivae.py:165-221)."""
        eng = self.engine()
        x = x.to(eng.device, torch.float32)
        if eps is None:
            eps = torch.empty(x.shape[0], self.z_dim).normal_()
        eng.loss_and_grads(x, eps.to(eng.device, torch.float32), kwargs.get("scale_factor", 1.),
                           None if y is None else y.to(eng.device, torch.float32), want_grads=False)
        s = eng.scalars.cpu().tolist()
        return dict(loss=s[0], ll=s[1], logpz=s[2], logqz=s[3])

    def model(self, x: torch.Tensor, y: Optional[torch.Tensor] = None, **kwargs: float) -> None:
        """p(x|z)p(z) as a Pyro program (models/ivae.py:165-202) — for users with pyro-ppl (poutine.trace, custom ELBOs,
        pyro.infer.SVI); raises NotImplementedError without it.  SVItrainer evaluates the same objective in HIP kernels
        and does not go through here (iVAE.elbo_terms does neither)."""
        from ._pyro_programs import ivae_model
        return ivae_model(self, x, y, **kwargs)

    def guide(self, x: torch.Tensor, y: Optional[torch.Tensor] = None, **kwargs: float) -> None:
        """q(z|x) as a Pyro program (models/ivae.py:204-221); see `model`."""
        from ._pyro_programs import ivae_guide
        return ivae_guide(self, x, y, **kwargs)

    def split_latent(self, z: torch.Tensor) -> Tuple[torch.Tensor]:
        """Split latent variable into parts associated with coordinate transformations
        (rotation and/or translation and/or scale) and image content."""
        return self._split_latent(z)

    # ---------------------------------------------------------------- inference
    def encode(self, x_new: torch.Tensor, y: torch.Tensor = None, **kwargs: int) -> torch.Tensor:
        """Encodes data with the trained encoder: returns (z_loc, z_scale) on the CPU.  The last
        latent_dim columns are the content latents; the first ones are rotation, dx, dy, scale
        (models/ivae.py:230-256).  kwargs: batch_size."""
        enc_args = [x_new, y] if y is not None else [x_new, ]
        z = self._encode(*enc_args, **kwargs)
        z_loc, z_scale = z.split(self.z_dim, 1)
        return z_loc, z_scale

    def decode(self, z: torch.Tensor, y: torch.Tensor = None, **kwargs: int) -> torch.Tensor:
        """Decodes a batch of (content) latent coordinates into the data space
        (models/ivae.py:258-275).  kwargs: batch_size, angle, shift, scale."""
        z = z.to(torch.float32)
        if y is not None:
            z = torch.cat([z.cpu(), y.to(torch.float32).cpu()], -1)
        return self._decode(z.cpu(), **kwargs)

    def manifold2d(self, d: int, y: torch.Tensor = None, plot: bool = True,
                   **kwargs: Union[str, int, float]) -> torch.Tensor:
        """Decodes a d x d grid of the 2-D latent space (models/ivae.py:277-310).  The grid is the
        reference's generate_latent_grid: inverse normal CDF of linspace(0.05, 0.95, d) unless
        z_coord=[z1, z2, z3, z4] is given; plot=True draws the mosaic (utils.plot_img_grid / plot_spect_grid)."""
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
        z = []
        for xi in grid_x:
            for yi in grid_y:
                z.append(torch.tensor([xi, yi]).float().unsqueeze(0))
        z = torch.cat(z)
        if self.c_dim > 0:
            if y is None:
                raise ValueError("To generate a manifold pass a conditional vector y")
            y = y.unsqueeze(1) if 0 < y.ndim < 2 else y
            loc = self.decode(z, y.expand(z.shape[0], *y.shape[1:]), **kwargs)
        else:
            loc = self.decode(z, **kwargs)
        if plot:
            _plot_manifold(self.ndim, loc, d, grid_x, grid_y, kwargs)
        return loc

    def predict_on_latent(self, train_data, gp_labels, gp_iterations: int = 1, d: int = 12, plot: bool = False):
        """Gaussian-process regression of labels over the latent space, evaluated on the d x d latent grid
        (models/ivae.py:312-364): encode, fit utils.gp_model on the encoded means, predict on
        utils.generate_latent_grid(d), decode the grid.  Returns ((z, z_decoded), predictions)."""
        from ..utils import generate_latent_grid
        from ..utils.gp import gp_model
        X = torch.as_tensor(train_data, dtype=torch.float32)
        y = torch.as_tensor(gp_labels, dtype=torch.float32)
        encoded_X = self.encode(X)[0]
        gpr = gp_model(input_dim=encoded_X.shape[1], encoded_X=encoded_X, y=y, gp_iterations=gp_iterations)
        z, _ = generate_latent_grid(d)
        z = torch.as_tensor(z, dtype=torch.float32)
        gpr.eval()
        with torch.no_grad():
            predictions, _ = gpr(z)
        z_decoded = self.manifold2d(d, plot=False)
        if plot:
            import matplotlib.pyplot as plt
            self.manifold2d(d=d, cmap='viridis')
            plt.figure(figsize=(8, 8))
            heatmap = plt.imshow(predictions.reshape(d, d), cmap='viridis', aspect='auto')
            plt.colorbar(heatmap, label='Prediction Value')
            plt.xticks(fontsize=14)
            plt.yticks(fontsize=14)
            plt.xlabel("$z_1$", fontsize=14)
            plt.ylabel("$z_2$", fontsize=14)
            plt.title('Predictions Visualization')
            plt.show()
        return (z, z_decoded), predictions
