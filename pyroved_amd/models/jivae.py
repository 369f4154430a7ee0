"""
jivae.py — joint (continuous + discrete) rotationally-, translationally- and scale-invariant
variational autoencoder; host-side mirror of pyroved/models/jivae.py:27-330.

Same constructor signature, attributes (encoder_z: jfcEncoderNet, decoder, sampler_d, z_dim,
discrete_dim, grid, t_prior, sc_prior, coord, ...), parameter initialisation order and inference
API (encode -> (z_loc, z_scale, classes), decode(z, y), manifold2d, manifold_traversal).

`model()` / `guide()` are Pyro programs in the reference, trained with
`SVItrainer(model, enumerate_parallel=True)` = TraceEnum_ELBO with the guide's OneHotCategorical
site enumerated in parallel (trainers/svi.py:83-90).  Here that objective,
    loss = -sum_b [ b0 (log p(z_b) - log q(z_b|x_b))
                    + sum_k alpha_bk ( log p(x_b|z_b,k) + b1 log(1/K) - b1 log alpha_bk ) ],
is evaluated by the HIP library: the spatial decoder runs on the K*B rows [k][b] in the same fused
kernel as iVAE, each row's gradient weighted by alpha_bk (engine.loss_and_grads ->
pv_ivae_loss_and_grads with plan.discrete_dim = K).
"""
from typing import List, Tuple, Union

import torch

from .base import baseVAE
from .ivae import _plot_manifold
from ..nets import fcDecoderNet, jfcEncoderNet, sDecoderNet
from ..utils import get_sampler, set_deterministic_mode, to_onehot


class jiVAE(baseVAE):
    """
    Args:
        data_dim: (height, width) or (length,)
        latent_dim: number of continuous latent dimensions (content)
        discrete_dim: number of classes of the discrete latent
        invariances: e.g. ['r'], ['r', 't'], ['r', 't', 's'], ['t'] (1D)
        hidden_dim_e / hidden_dim_d, activation, sampler_d, sigmoid_d, seed: as in models.iVAE

    Keyword Args:
        device, dx_prior, dy_prior, sc_prior, decoder_sig — as in the reference.
    """

    def __init__(self,
                 data_dim: Tuple[int],
                 latent_dim: int,
                 discrete_dim: int,
                 invariances: List[str] = None,
                 hidden_dim_e: List[int] = None,
                 hidden_dim_d: List[int] = None,
                 activation: str = "tanh",
                 sampler_d: str = "bernoulli",
                 sigmoid_d: bool = True,
                 seed: int = 1,
                 **kwargs: Union[str, float]
                 ) -> None:
        args = (data_dim, invariances)
        super(jiVAE, self).__init__(*args, **kwargs)
        # same RNG consumption as the reference (models/jivae.py:126-141)
        set_deterministic_mode(seed)
        self.data_dim = data_dim
        self.encoder_z = jfcEncoderNet(
            data_dim, latent_dim + self.coord, discrete_dim,
            hidden_dim_e, activation, softplus_out=True)
        dnet = sDecoderNet if 0 < self.coord < 5 else fcDecoderNet
        self.decoder = dnet(
            data_dim, latent_dim, discrete_dim, hidden_dim_d,
            activation, sigmoid_out=sigmoid_d, unflat=False)
        self.sampler_d = get_sampler(sampler_d, **kwargs)
        self.z_dim = latent_dim + self.coord
        self.discrete_dim = discrete_dim
        self.c_dim = 0
        self.to(self.device)

    def model(self, x: torch.Tensor, **kwargs: float) -> None:
        """p(x|z,c)p(z)p(c) as a Pyro program (models/jivae.py:152-197) — needs pyro-ppl; SVItrainer evaluates the same
        objective in HIP kernels and does not go through here."""
        from ._pyro_programs import jivae_model
        return jivae_model(self, x, **kwargs)

    def guide(self, x: torch.Tensor, **kwargs: float) -> None:
        """q(z,c|x) as a Pyro program (models/jivae.py:199-220); see `model`."""
        from ._pyro_programs import jivae_guide
        return jivae_guide(self, x, **kwargs)

    def split_latent(self, z: torch.Tensor) -> Tuple[torch.Tensor]:
        return self._split_latent(z)

    def encode(self, x_new: torch.Tensor, logits: bool = False, **kwargs: int) -> torch.Tensor:
        """Returns (z_loc, z_scale, classes): classes are the argmax of the class probabilities, or the
        probabilities themselves with logits=True (models/jivae.py:228-253).  kwargs: batch_size."""
        z = self._encode(x_new, **kwargs)
        z_loc = z[:, :self.z_dim]
        z_scale = z[:, self.z_dim:2 * self.z_dim]
        classes = z[:, 2 * self.z_dim:]
        if not logits:
            _, classes = torch.max(classes, 1)
        return z_loc, z_scale, classes

    def decode(self, z: torch.Tensor, y: torch.Tensor, **kwargs: int) -> torch.Tensor:
        """Decodes latent coordinates z (content part) for one-hot classes y (models/jivae.py:255-267).
        kwargs: batch_size, angle, shift, scale."""
        z = torch.cat([z.to(torch.float32).cpu(), y.to(torch.float32).cpu()], -1)
        loc = self._decode(z, **kwargs)
        return loc.view(-1, *self.data_dim)

    def manifold2d(self, d: int, disc_idx: int = 0, plot: bool = True, **kwargs) -> torch.Tensor:
        """Decodes a d x d grid of the continuous latent space for class disc_idx (models/jivae.py:269-296)."""
        import torch.distributions as td
        dd = [d, d] if isinstance(d, int) else d
        z_coord = kwargs.get("z_coord")
        if z_coord:
            z1, z2, z3, z4 = z_coord
            grid_x = torch.linspace(z2, z1, dd[0])
            grid_y = torch.linspace(z3, z4, dd[1])
        else:
            grid_x = td.Normal(0, 1).icdf(torch.linspace(0.95, 0.05, dd[0]))
            grid_y = td.Normal(0, 1).icdf(torch.linspace(0.05, 0.95, dd[1]))
        z = torch.cat([torch.tensor([xi, yi]).float().unsqueeze(0) for xi in grid_x for yi in grid_y])
        z_disc = to_onehot(torch.tensor(disc_idx).unsqueeze(0), self.discrete_dim).repeat(z.shape[0], 1)
        loc = self.decode(z, z_disc, **kwargs)
        if plot:
            _plot_manifold(self.ndim, loc, d, grid_x, grid_y, kwargs)
        return loc

    def manifold_traversal(self, d: int, cont_idx: int, cont_idx_fixed: int = 0, plot: bool = True,
                           **kwargs) -> torch.Tensor:
        """Latent traversal over one continuous variable for every class (models/jivae.py:298-330 with
        utils.generate_latent_grid_traversal)."""
        import torch.distributions as td
        disc_dim, cont_dim = self.discrete_dim, self.z_dim - self.coord
        # d x d grid: column j sweeps the chosen continuous variable, row i fixes a class (classes cycle over rows)
        samples_cont = torch.full((d * d, cont_dim), float(cont_idx_fixed))
        samples_cont[:, cont_idx] = td.Normal(0, 1).icdf(torch.linspace(0.95, 0.05, d)).repeat(d)
        row_class = torch.arange(d) % disc_dim
        samples_disc = to_onehot(row_class.repeat_interleave(d), disc_dim)
        decoded = self.decode(samples_cont, samples_disc, **kwargs)
        if plot:
            from ..utils.viz import plot_grid_traversal
            plot_grid_traversal(decoded, d, self.data_dim, disc_dim, **kwargs)
        return decoded
