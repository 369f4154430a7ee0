"""
ved.py — variational encoder-decoder (e.g. image -> spectrum); host-side mirror of pyroved/models/ved.py:26-243.

Same constructor signature, attributes (encoder_z: convEncoderNet, decoder: convDecoderNet, sampler_d, z_dim,
ndim), parameter initialisation order and inference API (encode / decode / predict / manifold2d).  `model()` /
`guide()` are Pyro programs in the reference; the objective they define,
    loss = -( sum_b log p(y_b | z_b) + beta*sum_b log N(z_b; 0, 1) - beta*sum_b log N(z_b; mu(x_b), sigma(x_b)) ),
is evaluated by the HIP library (trainers.SVItrainer -> engine_ved.VEDEngine -> pv_ved_loss_and_grads).
"""
from typing import List, Tuple, Union

import torch

from .base import baseVAE
from ..nets.conv import convEncoderNet, convDecoderNet
from ..utils import get_sampler, iter_batches, set_deterministic_mode


class VED(baseVAE):
    """
    Args:
        input_dim: (h, w) or (l,) of the inputs
        output_dim: (h, w) or (l,) of the targets
        input_channels / output_channels: channel counts (default 1)
        latent_dim: number of latent dimensions (default 2)
        hidden_dim_e: encoder conv filters per block (default [(32,), (64, 64), (128, 128)])
        hidden_dim_d: decoder conv filters per block (default [(128, 128), (64, 64), (32,)])
        activation: 'lrelu' (default), 'tanh', 'softplus', 'relu'
        batchnorm: batch normalisation after every conv activation (default False; op PV_OP_BATCHNORM of the HIP conv stack)
        sampler_d: 'bernoulli' (default) or 'gaussian'
        sigmoid_d: sigmoid at the decoder output (default True)
        seed: seed used in torch.manual_seed(seed)
    """

    def __init__(self,
                 input_dim: Tuple[int],
                 output_dim: Tuple[int],
                 input_channels: int = 1,
                 output_channels: int = 1,
                 latent_dim: int = 2,
                 hidden_dim_e: List[int] = None,
                 hidden_dim_d: List[int] = None,
                 activation: str = "lrelu",
                 batchnorm: bool = False,
                 sampler_d: str = "bernoulli",
                 sigmoid_d: bool = True,
                 seed: int = 1,
                 **kwargs: float
                 ) -> None:
        # (the reference forces device = cuda-if-available after super().__init__, ved.py:110; here the `device`
        #  keyword is honoured, default cuda-if-available as everywhere else)
        super(VED, self).__init__(output_dim, None, **kwargs)
        set_deterministic_mode(seed)
        self.ndim = len(output_dim)
        self.encoder_z = convEncoderNet(input_dim, latent_dim, input_channels, hidden_dim_e, batchnorm, activation)
        self.decoder = convDecoderNet(latent_dim, output_dim, output_channels, hidden_dim_d, batchnorm, activation,
                                      sigmoid_d)
        self.sampler_d = get_sampler(sampler_d, **kwargs)
        self.z_dim = latent_dim
        self.c_dim = 0
        self.to(self.device)

    def engine(self, **kw):
        from ..engine_ved import VEDEngine
        if self._engine is None:
            self._engine = VEDEngine(self, **kw)
            self.encoder_z._pv_engine = self.decoder._pv_engine = self._engine
        elif kw:
            self._engine.configure(**kw)      # an engine made earlier (encode, a previous trainer) takes the new settings
        return self._engine

    def model(self, x: torch.Tensor = None, y: torch.Tensor = None, **kwargs: float) -> None:
        """p(y|z)p(z) as a Pyro program (models/ved.py:122-145) — needs pyro-ppl; SVItrainer evaluates the same objective
        in HIP kernels and does not go through here."""
        from ._pyro_programs import ved_model
        return ved_model(self, x, y, **kwargs)

    def guide(self, x: torch.Tensor = None, y: torch.Tensor = None, **kwargs: float) -> None:
        """q(z|x) as a Pyro program (models/ved.py:147-163); see `model`."""
        from ._pyro_programs import ved_guide
        return ved_guide(self, x, y, **kwargs)

    def encode(self, x_new: torch.Tensor, **kwargs: int) -> torch.Tensor:
        """(z_loc, z_scale) of the encoded distributions, on the CPU (models/ved.py:165-181).  kwargs: batch_size.
        As in the reference this puts the module in eval() mode (batch-norm layers then use their running statistics)
        and nothing switches it back."""
        self.eval()
        z = self._encode(x_new, **kwargs)
        z_loc, z_scale = z.split(self.z_dim, 1)
        return z_loc, z_scale

    def decode(self, z: torch.Tensor, **kwargs: int) -> torch.Tensor:
        """Decodes latent coordinates into the target space (models/ved.py:183-196).  kwargs: batch_size."""
        self.eval()
        return self._decode(z.to(torch.float32).cpu(), **kwargs)

    def predict(self, x_new: torch.Tensor, **kwargs: int) -> torch.Tensor:
        """Forward prediction (encode -> 30 samples -> decode): mean and standard deviation (models/ved.py:198-216)."""
        eng = self.engine()
        loader = iter_batches(x_new, batch_size=kwargs.get("batch_size", 100))
        mus, sds = [], []
        for (x_i,) in loader:
            z_mu, z_sig = eng.encode(x_i.to(eng.device, torch.float32))
            z_mu, z_sig = z_mu.cpu(), z_sig.cpu()
            # the 30 draws come from the CPU generator as in the reference; they are decoded in ONE call (rows are
            # independent) and reduced on the device: one host copy per batch instead of 30 decode + .cpu() round trips
            z_samples = torch.distributions.Normal(z_mu, z_sig).rsample(sample_shape=(30,))
            if self.training and getattr(eng, "_bn_dec", None):
                # batch norm in train mode (the reference's predict never calls eval()): statistics are per decode CALL and
                # the running estimates move once per call — draw by draw, as models/ved.py:208-213 does
                y = torch.stack([eng.decode(z_samples[i].to(eng.device)) for i in range(30)])
            else:
                y = eng.decode(z_samples.reshape(-1, z_samples.shape[-1]).to(eng.device))
                y = y.reshape(30, z_mu.shape[0], *y.shape[1:])
            mus.append(y.mean(0).cpu())
            sds.append(y.std(0).cpu())
        return torch.cat(mus), torch.cat(sds)

    def manifold2d(self, d: int, plot: bool = True, **kwargs: Union[str, int]) -> torch.Tensor:
        """Decodes a d x d grid of the 2-D latent space (models/ved.py:218-243 with utils.generate_latent_grid)."""
        import torch.distributions as td
        self.eval()
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
        loc = self.decode(z)
        if plot:
            from .ivae import _plot_manifold
            _plot_manifold(self.ndim, loc, d, grid_x, grid_y, kwargs)
        return loc
