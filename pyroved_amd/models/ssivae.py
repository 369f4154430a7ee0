"""
ssivae.py — semi-supervised VAEs with rotational / translational / scale invariances; host-side mirror of
pyroved/models/ssivae.py:27-384 (ssiVAE: classification) and pyroved/models/ss_reg_ivae.py:26-346 (ss_reg_iVAE:
regression).  Same constructor signatures, sub-module names (encoder_z, encoder_y, decoder: hence state_dict keys),
parameter initialisation order and inference API (classifier / regressor, encode, decode, manifold2d,
manifold_traversal, set_classifier / set_regressor).

`model` / `guide` / `model_aux` / `guide_aux` are Pyro programs in the reference and (for users with pyro-ppl) here;
trainers.auxSVItrainer evaluates the objectives they define in the HIP library (engine_ss.SSEngine).
"""
from typing import Optional, Tuple, Union, Type, List

import torch

from .base import baseVAE
from .ivae import _plot_manifold
from ..nets import fcDecoderNet, fcEncoderNet, sDecoderNet, fcClassifierNet, fcRegressorNet
from ..utils import (get_sampler, set_deterministic_mode, to_onehot, iter_batches, generate_latent_grid,
                     generate_latent_grid_traversal)

tt = torch.tensor


class _ssBase(baseVAE):
    """What ssiVAE and ss_reg_iVAE share: z-encoder conditioned on the label, label network, conditioned decoder."""

    def _build(self, data_dim, latent_dim, label_dim, label_net, hidden_dim_e, hidden_dim_d, hidden_dim_y, activation,
               sampler_d, sigmoid_d, seed, kwargs):
        set_deterministic_mode(seed)
        self.data_dim = tuple(int(d) for d in data_dim)
        # same construction order as the reference (ssivae.py:137-152): z-encoder, label network, decoder
        self.encoder_z = fcEncoderNet(data_dim, latent_dim + self.coord, label_dim, hidden_dim_e, activation, flat=False)
        self.encoder_y = label_net(data_dim, label_dim, hidden_dim_y, activation)
        dnet = sDecoderNet if 0 < self.coord < 5 else fcDecoderNet
        self.decoder = dnet(data_dim, latent_dim, label_dim, hidden_dim_d, activation, sigmoid_out=sigmoid_d, unflat=False)
        self.sampler_d = get_sampler(sampler_d, **kwargs)
        self.z_dim = latent_dim + self.coord
        self.c_dim = label_dim               # the label vector is the conditioning input of encoder_z and decoder
        self.to(self.device)

    def engine(self, **kw):
        from ..engine_ss import SSEngine
        if self._engine is None:
            self._engine = SSEngine(self, **kw)
        elif kw:
            self._engine.configure(**kw)      # an engine made earlier (encode, a previous trainer) takes the new settings
        return self._engine

    # model / guide / model_aux / guide_aux: real Pyro programs for users who have pyro-ppl (models/_pyro_programs.py; same
    # sites, plates and scales as models/ssivae.py:153-234 and models/ss_reg_ivae.py:156-246); auxSVItrainer never goes
    # through them — it evaluates the same objectives in the HIP library (engine_ss.SSEngine)
    def model(self, xs, ys=None, **kwargs):
        from ._pyro_programs import ss_model
        return ss_model(self, xs, ys, **kwargs)

    def guide(self, xs, ys=None, **kwargs):
        from ._pyro_programs import ss_guide
        return ss_guide(self, xs, ys, **kwargs)

    def model_aux(self, xs, ys=None, **kwargs):
        from ._pyro_programs import ss_model_aux
        return ss_model_aux(self, xs, ys, **kwargs)

    def guide_aux(self, xs, ys=None, **kwargs):
        from ._pyro_programs import ss_guide_aux
        return ss_guide_aux(self, xs, ys, **kwargs)

    def split_latent(self, zs: torch.Tensor) -> Tuple[torch.Tensor]:
        """Split a latent variable into the transformation parts and the content (ssivae.py:203-213)."""
        zdims = list(zs.shape)
        zdims[-1] = zdims[-1] - self.coord
        zs = zs.view(-1, zs.size(-1))
        phi, dx, sc, zs = self._split_latent(zs)
        return phi, dx, sc, zs.view(*zdims)

    def _label_net_batches(self, x_new: torch.Tensor, **kwargs: int) -> torch.Tensor:
        eng = self.engine()
        loader = iter_batches(x_new, batch_size=kwargs.get("batch_size", 100))
        out = []
        for (x_i,) in loader:
            with torch.no_grad():
                out.append(eng.label_forward(x_i.to(eng.device, torch.float32)).cpu())
        return torch.cat(out)

    def decode(self, z: torch.Tensor, y: torch.Tensor, **kwargs: int) -> torch.Tensor:
        """Decodes a batch of (content) latent coordinates given label vectors y (ssivae.py:296-307)."""
        z = torch.cat([z.to(torch.float32).cpu(), y.to(torch.float32).cpu()], -1)
        loc = self._decode(z, **kwargs)
        return loc.view(-1, *self.data_dim)


class ssiVAE(_ssBase):
    """
    Semi-supervised VAE (classification) with rotational, translational and scale invariances
    (pyroved/models/ssivae.py).  Args as in the reference: data_dim, latent_dim, num_classes, invariances,
    hidden_dim_e, hidden_dim_d, hidden_dim_cls, activation, sampler_d, sigmoid_d, seed; kwargs device, dx_prior,
    dy_prior, sc_prior, decoder_sig.
    """
    def __init__(self, data_dim: Tuple[int], latent_dim: int, num_classes: int, invariances: List[str] = None,
                 hidden_dim_e: List[int] = None, hidden_dim_d: List[int] = None, hidden_dim_cls: List[int] = None,
                 activation: str = "tanh", sampler_d: str = "bernoulli", sigmoid_d: bool = True, seed: int = 1,
                 **kwargs: Union[str, float]) -> None:
        super(ssiVAE, self).__init__(data_dim, invariances, **kwargs)
        self._build(data_dim, latent_dim, num_classes, fcClassifierNet, hidden_dim_e, hidden_dim_d, hidden_dim_cls,
                    activation, sampler_d, sigmoid_d, seed, kwargs)
        self.num_classes = num_classes

    def set_classifier(self, cls_net: Type[torch.nn.Module]) -> None:
        """Sets a user-defined classification network (ssivae.py:236-240)."""
        self.encoder_y = cls_net.to(self.device)
        self._engine = None

    def classifier(self, x_new: torch.Tensor, **kwargs: int) -> torch.Tensor:
        """Classifies data with the trained label network (ssivae.py:242-266): predicted class indices (CPU)."""
        alpha = self._label_net_batches(x_new, **kwargs)
        _, predicted = torch.max(alpha, 1)
        return predicted

    def encode(self, x_new: torch.Tensor, y: Optional[torch.Tensor] = None, **kwargs: int):
        """(z_loc, z_scale, predicted class) (ssivae.py:268-294); y: one-hot classes or indices, predicted when omitted."""
        if y is None:
            y = self.classifier(x_new, **kwargs)
        if y.ndim < 2:
            y = to_onehot(y, self.num_classes)
        z = self._encode(x_new.reshape(x_new.shape[0], -1), y, **kwargs)
        z_loc, z_scale = z.split(self.z_dim, 1)
        _, y_pred = torch.max(y, 1)
        return z_loc, z_scale, y_pred

    def manifold2d(self, d: int, plot: bool = True, **kwargs: Union[str, int, float]) -> torch.Tensor:
        """Decoded d x d grid of the latent space for the class kwargs['label'] (ssivae.py:309-337)."""
        z, (grid_x, grid_y) = generate_latent_grid(d, **kwargs)
        cls = tt(kwargs.get("label", 0))
        if cls.ndim < 2:
            cls = to_onehot(cls.unsqueeze(0), self.num_classes)
        cls = cls.repeat(z.shape[0], 1)
        loc = self.decode(z, cls, **kwargs)
        if plot:
            _plot_manifold(self.ndim, loc, d, grid_x, grid_y, kwargs)
        return loc

    def manifold_traversal(self, d: int, cont_idx: int, cont_idx_fixed: int = 0, plot: bool = True,
                           **kwargs: Union[str, int, float]) -> torch.Tensor:
        """Latent-space traversal over one continuous latent and the classes (ssivae.py:339-384)."""
        samples_cont, samples_disc = generate_latent_grid_traversal(
            d, self.z_dim - self.coord, self.num_classes, cont_idx, cont_idx_fixed, d ** 2)
        decoded = self.decode(samples_cont, samples_disc, **kwargs)
        if plot:
            from ..utils.viz import plot_grid_traversal
            plot_grid_traversal(decoded, d, self.data_dim, self.num_classes, **kwargs)
        return decoded


class ss_reg_iVAE(_ssBase):
    """
    Semi-supervised VAE (regression) with rotational, translational and scale invariances
    (pyroved/models/ss_reg_ivae.py).  Args as in the reference: data_dim, latent_dim, reg_dim, invariances,
    hidden_dim_e, hidden_dim_d, hidden_dim_reg, activation, sampler_d, sigmoid_d, seed; kwargs device, dx_prior,
    dy_prior, sc_prior, decoder_sig, regressor_sig.
    """
    def __init__(self, data_dim: Tuple[int], latent_dim: int, reg_dim: int, invariances: List[str] = None,
                 hidden_dim_e: List[int] = None, hidden_dim_d: List[int] = None, hidden_dim_reg: List[int] = None,
                 activation: str = "tanh", sampler_d: str = "bernoulli", sigmoid_d: bool = True, seed: int = 1,
                 **kwargs: Union[str, float]) -> None:
        super(ss_reg_iVAE, self).__init__(data_dim, invariances, **kwargs)
        self._build(data_dim, latent_dim, reg_dim, fcRegressorNet, hidden_dim_e, hidden_dim_d, hidden_dim_reg,
                    activation, sampler_d, sigmoid_d, seed, kwargs)
        self.reg_sig = kwargs.get("regressor_sig", 0.5)
        self.reg_dim = reg_dim

    def set_regressor(self, reg_net: Type[torch.nn.Module]) -> None:
        """Sets a user-defined regression network (ss_reg_ivae.py:242-246)."""
        self.encoder_y = reg_net.to(self.device)
        self._engine = None

    def regressor(self, x_new: torch.Tensor, **kwargs: int) -> torch.Tensor:
        """Applies the trained regressor (ss_reg_ivae.py:248-272): predictions on the CPU."""
        return self._label_net_batches(x_new, **kwargs)

    def encode(self, x_new: torch.Tensor, y: Optional[torch.Tensor] = None, **kwargs: int):
        """(z_loc, z_scale, y) (ss_reg_ivae.py:274-298); y: the continuous variable(s), regressed when omitted."""
        if y is None:
            y = self.regressor(x_new, **kwargs)
        z = self._encode(x_new.reshape(x_new.shape[0], -1), y, **kwargs)
        z_loc, z_scale = z.split(self.z_dim, 1)
        return z_loc, z_scale, y

    def manifold2d(self, d: int, y: torch.Tensor, plot: bool = True, **kwargs: Union[str, int, float]) -> torch.Tensor:
        """Decoded d x d grid of the latent space conditioned on y (ss_reg_ivae.py:312-346)."""
        z, (grid_x, grid_y) = generate_latent_grid(d, **kwargs)
        y = y.unsqueeze(1) if 0 < y.ndim < 2 else y
        y = y.expand(z.shape[0], *y.shape[1:])
        loc = self.decode(z, y, **kwargs)
        if plot:
            _plot_manifold(self.ndim, loc, d, grid_x, grid_y, kwargs)
        return loc
