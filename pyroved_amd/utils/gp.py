"""
gp.py — the Gaussian-process helper behind iVAE.predict_on_latent (pyroved/utils/gp.py:5-29).  The reference builds
it from pyro.contrib.gp (RBF kernel + GPRegression, one Adam step per iteration on the negative log marginal
likelihood); Pyro is not a dependency here, so the same model is written out in plain torch: host-side linear algebra
on a few hundred encoded points, not part of the SVI path.
"""
import math

import torch
import torch.nn as nn


class GPRegression(nn.Module):
    """Exact GP regression, zero mean, RBF kernel k(a, b) = variance * exp(-|a - b|^2 / (2 lengthscale^2)), Gaussian
    noise — pyro.contrib.gp.models.GPRegression(X, y, kernels.RBF(input_dim)) with its defaults (variance =
    lengthscale = noise = 1, jitter 1e-6; the positive parameters are optimised through their logarithms).
    Calling the module on new inputs returns the predictive (mean, variance) of the noiseless latent function."""
    def __init__(self, X: torch.Tensor, y: torch.Tensor, jitter: float = 1e-6) -> None:
        super().__init__()
        self.X = X.detach().to(torch.float32)
        self.y = y.detach().to(torch.float32)
        self.jitter = jitter
        self.log_variance = nn.Parameter(torch.zeros(()))
        self.log_lengthscale = nn.Parameter(torch.zeros(()))
        self.log_noise = nn.Parameter(torch.zeros(()))

    def kernel(self, a: torch.Tensor, b: torch.Tensor = None) -> torch.Tensor:
        ls = self.log_lengthscale.exp()
        a = a / ls
        b = a if b is None else b / ls
        r2 = (a * a).sum(1, keepdim=True) - 2.0 * a @ b.t() + (b * b).sum(1, keepdim=True).t()
        return self.log_variance.exp() * torch.exp(-0.5 * r2.clamp(min=0))

    def _chol(self) -> torch.Tensor:
        Kff = self.kernel(self.X)
        Kff = Kff + (self.jitter + self.log_noise.exp()) * torch.eye(self.X.shape[0], dtype=Kff.dtype, device=Kff.device)
        return torch.linalg.cholesky(Kff)

    def loss(self) -> torch.Tensor:
        """Negative log marginal likelihood (what Trace_ELBO().differentiable_loss(gpr.model, gpr.guide) evaluates)."""
        Lff = self._chol()
        alpha = torch.linalg.solve_triangular(Lff, self.y.unsqueeze(-1), upper=False)
        n = self.X.shape[0]
        return 0.5 * (alpha * alpha).sum() + torch.log(torch.diagonal(Lff)).sum() + 0.5 * n * math.log(2 * math.pi)

    def forward(self, Xnew: torch.Tensor, full_cov: bool = False, noiseless: bool = True):
        Xnew = Xnew.to(torch.float32)
        Lff = self._chol()
        Kfs = self.kernel(self.X, Xnew)
        pack = torch.linalg.solve_triangular(Lff, torch.cat([self.y.unsqueeze(-1), Kfs], 1), upper=False)
        v, W = pack[:, :1], pack[:, 1:]
        loc = (W.t() @ v).squeeze(-1)
        if full_cov:
            cov = self.kernel(Xnew) - W.t() @ W
            if not noiseless:
                cov = cov + self.log_noise.exp() * torch.eye(Xnew.shape[0])
            return loc, cov
        var = (self.log_variance.exp() - (W * W).sum(0)).clamp(min=0)
        if not noiseless:
            var = var + self.log_noise.exp()
        return loc, var


def gp_model(input_dim: int = None, encoded_X: torch.Tensor = None, y: torch.Tensor = None, gp_iterations: int = 1):
    """A GP regression model trained on the encoded data (pyroved/utils/gp.py:5-29): Adam(lr=0.005) on the negative log
    marginal likelihood for `gp_iterations` steps.  (The reference evaluates the loss once and back-propagates it in
    every iteration, which only works for one iteration; here every step re-evaluates it.)"""
    print("Training GP model...")
    gpr = GPRegression(encoded_X, y)
    optimizer = torch.optim.Adam(gpr.parameters(), lr=0.005)
    for _ in range(gp_iterations):
        optimizer.zero_grad()
        gpr.loss().backward()
        optimizer.step()
    print("GP model trained.")
    return gpr
