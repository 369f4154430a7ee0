"""Decoder likelihood selection mirroring pyroved/utils/prob.py:5-37."""
import torch.distributions as td


class _Sampler:
    """Callable like the reference's lambdas (x -> distribution object); `.name` /
    `.decoder_sig` tell the HIP path which likelihood kernel (enum pv_lik) to run."""
    def __init__(self, name, decoder_sig):
        self.name = name
        self.decoder_sig = float(decoder_sig)

    def __call__(self, x):
        if self.name == "bernoulli":
            return td.Bernoulli(x, validate_args=False)
        if self.name == "continuous_bernoulli":
            return td.ContinuousBernoulli(x)
        return td.Normal(x, self.decoder_sig)


def get_sampler(sampler: str, **kwargs: float):
    """'bernoulli', 'continuous_bernoulli' or 'gaussian' (decoder_sig kwarg, default 0.5)."""
    names = ["bernoulli", "continuous_bernoulli", "gaussian"]
    if sampler not in names:
        raise KeyError("Select between the following decoder samplers: {}".format(names))
    return _Sampler(sampler, kwargs.get("decoder_sig", 0.5))
