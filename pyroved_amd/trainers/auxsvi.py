"""
auxsvi.py — SVI trainer for variational models with auxiliary losses; host-side mirror of
pyroved/trainers/auxsvi.py:14-225 (same constructor, compute_loss / train / evaluate / step / save_running_weights /
average_weights / print_statistics, history, current_epoch).

What the reference delegates to Pyro — two SVI objects sharing one optimizer: SVI(model.model, guide, Adam,
TraceEnum_ELBO | Trace_ELBO) and SVI(model.model_aux, model.guide_aux, Adam, Trace_ELBO) (auxsvi.py:60-84) — is executed
by the HIP library through engine_ss.SSEngine: per compute_loss call the ELBO step (loss + gradients + Adam) and the
auxiliary step (loss + gradients + Adam).  As in the reference, the auxiliary step of an UNLABELED batch has no loss
and no gradient but still runs the optimizer over every parameter (pyro.module registers them all in model_aux, and
zero_grads leaves zero tensors): a momentum-only Adam step.

RNG contract: the trainer re-seeds in its constructor and draws the guide's noise on the global CPU generator in the
guide's order (classification: one (K, B, z_dim) draw for an unlabeled batch; regression: the label's (B, reg_dim) draw,
then (B, z_dim)), so the same seed yields the reference's CPU stream.
"""
from collections import OrderedDict
from copy import deepcopy as dc
from typing import Type, Optional, Union, Dict

import torch
import torch.nn as nn

from ..utils import set_deterministic_mode, average_weights
from .. import dist as pvdist


class auxSVItrainer:
    """
    SVI trainer for variational models with auxiliary losses (models.ssiVAE, models.ss_reg_iVAE).

    Args:
        model: initialized model
        task: "classification" (models.ssiVAE) or "regression" (models.ss_reg_iVAE)
        optimizer: None (Adam, lr 5e-4) or a dict of Adam arguments {"lr", "betas", "eps"}
        seed: enforces reproducibility

    Keyword Args:
        lr: learning rate (Default: 5e-4)
        device: device of the model (defaults to the model's)
        precision / fused: as in trainers.SVItrainer
    """
    def __init__(self, model: Type[nn.Module], task: str = "classification", optimizer=None, seed: int = 1,
                 **kwargs: Union[str, float]) -> None:
        set_deterministic_mode(seed)
        if task not in ["classification", "regression"]:
            raise ValueError("Choose between 'classification' and 'regression' tasks")
        self.task = task
        self.device = kwargs.get("device", model.device)
        adam = {"lr": kwargs.get("lr", 5e-4), "betas": (0.9, 0.999), "eps": 1e-8}
        if optimizer is not None:
            if not isinstance(optimizer, dict):
                raise TypeError("optimizer must be None or a dict of Adam arguments (Pyro optimizer objects "
                                "cannot be used: Pyro is not a dependency of this build)")
            adam.update(optimizer)
        self.group = kwargs.get("process_group", None)
        precision = kwargs.get("precision", "fp32")
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16' (got %r)" % (precision,))
        self.model = model
        if kwargs.get("engine") is not None:
            # test hook: a stand-in engine (tests drive the data-parallel host logic on CPU/gloo with it)
            self.engine = kwargs["engine"]
        else:
            self.engine = model.engine(lr=adam["lr"], betas=adam["betas"], eps=adam["eps"],
                                       fused=int(kwargs.get("fused", 3 if precision == "bf16" else 2)))
        if self.engine.task != task:
            raise ValueError("task=%r does not match the model's label network (%s)" % (task, self.engine.task))
        self.engine.lr, self.engine.betas, self.engine.adam_eps = float(adam["lr"]), tuple(adam["betas"]), float(adam["eps"])
        if hasattr(self.engine, "reset_optimizer"):
            self.engine.reset_optimizer()          # every trainer starts a fresh Adam (auxsvi.py: a new optim.Adam)
        self.history = {"training_loss": [], "test": []}
        self.current_epoch = 0
        self.running_weights = {}
        self.device_feed = bool(kwargs.get("device_feed", True))    # TensorDataset loaders are served from device copies
        self._feed_cache = {}
        if pvdist.world(self.group)[1] > 1:
            pvdist.sync_replicas(self.engine, self.group)

    def _reduced(self, loss):
        """Data parallel: ONE all-reduce(SUM) of [flat gradient | loss] (the loss rides in the first of the gradient
        buffer's trailing scalar slots); single process: nothing to do."""
        eng = self.engine
        if pvdist.world(self.group)[1] == 1:
            return loss
        eng.scalars[0] = loss
        pvdist.allreduce_sum_(eng.grad, self.group)
        return eng.scalars[0].clone()

    def compute_loss(self, xs: torch.Tensor, ys: Optional[torch.Tensor] = None, **kwargs: float) -> float:
        """Computes the basic (ELBO) and the auxiliary loss and takes the two optimizer steps (auxsvi.py:88-99)."""
        return float(self._compute_loss(xs, ys, **kwargs).item())

    def _compute_loss(self, xs: torch.Tensor, ys: Optional[torch.Tensor] = None, **kwargs: float) -> torch.Tensor:
        """compute_loss without the device->host read: the loss stays a device scalar (train() reads an epoch's losses
        back in one copy and adds them up in step order, as the reference's `epoch_loss += loss` does)."""
        eng, m = self.engine, self.model
        beta = kwargs.get("scale_factor", 1.)
        mult = kwargs.get("aux_loss_multiplier", 20)
        b = xs.shape[0]
        # the guide's noise for the GLOBAL batch (identical on every rank), then this rank's contiguous slice of
        # samples: every term of both objectives is a sum over samples, so shard gradients and losses add up
        eps_y = None
        enum = self.task == "classification" and ys is None
        if self.task == "classification":
            shape = (m.num_classes, b, m.z_dim) if ys is None else (b, m.z_dim)
        else:
            if ys is None:
                eps_y = torch.empty(b, m.reg_dim).normal_()
            shape = (b, m.z_dim)
        eps = torch.empty(shape).normal_()
        rank, world = pvdist.world(self.group)
        lo, hi = pvdist.shard_bounds(b, rank, world)
        dev = eng.device
        zero = torch.zeros((), device=dev)
        if hi > lo:
            xs = xs[lo:hi].to(dev, torch.float32).reshape(hi - lo, -1)
            ys = None if ys is None else ys[lo:hi].to(dev, torch.float32)
            eps = (eps[:, lo:hi] if enum else eps[lo:hi]).contiguous().to(dev)
            eps_y = None if eps_y is None else eps_y[lo:hi].to(dev)
            loss = eng.elbo_loss_and_grads(xs, eps, ys, eps_y, beta)
        else:                                    # more ranks than samples: contribute zeros
            eng.grad.zero_()
            loss = zero
        loss = self._reduced(loss)
        eng.adam_step()
        if ys is not None:
            if hi > lo:
                aux = eng.aux_loss_and_grads(xs, ys, mult)
            else:
                eng.grad.zero_()
                aux = zero
            loss = loss + self._reduced(aux)
        if eng.grads_live:                       # (no parameter has a .grad before the very first backward)
            eng.adam_step()
        return loss

    def _index_feed(self, loader):
        """(index loader, device tensors) for a plain TensorDataset DataLoader, else None (the loader is iterated as is)."""
        from torch.utils.data import DataLoader, TensorDataset
        ds = getattr(loader, "dataset", None)
        dev = getattr(self.engine, "device", None)
        if (not self.device_feed or dev is None or torch.device(dev).type != "cuda" or not isinstance(loader, DataLoader)
                or not isinstance(ds, TensorDataset) or loader.num_workers != 0 or loader.batch_sampler is None):
            return None
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in ds.tensors)
        hit = self._feed_cache.get(id(loader))
        if hit is None or hit[0] != key:
            hit = (key, [t.to(dev, torch.float32) for t in ds.tensors])
            self._feed_cache[id(loader)] = hit
        return DataLoader(range(len(ds)), batch_sampler=loader.batch_sampler), hit[1]

    def train(self, loader_unsup, loader_sup, **kwargs: float) -> float:
        """Train a single epoch (auxsvi.py:101-127)."""
        sup_batches = len(loader_sup)
        unsup_batches = len(loader_unsup)
        p = (sup_batches + unsup_batches) // sup_batches
        # device-resident data: the loaders are iterated over sample INDICES with their own batch samplers (same
        # iterator creation order and laziness, so the global CPU generator is consumed exactly as by the loaders
        # themselves) and the minibatches are gathered on the device instead of being collated and copied per step
        feed_u, feed_s = self._index_feed(loader_unsup), self._index_feed(loader_sup)
        it_sup = iter(feed_s[0] if feed_s else loader_sup)
        losses = []
        unsup_count = 0
        for i, item in enumerate(feed_u[0] if feed_u else loader_unsup):
            xs = feed_u[1][0].index_select(0, item.to(feed_u[1][0].device)) if feed_u else item[0]
            losses.append(self._compute_loss(xs, **kwargs).reshape(1))
            unsup_count += xs.shape[0]
            if i % p == 1:
                item = next(it_sup)
                if feed_s:
                    idx = item.to(feed_s[1][0].device)
                    xs, ys = feed_s[1][0].index_select(0, idx), feed_s[1][1].index_select(0, idx)
                else:
                    xs, ys = item
                _ = self._compute_loss(xs, ys, **kwargs)
        epoch_loss = 0.
        for v in (torch.cat(losses).cpu().tolist() if losses else []):    # one read per epoch, same addition order
            epoch_loss += v
        return epoch_loss / unsup_count

    def evaluate(self, loader_val) -> float:
        """Evaluates the model's current state on labeled test data (auxsvi.py:129-163)."""
        if self.task == "classification":
            return self.evaluate_cls(loader_val)
        return self.evaluate_reg(loader_val)

    def evaluate_cls(self, loader_val) -> float:
        correct, total = 0, 0
        for data, labels in loader_val:
            predicted = self.model.classifier(data)
            _, lab_idx = torch.max(labels.cpu(), 1)
            correct += (predicted == lab_idx).sum().item()
            total += data.size(0)
        return correct / total

    def evaluate_reg(self, loader_val) -> float:
        correct, total = 0, 0
        for data, gt in loader_val:
            predicted = self.model.regressor(data)
            correct += nn.functional.mse_loss(predicted, gt.cpu())
            total += 1
        return correct / total

    def step(self, loader_unsup, loader_sup, loader_val=None, **kwargs: float) -> None:
        """Single train (and evaluation, if any) step (auxsvi.py:165-194).  kwargs: scale_factor (KL scale, default 1),
        aux_loss_multiplier (default 20)."""
        train_loss = self.train(loader_unsup, loader_sup, **kwargs)
        self.history["training_loss"].append(train_loss)
        if loader_val is not None:
            self.history["test"].append(self.evaluate(loader_val))
        self.current_epoch += 1

    def save_running_weights(self, net: str) -> None:
        """Saves the running weights of the specified network, e.g. "encoder_y" (auxsvi.py:196-205)."""
        net = getattr(self.model, net)
        state_dict_ = OrderedDict()
        for k, v in net.state_dict().items():
            state_dict_[k] = dc(v).cpu()
        self.running_weights[self.current_epoch] = state_dict_

    def average_weights(self, net: str) -> Dict[int, Dict[str, torch.Tensor]]:
        """Updates the selected network with the average of the saved weights (auxsvi.py:207-213)."""
        net = getattr(self.model, net)
        net.load_state_dict(average_weights(self.running_weights))

    def print_statistics(self) -> None:
        """Prints training loss and test metric (if any) of the current epoch (auxsvi.py:215-225)."""
        e = self.current_epoch
        if len(self.history["test"]) > 0:
            if self.task == "classification":
                template = 'Epoch: {} Training loss: {:.4f}, Test accuracy: {:.4f}'
            else:
                template = 'Epoch: {} Training loss: {:.4f}, Test MSE: {:.4f}'
            print(template.format(e, self.history["training_loss"][-1], self.history["test"][-1]))
        else:
            template = 'Epoch: {} Training loss: {:.4f}'
            print(template.format(e, self.history["training_loss"][-1]))
