"""
svi.py — stochastic variational inference trainer; host-side mirror of
pyroved/trainers/svi.py:11-175 (same constructor, train / evaluate / step /
print_statistics, loss_history, current_epoch).

What the reference delegates to Pyro (`infer.SVI(model.model, guide, optim.Adam, Trace_ELBO)`)
is executed here by the HIP library: per minibatch one `pv_ivae_loss_and_grads` + one
`pv_adam_step` on the current stream (engine.py), with exactly one gradient all-reduce in
between when torch.distributed is initialised (dist.py).

RNG / data-order contract: the trainer re-seeds in its constructor, iterates
the caller's DataLoader (whose shuffling consumes the global CPU generator exactly as in the
reference), and draws `eps = torch.empty(B, z_dim).normal_()` on the global CPU generator once
per step — the stream a CPU run of the reference consumes — so that the same seed yields the
same numbers as the reference's CPU path.  `rng="device"` draws eps on the GPU instead.
"""
from typing import Type, Optional, Union

import torch

from ..utils import set_deterministic_mode
from .. import dist as pvdist


class SVItrainer:
    """
    SVI trainer for unsupervised and class-conditioned VED models consisting
    of one encoder and one decoder.

    Args:
        model: initialized model (pyroved_amd.models.iVAE)
        optimizer: None (Adam, lr 1e-3) or a dict of Adam arguments {"lr", "betas", "eps"} — the fused HIP route; a
            pyro.optim object (needs pyro-ppl) selects the generic Pyro route over model.model / model.guide
        loss: None / "Trace_ELBO" (the one-particle ELBO the reference defaults to); a pyro.infer ELBO object
            (needs pyro-ppl) selects the generic Pyro route
        enumerate_parallel: exact enumeration of the discrete latent of models.jiVAE (False, the reference's default:
            the class is drawn by the guide — vanilla jiVAE only, as in the reference)
        seed: enforces reproducibility

    Keyword Args:
        lr: learning rate (Default: 1e-3)
        device: device of the model (defaults to the model's)
        rng: "cpu" (default; bit-compatible with the reference's CPU stream) or "device"
        precision: "fp32" (default: fp32-class results everywhere, the parity mode) or "bf16" (mixed precision: the
            spatial decoder's hidden-layer matrix products take bf16 operands with fp32 accumulation; ELBO within
            1e-5 of fp32, gradients to ~1e-2; ~1.5x the throughput)
        fused: spatial-decoder kernel path, overrides `precision`: 2 fused persistent kernel with bf16
            split-precision matrix math (fp32-class results), 3 the same kernel with plain bf16 operands,
            1 fused kernel on the f32-input MFMA, 0 layer-by-layer kernels
        process_group: torch.distributed group for data-parallel training (default: WORLD if initialised)
        device_feed: keep TensorDataset loaders' data on the device and gather minibatches there, in the order the
            loader's own sampler produces (default True; same numbers as iterating the loader)
        mirror_evaluate_update: keep the reference's behaviour of stepping the optimizer inside
            evaluate() (svi.py:126-135 calls svi.step under no_grad) — default True
    """
    def __init__(self,
                 model: Type[torch.nn.Module],
                 optimizer=None,
                 loss=None,
                 enumerate_parallel: bool = False,
                 seed: int = 1,
                 **kwargs: Union[str, float]
                 ) -> None:
        set_deterministic_mode(seed)
        self.device = kwargs.get("device", model.device)
        is_joint = int(getattr(model, "discrete_dim", 0)) > 0
        # jiVAE with the reference's DEFAULT enumerate_parallel=False: Trace_ELBO on a class drawn by the guide, score-function
        # gradient for the class logits (svi.py:83-91; models/jivae.py:213-220).  Defined for the vanilla decoder only: with
        # invariances the reference's own model raises (z repeated K times cannot broadcast against the (B, K) drawn class,
        # models/jivae.py:181-189) — the same RuntimeError, at construction instead of at the first step
        self._sampled_class = is_joint and not enumerate_parallel
        if self._sampled_class and int(getattr(model, "coord", 0)) > 0:
            raise RuntimeError("jiVAE with invariances cannot be trained without enumeration (the reference's model cannot "
                               "broadcast its K-times repeated latent against the drawn class, models/jivae.py:181-189): "
                               "pass enumerate_parallel=True")
        if enumerate_parallel and not is_joint:
            raise ValueError("enumerate_parallel=True needs a model with a discrete latent (models.jiVAE)")
        self.model = model
        self.svi = None
        # (attributes every route leaves behind, so that code probing a trainer never meets a missing one)
        self.engine, self.group, self._hist, self._feed_cache = None, kwargs.get("process_group", None), None, None
        self.loss_history = {"training_loss": [], "test_loss": []}
        self.current_epoch = 0
        pyro_objects = (optimizer is not None and not isinstance(optimizer, dict)) or \
                       (loss is not None and loss != "Trace_ELBO")
        if pyro_objects:
            # Pyro optimizer / ELBO OBJECTS (the reference's signature, svi.py:66-67): the generic Pyro route —
            # pyro.infer.SVI over model.model / model.guide (models/_pyro_programs.py), the networks as differentiable
            # operators over the library's GEMMs.  Any objective Pyro can express; none of the fused kernels.
            try:
                import pyro.infer as infer
                import pyro.optim as poptim
                import pyro
            except ImportError as e:
                raise TypeError("optimizer / loss objects are Pyro objects and need pyro-ppl, which is not installed; "
                                "pass None or a dict of Adam arguments (the default Trace_ELBO objective runs in the "
                                "HIP library)") from e
            # this route is plain single-process Pyro: nothing synchronises replicas and the HIP precision modes do not apply
            if pvdist.world(self.group)[1] > 1:
                raise ValueError("Pyro optimizer / loss objects cannot be combined with data-parallel training: replicas "
                                 "would train unsynchronised (pass optimizer=None or a dict of Adam arguments)")
            bad = [k for k in ("precision", "fused", "rng", "device_feed", "mirror_evaluate_update") if k in kwargs]
            if bad:
                raise ValueError("%s only apply to the HIP training path, not to Pyro optimizer / loss objects" % bad)
            pyro.clear_param_store()
            opt = optimizer if optimizer is not None and not isinstance(optimizer, dict) else \
                poptim.Adam({"lr": kwargs.get("lr", 1e-3), **(optimizer or {})})
            if loss is None or loss == "Trace_ELBO":
                loss = (infer.TraceEnum_ELBO(max_plate_nesting=1, strict_enumeration_warning=False)
                        if enumerate_parallel else infer.Trace_ELBO())
            guide = infer.config_enumerate(model.guide, "parallel", expand=True) if enumerate_parallel else model.guide
            self.svi = infer.SVI(model.model, guide, opt, loss=loss)
            self.loss_history = {"training_loss": [], "test_loss": []}
            self.current_epoch = 0
            self._sampled_class = False
            return
        adam = {"lr": kwargs.get("lr", 1e-3), "betas": (0.9, 0.999), "eps": 1e-8}
        if optimizer is not None:
            adam.update(optimizer)
        self.rng = kwargs.get("rng", "cpu")
        self.group = kwargs.get("process_group", None)
        self.mirror_evaluate_update = bool(kwargs.get("mirror_evaluate_update", True))
        self.device_feed = bool(kwargs.get("device_feed", True))
        self._feed_cache = None
        if kwargs.get("engine") is not None:
            # test hook: a stand-in engine (tests drive the data-parallel host logic on CPU/gloo with it)
            self.engine = kwargs["engine"]
        else:
            precision = kwargs.get("precision", "fp32")
            if precision not in ("fp32", "bf16"):
                raise ValueError("precision must be 'fp32' or 'bf16' (got %r)" % (precision,))
            self.engine = model.engine(lr=adam["lr"], betas=adam["betas"], eps=adam["eps"],
                                       fused=int(kwargs.get("fused", 3 if precision == "bf16" else 2)))
        self.engine.lr, self.engine.betas, self.engine.adam_eps = float(adam["lr"]), tuple(adam["betas"]), float(adam["eps"])
        if hasattr(self.engine, "reset_optimizer"):
            self.engine.reset_optimizer()          # every trainer starts a fresh Adam (svi.py:75-81)
        self.loss_history = {"training_loss": [], "test_loss": []}
        self.current_epoch = 0
        self._hist = None
        rank, world = pvdist.world(self.group)
        # data parallel over the nccl (= RCCL) backend: the step's one collective is enqueued by the LIBRARY on the compute
        # stream (dist.NativeComm, ABI v16 pv_ivae_dp_step / pv_ved_dp_step) — `collective="torch"` keeps torch.distributed's
        # all_reduce (its own stream, two event hand-offs per step); gloo groups (CPU tests) always do.
        self._comm = None
        which = kwargs.get("collective", "native")
        if which not in ("native", "torch"):
            raise ValueError("collective must be 'native' or 'torch' (got %r)" % (which,))
        if world > 1:
            pvdist.sync_replicas(self.engine, self.group)
            if (which == "native" and kwargs.get("engine") is None and pvdist.native_available(self.group)
                    and getattr(self.engine, "supports_dp_step", False)):
                self._comm = pvdist.native_comm(self.engine.device, self.group)

    # ------------------------------------------------------------------ one minibatch
    def _draw_eps(self, b: int) -> torch.Tensor:
        z_dim = self.model.z_dim
        if self.rng == "cpu":
            return torch.empty(b, z_dim).normal_()
        return torch.empty(b, z_dim, device=self.engine.device).normal_()

    def _svi_step(self, i: int, x: torch.Tensor, y: Optional[torch.Tensor], train: bool, eps=None, shard=None,
                  **kwargs) -> None:
        """SVI.step on one (global) minibatch; the loss lands in slot i of the device history.
        x / y / eps may already live on the device (device feed, _epoch_device_feed).
        shard = (lo, hi, b): x / y / eps are already THIS rank's rows [lo, hi) of a global minibatch of b samples
        (the data-parallel device feed gathers nothing else)."""
        eng = self.engine
        beta = kwargs.get("scale_factor", 1.)          # jiVAE: scalar or [continuous, discrete] (jivae.py:161-165)
        if torch.is_tensor(beta):
            beta = beta.tolist()
        beta = [float(v) for v in beta] if isinstance(beta, (list, tuple)) else float(beta)
        rank, world = pvdist.world(self.group)
        if shard is not None:
            lo, hi, b = shard
            x0 = 0                                    # offset of row `lo` inside x / y / eps
        else:
            b = x.shape[0]
            if eps is None:
                eps = self._draw_eps(b)               # global batch: identical on every rank
            lo, hi = pvdist.shard_bounds(b, rank, world)
            x0 = lo
        dev = eng.device
        if self._hist is None or self._hist.shape[0] <= i:
            new = torch.zeros(max(64, 2 * (i + 1)), 4, device=dev, dtype=torch.float32)
            if self._hist is not None:
                new[:self._hist.shape[0]].copy_(self._hist)
            self._hist = new
        direct = world == 1 and getattr(eng, "supports_scalars_out", False)   # single process: the loss lands in the history slot
        extra = {}
        if self._sampled_class:
            # the guide's second draw (after eps): y_b ~ OneHotCategorical(alpha_b), alpha = the encoder's class
            # probabilities for the GLOBAL batch (every rank computes them, so every rank draws the same classes).
            # Cost of this mode, by design: one extra encoder pass and — with rng="cpu", the default — one host sync per
            # step (alpha.cpu()), because the class must come out of the global CPU generator right after eps for the
            # trainer to reproduce the reference's stream (fixtures jsivae_*); the device feed is off for the same reason.
            # rng="device" draws on the GPU without the sync (a different stream, like the reference on a CUDA device).
            xg = x.to(dev, torch.float32)
            alpha = eng.encode(xg)[2]
            if self.rng == "cpu":
                y1h = torch.distributions.OneHotCategorical(probs=alpha.cpu()).sample().to(dev)
            else:
                y1h = torch.distributions.OneHotCategorical(probs=alpha).sample()
            extra["class_onehot"] = y1h[x0:x0 + hi - lo].contiguous()
        if hi > lo:
            xs = x[x0:x0 + hi - lo].to(dev, torch.float32)
            es = eps[x0:x0 + hi - lo].to(dev, torch.float32)
            ys = None if y is None else y[x0:x0 + hi - lo].to(dev, torch.float32)
            one_call = (direct and train and getattr(eng, "supports_step", False)
                        and not (getattr(eng, "ext_enc", False) or getattr(eng, "ext_dec", False)))
            if one_call:                              # loss, gradients and Adam in one library call (pv_ivae_step)
                eng.loss_and_grads(xs, es, beta, ys, scalars_out=self._hist[i], step=True, **extra)
                return
            if (self._comm is not None and train and world > 1
                    and not (getattr(eng, "ext_enc", False) or getattr(eng, "ext_dec", False) or getattr(eng, "ext_y", False))):
                # the data-parallel SVI.step as ONE library call on this stream: shard gradients -> ncclAllReduce -> Adam + history
                eng.loss_and_grads(xs, es, beta, ys, step=True, comm=self._comm, hist_out=self._hist[i], **extra)
                return
            if direct:
                eng.loss_and_grads(xs, es, beta, ys, want_grads=train, scalars_out=self._hist[i], **extra)
            else:
                eng.loss_and_grads(xs, es, beta, ys, want_grads=train, **extra)
        else:                                          # more ranks than samples: contribute zeros
            eng.grad.zero_()
        if world > 1:
            reduce_ = self._comm.allreduce_sum_ if self._comm is not None else (lambda t_: pvdist.allreduce_sum_(t_, self.group))
            if not train:                              # only the scalars need reducing
                reduce_(eng.scalars)
            else:
                reduce_(eng.grad)
                for g_ in (eng.extra_grads() if hasattr(eng, "extra_grads") else []):
                    reduce_(g_)                        # a user-defined encoder's gradients
        step_opt = train or (self.mirror_evaluate_update and eng.grads_live)
        if step_opt and not direct and hasattr(eng, "adam_step_hist"):
            eng.adam_step_hist(self._hist[i])          # Adam + the reduced loss into the history, one launch
            return
        if step_opt:
            eng.adam_step()
        if not direct:
            self._hist[i].copy_(eng.scalars)

    # ---- the epoch's minibatch order and noise, with the global CPU generator consumed exactly as the reference's
    # `for data in loader: svi.step(...)` consumes it, but without per-batch Python work
    def _epoch_batches(self, loader, n: int):
        """Index batches of one pass over `loader`.  Standard samplers are reproduced directly: iter(DataLoader) draws
        the base seed (torch/utils/data/dataloader.py: _BaseDataLoaderIter.__init__), RandomSampler.__iter__ draws its
        seed and permutes on a private generator (torch/utils/data/sampler.py) — the permutation itself does not touch
        the global stream, so the NEXT epoch's is computed ahead on a background thread from the seed the global
        generator will produce if nobody draws from it in between, and is used only if the seed really drawn matches
        (tests/test_host_cpu.py checks order and generator state against a real DataLoader pass).  Anything else
        (custom samplers / generators / replacement) iterates a DataLoader over the indices with the caller's sampler."""
        from torch.utils.data import DataLoader, BatchSampler, RandomSampler, SequentialSampler
        bs = loader.batch_sampler
        smp = getattr(bs, "sampler", None)
        std = (type(bs) is BatchSampler and loader.generator is None and len(smp) == n if smp is not None else False)
        if std and type(smp) is SequentialSampler:
            torch.empty((), dtype=torch.int64).random_()                    # iter(DataLoader): base seed
            order = torch.arange(n)
        elif (std and type(smp) is RandomSampler and not smp.replacement and smp.generator is None
              and smp._num_samples is None):
            torch.empty((), dtype=torch.int64).random_()                    # iter(DataLoader): base seed
            seed = int(torch.empty((), dtype=torch.int64).random_().item())  # RandomSampler.__iter__
            order = self._perm_for(seed, n)
            self._rand_n = n                                                # (what _prefetch_perm prepares for)
        else:
            idx_loader = DataLoader(range(n), batch_sampler=bs)             # same sampler object
            return [b for b in idx_loader]
        batches = list(order.split(bs.batch_size))
        if bs.drop_last and batches and len(batches[-1]) < bs.batch_size:
            batches.pop()
        return batches

    def _perm_for(self, seed: int, n: int) -> torch.Tensor:
        job = getattr(self, "_perm_job", None)
        self._perm_job = None
        if job is not None:
            job[0].join()
            if job[1] == (seed, n) and job[2]:
                return job[2][0]
        g = torch.Generator()
        g.manual_seed(seed)
        return torch.randperm(n, generator=g)

    def _prefetch_perm(self, n: int) -> None:
        """Starts the next epoch's permutation on a thread, assuming the next two draws from the global CPU generator
        are that epoch's (base seed, sampler seed); the generator itself is left untouched."""
        import threading
        probe = torch.Generator()
        probe.set_state(torch.get_rng_state())
        torch.empty((), dtype=torch.int64).random_(generator=probe)
        seed = int(torch.empty((), dtype=torch.int64).random_(generator=probe).item())
        out = []

        def work():
            g = torch.Generator()
            g.manual_seed(seed)
            out.append(torch.randperm(n, generator=g))
        t = threading.Thread(target=work, daemon=True)
        t.start()
        self._perm_job = (t, (seed, n), out)

    def _draw_eps_epoch(self, sizes) -> torch.Tensor:
        """Every step's eps, drawn on the global CPU generator in step order, as ONE (sum(sizes), z_dim) tensor.  CPU
        normal_ fills 16 values at a time and re-draws the last 16 of a tensor whose size is not a multiple of 16, so a
        run of batches with numel % 16 == 0 is one draw of their total; the others are drawn one by one."""
        z = self.model.z_dim
        if self.rng != "cpu":
            return torch.empty(sum(sizes), z, device=self.engine.device).normal_()
        out, i = [], 0
        while i < len(sizes):
            j = i
            while j < len(sizes) and (sizes[j] * z) % 16 == 0 and sizes[j] * z >= 16:
                j += 1
            if j > i:
                out.append(torch.empty(sum(sizes[i:j]), z).normal_())
                i = j
            else:
                out.append(torch.empty(sizes[i], z).normal_())
                i += 1
        return out[0] if len(out) == 1 else torch.cat(out)

    # ------------------------------------------------------------------ data feed
    def _device_dataset(self, tensors):
        """Device-resident copy of the loader's TensorDataset, cached until the tensors change."""
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in tensors)
        if self._feed_cache is None or self._feed_cache[0] != key:
            dev = self.engine.device
            self._feed_cache = (key, [t.to(dev, torch.float32) for t in tensors])
        return self._feed_cache[1]

    def _epoch_device_feed(self, loader, train: bool, **kwargs) -> Optional[int]:
        """One epoch without per-batch host work on the data: the dataset lives on the device, the
        minibatch ORDER is still produced by the caller's own sampler — iterated through a DataLoader over the
        indices, so the global CPU generator is consumed exactly as `for data in loader` consumes it (one base-seed
        draw per iterator, one sampler-seed draw per epoch) — and every step's eps is drawn on the CPU generator in
        step order (same stream as the per-batch path), then shipped in one copy.  Returns the number of steps, or
        None when the loader is not a plain TensorDataset loader (the caller falls back to iterating it)."""
        from torch.utils.data import DataLoader, TensorDataset
        ds = getattr(loader, "dataset", None)
        if (self._sampled_class or not self.device_feed or not isinstance(loader, DataLoader) or not isinstance(ds, TensorDataset)
                or loader.num_workers != 0 or loader.batch_sampler is None or len(ds.tensors) not in (1, 2)
                or getattr(self.engine, "device", None) is None or self.rng != "cpu"):
            return None
        batches = self._epoch_batches(loader, len(ds))         # LongTensors of sample indices, the loader's order
        if not batches:
            return 0
        eps = self._draw_eps_epoch([len(b) for b in batches])
        if getattr(self, "_rand_n", None) and self.rng == "cpu":
            self._prefetch_perm(self._rand_n)                  # the next shuffled epoch's order, off the critical path
        dev = self.engine.device
        data = self._device_dataset(ds.tensors)
        sizes = [len(b) for b in batches]
        idx_dev = torch.cat(batches).to(dev)
        eps_dev = eps.to(dev, torch.float32)
        # minibatches are gathered a chunk of steps at a time (one index_select per ~256 MB of samples instead of one
        # small gather kernel in front of every step); a step then takes a contiguous view
        per_sample = sum(t[0].numel() for t in data) * 4
        chunk = max(1, int((256 << 20) // max(per_sample * max(sizes), 1)))
        rank, world = pvdist.world(self.group)
        off, n = 0, 0
        while n < len(sizes):
            m = min(chunk, len(sizes) - n)
            rows = sum(sizes[n:n + m])
            if world > 1:
                # data parallel: this rank gathers ONLY its rows [lo, hi) of every global minibatch of the chunk (the
                # order is the shared permutation, identical on every rank): 1/world of the gather and of its memory
                bounds, parts, r0 = [], [], 0
                for k in range(m):
                    bsz = sizes[n + k]
                    lo, hi = pvdist.shard_bounds(bsz, rank, world)
                    bounds.append((lo, hi, bsz, r0))
                    parts.append(idx_dev[off + r0 + lo:off + r0 + hi])
                    r0 += bsz
                idx = torch.cat(parts)
                xs = data[0].index_select(0, idx)
                ys = data[1].index_select(0, idx) if len(data) > 1 else None
                c0 = 0
                for k, (lo, hi, bsz, r0) in enumerate(bounds):
                    w_ = hi - lo
                    self._svi_step(n + k, xs[c0:c0 + w_], None if ys is None else ys[c0:c0 + w_], train,
                                   eps=eps_dev[off + r0 + lo:off + r0 + hi], shard=(lo, hi, bsz), **kwargs)
                    c0 += w_
            else:
                idx = idx_dev[off:off + rows]
                xs = data[0].index_select(0, idx)
                ys = data[1].index_select(0, idx) if len(data) > 1 else None
                r0 = 0
                for k in range(m):
                    bsz = sizes[n + k]
                    self._svi_step(n + k, xs[r0:r0 + bsz], None if ys is None else ys[r0:r0 + bsz], train,
                                   eps=eps_dev[off + r0:off + r0 + bsz], **kwargs)
                    r0 += bsz
            off += rows
            n += m
        return len(sizes)

    def _epoch_pyro(self, loader, train: bool, **kwargs) -> float:
        """The reference's own loop (svi.py:95-137) around pyro.infer.SVI: used when the trainer was given Pyro objects."""
        epoch_loss = 0.
        ctx = torch.enable_grad() if train else torch.no_grad()
        with ctx:
            for data in loader:
                args = [d.to(self.device, torch.float32) for d in data]
                epoch_loss += self.svi.step(*args, **kwargs)
        return epoch_loss / len(loader.dataset)

    def _epoch(self, loader, train: bool, **kwargs) -> float:
        if self.svi is not None:
            return self._epoch_pyro(loader, train, **kwargs)
        n = self._epoch_device_feed(loader, train, **kwargs)
        if n is None:
            n = 0
            for data in loader:
                if len(data) == 1:  # VAE mode
                    self._svi_step(n, data[0], None, train, **kwargs)
                else:  # VED or cVAE mode
                    self._svi_step(n, data[0], data[1], train, **kwargs)
                n += 1
        # one device->host read per epoch; python-float accumulation like the reference's `epoch_loss += loss`
        losses = self._hist[:n, 0].cpu().tolist() if n else []
        epoch_loss = 0.
        for v in losses:
            epoch_loss += v
        return epoch_loss / len(loader.dataset)

    # ------------------------------------------------------------------ reference API
    def train(self, train_loader: Type[torch.utils.data.DataLoader], **kwargs: float) -> float:
        """Trains a single epoch; returns the summed loss divided by the dataset size."""
        return self._epoch(train_loader, True, **kwargs)

    def evaluate(self, test_loader: Type[torch.utils.data.DataLoader], **kwargs: float) -> float:
        """Evaluates the current model state on a single epoch (no gradients)."""
        return self._epoch(test_loader, False, **kwargs)

    def step(self,
             train_loader: Type[torch.utils.data.DataLoader],
             test_loader: Optional[Type[torch.utils.data.DataLoader]] = None,
             **kwargs: float) -> None:
        """Single training and (optionally) evaluation step.

        Keyword Args:
            scale_factor: scale factor for the KL terms (default 1)
        """
        train_loss = self.train(train_loader, **kwargs)
        self.loss_history["training_loss"].append(train_loss)
        if test_loader is not None:
            test_loss = self.evaluate(test_loader, **kwargs)
            self.loss_history["test_loss"].append(test_loss)
        self.current_epoch += 1

    def print_statistics(self) -> None:
        """Prints training and test (if any) losses for current epoch."""
        e = self.current_epoch
        if len(self.loss_history["test_loss"]) > 0:
            template = 'Epoch: {} Training loss: {:.4f}, Test loss: {:.4f}'
            print(template.format(e, self.loss_history["training_loss"][-1],
                                  self.loss_history["test_loss"][-1]))
        else:
            template = 'Epoch: {} Training loss: {:.4f}'
            print(template.format(e, self.loss_history["training_loss"][-1]))
