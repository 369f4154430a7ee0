"""DataLoader helpers with the call signatures of pyroved/utils/data.py:6-52.

Behaviour contract kept from the reference (it defines the data order parity depends on):
`TensorDataset(*tensors)`, batch_size 100 unless given, `shuffle=True` unless a `RandomSampler` is asked for,
a `torch.Generator(device)` handed to the loader only when `device=` is passed.  SVItrainer recognises loaders
built here (plain TensorDataset + standard samplers) and feeds them from device memory (trainers/svi.py).
"""
from typing import Iterator, Tuple

import torch
from torch.utils.data import DataLoader, RandomSampler, TensorDataset


def init_dataloader(*args: torch.Tensor, random_sampler: bool = False, shuffle: bool = True,
                    **kwargs: int) -> DataLoader:
    """DataLoader over the given tensors.  kwargs: batch_size (100), device (generator device)."""
    dev = kwargs.get("device")
    opts = dict(batch_size=kwargs.get("batch_size", 100),
                generator=torch.Generator(dev) if dev else None)
    ds = TensorDataset(*args)
    if random_sampler:
        opts["sampler"] = RandomSampler(ds)
    else:
        opts["shuffle"] = shuffle
    return DataLoader(ds, **opts)


def iter_batches(*args: torch.Tensor, batch_size: int = 100) -> Iterator[Tuple[torch.Tensor, ...]]:
    """The batches `init_dataloader(*args, shuffle=False, batch_size=...)` yields — same boundaries, same order, the last
    one partial — as views of the callers' tensors.  The inference loops (encode / decode / predict / classifier) walk
    their input in order, where a DataLoader over a TensorDataset indexes and collates SAMPLE BY SAMPLE on the host
    (0.3-1.5 s per 65 536 images of 28x28, ~50x the kernels' time)."""
    n = args[0].shape[0]
    if any(a.shape[0] != n for a in args):
        raise ValueError("Size mismatch between tensors")          # (TensorDataset's assertion)
    # iter(DataLoader) draws its base seed from the global CPU generator even when nothing is shuffled; an encode /
    # classifier call between two epochs therefore moves the stream the NEXT epoch's permutation and noise come from
    # (the reference trainers' recorded histories include that draw): consume exactly the same value here
    torch.empty((), dtype=torch.int64).random_()
    for lo in range(0, n, batch_size):
        yield tuple(a[lo:lo + batch_size] for a in args)


def init_ssvae_dataloaders(data_unsup: torch.Tensor, data_sup: Tuple[torch.Tensor], data_val: Tuple[torch.Tensor],
                           **kwargs: int) -> Tuple[DataLoader, DataLoader, DataLoader]:
    """(unlabeled, labeled, validation) loaders for the semi-supervised trainers (pyroved/utils/data.py:41-52).
    The reference passes `sampler=True` for the labeled set, a keyword its init_dataloader swallows in **kwargs —
    i.e. all three are plain shuffling loaders; kept so."""
    return (init_dataloader(data_unsup, **kwargs),
            init_dataloader(*data_sup, **kwargs),
            init_dataloader(*data_val, **kwargs))
