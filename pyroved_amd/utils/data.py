"""DataLoader helpers with the call signatures of pyroved/utils/data.py:6-52.

Behaviour contract kept from the reference (it defines the data order parity depends on):
`TensorDataset(*tensors)`, batch_size 100 unless given, `shuffle=True` unless a `RandomSampler` is asked for,
a `torch.Generator(device)` handed to the loader only when `device=` is passed.  SVItrainer recognises loaders
built here (plain TensorDataset + standard samplers) and feeds them from device memory (trainers/svi.py).
"""
from typing import Tuple

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


def init_ssvae_dataloaders(data_unsup: torch.Tensor, data_sup: Tuple[torch.Tensor], data_val: Tuple[torch.Tensor],
                           **kwargs: int) -> Tuple[DataLoader, DataLoader, DataLoader]:
    """(unlabeled, labeled, validation) loaders for the semi-supervised trainers (pyroved/utils/data.py:41-52).
    The reference passes `sampler=True` for the labeled set, a keyword its init_dataloader swallows in **kwargs —
    i.e. all three are plain shuffling loaders; kept so."""
    return (init_dataloader(data_unsup, **kwargs),
            init_dataloader(*data_sup, **kwargs),
            init_dataloader(*data_val, **kwargs))
