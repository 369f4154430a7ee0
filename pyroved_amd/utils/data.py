"""DataLoader helper mirroring pyroved/utils/data.py:6-38 (pure torch.utils.data plumbing)."""
from typing import Type

import torch


def init_dataloader(*args: torch.Tensor,
                    random_sampler: bool = False,
                    shuffle: bool = True,
                    **kwargs: int
                    ) -> Type[torch.utils.data.DataLoader]:
    """Returns an initialized PyTorch dataloader over the given tensors
    (batch_size defaults to 100, shuffling on)."""
    device_ = kwargs.get("device")
    generator_ = torch.Generator(device_) if device_ else None
    batch_size = kwargs.get("batch_size", 100)
    tensor_set = torch.utils.data.dataset.TensorDataset(*args)
    if random_sampler:
        sampler = torch.utils.data.RandomSampler(tensor_set)
        return torch.utils.data.DataLoader(
            dataset=tensor_set, batch_size=batch_size, sampler=sampler, generator=generator_)
    return torch.utils.data.DataLoader(
        dataset=tensor_set, batch_size=batch_size, shuffle=shuffle, generator=generator_)


def init_ssvae_dataloaders(data_unsup: torch.Tensor, data_sup, data_val, **kwargs: int):
    """Dataloaders for the semi-supervised models (pyroved/utils/data.py:41-52): unlabeled, labeled, validation."""
    loader_unsup = init_dataloader(data_unsup, **kwargs)
    loader_sup = init_dataloader(*data_sup, sampler=True, **kwargs)
    loader_val = init_dataloader(*data_val, **kwargs)
    return loader_unsup, loader_sup, loader_val
