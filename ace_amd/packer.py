"""Packer (fme/core/packer.py:13-52): names -> channel order."""

from typing import Dict, List

import torch


class DataShapesNotUniform(ValueError):
    """Indicates that a set of tensors do not all have the same shape."""


class Packer:
    def __init__(self, names: List[str]):
        self.names = names

    def pack(self, tensors: Dict[str, torch.Tensor], axis=0) -> torch.Tensor:
        shape = next(iter(tensors.values())).shape
        for name in tensors:
            if tensors[name].shape != shape:
                raise DataShapesNotUniform(
                    f'Cannot pack tensors of different shapes. Expected "{shape}" got "{tensors[name].shape}"'
                )
        return torch.cat([tensors[n].unsqueeze(axis) for n in self.names], dim=axis)

    def unpack(self, tensor: torch.Tensor, axis=0) -> Dict[str, torch.Tensor]:
        return {n: tensor.select(axis, index=i) for i, n in enumerate(self.names)}
