"""Name <-> channel-slot bookkeeping of the step (the role of fme/core/packer.py:13-52): ``names`` fixes the channel order of
the network input / output; packing stacks the per-name fields along a new axis, unpacking hands back per-name views."""

from typing import Dict, List, Mapping

import torch


class DataShapesNotUniform(ValueError):
    """Raised when the fields handed to ``Packer.pack`` do not share one shape."""


class Packer:
    def __init__(self, names: List[str]):
        self.names = names

    def pack(self, tensors: Mapping[str, torch.Tensor], axis: int = 0) -> torch.Tensor:
        shapes = {tuple(t.shape) for t in tensors.values()}
        if len(shapes) > 1:
            first = next(iter(tensors.values())).shape
            odd = next(t.shape for t in tensors.values() if t.shape != first)
            raise DataShapesNotUniform(f'Cannot pack tensors of different shapes. Expected "{first}" got "{odd}"')
        return torch.stack([tensors[name] for name in self.names], dim=axis)

    def unpack(self, tensor: torch.Tensor, axis: int = 0) -> Dict[str, torch.Tensor]:
        return dict(zip(self.names, tensor.unbind(dim=axis)))
