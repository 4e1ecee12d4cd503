"""Prescribed sea-surface temperature: fme/core/ocean.py:95-222, fme/core/prescriber.py:54-117.

The slab ocean (ocean.py:64-92) is not on the path and raises ``NotImplementedError``.
"""
import dataclasses
from typing import Any, Dict, List, Mapping, Optional

import torch

TensorMapping = Mapping[str, torch.Tensor]
TensorDict = Dict[str, torch.Tensor]


def replace_on_mask(original: torch.Tensor, replacement: torch.Tensor, mask: torch.Tensor, mask_value: int):
    """fme/core/spatial_masking.py:11-30."""
    rounded_mask = torch.round(mask).to(int)
    return torch.where(rounded_mask == mask_value, replacement, original)


class Prescriber:
    """fme/core/prescriber.py:54-117."""

    def __init__(self, prescribed_name: str, mask_name: str, mask_value: int, interpolate: bool = False):
        if interpolate and mask_value != 1:
            raise ValueError(f"Interpolation requires mask_value to be 1, but it is set to {mask_value}.")
        self.prescribed_name = prescribed_name
        self.mask_name = mask_name
        self.mask_value = mask_value
        self.interpolate = interpolate

    def __call__(self, mask_data: TensorMapping, gen: TensorMapping, target: TensorMapping) -> TensorDict:
        for name, named in (("gen", gen), ("target", target)):
            if self.prescribed_name not in named:
                raise ValueError(f'Prescribed variable "{self.prescribed_name}" is missing from "{name}"')
        if self.interpolate:
            mask = mask_data[self.mask_name]
            output = mask * target[self.prescribed_name] + (1 - mask) * gen[self.prescribed_name]
        else:
            output = replace_on_mask(gen[self.prescribed_name], target[self.prescribed_name], mask_data[self.mask_name],
                                     self.mask_value)
        return {**gen, self.prescribed_name: output}


@dataclasses.dataclass
class OceanConfig:
    surface_temperature_name: str
    ocean_fraction_name: str
    interpolate: bool = False
    slab: Optional[Any] = None

    @classmethod
    def from_state(cls, state: Optional[Mapping[str, Any]]) -> Optional["OceanConfig"]:
        if state is None:
            return None
        unknown = set(state) - {f.name for f in dataclasses.fields(cls)}
        if unknown:
            raise ValueError(f"unknown ocean fields: {sorted(unknown)}")
        return cls(**state)

    def build(self, in_names: List[str], out_names: List[str], timestep=None) -> "Ocean":
        if self.slab is not None:
            raise NotImplementedError("the slab ocean is outside the accelerated hot path")
        if not (self.surface_temperature_name in in_names and self.surface_temperature_name in out_names):
            raise ValueError("To use a surface ocean model, the surface temperature must be present in_names and "
                             f"out_names, but {self.surface_temperature_name} is not.")
        return Ocean(self)

    @property
    def forcing_names(self) -> List[str]:
        return list({self.ocean_fraction_name, self.surface_temperature_name})


class Ocean:
    """Overwrite the generated SST with the prescribed one over ocean (ocean.py:167-215)."""

    def __init__(self, config: OceanConfig):
        self.surface_temperature_name = config.surface_temperature_name
        self.ocean_fraction_name = config.ocean_fraction_name
        self.prescriber = Prescriber(config.surface_temperature_name, config.ocean_fraction_name, 1, config.interpolate)
        self._forcing_names = config.forcing_names

    def __call__(self, input_data: TensorMapping, gen_data: TensorMapping, target_data: TensorMapping) -> TensorDict:
        next_step_temperature = target_data[self.surface_temperature_name]     # PrescribedSurfaceTemperature
        return self.prescriber(target_data, gen_data, {self.surface_temperature_name: next_step_temperature})

    @property
    def forcing_names(self) -> List[str]:
        return self._forcing_names
