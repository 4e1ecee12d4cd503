"""Sea-surface temperature from an ocean model: fme/core/ocean.py, fme/core/prescriber.py:54-117 -
the prescribed SST (ocean.py:48-61) and the slab ocean (ocean.py:14-29, 64-92, 233-254: mixed-layer temperature tendency from the
generated net surface energy flux, the prescribed Q-flux and mixed-layer depth).  Torch ops on the device the state lives on; the
prescribed form also exists fused into csrc/physics.hip (ace_amd/physics.py), the slab form runs as these ops.
"""
import dataclasses
import datetime
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


DENSITY_OF_WATER = 1000.0          # kg/m^3   (fme/core/constants.py:6)
SPECIFIC_HEAT_OF_WATER = 4000.0    # J/kg/K   (fme/core/constants.py:5)


def mixed_layer_temperature_tendency(f_net, q_flux, depth, density=DENSITY_OF_WATER, specific_heat=SPECIFIC_HEAT_OF_WATER):
    """ocean.py:233-254: K/s of a mixed layer of `depth` m under f_net + q_flux W/m^2."""
    return (f_net + q_flux) / (density * depth * specific_heat)


@dataclasses.dataclass(frozen=True)
class SlabOceanConfig:
    """ocean.py:14-29."""
    mixed_layer_depth_name: str
    q_flux_name: str

    @property
    def names(self) -> List[str]:
        return [self.mixed_layer_depth_name, self.q_flux_name]


@dataclasses.dataclass
class OceanConfig:
    surface_temperature_name: str
    ocean_fraction_name: str
    interpolate: bool = False
    slab: Optional[Any] = None

    def __post_init__(self):
        if isinstance(self.slab, Mapping):
            unknown = set(self.slab) - {"mixed_layer_depth_name", "q_flux_name"}
            if unknown:
                raise ValueError(f"unknown slab ocean fields: {sorted(unknown)}")
            self.slab = SlabOceanConfig(**self.slab)
        if self.slab is not None and not isinstance(self.slab, SlabOceanConfig):
            raise TypeError("OceanConfig.slab must be a SlabOceanConfig or its state")

    @property
    def is_slab(self) -> bool:
        return self.slab is not None

    @classmethod
    def from_state(cls, state: Optional[Mapping[str, Any]]) -> Optional["OceanConfig"]:
        if state is None:
            return None
        unknown = set(state) - {f.name for f in dataclasses.fields(cls)}
        if unknown:
            raise ValueError(f"unknown ocean fields: {sorted(unknown)}")
        return cls(**state)

    def build(self, in_names: List[str], out_names: List[str], timestep=None) -> "Ocean":
        if not (self.surface_temperature_name in in_names and self.surface_temperature_name in out_names):
            raise ValueError("To use a surface ocean model, the surface temperature must be present in_names and "
                             f"out_names, but {self.surface_temperature_name} is not.")
        if self.slab is not None and not isinstance(timestep, datetime.timedelta):
            raise ValueError("the slab ocean needs the dataset's timestep (a datetime.timedelta)")
        return Ocean(self, timestep)

    @property
    def forcing_names(self) -> List[str]:
        names = [self.ocean_fraction_name]
        if self.slab is None:
            names.append(self.surface_temperature_name)
        else:
            names.extend(self.slab.names)
        return list(set(names))


class Ocean:
    """Overwrite the generated SST over ocean with the one an ocean model predicts (ocean.py:167-215): the prescribed next-step
    SST, or the slab ocean's mixed-layer update of the input SST."""

    def __init__(self, config: OceanConfig, timestep: Optional[datetime.timedelta] = None):
        self.surface_temperature_name = config.surface_temperature_name
        self.ocean_fraction_name = config.ocean_fraction_name
        self.prescriber = Prescriber(config.surface_temperature_name, config.ocean_fraction_name, 1, config.interpolate)
        self._forcing_names = config.forcing_names
        self.slab: Optional[SlabOceanConfig] = config.slab
        self._dt_seconds = timestep.total_seconds() if (config.slab is not None and timestep is not None) else None

    @property
    def is_slab(self) -> bool:
        return self.slab is not None

    def __call__(self, input_data: TensorMapping, gen_data: TensorMapping, target_data: TensorMapping) -> TensorDict:
        if self.slab is None:
            next_step_temperature = target_data[self.surface_temperature_name]     # PrescribedSurfaceTemperature, ocean.py:48-61
        else:                                                                      # SlabOceanSurfaceTemperature, ocean.py:64-92
            from .atmosphere import AtmosphereData
            tendency = mixed_layer_temperature_tendency(AtmosphereData(gen_data).net_surface_energy_flux_without_frozen_precip,
                                                        target_data[self.slab.q_flux_name],
                                                        target_data[self.slab.mixed_layer_depth_name])
            next_step_temperature = input_data[self.surface_temperature_name] + tendency * self._dt_seconds
        return self.prescriber(target_data, gen_data, {self.surface_temperature_name: next_step_temperature})

    @property
    def forcing_names(self) -> List[str]:
        return self._forcing_names
