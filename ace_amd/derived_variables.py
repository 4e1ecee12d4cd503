"""Derived variables of the inference outputs (fme/core/derived_variables.py:14-236), computed on the device the series
lives on from ``AtmosphereData`` (ace_amd/atmosphere.py): same registry names, order and skip rule (a variable whose inputs are
missing is silently not computed; an existing name is never overwritten).  ``AtmosphericDeriveFn`` is what
``HybridSigmaPressureCoordinate.build_derive_function`` returns in the reference (fme/core/coordinates.py:51-72, 194-199)."""
import datetime
from typing import Callable, Dict, Mapping, MutableMapping, Optional

import torch

from .atmosphere import AtmosphereData

TensorDict = Dict[str, torch.Tensor]
DerivedVariableFunc = Callable[[AtmosphereData, datetime.timedelta], torch.Tensor]

_DERIVED_VARIABLE_REGISTRY: MutableMapping[str, DerivedVariableFunc] = {}


def register(func: DerivedVariableFunc) -> DerivedVariableFunc:
    label = func.__name__
    if label in _DERIVED_VARIABLE_REGISTRY:
        raise ValueError(f"Function {label} has already been added to registry.")
    _DERIVED_VARIABLE_REGISTRY[label] = func
    return func


def get_derived_variable_names():
    return list(_DERIVED_VARIABLE_REGISTRY)


@register
def surface_pressure_due_to_dry_air(data, timestep):
    return data.surface_pressure_due_to_dry_air


@register
def surface_pressure_due_to_dry_air_absolute_tendency(data, timestep):
    ps_dry = data.surface_pressure_due_to_dry_air
    out = torch.zeros_like(ps_dry)
    out[:, 1:] = torch.diff(ps_dry, n=1, dim=1).abs()
    return out


@register
def total_water_path(data, timestep):
    return data.total_water_path


@register
def total_water_path_budget_residual(data, timestep):
    twp = data.total_water_path
    tendency = (twp[:, 1:] - twp[:, :-1]) / (timestep.total_seconds())
    out = torch.zeros_like(twp)          # no budget residual on the initial step
    out[:, 1:] = tendency - (data.evaporation_rate[:, 1:] - data.precipitation_rate[:, 1:]
                             + data.tendency_of_total_water_path_due_to_advection[:, 1:])
    return out


@register
def net_energy_flux_toa_into_atmosphere(data, timestep):
    return data.net_top_of_atmosphere_energy_flux


@register
def net_energy_flux_sfc_into_atmosphere(data, timestep):
    return -data.net_surface_energy_flux   # the property is positive into the surface


@register
def net_energy_flux_into_atmospheric_column(data, timestep):
    return data.net_energy_flux_into_atmosphere


@register
def total_energy_ace2_path(data, timestep):
    return data.total_energy_ace2_path


@register
def total_energy_ace2_path_tendency(data, timestep):
    mse = total_energy_ace2_path(data, timestep)
    out = torch.zeros_like(mse)
    out[:, 1:] = torch.diff(mse, n=1, dim=1) / timestep.total_seconds()
    return out


@register
def implied_tendency_of_total_energy_ace2_path_due_to_advection(data, timestep):
    return total_energy_ace2_path_tendency(data, timestep) - data.net_energy_flux_into_atmosphere


@register
def windspeed_at_10m(data, timestep):
    return data.windspeed_at_10m


def _compute_derived_variable(data: TensorDict, vertical_coordinate, timestep: datetime.timedelta, label: str,
                              func: DerivedVariableFunc, forcing_data: Optional[Mapping[str, torch.Tensor]] = None) -> TensorDict:
    """derived_variables.py:170-218."""
    if label in data:
        raise ValueError(f"Variable {label} already exists. It is not permitted "
                         "to overwrite existing variables with derived variables.")
    new_data = data.copy()
    if forcing_data is not None:
        for key, value in forcing_data.items():
            if key not in data:
                data[key] = value
    try:
        output = func(AtmosphereData(data, vertical_coordinate), timestep)
    except KeyError:
        return new_data
    new_data[label] = output
    return new_data


def compute_derived_quantities(data: TensorDict, vertical_coordinate, timestep: datetime.timedelta,
                               forcing_data: Optional[Mapping[str, torch.Tensor]] = None) -> TensorDict:
    """derived_variables.py:221-236: every registered variable, in registration order."""
    for label, func in _DERIVED_VARIABLE_REGISTRY.items():
        data = _compute_derived_variable(data, vertical_coordinate, timestep, label, func, forcing_data=forcing_data)
    return data


class AtmosphericDeriveFn:
    """coordinates.py:51-72.  With ``vertical_coordinate`` None the variables that need a vertical integral raise
    ``ValueError`` (only missing *names* are skipped), exactly as ``AtmosphereData`` does in the reference."""

    def __init__(self, vertical_coordinate, timestep: datetime.timedelta):
        self.vertical_coordinate = vertical_coordinate
        self.timestep = timestep

    def __call__(self, data: Mapping[str, torch.Tensor], forcing_data: Mapping[str, torch.Tensor]) -> TensorDict:
        vc = self.vertical_coordinate
        if vc is not None:
            vc = vc.to(next(iter(data.values())).device)
        return compute_derived_quantities(dict(data), vertical_coordinate=vc, timestep=self.timestep,
                                          forcing_data=dict(forcing_data))
