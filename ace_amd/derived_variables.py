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

def _stepwise_change(series: torch.Tensor) -> torch.Tensor:
    """x[t] - x[t-1] along the time axis, zero at the first time level (nothing to difference against)."""
    out = torch.zeros_like(series)
    out[:, 1:] = series[:, 1:] - series[:, :-1]
    return out


def _twp_budget_residual(d: AtmosphereData, dt: datetime.timedelta) -> torch.Tensor:
    """d(TWP)/dt - (E - P + advective tendency); defined from the second time level on."""
    sources = d.evaporation_rate - d.precipitation_rate + d.tendency_of_total_water_path_due_to_advection
    twp = d.total_water_path
    out = torch.zeros_like(twp)
    out[:, 1:] = (twp[:, 1:] - twp[:, :-1]) / dt.total_seconds() - sources[:, 1:]
    return out


def _energy_path_tendency(d: AtmosphereData, dt: datetime.timedelta) -> torch.Tensor:
    return _stepwise_change(d.total_energy_ace2_path) / dt.total_seconds()


# name -> function, in the reference's registration order (derived_variables.py:44-167): the order is the order in which the
# variables are appended to the output dict, which downstream writers keep
_DERIVED_VARIABLE_REGISTRY: MutableMapping[str, DerivedVariableFunc] = {
    "surface_pressure_due_to_dry_air": lambda d, dt: d.surface_pressure_due_to_dry_air,
    "surface_pressure_due_to_dry_air_absolute_tendency": lambda d, dt: _stepwise_change(d.surface_pressure_due_to_dry_air).abs(),
    "total_water_path": lambda d, dt: d.total_water_path,
    "total_water_path_budget_residual": _twp_budget_residual,
    "net_energy_flux_toa_into_atmosphere": lambda d, dt: d.net_top_of_atmosphere_energy_flux,
    "net_energy_flux_sfc_into_atmosphere": lambda d, dt: -d.net_surface_energy_flux,     # stored positive into the surface
    "net_energy_flux_into_atmospheric_column": lambda d, dt: d.net_energy_flux_into_atmosphere,
    "total_energy_ace2_path": lambda d, dt: d.total_energy_ace2_path,
    "total_energy_ace2_path_tendency": _energy_path_tendency,
    "implied_tendency_of_total_energy_ace2_path_due_to_advection":
        lambda d, dt: _energy_path_tendency(d, dt) - d.net_energy_flux_into_atmosphere,   # residual of the column budget
    "windspeed_at_10m": lambda d, dt: d.windspeed_at_10m,
}


def get_derived_variable_names():
    return list(_DERIVED_VARIABLE_REGISTRY)


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
