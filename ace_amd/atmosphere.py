"""Atmosphere bookkeeping needed by the conservation correctors.

Restates, for plain dicts of torch tensors on any device:
  * the variable-name conventions and derived quantities of ``AtmosphereData`` (fme/core/atmosphere_data.py:18-416),
    with the level stacking of ``Stacker`` (fme/core/stacker.py:38-160),
  * ``HybridSigmaPressureCoordinate.interface_pressure / vertical_integral`` (fme/core/coordinates.py:241-280),
  * the area weights and weighted mean of a regular lat-lon grid (fme/core/metrics.py:14-32, 63-90;
    fme/core/gridded_ops.py:350-359),
  * the energy / moisture helper formulas (fme/core/metrics.py:283-355) and constants (fme/core/constants.py).
Pinned against the reference itself by tests/golden/gen_corrector_*.pt (tests/golden/make_golden_corrector.py).
"""
import dataclasses
import re
from typing import Dict, List, Mapping, Optional

import torch

TensorMapping = Mapping[str, torch.Tensor]
TensorDict = Dict[str, torch.Tensor]

# fme/core/constants.py
LATENT_HEAT_OF_VAPORIZATION = 2.5e6
LATENT_HEAT_OF_FREEZING = 334000.0
GRAVITY = 9.80665
SPECIFIC_HEAT_OF_DRY_AIR_CONST_PRESSURE = 1004.6
RVGAS = 461.5
RDGAS = 287.05
SPECIFIC_HEAT_OF_DRY_AIR_CONST_VOLUME = SPECIFIC_HEAT_OF_DRY_AIR_CONST_PRESSURE - RDGAS

# fme/core/atmosphere_data.py:18-43
ATMOSPHERE_FIELD_NAME_PREFIXES = {
    "specific_total_water": ["specific_total_water_"],
    "surface_pressure": ["PRESsfc", "PS"],
    "surface_height": ["HGTsfc"],
    "surface_geopotential": ["PHIS"],
    "tendency_of_total_water_path_due_to_advection": ["tendency_of_total_water_path_due_to_advection"],
    "latent_heat_flux": ["LHTFLsfc", "LHFLX"],
    "sensible_heat_flux": ["SHTFLsfc", "SHFLX"],
    "precipitation_rate": ["PRATEsfc", "surface_precipitation_rate"],
    "sfc_down_sw_radiative_flux": ["DSWRFsfc", "FSDS"],
    "sfc_up_sw_radiative_flux": ["USWRFsfc", "surface_upward_shortwave_flux"],
    "sfc_down_lw_radiative_flux": ["DLWRFsfc", "FLDS"],
    "sfc_up_lw_radiative_flux": ["ULWRFsfc", "surface_upward_longwave_flux"],
    "toa_up_lw_radiative_flux": ["ULWRFtoa", "FLUT"],
    "toa_up_sw_radiative_flux": ["USWRFtoa", "top_of_atmos_upward_shortwave_flux"],
    "toa_down_sw_radiative_flux": ["DSWRFtoa", "SOLIN"],
    "air_temperature": ["air_temperature_", "T_"],
    "frozen_precipitation_rate": ["total_frozen_precipitation_rate"],
    "eastward_wind_at_10m": ["UGRD10m"],
    "northward_wind_at_10m": ["VGRD10m"],
}
_LEVEL = re.compile(r"_(\d+)$")


def spherical_area_weights(lat: torch.Tensor, num_lon: int) -> torch.Tensor:
    """fme/core/metrics.py:14-32: cos(latitude in degrees), normalised to sum 1 over the grid."""
    w = torch.cos(torch.deg2rad(lat)).unsqueeze(-1).expand(-1, num_lon)
    return w / w.sum(dim=(-1, -2), keepdim=True)


class AreaWeightedMean:
    """LatLonOperations.area_weighted_mean (gridded_ops.py:350-359) -> metrics.weighted_mean (metrics.py:63-90)."""

    def __init__(self, area_weights: torch.Tensor):
        if not torch.allclose(area_weights, area_weights[..., :1].expand_as(area_weights)):
            raise ValueError("Area weights must be longitudinally uniform, as assumed for zonal mean.")
        self._weights = {}
        self._cpu = area_weights.detach().to("cpu", copy=True)

    def _w(self, device) -> torch.Tensor:
        key = str(device)
        if key not in self._weights:
            self._weights[key] = self._cpu.to(device)
        return self._weights[key]

    def __call__(self, data: torch.Tensor, keepdim: bool = False, name: Optional[str] = None) -> torch.Tensor:
        w = self._w(data.device).expand(data.shape)
        data = data.where(w != 0.0, 0.0)
        return (data * w).sum(dim=(-2, -1), keepdim=keepdim) / w.sum(dim=(-2, -1), keepdim=keepdim)


@dataclasses.dataclass
class HybridSigmaPressureCoordinate:
    """p(k) = a(k) + b(k) ps at the layer interfaces (fme/core/coordinates.py:150-280)."""

    ak: torch.Tensor
    bk: torch.Tensor

    def __post_init__(self):
        if self.ak.dim() != 1 or self.bk.dim() != 1 or len(self.ak) != len(self.bk):
            raise ValueError("ak and bk must be 1-dimensional tensors of the same length")

    def to(self, device) -> "HybridSigmaPressureCoordinate":
        return HybridSigmaPressureCoordinate(self.ak.to(device), self.bk.to(device))

    def get_ak(self) -> torch.Tensor:
        return self.ak

    def get_bk(self) -> torch.Tensor:
        return self.bk

    def interface_pressure(self, surface_pressure: torch.Tensor) -> torch.Tensor:
        return torch.stack([ak + bk * surface_pressure for ak, bk in zip(self.ak, self.bk)], dim=-1)

    def vertical_integral(self, integrand: torch.Tensor, surface_pressure: torch.Tensor) -> torch.Tensor:
        if len(self.ak) != integrand.shape[-1] + 1:
            raise ValueError("The last dimension of integrand must match the number of vertical layers in the "
                             "hybrid sigma-pressure vertical coordinate.")
        thickness = self.interface_pressure(surface_pressure).diff(dim=-1)
        return (integrand * thickness).sum(dim=-1) / GRAVITY


def compute_layer_thickness(pressure_at_interface, air_temperature, specific_total_water) -> torch.Tensor:
    """fme/core/atmosphere_data.py:380-398."""
    tv = air_temperature * (1 + (RVGAS / RDGAS - 1.0) * specific_total_water)
    dlogp = torch.log(torch.clamp(pressure_at_interface, min=1.0)).diff(dim=-1)
    return dlogp * RDGAS * tv / GRAVITY


def _height_at_interface(layer_thickness: torch.Tensor, surface_height: torch.Tensor) -> torch.Tensor:
    """fme/core/atmosphere_data.py:401-416."""
    cumulative = torch.cumsum(layer_thickness.flip(dims=(-1,)), dim=-1).flip(dims=(-1,))
    hsfc = torch.where(surface_height < 0.0, 0, surface_height).reshape(*surface_height.shape, 1)
    return torch.concat([cumulative + hsfc.broadcast_to(cumulative.shape), hsfc], dim=-1)


class AtmosphereData:
    """The accessors of fme/core/atmosphere_data.py:59-377 that the correctors use."""

    def __init__(self, atmosphere_data: TensorMapping, vertical_coordinate: Optional[HybridSigmaPressureCoordinate] = None):
        self._data = dict(atmosphere_data)
        self._vc = vertical_coordinate
        self._modified_keys = set()

    @property
    def data(self) -> TensorDict:
        return self._data

    @property
    def modified_data(self) -> TensorDict:
        return {k: self._data[k] for k in self._modified_keys}

    # ---- name resolution (Stacker semantics)
    def _get(self, name: str) -> torch.Tensor:
        for prefix in ATMOSPHERE_FIELD_NAME_PREFIXES[name]:
            if prefix in self._data:
                return self._data[prefix]
        raise KeyError(name)

    def _set(self, name: str, value: torch.Tensor) -> None:
        for prefix in ATMOSPHERE_FIELD_NAME_PREFIXES[name]:
            if prefix in self._data:
                self._data[prefix] = value
                self._modified_keys.add(prefix)
                return
        raise KeyError(name)

    def _level_names(self, prefix: str) -> List[str]:
        names = [n for n in self._data if n.startswith(prefix)]
        levels = []
        for n in names:
            m = _LEVEL.search(n)
            if m is None:
                raise ValueError(f"Invalid field name {n}, is a prefix variable but does not end in _{{number}}.")
            levels.append(int(m.group(1)))
        for i, level in enumerate(sorted(levels)):
            if i != level:
                raise ValueError(f"Missing level {i} in {prefix} levels {levels}.")
        if not names:
            raise KeyError(prefix)
        return sorted(names, key=lambda n: levels[names.index(n)])

    def get_all_vertical_level_names(self, standard_name: str) -> List[str]:
        for prefix in ATMOSPHERE_FIELD_NAME_PREFIXES[standard_name]:
            if prefix in self._data:
                return [prefix]
            try:
                return self._level_names(prefix)
            except KeyError:
                pass
        raise KeyError(f"No prefix associated with '{standard_name}' was found in data keys.")

    def _stack(self, standard_name: str) -> torch.Tensor:
        for prefix in ATMOSPHERE_FIELD_NAME_PREFIXES[standard_name]:
            if prefix in self._data:
                return self._data[prefix].unsqueeze(-1)
            try:
                return torch.stack([self._data[n] for n in self._level_names(prefix)], dim=-1)
            except KeyError:
                pass
        raise KeyError(f"Found no matches for any of {ATMOSPHERE_FIELD_NAME_PREFIXES[standard_name]} among the data "
                       f"names {list(self._data.keys())}.")

    # ---- fields
    @property
    def air_temperature(self) -> torch.Tensor:
        return self._stack("air_temperature")

    @property
    def specific_total_water(self) -> torch.Tensor:
        return self._stack("specific_total_water")

    @property
    def surface_height(self) -> torch.Tensor:
        try:
            return self._get("surface_height")
        except KeyError:
            return self._get("surface_geopotential") / 9.80616

    @property
    def surface_pressure(self) -> torch.Tensor:
        return self._get("surface_pressure")

    def set_surface_pressure(self, value):
        self._set("surface_pressure", value)

    @property
    def toa_down_sw_radiative_flux(self) -> torch.Tensor:
        return self._get("toa_down_sw_radiative_flux")

    def _need_vc(self, what: str):
        if self._vc is None:
            raise ValueError(f"Vertical coordinate must be provided to compute {what}.")
        return self._vc

    @property
    def total_water_path(self) -> torch.Tensor:
        return self._need_vc("total water path").vertical_integral(self.specific_total_water, self.surface_pressure)

    @property
    def surface_pressure_due_to_dry_air(self) -> torch.Tensor:
        self._need_vc("dry air")
        return self.surface_pressure - GRAVITY * self.total_water_path          # metrics.py:283-296

    @property
    def frozen_precipitation_rate(self) -> torch.Tensor:
        try:
            return self._get("frozen_precipitation_rate")
        except KeyError:
            try:
                return self._data["ICEsfc"] + self._data["GRAUPELsfc"] + self._data["SNOWsfc"]
            except KeyError:
                return torch.zeros_like(self.surface_pressure)

    def set_frozen_precipitation_rate(self, value):
        self._set("frozen_precipitation_rate", value)

    @property
    def net_surface_energy_flux_without_frozen_precip(self) -> torch.Tensor:
        """metrics.py:299-334 without the frozen-precipitation term (atmosphere_data.py:217-226; the slab ocean's forcing)."""
        radiative = (self._get("sfc_down_sw_radiative_flux") - self._get("sfc_up_sw_radiative_flux")
                     + self._get("sfc_down_lw_radiative_flux") - self._get("sfc_up_lw_radiative_flux"))
        turbulent = -self._get("latent_heat_flux") - self._get("sensible_heat_flux")
        return radiative + turbulent - 0.0

    @property
    def net_surface_energy_flux(self) -> torch.Tensor:
        """metrics.py:299-334 with the frozen-precipitation term."""
        radiative = (self._get("sfc_down_sw_radiative_flux") - self._get("sfc_up_sw_radiative_flux")
                     + self._get("sfc_down_lw_radiative_flux") - self._get("sfc_up_lw_radiative_flux"))
        turbulent = -self._get("latent_heat_flux") - self._get("sensible_heat_flux")
        return radiative + turbulent - self.frozen_precipitation_rate * LATENT_HEAT_OF_FREEZING

    @property
    def net_top_of_atmosphere_energy_flux(self) -> torch.Tensor:
        """metrics.py:337-355."""
        return (self._get("toa_down_sw_radiative_flux") - self._get("toa_up_sw_radiative_flux")
                - self._get("toa_up_lw_radiative_flux"))

    @property
    def net_energy_flux_into_atmosphere(self) -> torch.Tensor:
        return self.net_top_of_atmosphere_energy_flux - self.net_surface_energy_flux

    @property
    def precipitation_rate(self) -> torch.Tensor:
        return self._get("precipitation_rate")

    def set_precipitation_rate(self, value):
        self._set("precipitation_rate", value)

    @property
    def evaporation_rate(self) -> torch.Tensor:
        return self._get("latent_heat_flux") / LATENT_HEAT_OF_VAPORIZATION

    def set_evaporation_rate(self, value):
        self._set("latent_heat_flux", value * LATENT_HEAT_OF_VAPORIZATION)

    @property
    def tendency_of_total_water_path_due_to_advection(self) -> torch.Tensor:
        return self._get("tendency_of_total_water_path_due_to_advection")

    def set_tendency_of_total_water_path_due_to_advection(self, value):
        self._set("tendency_of_total_water_path_due_to_advection", value)

    @property
    def height_at_midpoint(self) -> torch.Tensor:
        vc = self._need_vc("height at midpoint")
        thickness = compute_layer_thickness(vc.interface_pressure(self.surface_pressure), self.air_temperature,
                                            self.specific_total_water)
        h = _height_at_interface(thickness, self.surface_height)
        return 0.5 * (h[..., :-1] + h[..., 1:])

    @property
    def total_energy_ace2(self) -> torch.Tensor:
        return (self.air_temperature * SPECIFIC_HEAT_OF_DRY_AIR_CONST_VOLUME
                + self.specific_total_water * LATENT_HEAT_OF_VAPORIZATION + self.height_at_midpoint * GRAVITY)

    @property
    def windspeed_at_10m(self) -> torch.Tensor:
        """atmosphere_data.py:367-373."""
        return torch.sqrt(self._get("eastward_wind_at_10m") ** 2 + self._get("northward_wind_at_10m") ** 2)

    @property
    def total_energy_ace2_path(self) -> torch.Tensor:
        return self._need_vc("total energy ACE2 path").vertical_integral(self.total_energy_ace2, self.surface_pressure)
