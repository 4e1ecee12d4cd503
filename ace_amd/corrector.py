"""Post-step atmosphere corrector: fme/core/corrector/atmosphere.py:223-692.

Mirror of ``AtmosphereCorrectorConfig`` (same field set and defaults, so a reference step config round-trips) and of the
corrections it builds, applied in the reference's order (atmosphere.py:349-398): force positive -> conserve dry air ->
zero global-mean moisture advection -> moisture budget (+ frozen-precipitation clip) -> total energy budget.
Global means are area weighted (cos latitude, fme/core/metrics.py:14-32) and, for dry air, taken in fp64; the dry-air
reference mass is seeded from the first step's input and carried in ``CorrectorState`` (fme/core/corrector/state.py).
The conservation closures need the dataset's latitudes and hybrid-sigma coefficients (``DatasetInfo.area_weights``,
``DatasetInfo.vertical_coordinate``); without them they raise, they are never silently skipped.
Plain torch ops on whatever device the state lives on; pinned against the reference itself by
tests/golden/gen_corrector_*.pt.
"""
import dataclasses
from typing import Any, Dict, List, Mapping, Optional, Tuple

import torch

from .atmosphere import (GRAVITY, SPECIFIC_HEAT_OF_DRY_AIR_CONST_VOLUME, AreaWeightedMean, AtmosphereData,
                         HybridSigmaPressureCoordinate, compute_layer_thickness)

TensorMapping = Mapping[str, torch.Tensor]
TensorDict = Dict[str, torch.Tensor]

_MOISTURE_TERMS = ("precipitation", "evaporation", "advection_and_precipitation", "advection_and_evaporation")


@dataclasses.dataclass
class CorrectorState:
    """fme/core/corrector/state.py: per-sample state owned by the corrector, threaded through StepperState."""
    global_dry_air_mass: Optional[torch.Tensor] = None


def force_positive(data: TensorMapping, names: List[str]) -> TensorDict:
    """fme/core/corrector/utils.py:26-44 (inference: no straight-through gradient): only the clamped fields."""
    return {name: torch.clamp(data[name], min=0.0) for name in names}


# ---------------------------------------------------------------------------------------------------------------------
# the corrections (atmosphere.py:404-692)
def _seed_global_dry_air_mass(input_data, corrector_state, area_weighted_mean, vertical_coordinate, precision):
    if corrector_state is not None and corrector_state.global_dry_air_mass is not None:
        return corrector_state
    ic = AtmosphereData(input_data, vertical_coordinate)
    target = area_weighted_mean(ic.surface_pressure_due_to_dry_air.to(precision), keepdim=True)
    return CorrectorState(global_dry_air_mass=target)


def _adjust_gen_dry_air_to_target(gen_data, target_global_dry_air, area_weighted_mean, vertical_coordinate, precision):
    """ps = (dry_air + sum_k(dak_k wat_k)) / (1 - sum_k(dbk_k wat_k)) after shifting the dry-air pressure of every column
    by the global-mean error (atmosphere.py:431-467)."""
    gen = AtmosphereData(gen_data, vertical_coordinate)
    gen_dry_air = gen.surface_pressure_due_to_dry_air
    global_gen_dry_air = area_weighted_mean(gen_dry_air.to(precision), keepdim=True)
    error = global_gen_dry_air - target_global_dry_air.to(precision)
    new_gen_dry_air = gen_dry_air.to(precision) - error
    try:
        wat = gen.specific_total_water.to(precision)
    except KeyError:
        raise ValueError("specific_total_water is required for conservation")
    ak_diff = vertical_coordinate.get_ak().diff().to(precision)
    bk_diff = vertical_coordinate.get_bk().diff().to(precision)
    new_pressure = (new_gen_dry_air + (ak_diff * wat).sum(-1)) / (1 - (bk_diff * wat).sum(-1))
    gen.set_surface_pressure(new_pressure.to(dtype=gen.surface_pressure.dtype))
    return gen.modified_data


def _force_zero_global_mean_moisture_advection(gen_data, area_weighted_mean):
    gen = AtmosphereData(gen_data)
    adv = gen.tendency_of_total_water_path_due_to_advection
    gen.set_tendency_of_total_water_path_due_to_advection(adv - area_weighted_mean(adv)[..., None, None])
    return gen.modified_data


def _clip_frozen_precipitation(gen_data):
    if "total_frozen_precipitation_rate" not in gen_data:
        return {}
    gen = AtmosphereData(gen_data)
    gen.set_frozen_precipitation_rate(torch.minimum(gen.frozen_precipitation_rate, gen.precipitation_rate))
    return gen.modified_data


def _force_conserve_moisture(input_data, gen_data, area_weighted_mean, vertical_coordinate, timestep_seconds,
                             terms_to_modify):
    """atmosphere.py:511-608."""
    inp = AtmosphereData(input_data, vertical_coordinate)
    gen = AtmosphereData(gen_data, vertical_coordinate)
    twp_total_tendency = (gen.total_water_path - inp.total_water_path) / timestep_seconds
    twp_tendency_global_mean = area_weighted_mean(twp_total_tendency, keepdim=True)
    evaporation_global_mean = area_weighted_mean(gen.evaporation_rate, keepdim=True)
    precipitation_global_mean = area_weighted_mean(gen.precipitation_rate, keepdim=True)
    if terms_to_modify.endswith("precipitation"):
        new_precipitation_global_mean = evaporation_global_mean - twp_tendency_global_mean
        gen.set_precipitation_rate(gen.precipitation_rate * (new_precipitation_global_mean / precipitation_global_mean))
    elif terms_to_modify.endswith("evaporation"):
        new_evaporation_global_mean = twp_tendency_global_mean + precipitation_global_mean
        gen.set_evaporation_rate(gen.evaporation_rate * (new_evaporation_global_mean / evaporation_global_mean))
    if terms_to_modify.startswith("advection"):
        gen.set_tendency_of_total_water_path_due_to_advection(
            twp_total_tendency - (gen.evaporation_rate - gen.precipitation_rate))
    return gen.modified_data


def _energy_correction_factor(gen: AtmosphereData, vertical_coordinate) -> torch.Tensor:
    """atmosphere.py:666-692."""
    interface_pressure = vertical_coordinate.interface_pressure(gen.surface_pressure)
    q_times_dlogp = (compute_layer_thickness(interface_pressure, gen.air_temperature, gen.specific_total_water)
                     * GRAVITY / gen.air_temperature)
    cumulative = torch.cumsum(q_times_dlogp.flip(dims=(-1,)), dim=-1).flip(dims=(-1,))
    total_integrand = SPECIFIC_HEAT_OF_DRY_AIR_CONST_VOLUME - 0.5 * q_times_dlogp + cumulative
    return vertical_coordinate.vertical_integral(total_integrand, gen.surface_pressure)


def _force_conserve_total_energy(input_data, gen_data, forcing_data, area_weighted_mean, vertical_coordinate,
                                 timestep_seconds, method="constant_temperature", unaccounted_heating=0.0):
    """atmosphere.py:611-663: a spatially and vertically uniform temperature increment closes the global energy budget."""
    if method != "constant_temperature":
        raise NotImplementedError(f"Method {method} not implemented for total energy conservation")
    inp = AtmosphereData(input_data, vertical_coordinate)
    forcing = AtmosphereData(forcing_data)
    atmosphere_data = dict(gen_data)
    atmosphere_data["DSWRFtoa"] = forcing.toa_down_sw_radiative_flux
    atmosphere_data["HGTsfc"] = forcing.surface_height
    gen = AtmosphereData(atmosphere_data, vertical_coordinate)
    gen_energy_path_gm = area_weighted_mean(gen.total_energy_ace2_path, keepdim=True)
    input_energy_path_gm = area_weighted_mean(inp.total_energy_ace2_path, keepdim=True)
    energy_flux_gm = area_weighted_mean(gen.net_energy_flux_into_atmosphere, keepdim=True)
    desired = input_energy_path_gm + (energy_flux_gm + unaccounted_heating) * timestep_seconds
    energy_correction = desired - gen_energy_path_gm
    factor_gm = area_weighted_mean(_energy_correction_factor(gen, vertical_coordinate), True)
    temperature_correction = energy_correction / factor_gm
    return {name: gen.data[name] + temperature_correction for name in gen.get_all_vertical_level_names("air_temperature")}


# ---------------------------------------------------------------------------------------------------------------------
@dataclasses.dataclass
class EnergyBudgetConfig:
    method: str
    constant_unaccounted_heating: float = 0.0


@dataclasses.dataclass
class AtmosphereCorrectorConfig:
    conserve_dry_air: bool = False
    zero_global_mean_moisture_advection: bool = False
    moisture_budget_correction: Optional[str] = None
    force_positive_names: List[str] = dataclasses.field(default_factory=list)
    total_energy_budget_correction: Optional[Any] = None
    keep_gradient_through_clamps: bool = False
    clip_frozen_precipitation: bool = False
    # CorrectorConfigABC (fme/core/corrector/registry.py:14-62): a training-epoch schedule; the corrector is always
    # applied in eval mode (validation, inline and standalone inference), so inference ignores the value.
    corrector_disabled_epochs: int = 0

    def __post_init__(self):
        if self.corrector_disabled_epochs < 0:
            raise ValueError(f"corrector_disabled_epochs must be non-negative, got {self.corrector_disabled_epochs}")
        if self.moisture_budget_correction is not None and self.moisture_budget_correction not in _MOISTURE_TERMS:
            raise ValueError(f"moisture_budget_correction must be one of {_MOISTURE_TERMS}")
        if isinstance(self.total_energy_budget_correction, Mapping):
            self.total_energy_budget_correction = EnergyBudgetConfig(**self.total_energy_budget_correction)

    @classmethod
    def from_state(cls, state: Optional[Mapping[str, Any]]) -> "AtmosphereCorrectorConfig":
        if state is None:
            return cls()
        state = dict(state)
        if "type" in state and "config" in state:      # CorrectorSelector form (fme/core/registry/corrector.py:11-45)
            if set(state) - {"type", "config", "corrector_disabled_epochs"}:
                raise ValueError(f"unknown corrector selector fields: {sorted(state)}")
            if state.get("corrector_disabled_epochs", 0) != 0:
                raise ValueError("corrector_disabled_epochs must be set on the wrapped corrector config (inside "
                                 "`config:`), not on the CorrectorSelector.")
            if state["type"] != "atmosphere_corrector":
                raise NotImplementedError(f"corrector type '{state['type']}' is outside the accelerated hot path")
            state = dict(state["config"])
        unknown = set(state) - {f.name for f in dataclasses.fields(cls)}
        if unknown:
            raise ValueError(f"unknown corrector fields: {sorted(unknown)}")
        return cls(**state)

    def needs_geometry(self) -> List[str]:
        """the options that need area weights (and, except the advection one, a vertical coordinate)"""
        out = []
        if self.conserve_dry_air:
            out.append("conserve_dry_air")
        if self.zero_global_mean_moisture_advection:
            out.append("zero_global_mean_moisture_advection")
        if self.moisture_budget_correction is not None:
            out.append("moisture_budget_correction")
        if self.total_energy_budget_correction is not None:
            out.append("total_energy_budget_correction")
        return out

    def unsupported(self, dataset_info=None) -> List[str]:
        """options that cannot be honoured with this dataset_info (no latitudes / no vertical coordinate)"""
        need = self.needs_geometry()
        if not need:
            return []
        has_area = getattr(dataset_info, "area_weights", None) is not None
        has_vc = getattr(dataset_info, "vertical_coordinate", None) is not None
        return [n for n in need if not has_area or (n != "zero_global_mean_moisture_advection" and not has_vc)]

    def get_corrector(self, dataset_info=None, ignore_unsupported: bool = False) -> Optional["AtmosphereCorrector"]:
        missing = self.unsupported(dataset_info)
        if missing and not ignore_unsupported:
            raise NotImplementedError(
                "corrector options that need the dataset's latitudes / hybrid-sigma vertical coordinate, which this "
                "dataset_info does not carry: " + ", ".join(missing))
        corrector = AtmosphereCorrector(self, dataset_info, skip=set(missing))
        return corrector if corrector.corrections else None


class AtmosphereCorrector:
    """CorrectionSequence (fme/core/corrector/registry.py:161-198) built as in atmosphere.py:349-398."""

    def __init__(self, config: AtmosphereCorrectorConfig, dataset_info=None, skip=frozenset()):
        self.force_positive_names = list(config.force_positive_names)
        area = getattr(dataset_info, "area_weights", None)
        self._mean = AreaWeightedMean(area) if area is not None else None
        self._vc: Optional[HybridSigmaPressureCoordinate] = getattr(dataset_info, "vertical_coordinate", None)
        ts = getattr(dataset_info, "timestep", None)
        self._dt = ts.total_seconds() if ts is not None else None
        self._vc_dev: Dict[str, HybridSigmaPressureCoordinate] = {}
        self.corrections: List[str] = []
        if self.force_positive_names:
            self.corrections.append("force_positive")
        for name in ("conserve_dry_air", "zero_global_mean_moisture_advection", "moisture_budget_correction",
                     "total_energy_budget_correction"):
            if name in config.needs_geometry() and name not in skip:
                self.corrections.append(name)
        self._cfg = config

    def _vcoord(self, device):
        key = str(device)
        if key not in self._vc_dev:
            self._vc_dev[key] = self._vc.to(device)
        return self._vc_dev[key]

    def __call__(self, input_data: TensorMapping, gen_data: TensorMapping, forcing_data: TensorMapping,
                 corrector_state: Optional[CorrectorState] = None) -> Tuple[TensorDict, Optional[CorrectorState]]:
        gen = dict(gen_data)
        dev = next(iter(gen.values())).device
        vc = self._vcoord(dev) if self._vc is not None else None
        for name in self.corrections:
            if name == "force_positive":
                changed = force_positive(gen, self.force_positive_names)
            elif name == "conserve_dry_air":
                corrector_state = _seed_global_dry_air_mass(input_data, corrector_state, self._mean, vc, torch.float64)
                changed = _adjust_gen_dry_air_to_target(gen, corrector_state.global_dry_air_mass, self._mean, vc,
                                                        torch.float64)
            elif name == "zero_global_mean_moisture_advection":
                changed = _force_zero_global_mean_moisture_advection(gen, self._mean)
            elif name == "moisture_budget_correction":
                changed = _force_conserve_moisture(input_data, gen, self._mean, vc, self._dt,
                                                   self._cfg.moisture_budget_correction)
                if self._cfg.clip_frozen_precipitation:
                    changed = {**changed, **_clip_frozen_precipitation({**gen, **changed})}
            else:
                eb = self._cfg.total_energy_budget_correction
                changed = _force_conserve_total_energy(input_data, gen, forcing_data, self._mean, vc, self._dt, eb.method,
                                                       eb.constant_unaccounted_heating)
            gen.update(changed)
        return gen, corrector_state
