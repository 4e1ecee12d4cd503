"""Host side of the fused post-step physics kernels (include/ace_sfno.h: ace_physics_*; ace_amd/csrc/physics.hip).

``FusedPhysics`` lowers what ``step_with_adjustments`` does after the network (fme/core/step/single_module.py:669-716:
AtmosphereCorrector -> Ocean -> prescribed prognostics) onto the four HIP kernels: it resolves the reference's variable
names (fme/core/atmosphere_data.py:18-43, the ``AtmosphereData`` accessors the corrections use) to device planes of the
RolloutEngine's static buffers, one ``ace_phys_fields`` struct per step of the window.  The torch implementation
(ace_amd/corrector.py, ace_amd/ocean.py: what ``Stepper.predict`` runs) stays the readable restatement and the CPU-testable
one; both are held to the reference's own golden rollouts."""
import ctypes
from ctypes import c_int, c_void_p
from typing import Callable, Dict, List, Mapping, Optional, Tuple

import torch

from . import _lib
from .atmosphere import ATMOSPHERE_FIELD_NAME_PREFIXES, _LEVEL

from ._lib import MAX_LEVELS, MAX_POSITIVE, MAX_PRESCRIBED, PhysConfig, PhysFields, Plane  # noqa: F401  (the C structs)

_MOISTURE = {None: 0, "precipitation": 1, "evaporation": 2, "advection_and_precipitation": 3, "advection_and_evaporation": 4}


def _check(rc: int) -> None:
    if rc != 0:
        msg = _lib.lib().ace_physics_last_error().decode()
        raise (ValueError if rc == _lib.ACE_ERR_INVALID else RuntimeError)(msg)


# ---- name resolution with AtmosphereData's rules (first prefix present; level-stacked prefixes end in _{k}) -----------------
def _single(names, standard: str) -> Optional[str]:
    for prefix in ATMOSPHERE_FIELD_NAME_PREFIXES[standard]:
        if prefix in names:
            return prefix
    return None


def _levels(names, standard: str) -> Optional[List[str]]:
    for prefix in ATMOSPHERE_FIELD_NAME_PREFIXES[standard]:
        if prefix in names:
            return [prefix]
        found = [n for n in names if n.startswith(prefix)]
        if found:
            lev = {}
            for n in found:
                m = _LEVEL.search(n)
                if m is None:
                    raise ValueError(f"Invalid field name {n}, is a prefix variable but does not end in _{{number}}.")
                lev[int(m.group(1))] = n
            if sorted(lev) != list(range(len(lev))):
                raise ValueError(f"Missing level in {prefix} levels {sorted(lev)}.")
            return [lev[k] for k in range(len(lev))]
    return None


class FusedPhysics:
    """One handle per RolloutEngine: configuration from the step's corrector / ocean objects, one field table per window step.

    ``locate_gen(name, s)``, ``locate_in(name, s)``, ``locate_next(name, s)`` return (data_ptr, per-sample stride in floats) of
    the plane holding that variable as the step's output / input / next-step data, or None when it has none."""

    def __init__(self, corrector, ocean, prescribed: List[str], batch: int, img_shape: Tuple[int, int], n_steps: int,
                 gen_names: List[str], in_names: List[str], next_names: List[str],
                 locate_gen: Callable, locate_in: Callable, locate_next: Callable, device):
        self.handle = c_void_p()
        self.batch = batch
        self._device = device
        cfgc = corrector._cfg if corrector is not None else None
        active = set(corrector.corrections) if corrector is not None else set()
        H, W = img_shape
        cfg = PhysConfig()
        cfg.nlat, cfg.nlon, cfg.max_batch = H, W, batch
        cfg.conserve_dry_air = int("conserve_dry_air" in active)
        cfg.zero_global_mean_moisture_advection = int("zero_global_mean_moisture_advection" in active)
        cfg.moisture_budget = _MOISTURE[cfgc.moisture_budget_correction] if "moisture_budget_correction" in active else 0
        cfg.clip_frozen_precipitation = int(bool(cfgc.clip_frozen_precipitation)) if cfg.moisture_budget else 0
        cfg.energy_budget = 0
        if "total_energy_budget_correction" in active:
            eb = cfgc.total_energy_budget_correction
            if eb.method != "constant_temperature":
                raise NotImplementedError(f"Method {eb.method} not implemented for total energy conservation")
            cfg.energy_budget, cfg.unaccounted_heating = 1, float(eb.constant_unaccounted_heating)
        cfg.timestep_seconds = float(corrector._dt) if (corrector is not None and corrector._dt is not None) else 0.0
        cfg.ocean = 0 if ocean is None else (2 if ocean.prescriber.interpolate else 1)
        if ocean is not None and getattr(ocean, "is_slab", False):
            raise NotImplementedError("fused ocean kernel: prescribed SST only (the slab ocean runs as torch ops)")
        if ocean is not None and ocean.prescriber.mask_value != 1:
            raise NotImplementedError("fused ocean kernel: mask_value must be 1 (what Ocean builds)")
        vc = corrector._vc if corrector is not None else None
        need_vc = cfg.conserve_dry_air or cfg.moisture_budget or cfg.energy_budget
        ak = bk = None
        if need_vc:
            ak = vc.get_ak().detach().to("cpu", torch.float32).contiguous()
            bk = vc.get_bk().detach().to("cpu", torch.float32).contiguous()
            cfg.nlev = len(ak) - 1
            if cfg.nlev > MAX_LEVELS:
                raise NotImplementedError(f"fused physics kernels support up to {MAX_LEVELS} vertical layers, got {cfg.nlev}")
        wl = None
        if corrector is not None and corrector._mean is not None:
            wl = corrector._mean._cpu.to(torch.float32).reshape(H, W)[:, 0].contiguous()
        self.force_positive = list(corrector.force_positive_names) if "force_positive" in active else []
        if len(self.force_positive) > MAX_POSITIVE or len(prescribed) > MAX_PRESCRIBED:
            raise NotImplementedError("too many force-positive / prescribed fields for the fused physics kernels")
        self.config = cfg
        self._keep = (ak, bk, wl)
        with torch.cuda.device(device):
            _check(_lib.lib().ace_physics_create(ctypes.byref(cfg), wl.data_ptr() if wl is not None else None,
                                                 ak.data_ptr() if ak is not None else None,
                                                 bk.data_ptr() if bk is not None else None, ctypes.byref(self.handle)))
        self._ref = torch.zeros(batch, dtype=torch.float64, device=device)
        # ---- one field table per step
        gen, inn, nxt = set(gen_names), set(in_names), set(next_names)
        self.fields: List[PhysFields] = []
        for s in range(n_steps):
            f = PhysFields()

            def put(dst, loc):
                if loc is not None:
                    dst.p, dst.stride = int(loc[0]), int(loc[1])

            def gen1(standard):
                n = _single(gen, standard)
                return locate_gen(n, s) if n is not None else None

            if need_vc or cfg.zero_global_mean_moisture_advection:
                put(f.ps, gen1("surface_pressure"))
                n = _single(inn, "surface_pressure")
                put(f.ps_in, locate_in(n, s) if n else None)
                for dst, names, loc in ((f.wat, _levels(gen, "specific_total_water"), locate_gen),
                                        (f.wat_in, _levels(inn, "specific_total_water"), locate_in),
                                        (f.T, _levels(gen, "air_temperature") if cfg.energy_budget else None, locate_gen),
                                        (f.T_in, _levels(inn, "air_temperature") if cfg.energy_budget else None, locate_in)):
                    if names is not None:
                        if need_vc and len(names) != cfg.nlev:
                            raise ValueError(f"{len(names)} vertical levels in the data but {cfg.nlev} layers in the vertical coordinate")
                        for k, nm in enumerate(names):
                            put(dst[k], loc(nm, s))
                put(f.adv, gen1("tendency_of_total_water_path_due_to_advection"))
                put(f.precip, gen1("precipitation_rate"))
                put(f.lhf, gen1("latent_heat_flux"))
                put(f.shf, gen1("sensible_heat_flux"))
                for attr, std in (("dswsfc", "sfc_down_sw_radiative_flux"), ("uswsfc", "sfc_up_sw_radiative_flux"),
                                  ("dlwsfc", "sfc_down_lw_radiative_flux"), ("ulwsfc", "sfc_up_lw_radiative_flux"),
                                  ("ulwtoa", "toa_up_lw_radiative_flux"), ("uswtoa", "toa_up_sw_radiative_flux")):
                    put(getattr(f, attr), gen1(std))
                if "total_frozen_precipitation_rate" in gen:
                    put(f.frozen, locate_gen("total_frozen_precipitation_rate", s))
                elif {"ICEsfc", "GRAUPELsfc", "SNOWsfc"} <= gen:
                    for k, nm in enumerate(("ICEsfc", "GRAUPELsfc", "SNOWsfc")):
                        put(f.frozen_parts[k], locate_gen(nm, s))
                if cfg.energy_budget:   # surface height of the input; next step's forcing: surface height and insolation
                    f.hgt_in_scale = f.hgt_next_scale = 1.0
                    n = _single(inn, "surface_height")
                    if n is None:
                        n = _single(inn, "surface_geopotential")
                        f.hgt_in_scale = 1.0 / 9.80616
                    put(f.hgt_in, locate_in(n, s) if n else None)
                    n = _single(nxt, "surface_height")
                    if n is None:
                        n = _single(nxt, "surface_geopotential")
                        f.hgt_next_scale = 1.0 / 9.80616
                    put(f.hgt_next, locate_next(n, s) if n else None)
                    n = _single(nxt, "toa_down_sw_radiative_flux")
                    put(f.dswtoa_next, locate_next(n, s) if n else None)
            for k, nm in enumerate(self.force_positive):
                if nm not in gen:
                    raise KeyError(nm)
                put(f.positive[k], locate_gen(nm, s))
            f.npositive = len(self.force_positive)
            if ocean is not None:
                put(f.sst, locate_gen(ocean.surface_temperature_name, s))
                put(f.sst_target, locate_next(ocean.surface_temperature_name, s))
                put(f.ocean_fraction, locate_next(ocean.ocean_fraction_name, s))
            for k, nm in enumerate(prescribed):
                src = locate_next(nm, s)
                if src is None:
                    raise ValueError(f"prescribed_prognostic_name '{nm}' not in next_step_input_data")
                put(f.prescribed_dst[k], locate_gen(nm, s))
                put(f.prescribed_src[k], src)
            f.nprescribed = len(prescribed)
            self.fields.append(f)

    @property
    def tracks_dry_air(self) -> bool:
        return bool(self.config.conserve_dry_air)

    def apply(self, s: int, stream: int) -> None:
        _check(_lib.lib().ace_physics_apply(self.handle, ctypes.byref(self.fields[s]), self.batch, stream))

    def reset(self, stream: int) -> None:
        _check(_lib.lib().ace_physics_reset(self.handle, stream))

    def set_reference(self, mass: torch.Tensor, stream: int) -> None:
        self._ref.copy_(mass.reshape(-1).to(self._ref))
        _check(_lib.lib().ace_physics_set_reference(self.handle, self._ref.data_ptr(), self.batch, stream))

    def get_reference(self, stream: int) -> Optional[torch.Tensor]:
        have = c_int(0)
        _check(_lib.lib().ace_physics_get_reference(self.handle, self._ref.data_ptr(), ctypes.byref(have), self.batch, stream))
        return self._ref.clone().reshape(self.batch, 1, 1) if have.value else None

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().ace_physics_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
