"""Checkpoint ingestion: build an ``ace_amd.Stepper`` from a checkpoint written by ``fme``.

Follows ``load_stepper`` / ``Stepper.from_state`` (fme/ace/stepper/single_module.py:1909-1927, 1358-1429):

    checkpoint = torch.load(path)["stepper"]
    new format   {"config": {"step": {"type", "config"}, ...}, "dataset_info": {...}, "step": <step state>, ...}
                 step type "single_module"/"default" (fme/core/step/single_module.py:48-49) possibly wrapped in
                 "multi_call" (fme/core/step/multi_call.py:60-85, state {"wrapped_step": ...}: :330-345)
    legacy       {"config": SingleModuleStepperConfig fields, "module": state_dict, "normalizer", "img_shape" |
                 "data_shapes", ...}   (single_module.py:1370-1413)

Module weights are loaded with the strict ``load_state_dict`` the reference uses (fme/core/registry/module.py:102-112);
the "module." prefix of wrapped modules is accepted (fme/core/distributed/non_distributed.py:15-28).

What is NOT carried over (and why) is returned in ``LoadedStepper.ignored``: training history, loss configuration,
parameter-init configuration.  Input masking and the dataset's mask provider are carried in (ace_amd/masking.py).  ``derived_forcings`` (the insolation computed from the time axis) is built
(ace_amd/derived_forcings.py): the stepper then wants the times of every forcing window.  Latitudes / area weights and the hybrid-sigma
coefficients are kept for the conservation correctors (ace_amd/corrector.py).  Features that change the rollout and are not
implemented raise ``NotImplementedError`` unless ``ignore_unsupported=True``.
"""
import dataclasses
import datetime
import pathlib
from typing import Any, Dict, List, Mapping, Optional, Tuple, Union

import torch

from .dataset_info import DatasetInfo
from .registry import ModuleSelector
from .step import NormalizationConfig, SingleModuleStepConfig
from .stepper import Stepper

_STEP_TYPES = ("single_module", "default")


@dataclasses.dataclass
class LoadedStepper:
    stepper: Stepper
    config: SingleModuleStepConfig
    dataset_info: DatasetInfo
    ignored: List[str]


def _img_shape_from_dataset_state(ds: Mapping[str, Any]) -> Tuple[int, int]:
    """DatasetInfo.from_state (fme/core/dataset_info.py:291-345): img_shape, or the lat/lon sizes of the coordinates."""
    if ds.get("img_shape") is not None:
        return tuple(int(v) for v in ds["img_shape"])
    hc = ds.get("horizontal_coordinates")
    if hc is not None:
        if "lat" in hc and "lon" in hc:                      # LatLonCoordinates.get_state (coordinates.py:708-709)
            return (int(len(hc["lat"])), int(len(hc["lon"])))
        if "face" in hc and "height" in hc and "width" in hc:    # HEALPixCoordinates (coordinates.py:716-799): the face's (height, width)
            if len(hc["face"]) != 12:
                raise ValueError("HEALPixCoordinates must have 12 faces.")
            return (int(len(hc["height"])), int(len(hc["width"])))
        raise ValueError(f"unknown horizontal coordinates in the checkpoint's dataset_info: {sorted(hc)}")
    go = ds.get("gridded_operations")
    if go is not None and "state" in go and "area_weights" in go["state"]:
        aw = go["state"]["area_weights"]
        return (int(aw.shape[-2]), int(aw.shape[-1]))
    raise ValueError("checkpoint dataset_info carries neither img_shape nor horizontal coordinates")


def _timestep_from_dataset_state(ds: Mapping[str, Any]) -> datetime.timedelta:
    ts = ds.get("timestep")
    if ts is None:
        return datetime.timedelta(hours=6)
    if isinstance(ts, datetime.timedelta):
        return ts
    return datetime.timedelta(microseconds=int(ts))          # encode_timestep: integer microseconds (fme/core/dataset/utils.py)


def _normalization_from_state(norm: Mapping[str, Any]) -> NormalizationConfig:
    """NetworkAndLossNormalizationConfig (fme/core/normalizer.py:318-358) -> the network normaliser; also accepts a
    bare NormalizationConfig / StandardNormalizer state {"means", "stds"}."""
    if "network" in norm:
        norm = norm["network"]
    means, stds = norm.get("means"), norm.get("stds")
    if not means or not stds:
        raise ValueError("checkpoint normalization carries no loaded means/stds (newer checkpoints embed them: "
                         "StepperConfig.as_loaded_dict, single_module.py:582-584)")
    f = lambda v: float(v.item()) if isinstance(v, torch.Tensor) else float(v)
    # fme/core/normalizer.py:41-42, 212-242: the NaN fills travel with the normaliser and change what the network sees
    return NormalizationConfig(means={k: f(v) for k, v in means.items()}, stds={k: f(v) for k, v in stds.items()},
                               fill_nans_on_normalize=bool(norm.get("fill_nans_on_normalize", False)),
                               fill_nans_on_denormalize=bool(norm.get("fill_nans_on_denormalize", False)))


def stepper_config_from_state(state: Mapping[str, Any], ignore_unsupported: bool = False):
    """-> (SingleModuleStepConfig, DatasetInfo, module state dict, list of ignored items).  The stepper-level
    ``derived_forcings`` configuration is attached to the step config as ``_derived_forcings`` (a DerivedForcingsConfig), the
    multi-call configuration of a ``multi_call`` wrapper as ``_multi_call`` (a MultiCallConfig or None)."""
    from .derived_forcings import DerivedForcingsConfig
    from .multi_call import MultiCallConfig
    ignored: List[str] = []
    multi_call = None
    input_masking = None
    cfg = state["config"]
    derived_forcings = DerivedForcingsConfig.from_state(cfg.get("derived_forcings") if "step" in cfg else None)
    if "step" in cfg:                                         # ---- new format
        sel = cfg["step"]
        step_type, step_cfg = sel["type"], dict(sel["config"])
        step_state = state["step"]
        if step_type == "multi_call":
            multi_call = MultiCallConfig.from_state(step_cfg.get("config"))      # run by Stepper.step (ace_amd/multi_call.py)
            inner = step_cfg["wrapped_step"]
            step_type, step_cfg = inner["type"], dict(inner["config"])
            step_state = step_state["wrapped_step"]
        if step_type not in _STEP_TYPES:
            raise NotImplementedError(f"step type '{step_type}' is outside the accelerated hot path")
        # static input masking (input_process_func on every step, single_module.py:615-632): ace_amd/masking.py
        input_masking = cfg.get("input_masking")
        ds_state = state["dataset_info"]
        normalization = _normalization_from_state(step_cfg["normalization"])
    else:                                                     # ---- legacy single-module stepper
        step_cfg = dict(cfg)
        multi_call = MultiCallConfig.from_state(step_cfg.pop("multi_call", None))      # single_module.py:1370-1413 -> MultiCallStep
        for k in ("parameter_init", "loss", "loss_normalization", "residual_normalization",
                  "include_multi_call_in_loss", "crps_training"):
            if step_cfg.pop(k, None) not in (None, {}, True, False):
                ignored.append(k)
        step_state = {"module": state["module"]}
        ds_state = {"timestep": state.get("encoded_timestep"),
                    "vertical_coordinate": state.get("sigma_coordinates", state.get("vertical_coordinate"))}
        if "area" in state:
            ds_state["gridded_operations"] = {"type": "LatLonOperations", "state": {"area_weights": state["area"]}}
        elif "gridded_operations" in state:
            ds_state["gridded_operations"] = state["gridded_operations"]
        if "img_shape" in state:
            ds_state["img_shape"] = state["img_shape"]
        else:
            for shape in state.get("data_shapes", {}).values():
                if len(shape) == 4:
                    ds_state["img_shape"] = shape[-2:]
                    break
        norm_state = state.get("normalizer", state.get("normalization"))
        if norm_state is None:
            raise ValueError(f"No normalizer state found, keys include {list(state.keys())}")
        normalization = _normalization_from_state(norm_state)
    step_cfg.pop("crps_training", None)
    step_cfg.pop("normalization", None)
    builder = step_cfg.pop("builder")
    unsupported = []
    for k in ("global_mean_removal", "input_dropout"):
        if step_cfg.get(k) is not None:
            unsupported.append(k)
    if step_cfg.get("include_channel_mask_inputs"):
        unsupported.append("include_channel_mask_inputs")
    if unsupported and not ignore_unsupported:
        raise NotImplementedError("step options outside the accelerated hot path: " + ", ".join(unsupported))
    for k in unsupported:
        ignored.append(k)
        step_cfg[k] = None if k != "include_channel_mask_inputs" else False
    known = {f.name for f in dataclasses.fields(SingleModuleStepConfig)}
    unknown = set(step_cfg) - known
    if unknown:
        raise ValueError(f"unknown step config fields: {sorted(unknown)}")
    config = SingleModuleStepConfig(builder=ModuleSelector(type=builder["type"], config=dict(builder["config"])),
                                    normalization=normalization, **step_cfg)
    from .masking import SpatialMaskProvider
    mp = ds_state.get("mask_provider")
    provider = SpatialMaskProvider.from_state(mp) if isinstance(mp, Mapping) else None      # drives the output masking (and input_masking)
    if ds_state.get("variable_metadata") is not None:
        ignored.append("dataset_info.variable_metadata")
    labels = ds_state.get("all_labels") or None
    # geometry for the conservation correctors: latitudes (or legacy area weights) and hybrid-sigma coefficients
    hc = ds_state.get("horizontal_coordinates") or {}
    vc = ds_state.get("vertical_coordinate") or {}
    go = ds_state.get("gridded_operations") or {}
    area = go.get("state", {}).get("area_weights") if isinstance(go, Mapping) else None
    dataset_info = DatasetInfo(_img_shape_from_dataset_state(ds_state), all_labels=set(labels) if labels else None,
                               timestep=_timestep_from_dataset_state(ds_state), lat=hc.get("lat"), lon=hc.get("lon"),
                               ak=vc.get("ak"), bk=vc.get("bk"), area_weights=area, mask_provider=provider)
    missing = config.corrector.unsupported(dataset_info)
    if missing:
        if not ignore_unsupported:
            raise NotImplementedError("corrector options that need latitudes / a hybrid-sigma vertical coordinate the "
                                      "checkpoint's dataset_info does not carry: " + ", ".join(missing))
        ignored.extend(f"corrector.{m}" for m in missing)
        config._ignore_unsupported = True
    if state.get("training_history"):
        ignored.append("training_history")
    if derived_forcings.insolation is not None and dataset_info.horizontal_coordinates is None:
        if not ignore_unsupported:
            raise NotImplementedError("derived_forcings.insolation needs the latitudes and longitudes of the grid, which the "
                                      "checkpoint's dataset_info does not carry (pass ignore_unsupported=True to drop it)")
        ignored.append("derived_forcings")
        derived_forcings = DerivedForcingsConfig()
    config._derived_forcings = derived_forcings
    config._multi_call = multi_call
    config._input_masking = input_masking
    return config, dataset_info, step_state, ignored


@dataclasses.dataclass
class StepperOverrideConfig:
    """single_module.py:1848-1870: inference-time overrides of a serialized stepper; ``"keep"`` leaves the option alone.
    ``multi_call``: "keep", None (no multi-call diagnostics) or a MultiCallConfig / its state dict;
    ``derived_forcings``: "keep" or a DerivedForcingsConfig (its state dict) with the insolation name the network was trained on."""

    ocean: Any = "keep"
    multi_call: Any = "keep"
    derived_forcings: Any = "keep"
    prescribed_prognostic_names: Any = "keep"


def apply_stepper_override(stepper: Stepper, override_config: Optional[StepperOverrideConfig] = None) -> None:
    """single_module.py:1929-1960."""
    if override_config is None:
        override_config = StepperOverrideConfig()
    if override_config.ocean != "keep":
        stepper.replace_ocean(override_config.ocean)
    if not (isinstance(override_config.multi_call, str) and override_config.multi_call == "keep"):
        stepper.replace_multi_call(override_config.multi_call)
    if not (isinstance(override_config.derived_forcings, str) and override_config.derived_forcings == "keep"):
        stepper.replace_derived_forcings(override_config.derived_forcings)
    if override_config.prescribed_prognostic_names != "keep":
        stepper.replace_prescribed_prognostic_names(override_config.prescribed_prognostic_names)


def load_stepper(checkpoint: Union[str, pathlib.Path, Mapping[str, Any]],
                 override_config: Optional[StepperOverrideConfig] = None, device=None,
                 ignore_unsupported: bool = False) -> LoadedStepper:
    """``fme.ace.stepper.load_stepper`` (single_module.py:1909-1927) for the accelerated stepper.  ``checkpoint``: a path
    (torch.load) or the already-loaded dict; either the whole checkpoint ({"stepper": ...}) or the stepper state itself."""
    if not isinstance(checkpoint, Mapping):
        checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=False)
    state = checkpoint["stepper"] if "stepper" in checkpoint else checkpoint
    config, dataset_info, step_state, ignored = stepper_config_from_state(state, ignore_unsupported)
    stepper = Stepper.from_config(config, dataset_info, device=device, derived_forcings=getattr(config, "_derived_forcings", None),
                                  multi_call=getattr(config, "_multi_call", None), input_masking=getattr(config, "_input_masking", None))
    stepper.load_state({"step": step_state})
    apply_stepper_override(stepper, override_config)
    stepper.set_eval()
    return LoadedStepper(stepper=stepper, config=stepper._step_obj.config, dataset_info=dataset_info, ignored=ignored)
