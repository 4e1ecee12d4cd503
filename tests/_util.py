"""Shared helpers of the parity tests."""

import dataclasses
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=False)


def rel_max(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| / max|b|  (the per-step tolerance of BASELINE.json's north_star is 1e-5 in this norm)."""
    a = a.detach().cpu()
    b = b.detach().cpu()
    if a.is_complex() or b.is_complex():
        a = torch.view_as_real(a.to(torch.complex128))
        b = torch.view_as_real(b.to(torch.complex128))
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300))


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().cpu()
    b = b.detach().cpu()
    if a.is_complex() or b.is_complex():
        a = torch.view_as_real(a.to(torch.complex128))
        b = torch.view_as_real(b.to(torch.complex128))
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


def build_native_net(cfg, state, device="cuda", precision=None):
    """ace_amd network for an oracle SFNOConfig, loaded with `state` (strict).  precision: None (library default),
    "fp32" or "f16x3"."""
    import types

    from ace_amd.sfno import SphericalFourierNeuralOperatorNet

    d = dataclasses.asdict(cfg)
    params = types.SimpleNamespace(
        operator_type=d["operator_type"], scale_factor=d["scale_factor"], embed_dim=d["embed_dim"],
        num_layers=d["num_layers"], hard_thresholding_fraction=d["hard_thresholding_fraction"],
        normalization_layer=d["normalization_layer"], use_mlp=d["use_mlp"],
        activation_function=d["activation_function"], encoder_layers=d["encoder_layers"],
        pos_embed=d["pos_embed"], big_skip=d["big_skip"], data_grid=d["data_grid"],
    )
    net = SphericalFourierNeuralOperatorNet(params=params, in_chans=cfg.in_chans, out_chans=cfg.out_chans,
                                            img_shape=tuple(cfg.img_shape), mlp_ratio=cfg.mlp_ratio)
    net.load_state_dict({k: v for k, v in state.items()}, strict=True)
    if precision is not None:
        net.set_precision(precision)
    return net.to(device).eval()
