"""Autoregressive stepper loop on CPU (oracle; test infrastructure only).

Restates, for the default configuration (no corrector corrections, no ocean,
no residual prediction, no masks):

* ``StandardNormalizer``          fme/core/normalizer.py:213-242
* ``Packer.pack/unpack``          fme/core/packer.py:45-52
* ``SingleModuleStep.step`` / ``step_with_adjustments``
                                  fme/core/step/single_module.py:396-449, 595-733
* ``Stepper.predict_generator``   fme/ace/stepper/single_module.py:1124-1167
"""

from typing import Callable, Dict, List

import torch


def normalize(tensors, means, stds):
    # normalizer.py:213-221
    return {k: (t - means[k]) / stds[k] for k, t in tensors.items()}


def denormalize(tensors, means, stds):
    # normalizer.py:230-236
    return {k: t * stds[k] + means[k] for k, t in tensors.items()}


def pack(tensors, names, axis=-3):
    # packer.py:45-47
    return torch.cat([tensors[n].unsqueeze(axis) for n in names], dim=axis)


def unpack(tensor, names, axis=-3):
    # packer.py:50-52
    return {n: tensor.select(axis, index=i) for i, n in enumerate(names)}


def step(network: Callable, inp: Dict[str, torch.Tensor], in_names: List[str], out_names: List[str],
         means, stds):
    """One step: normalize -> pack -> network -> unpack -> denormalize."""
    input_norm = normalize({k: inp[k] for k in in_names}, means, stds)
    x = pack(input_norm, in_names)
    y = network(x)
    out_norm = unpack(y, out_names)
    return denormalize(out_norm, means, stds)


def predict(network: Callable, ic: Dict[str, torch.Tensor], forcing: Dict[str, torch.Tensor],
            n_forward_steps: int, in_names: List[str], out_names: List[str], means, stds,
            next_step_forcing_names=()):
    """fme/ace/stepper/single_module.py:1135-1167.

    ic: name -> (B, 1, H, W) prognostic state; forcing: name -> (B, T+1, H, W).
    Returns list of per-step output dicts (name -> (B, H, W))."""
    input_only = set(in_names) - set(out_names)
    state = {k: v.squeeze(1) for k, v in ic.items()}
    outs = []
    for s in range(n_forward_steps):
        input_forcing = {
            k: (forcing[k][:, s] if k not in next_step_forcing_names else forcing[k][:, s + 1])
            for k in input_only
        }
        input_data = {**state, **input_forcing}
        out = step(network, input_data, in_names, out_names, means, stds)
        state = out
        outs.append(out)
    return outs
