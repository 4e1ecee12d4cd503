"""Shared helpers of the parity tests."""

import dataclasses
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=False, map_location="cpu")   # (the reference-held csfno checkpoint was saved from a GPU)


def rel_max(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| / max|b|  (the per-step tolerance of BASELINE.json's north_star is 1e-5 in this norm)."""
    a = a.detach().cpu()
    b = b.detach().cpu()
    if a.is_complex() or b.is_complex():
        a = torch.view_as_real(a.to(torch.complex128))
        b = torch.view_as_real(b.to(torch.complex128))
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300))


def rel_max_per_channel(a: torch.Tensor, b: torch.Tensor, channel_dim: int = 1) -> float:
    """worst channel of max|a - b| over the channel / max|b| over THAT channel: a max norm over the whole tensor does not see
    an output channel of small magnitude"""
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    dims = [d for d in range(b.dim()) if d != channel_dim % b.dim()]
    err = (a - b).abs().amax(dim=dims)
    return float((err / b.abs().amax(dim=dims).clamp_min(1e-300)).max())


def assert_net_close(y: torch.Tensor, ref: torch.Tensor, tol: float = 1e-5, channel_factor: float = 3.0, channel_dim: int = 1, what=""):
    """the network-level bar of the GPU parity tests: max|err| / max|ref| <= tol over the whole output (north_star: 1e-5 per step)
    AND channel by channel within channel_factor x tol of the channel's own maximum (the first alone lets a
    small-magnitude channel through)"""
    whole = rel_max(y, ref)
    assert whole <= tol, (what, "whole-tensor", whole)
    if y.dim() > channel_dim:
        chan = rel_max_per_channel(y, ref, channel_dim)
        assert chan <= channel_factor * tol, (what, "per-channel", chan)


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
        operator_type=d["operator_type"], scale_factor=d["scale_factor"], residual_filter_factor=d["residual_filter_factor"],
        embed_dim=d["embed_dim"],
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


def checkpoint_case(golden: dict, name: str) -> dict:
    """one case of tests/golden/gen_checkpoint.pt; "<base>_override" re-uses the state / inputs of <base> with the
    StepperOverrideConfig fields stored under "override" and its own reference rollout"""
    case = golden[name]
    if "override" in case:
        base = golden[name[: -len("_override")]]
        case = {**base, "steps": case["steps"], "override": case["override"]}
    return case


def checkpoint_override(case: dict):
    import ace_amd
    return ace_amd.StepperOverrideConfig(**case["override"]) if "override" in case else None


def oracle_checkpoint_rollout(case: dict, dtype=torch.float32):
    """CPU rollout of one case of tests/golden/gen_checkpoint.pt: the oracle network (oracle/sfno.py) evaluated in
    `dtype` under the host-side step logic (ace_amd.step.step_with_adjustments: normalise, residual, corrector, ocean,
    prescribed prognostics - plain torch).  dtype=float32 restates what the reference stepper computed for the fixture;
    dtype=float64 gives the exact-arithmetic result both are approximations of (the conditioning floor of each field)."""
    import ace_amd
    from ace_amd.step import step_with_adjustments
    from oracle.sfno import SFNOConfig, SFNOOracle

    loaded = ace_amd.load_stepper(case["state"], override_config=checkpoint_override(case), device="cpu")
    step, cfg = loaded.stepper._step_obj, loaded.config
    fields = {f.name for f in dataclasses.fields(SFNOConfig)}
    ocfg = SFNOConfig(in_chans=len(cfg.in_names), out_chans=len(cfg.out_names), img_shape=loaded.dataset_info.img_shape,
                      **{k: v for k, v in cfg.builder.config.items() if k in fields})
    weights = {k: v for k, v in case["state"]["step"]["module"].items() if isinstance(v, torch.Tensor)}
    net = SFNOOracle(ocfg, weights, dtype=dtype)
    means = {k: v.cpu().to(dtype) for k, v in step.normalizer.means.items()}
    stds = {k: v.cpu().to(dtype) for k, v in step.normalizer.stds.items()}

    class Normalizer:
        def normalize(self, d):
            return {k: (v - means[k]) / stds[k] for k, v in d.items()}

        def denormalize(self, d):
            return {k: v * stds[k] + means[k] for k, v in d.items()}

    def network_calls(input_norm):
        y = net(torch.stack([input_norm[n] for n in cfg.in_names], dim=1))
        return {n: y[:, i] for i, n in enumerate(cfg.out_names)}

    forcing = {k: v.to(dtype) for k, v in case["forcing"].items()}
    state = {k: v[:, 0].to(dtype) for k, v in case["ic"].items()}
    input_only = [n for n in cfg.in_names if n not in cfg.out_names]
    stepper_state, outs = None, []
    for s in range(len(case["steps"])):
        f = {k: forcing[k][:, s + 1 if k in cfg.next_step_forcing_names else s] for k in input_only}
        nxt = {k: forcing[k][:, s + 1] for k in step.next_step_input_names}
        r = step_with_adjustments({**state, **f}, nxt, network_calls, Normalizer(), cfg.residual_prediction,
                                  cfg.prognostic_names, cfg.prescribed_prognostic_names, stepper_state,
                                  step._corrector, step._ocean)
        stepper_state = r.stepper_state
        state = {k: r.output[k] for k in cfg.prognostic_names}
        outs.append(r.output)
    return outs


def conditioning_floor(case: dict):
    """per step, per field: max|reference fp32 - exact| / max|reference| - how far the reference's own fp32 rollout is
    from exact arithmetic (fields that close a budget are differences of nearly cancelling terms and sit far above eps)."""
    exact = oracle_checkpoint_rollout(case, torch.float64)
    return [{k: float((want.double() - exact[s][k]).abs().max() / want.double().abs().max()) for k, want in st.items()}
            for s, st in enumerate(case["steps"])]


def csfno_reference_checkpoint_case():
    """The reference-HELD, RNG-free golden of the conditional SFNO (fme/core/models/conditional_sfno/test_sfnonet.py:162-191,
    testdata/test_sfnonet_checkpoint_{input,output}.pt, copied as tests/golden/ref_csfno_checkpoint_*.pt): a raw conditional
    SphericalFourierNeuralOperatorNet (embed 16, 2 layers, 9 x 18 equiangular, big skip, pos_embed) with a SCALAR context embedding
    (8), labels (4) and a 16-channel 2-D noise context, the spectral filter stored in the pre-grouping layout (1, in, out, L, 2).

    The accelerated family (NoiseConditionedSFNO, stochastic_sfno.py) has no scalar embedding, so the case is mapped onto it
    exactly: the scalar embedding and the labels are both per-sample vectors entering every norm through a Linear layer
    (layers.py:262-318), i.e. ONE 12-dimensional label vector with the concatenated weights; with a scalar embedding the
    reference's scale has no leading 1 (scale = W_scale(e) + ..., layers.py:270-281), which goes into the merged bias.
    Returns dict(kwargs, labels (12 names), state (ace_amd NoiseConditionedSFNO names), x, label_vector, noise, y)."""
    d = load_golden("ref_csfno_checkpoint_input.pt")
    y = load_golden("ref_csfno_checkpoint_output.pt")
    x = d.pop("x")
    ctx = d.pop("context")
    state = {}
    for k, v in d.items():
        if ".W_scale." in k or ".W_bias." in k or "_labels." in k:
            continue
        state["conditional_model." + k] = v.clone()
    for k in list(d):
        if k.endswith(".W_scale.weight") or k.endswith(".W_bias.weight"):
            which = "scale" if k.endswith(".W_scale.weight") else "bias"
            pre = k[: -len(f"W_{which}.weight")]
            w = torch.cat([d[pre + f"W_{which}.weight"], d[pre + f"W_{which}_labels.weight"]], dim=1)
            b = d[pre + f"W_{which}.bias"] + d[pre + f"W_{which}_labels.bias"] - (1.0 if which == "scale" else 0.0)
            state["conditional_model." + pre + f"W_{which}_labels.weight"] = w
            state["conditional_model." + pre + f"W_{which}_labels.bias"] = b
    kwargs = dict(embed_dim=16, num_layers=2, noise_embed_dim=16, noise_type="gaussian", data_grid="equiangular",
                  filter_type="linear", pos_embed=True, big_skip=True, encoder_layers=1, use_mlp=True)
    labels = [f"context_{i:02d}" for i in range(12)]
    return dict(kwargs=kwargs, labels=labels, state=state, x=x, label_vector=torch.cat([ctx["embedding_scalar"], ctx["labels"]], dim=1),
                noise=ctx["noise"], y=y)


def csfno_block_case(name):
    """The reference's block-level regression golden `name` (csfno_block / csfno_block_8_groups; output held by the reference,
    parameters and inputs re-drawn by tests/golden/make_golden_csfno_block.py) mapped onto a ONE-block NoiseConditionedSFNO the
    C ABI can run: identity encoder / decoder convolutions (encoder_layers 0, no big skip, no pos_embed), the given positional
    context as the wrapper's learned one, lobatto data grid, and the filter padded by a zero row from the benchmark's lmax = 8 to
    the network's lmax = nlat = 9 (coefficients are computed per degree, so the first 8 are unchanged and the 9th is filtered to
    zero).  The network adds the identity outer skip the stand-alone block does not have (conditional_sfno/sfnonet.py:429-435 with
    outer_skip=None vs 'identity' at :653): block output = network output - norm0(x), see `csfno_block_norm0`."""
    g = load_golden("gen_csfno_block.pt")[name]
    C, G = 16, g["groups"]
    kwargs = dict(embed_dim=C, num_layers=1, encoder_layers=0, big_skip=False, pos_embed=False, noise_embed_dim=4,
                  noise_type="gaussian", context_pos_embed_dim=2, filter_num_groups=G, data_grid="lobatto", mlp_ratio=2.0,
                  use_mlp=True, affine_norms=False)
    labels = ["l0", "l1", "l2"]
    state = {}
    for k, v in g["state"].items():
        if k == "filter.filter.weight":          # (G, L = 8, C/G, C/G, 2) -> L = 9
            v = torch.cat([v, torch.zeros(G, 1, *v.shape[2:])], dim=1)
        state["conditional_model.blocks.0." + k] = v.clone()
    eye = torch.eye(C)[:, :, None, None]
    state["conditional_model.encoder.0.weight"] = eye.clone()
    state["conditional_model.decoder.0.weight"] = eye.clone()
    state["pos_embed"] = g["embedding_pos"].clone()                       # (1, 2, 9, 18): the wrapper's learned positional context
    state["label_pos_embed"] = torch.zeros(len(labels), 2, 9, 18)         # ... without its label interaction
    return dict(kwargs=kwargs, labels=labels, state=state, x=g["x"], noise=g["noise"], label_vector=g["labels"],
                embedding_pos=g["embedding_pos"], y=g["held_output"], block_state=g["state"])


def csfno_block_norm0(case, dtype=torch.float64):
    """norm0(x, context) of the block case: the reference formula (conditional_sfno/layers.py:262-318) in torch, fp64"""
    s = {k: v.to(dtype) for k, v in case["block_state"].items()}
    x, noise, lab, pos = (case[k].to(dtype) for k in ("x", "noise", "label_vector", "embedding_pos"))
    F = torch.nn.functional
    scale = 1.0 + F.conv2d(noise, s["norm0.W_scale_2d.weight"]) + F.linear(lab, s["norm0.W_scale_labels.weight"], s["norm0.W_scale_labels.bias"])[:, :, None, None] \
        + F.conv2d(pos, s["norm0.W_scale_pos.weight"])
    bias = F.conv2d(noise, s["norm0.W_bias_2d.weight"]) + F.linear(lab, s["norm0.W_bias_labels.weight"], s["norm0.W_bias_labels.bias"])[:, :, None, None] \
        + F.conv2d(pos, s["norm0.W_bias_pos.weight"])
    mean = x.mean(dim=1, keepdim=True)
    var = x.var(dim=1, keepdim=True, unbiased=False)
    return (x - mean) * torch.rsqrt(var + 1e-5) * scale + bias
