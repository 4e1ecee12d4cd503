"""CPU-only checks (no GPU): the C-ABI library loads and exports every symbol include/ace_sfno.h declares,
the native fp64 table builder agrees with the oracle, and the host-side mirror of the reference interfaces
(registry, builder, packer, normaliser, step/stepper bookkeeping) behaves like the reference."""

import ctypes
import os
import re

import numpy as np
import pytest
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ace_amd import build, _lib
    build.build()  # no-op when the in-tree .so is current (hipcc cross-compiles without a GPU)
    return _lib.lib()


def test_library_exports_every_declared_symbol(lib):
    from ace_amd import _lib
    header = open(os.path.join(ROOT, "include", "ace_sfno.h")).read()
    declared = set(re.findall(r"\b(ace_[a-z0-9_]+)\s*\(", header))
    declared -= {"ace_sht_plan", "ace_sfno", "ace_sfno_config"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ace_version() >= 100


def test_no_torch_types_in_the_abi():
    header = open(os.path.join(ROOT, "include", "ace_sfno.h")).read()
    code = re.sub(r"/\*.*?\*/", "", header, flags=re.S)  # declarations only, comments stripped
    assert "torch" not in code.lower() and "at::" not in code and "Tensor" not in code


@pytest.mark.parametrize("nlat,nlon,lmax,mmax,grid", [
    (9, 18, 0, 0, "lobatto"), (9, 18, 9, 10, "equiangular"), (6, 12, 0, 0, "legendre-gauss"),
    (180, 360, 180, 181, "legendre-gauss"), (45, 90, 30, 31, "legendre-gauss"), (13, 27, 0, 0, "equiangular"),
])
def test_native_tables_match_oracle(lib, nlat, nlon, lmax, mmax, grid):
    import oracle
    from oracle.quadrature import quadrature
    L = lmax or (nlat - 1 if grid == "lobatto" else nlat)
    M = mmax or nlon // 2 + 1

    def table(which):
        out = np.zeros(nlat, dtype=np.float64) if which in (2, 3) else np.zeros((M, L, nlat), dtype=np.float32)
        rc = lib.ace_sht_tables_host(nlat, nlon, lmax, mmax, grid.encode(), which, out.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0, lib.ace_last_error()
        return out

    x, w = quadrature(grid, nlat)
    np.testing.assert_allclose(table(2), x, rtol=0, atol=1e-15)
    np.testing.assert_allclose(table(3), w, rtol=0, atol=1e-14)
    f = oracle.RealSHT(nlat, nlon, lmax or None, mmax or None, grid)
    i = oracle.InverseRealSHT(nlat, nlon, lmax or None, mmax or None, grid)
    # fp32 tables: identical up to one rounding of the fp64 value
    np.testing.assert_allclose(table(0), f.weights.numpy(), rtol=0, atol=1e-9)
    np.testing.assert_allclose(table(1), i.pct.numpy(), rtol=0, atol=1e-9)


def test_table_errors(lib):
    buf = np.zeros(16, dtype=np.float64)
    rc = lib.ace_sht_tables_host(4, 8, 0, 0, b"nonsense", 2, buf.ctypes.data_as(ctypes.c_void_p))
    assert rc == -1 and b"Unknown quadrature" in lib.ace_last_error()


def test_ops_fail_loudly_without_gpu():
    """No CPU fallback: a CPU tensor is an error, not a silent slow path."""
    import ace_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        ace_amd.RealSHT(6, 12, grid="legendre-gauss")(torch.zeros(1, 6, 12))
    net = ace_amd.SphericalFourierNeuralOperatorNet(params=None, embed_dim=8, num_layers=1, img_shape=(6, 12))
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 2, 6, 12))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ace_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                # ... nor the C-ABI emulations of the test suite (tests/_fake_sfno.py, tests/_fake_hpx.py), nor anything under tests/
                assert "_fake_" not in src and "import tests" not in src and "from tests" not in src, f


# ----------------------------------------------------------------------------- registry / builder
def test_registry_and_builder_defaults():
    import ace_amd
    sel = ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet", config={"embed_dim": 8, "num_layers": 2})
    assert sel.config["operator_type"] == "diagonal" and sel.config["data_grid"] == "legendre-gauss"  # sfno.py:23,42
    assert "SphericalFourierNeuralOperatorNet" in ace_amd.ModuleSelector.get_available_types()
    mod = sel.build(3, 4, ace_amd.DatasetInfo((6, 12)))
    sd = mod.torch_module.state_dict()
    assert sd["pos_embed"].shape == (1, 8, 6, 12)
    assert sd["blocks.1.filter.filter.weight"].shape == (8, 8, 6, 7, 2)
    assert sd["decoder.0.weight"].shape == (8, 11, 1, 1) and sd["decoder.2.weight"].shape == (4, 8, 1, 1)
    state = mod.get_state()
    assert state["label_encoding"] is None
    mod.load_state(state)
    with pytest.raises(KeyError):
        ace_amd.ModuleSelector(type="NoSuchNet", config={})
    with pytest.raises(ValueError):
        ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet", config={"bogus": 1})
    with pytest.raises(ValueError):
        ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet", config={}, conditional=True)
    with pytest.raises(ValueError):
        sel.build(3, 4, ace_amd.DatasetInfo((6, 12), all_labels={"x"}))


def test_state_dict_matches_reference_names_at_ace2_shape():
    """exact names/shapes so reference checkpoints load strictly (meta device: no memory)."""
    import ace_amd
    with torch.device("meta"):
        net = ace_amd.SphericalFourierNeuralOperatorNet(
            params=ace_amd.SphericalFourierNeuralOperatorBuilder(embed_dim=384, num_layers=8, operator_type="dhconv"),
            in_chans=44, out_chans=50, img_shape=(180, 360))
    sd = net.state_dict()
    expect = {"pos_embed": (1, 384, 180, 360), "encoder.0.weight": (384, 44, 1, 1), "encoder.0.bias": (384,),
              "encoder.2.weight": (384, 384, 1, 1), "blocks.7.norm0.weight": (384,), "blocks.7.norm1.bias": (384,),
              "blocks.0.filter.filter.weight": (384, 384, 180, 2), "blocks.0.filter.filter.bias": (1, 384, 1, 1),
              "blocks.3.inner_skip.weight": (384, 384, 1, 1), "blocks.3.inner_skip.bias": (384,),
              "blocks.5.mlp.fwd.0.weight": (768, 384, 1, 1), "blocks.5.mlp.fwd.0.bias": (768,),
              "blocks.5.mlp.fwd.2.weight": (384, 768, 1, 1), "blocks.5.mlp.fwd.2.bias": (384,),
              "decoder.0.weight": (384, 428, 1, 1), "decoder.0.bias": (384,), "decoder.2.weight": (50, 384, 1, 1)}
    for k, shp in expect.items():
        assert tuple(sd[k].shape) == shp, k
    assert sum(v.numel() for v in sd.values()) == 455_831_552 + 0 or sum(v.numel() for v in sd.values()) > 4.5e8


def test_layer_norm_state_dict_has_the_reference_shapes():
    """normalization_layer = "layer_norm" (sfnonet.py:584-592): nn.LayerNorm over (H, W) - norm0 / norm1 carry (H, W) weights and
    biases under the reference's names (the real reference net loads oracle.init_state's tensors strictly: make_golden_layer_norm.py)."""
    import ace_amd
    from oracle.sfno import SFNOConfig, init_state
    with torch.device("meta"):
        net = ace_amd.SphericalFourierNeuralOperatorNet(
            params=ace_amd.SphericalFourierNeuralOperatorBuilder(embed_dim=16, num_layers=2, operator_type="dhconv", normalization_layer="layer_norm"),
            in_chans=5, out_chans=7, img_shape=(12, 24))
    sd = net.state_dict()
    st = init_state(SFNOConfig(in_chans=5, out_chans=7, img_shape=(12, 24), embed_dim=16, num_layers=2, operator_type="dhconv",
                               normalization_layer="layer_norm"), seed=0)
    assert set(sd) == set(st)
    for k, v in st.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    assert tuple(sd["blocks.1.norm1.weight"].shape) == (12, 24)


def test_seeded_init_reproduces_reference_draw_order(golden_dir):
    """torch.manual_seed(0) + construction gives the reference's parameters bit for bit
    (fme/ace/models/modulus/test_sfnonet.py:13-36; state captured from the reference itself)."""
    import ace_amd
    d = torch.load(os.path.join(golden_dir, "gen_modulus_sfnonet_case.pt"), weights_only=False)
    torch.manual_seed(0)
    net = ace_amd.SphericalFourierNeuralOperatorNet(params=None, embed_dim=16, num_layers=2, img_shape=(9, 18),
                                                     in_chans=2, out_chans=3)
    x = torch.randn(4, 2, 9, 18)
    sd = net.state_dict()
    assert list(sd) == list(d["state"])
    for k in sd:
        assert torch.equal(sd[k], d["state"][k]), k
    assert torch.equal(x, d["x"])


def test_unsupported_options_are_rejected():
    import ace_amd
    Net = ace_amd.SphericalFourierNeuralOperatorNet
    with pytest.raises(NotImplementedError):
        Net(params=None, filter_type="non-linear", img_shape=(6, 12))
    with pytest.raises(ValueError):
        Net(params=None, operator_type="block-diagonal", img_shape=(6, 12))
    with pytest.raises(ValueError):
        Net(params=None, scale_factor=0, img_shape=(6, 12))
    with pytest.raises(ValueError):
        Net(params=None, activation_function="tanh", img_shape=(6, 12))
    # built since round 6 (sfnonet.py:467-515, 584-592): the reference's names and shapes
    net = Net(params=None, scale_factor=2, residual_filter_factor=2, normalization_layer="layer_norm", img_shape=(12, 24), embed_dim=8, num_layers=3,
              in_chans=2, out_chans=2)
    sd = net.state_dict()
    assert tuple(sd["blocks.0.norm0.weight"].shape) == (12, 24) and tuple(sd["blocks.0.norm1.weight"].shape) == (6, 12)
    assert tuple(sd["blocks.1.norm0.weight"].shape) == (6, 12) and tuple(sd["blocks.2.norm1.bias"].shape) == (12, 24)
    assert tuple(sd["blocks.1.filter.filter.weight"].shape) == (8, 8, 6, 7, 2) and tuple(sd["pos_embed"].shape) == (1, 8, 12, 24)   # "diagonal": (L, M) of the inner grid


# ----------------------------------------------------------------------------- packer / normaliser / step bookkeeping
def test_packer_and_normalizer():
    from ace_amd.normalizer import StandardNormalizer
    from ace_amd.packer import DataShapesNotUniform, Packer
    p = Packer(["a", "b"])
    t = {"a": torch.zeros(2, 3, 4), "b": torch.ones(2, 3, 4), "c": torch.full((2, 3, 4), 2.0)}
    x = p.pack({k: t[k] for k in ["a", "b"]}, axis=-3)
    assert x.shape == (2, 2, 3, 4) and torch.equal(x[:, 1], t["b"])
    assert torch.equal(p.unpack(x, axis=-3)["a"], t["a"])
    with pytest.raises(DataShapesNotUniform):
        p.pack({"a": torch.zeros(1), "b": torch.zeros(2)})
    n = StandardNormalizer({"a": torch.tensor(1.0), "b": torch.tensor(2.0)}, {"a": torch.tensor(2.0), "b": torch.tensor(4.0)},
                           device="cpu")
    out = n.normalize(t)
    assert set(out) == {"a", "b"}                      # names without constants are dropped
    assert torch.allclose(n.denormalize(out)["b"], t["b"])
    assert StandardNormalizer.from_state(n.get_state()).get_state() == n.get_state()


def test_step_config_validation_and_names():
    import ace_amd
    from ace_amd.step import NormalizationConfig
    norm = NormalizationConfig(means={k: 0.0 for k in "abcf"}, stds={k: 1.0 for k in "abcf"})
    b = ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet", config={"embed_dim": 8, "num_layers": 1})
    cfg = ace_amd.SingleModuleStepConfig(builder=b, in_names=["f", "a", "b"], out_names=["b", "c", "a"],
                                         normalization=norm, next_step_forcing_names=["f"])
    assert cfg.prognostic_names == ["b", "a"]            # out_names order
    assert set(cfg.next_step_input_names) == {"f"}
    with pytest.raises(ValueError):
        ace_amd.SingleModuleStepConfig(builder=b, in_names=["a"], out_names=["a"], normalization=norm,
                                       next_step_forcing_names=["a"])
    with pytest.raises(ValueError):
        ace_amd.SingleModuleStepConfig(builder=b, in_names=["a"], out_names=["a"], normalization=norm,
                                       prescribed_prognostic_names=["zz"])
    with pytest.raises(NotImplementedError):
        ace_amd.SingleModuleStepConfig(builder=b, in_names=["a"], out_names=["a"], normalization=norm, ocean=object())
    with pytest.raises(ValueError):           # built since round 3 (tests/test_step_options.py): a malformed state is a ValueError
        ace_amd.SingleModuleStepConfig(builder=b, in_names=["a"], out_names=["a"], normalization=norm,
                                       secondary_decoder={"x": 1})
    with pytest.raises(NotImplementedError):
        ace_amd.SingleModuleStepConfig(builder=b, in_names=["a"], out_names=["a"], normalization=norm,
                                       global_mean_removal={"x": 1})


def test_stepper_loop_bookkeeping_with_a_stub_module():
    """The loop semantics of predict_generator (state feedback, forcing index, next-step forcing) with the
    network replaced by a CPU stub, as the reference tests do with 'prebuilt' modules
    (fme/ace/stepper/test_single_module.py:1146-1223)."""
    import ace_amd
    from ace_amd.registry import Module
    from ace_amd.step import NormalizationConfig, SingleModuleStep
    from oracle import stepper as ostep

    class AddOne(torch.nn.Module):  # (B, 3, H, W) -> (B, 2, H, W): out0 = in1 + in0, out1 = in2 + 1
        def forward(self, x):
            return torch.stack([x[:, 1] + x[:, 0], x[:, 2] + 1.0], dim=1)

    names = ["f", "g", "p", "d"]
    norm = NormalizationConfig(means={k: 0.5 for k in names}, stds={k: 2.0 for k in names})
    cfg = ace_amd.SingleModuleStepConfig(
        builder=ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet", config={"embed_dim": 4, "num_layers": 1}),
        in_names=["f", "p", "g"], out_names=["p", "d"], normalization=norm, next_step_forcing_names=["g"])
    step = SingleModuleStep(cfg, ace_amd.DatasetInfo((4, 8)), cfg.normalization.build(names), device="cpu")
    step.module = Module(AddOne(), None)
    stepper = ace_amd.Stepper(step)
    g = torch.Generator().manual_seed(0)
    ic = {"p": torch.randn(2, 1, 4, 8, generator=g)}
    forcing = {k: torch.randn(2, 4, 4, 8, generator=g) for k in ["f", "g"]}
    out, state = stepper.predict(ic, forcing)
    means = {k: torch.tensor(0.5) for k in names}
    stds = {k: torch.tensor(2.0) for k in names}
    ref = ostep.predict(AddOne(), ic, forcing, 3, ["f", "p", "g"], ["p", "d"], means, stds, next_step_forcing_names=["g"])
    for k in ["p", "d"]:
        torch.testing.assert_close(out[k], torch.stack([o[k] for o in ref], 1))
    assert torch.equal(state["p"], out["p"][:, -1:])


def test_stepper_derives_the_insolation_from_the_window_times():
    """Stepper.predict with StepperConfig.derived_forcings (single_module.py:1202-1203): the forcing the network reads is computed
    from the time axis - same rollout as with that forcing handed in, an error when the times are missing."""
    import datetime

    import ace_amd
    from ace_amd.derived_forcings import ForcingWindow
    from ace_amd.registry import Module
    from ace_amd.step import NormalizationConfig, SingleModuleStep
    from ace_amd.timeaxis import TimeAxis

    class Mix(torch.nn.Module):  # in: [sun, p] -> out: [p]
        def forward(self, x):
            return (0.5 * x[:, 1] + 0.001 * x[:, 0]).unsqueeze(1)

    names = ["sun", "p"]
    norm = NormalizationConfig(means={k: 0.0 for k in names}, stds={k: 1.0 for k in names})
    cfg = ace_amd.SingleModuleStepConfig(
        builder=ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet", config={"embed_dim": 4, "num_layers": 1}),
        in_names=["sun", "p"], out_names=["p"], normalization=norm, next_step_forcing_names=["sun"])
    info = ace_amd.DatasetInfo((4, 8), lat=torch.linspace(-67.5, 67.5, 4), lon=torch.arange(8.0) * 45.0)
    step = SingleModuleStep(cfg, info, cfg.normalization.build(names), device="cpu")
    step.module = Module(Mix(), None)
    derived = {"insolation": {"insolation_name": "sun", "solar_constant": {"value": 1360.0}}}
    stepper = ace_amd.Stepper(step, derived_forcings=derived, dataset_info=info)
    assert stepper.forcing_names_from_data() == []
    ic = {"p": torch.ones(2, 1, 4, 8)}
    time = TimeAxis.regular((2021, 6, 1), datetime.timedelta(hours=6), 4, 2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                     # (the 4 x 8 grid's latitude range looks like radians to the check)
        out, _ = stepper.predict(ic, {}, n_forward_steps=3, time=time)
        sun = stepper.forcing_deriver({}, time)["sun"]
        again, _ = stepper.predict(ic, ForcingWindow({}, time), n_forward_steps=3)
    explicit, _ = stepper.predict(ic, {"sun": sun}, compute_derived_forcings=False)
    assert out["p"].shape == (2, 3, 4, 8) and torch.equal(out["p"], explicit["p"]) and torch.equal(again["p"], out["p"])
    assert (out["p"][:, 0] - 0.5).abs().max() > 0.1          # the sun was seen
    with pytest.raises(ValueError, match="time axis"):
        stepper.predict(ic, {}, n_forward_steps=3)


def test_counter_file_is_tied_to_a_build():
    """bench.py takes `roofline.traffic` / `mfma_busy_pmc` from the newest profiles/rNN_pmc_traffic.json only when that file was collected on
    the library that is loaded now or on one built from the same kernel sources and flags (tools/pmc_collect.sh stamps both
    hashes); otherwise it says why it reports null.  The stamps must be present and well-formed."""
    import json
    import bench
    from ace_amd import build
    with open(bench.PMC_FILE) as f:
        d = json.load(f)
    for k in ("_lib_sha256", "_src_sha256"):
        assert isinstance(d.get(k), str) and len(d[k]) == 64 and int(d[k], 16) >= 0, k
    assert len(build.source_sha256()) == 64
    data, reason = bench.measured_counters()
    assert (reason is None and ("f16x3", "mlp.fc2+outer_skip") in data) or (isinstance(reason, str) and data == {})
