"""HEALPix variant, CPU side: the native padding gather table (ace_hpx_pad_table_host, csrc/healpix.hip - host code) against
the reference's HEALPixPadding outputs (tests/golden/gen_healpix.pt, emitted by the reference itself), and the registry /
configuration / state_dict surface of the HEALPixUNet builder."""
import os

import numpy as np
import pytest
import torch

import ace_amd
from ace_amd import _lib

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gen_healpix.pt")


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD, map_location="cpu", weights_only=False)


def test_pad_table_matches_reference_padding(gold):
    """every halo cell is one cell of a neighbouring face (rotated at the poles) or the mean of two (equatorial corners):
    gathering with the native table reproduces the reference's padded faces BIT FOR BIT, for padding 1 .. nside"""
    L = _lib.lib()
    for (nside, p), d in gold["padding"].items():
        m = nside + 2 * p
        ia = np.zeros(12 * m * m, dtype=np.int32)
        ib = np.zeros_like(ia)
        assert L.ace_hpx_pad_table_host(nside, p, ia.ctypes.data, ib.ctypes.data) == 0, L.ace_hpx_last_error()
        x = d["x"].reshape(2, 12, 3, nside, nside)

        def gather(idx):
            idx = torch.from_numpy(idx.astype(np.int64)).reshape(12, m, m)
            return x[:, idx >> 24, :, (idx >> 12) & 4095, idx & 4095].permute(3, 0, 4, 1, 2)     # (12, m, m, 2, 3) -> (2, 12, 3, m, m)

        a, b = gather(ia), gather(ib)
        got = torch.where(torch.from_numpy(ia == ib).reshape(1, 12, 1, m, m), a, 0.5 * a + 0.5 * b)
        assert torch.equal(got.reshape(24, 3, m, m), d["padded"]), (nside, p)
        assert int((ia != ib).sum()) == 8 * p              # the diagonals of the two synthetic corners of the 4 equatorial faces
    assert L.ace_hpx_pad_table_host(4, 5, ia.ctypes.data, ib.ctypes.data) != 0          # padding > nside


def test_builder_registry_and_state_dict(gold):
    """ModuleSelector(type="HEALPixUNet") builds a module whose state_dict has the reference's names and shapes (strict load)."""
    for name, g in gold["unet"].items():
        case = g["case"]
        sel = ace_amd.ModuleSelector(type="HEALPixUNet", config=case["config"])
        mod = sel.build(case["n_in"], case["n_out"], ace_amd.DatasetInfo((case["nside"], case["nside"])))
        net = mod.torch_module
        ref_sd = g["state_dict"]
        sd = net.state_dict()
        assert list(sd) == list(ref_sd), name
        assert all(tuple(sd[k].shape) == tuple(ref_sd[k].shape) for k in sd), name
        net.load_state_dict(ref_sd, strict=True)


def test_unbuilt_variants_are_loud(gold):
    cfg = dict(gold["unet"]["basic_maxpool"]["case"]["config"])
    iso = ace_amd.ModuleSelector(type="HEALPixUNet", config={**cfg, "hpx_padding_mode": "isolatitude"}).build(3, 2, ace_amd.DatasetInfo((8, 8)))
    assert any(getattr(m, "mode", None) == "isolatitude" for m in iso.torch_module.modules())      # built since round 3 (own table)
    enc = dict(cfg["encoder"])
    enc["down_sampling_block"] = {"block_type": "DealiasedDownsample"}       # built since round 3 (same operators, tests/test_healpix_resamplers.py)
    ace_amd.ModuleSelector(type="HEALPixUNet", config={**cfg, "encoder": enc}).build(3, 2, ace_amd.DatasetInfo((8, 8)))
    dec = dict(cfg["decoder"])
    dec["up_sampling_block"] = {"block_type": "SmoothedInterpolateConv", "upsample_mode": "bilinear"}      # built since round 6 (ace_hpx_upsample2)
    ace_amd.ModuleSelector(type="HEALPixUNet", config={**cfg, "decoder": dec}).build(3, 2, ace_amd.DatasetInfo((8, 8)))
    dec["up_sampling_block"] = {"block_type": "SmoothedInterpolateConv", "upsample_mode": "bicubic"}
    with pytest.raises(NotImplementedError, match="bilinear"):
        ace_amd.ModuleSelector(type="HEALPixUNet", config={**cfg, "decoder": dec}).build(3, 2, ace_amd.DatasetInfo((8, 8)))
    enc["down_sampling_block"] = {"block_type": "FancyPool"}
    with pytest.raises(ValueError, match="FancyPool"):
        ace_amd.ModuleSelector(type="HEALPixUNet", config={**cfg, "encoder": enc})
    with pytest.raises(ValueError):
        ace_amd.ModuleSelector(type="HEALPixUNet", config={**cfg, "hpx_padding_mode": "nearest"})
    net = ace_amd.ModuleSelector(type="HEALPixUNet", config=cfg).build(3, 2, ace_amd.DatasetInfo((8, 8))).torch_module
    with pytest.raises(RuntimeError, match="MI355X"):
        net(torch.zeros(1, 12, 3, 8, 8))
    with pytest.raises(ValueError, match="5D"):
        net(torch.zeros(12, 3, 8, 8))


@pytest.mark.parametrize("k,dil,cin,W", [(3, 1, 5, 8), (3, 2, 4, 8), (3, 4, 3, 16), (5, 1, 2, 8)])
def test_row_offset_table_is_the_convolution(k, dil, cin, W):
    """The k x k (dilated) convolution runs as ONE contraction over (tap, channel) whose B rows are the padded tensor shifted by
    the tap's offset (GemmArgs::brow): the host side builds the row-offset table and lays the weight out as [cout][(ky, kx, cin)]
    (HEALPixLayer._row_offsets / ._weight).  Evaluate exactly that contraction with torch on the CPU - rows gathered through the
    table from the flat padded buffer, all H * pitch columns incl. the gap columns - against nn.functional.conv2d."""
    from ace_amd.healpix import HEALPixLayer, _round4
    torch.manual_seed(k * 100 + dil)
    layer = HEALPixLayer(layer=torch.nn.Conv2d, in_channels=cin, out_channels=6, kernel_size=k, dilation=dil, hpx_padding_mode="karlbauer")
    p = layer._pad
    assert p == (k - 1) // 2 * dil
    H, m = W, W + 2 * p
    mp = _round4(m) + 4                                   # a pitch wider than the padded face
    xp = torch.zeros(cin, m, mp)
    xp[:, :, :m] = torch.randn(cin, m, m)
    flat = torch.cat([xp.reshape(-1), torch.zeros(16)])   # + ACE_HPX_SLACK_FLOATS
    rows = layer._row_offsets(cin, m, mp, torch.device("cpu"))
    w = layer.base.weight.detach()
    A = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)     # what _weight() hands to ace_hpx_weight_create
    assert rows.shape == (k * k * cin,) and A.shape[1] == rows.numel()
    N = H * mp
    cols = torch.arange(N)
    B = flat[rows[:, None] + cols[None, :]]               # never reads past the slack
    y = (A.double() @ B.double()).reshape(-1, H, mp)[:, :, :W]
    ref = torch.nn.functional.conv2d(xp[None, :, :, :m].double(), w.double(), dilation=dil)[0]
    assert ref.shape == y.shape
    assert float((y - ref).abs().max()) <= 1e-12
    assert int(rows.max()) + N <= flat.numel()


ISO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gen_healpix_isolatitude.pt")


def test_isolatitude_pad_table_matches_reference_gather_indices():
    """isolatitude padding (healpix_paddings.py:613-1140): the table built on the host (ace_amd.healpix.isolatitude_pad_table -
    shifted polar strips, diagonal-mean equatorial corners) names, for EVERY padded cell, exactly the source cells of the
    reference's own precomputed gather index (build_isolatitude_gather_index), and gathering with it reproduces the reference's
    padded faces (the reference evaluates a + (b - a) / 2, the kernel 0.5 a + 0.5 b: one rounding apart)."""
    from ace_amd.healpix import isolatitude_pad_table
    gold = torch.load(ISO, map_location="cpu", weights_only=False)
    for (nside, p), d in gold["padding"].items():
        m = nside + 2 * p
        ia, ib = isolatitude_pad_table(nside, p)
        lin = lambda s: (s >> 24) * nside * nside + ((s >> 12) & 4095) * nside + (s & 4095)      # noqa: E731
        a, b = lin(ia.astype(np.int64)), lin(ib.astype(np.int64))
        i0 = d["index"][0].numpy()
        i1 = d["index"][1].numpy()
        i1 = np.where(i1 < 0, i0, i1)
        assert bool((((a == i0) & (b == i1)) | ((a == i1) & (b == i0))).all()), (nside, p)
        assert int((ia != ib).sum()) == int(d["valid"][1].sum())
        x = d["x"].reshape(2, 12, 3, nside, nside)

        def gather(idx):
            idx = torch.from_numpy(idx.astype(np.int64)).reshape(12, m, m)
            return x[:, idx >> 24, :, (idx >> 12) & 4095, idx & 4095].permute(3, 0, 4, 1, 2)

        ga, gb = gather(ia), gather(ib)
        got = torch.where(torch.from_numpy(ia == ib).reshape(1, 12, 1, m, m), ga, 0.5 * ga + 0.5 * gb).reshape(24, 3, m, m)
        assert float((got - d["padded"]).abs().max()) <= 2.5e-7, (nside, p)
    with pytest.raises(ValueError):
        isolatitude_pad_table(4, 3)                        # padding > nside / 2 (the reference's bound)


def test_isolatitude_builder_surface():
    """hpx_padding_mode="isolatitude" needs nside per level (fme/ace/registry/hpx.py:80-83) and builds padding layers that know it"""
    gold = torch.load(ISO, map_location="cpu", weights_only=False)["unet"]["isolatitude"]
    case = gold["case"]
    net = ace_amd.ModuleSelector(type="HEALPixUNet", config=case["config"]).build(
        case["n_in"], case["n_out"], ace_amd.DatasetInfo((case["nside"], case["nside"]))).torch_module
    assert list(net.state_dict()) == list(gold["state_dict"])
    net.load_state_dict(gold["state_dict"], strict=True)
    pads = [m for m in net.modules() if type(m).__name__ == "HEALPixPadding"]
    assert pads and all(m.mode == "isolatitude" and m._nside in (8, 4) for m in pads)
    bad = dict(case["config"])
    bad.pop("nside")
    with pytest.raises(ValueError):
        ace_amd.ModuleSelector(type="HEALPixUNet", config=bad).build(case["n_in"], case["n_out"], ace_amd.DatasetInfo((8, 8)))


def test_symmetric_convnext_blocks_mirror_the_reference_parameters():
    """SymmetricConvNeXtBlock / Multi_SymmetricConvNeXtBlock (healpix_blocks.py:1214-1402): the reference's state_dict names, order
    and shapes (identity skip iff in_channels == latent_channels, else a 1 x 1 convolution registered BEFORE the conv block)"""
    gold = torch.load(ISO, map_location="cpu", weights_only=False)["unet"]["symmetric"]
    case = gold["case"]
    net = ace_amd.ModuleSelector(type="HEALPixUNet", config=case["config"]).build(
        case["n_in"], case["n_out"], ace_amd.DatasetInfo((case["nside"], case["nside"]))).torch_module
    assert list(net.state_dict()) == list(gold["state_dict"])
    assert all(tuple(v.shape) == tuple(gold["state_dict"][k].shape) for k, v in net.state_dict().items())
    net.load_state_dict(gold["state_dict"], strict=True)
    from ace_amd.healpix import Multi_SymmetricConvNeXtBlock, SymmetricConvNeXtBlock
    blocks = [m for m in net.modules() if isinstance(m, SymmetricConvNeXtBlock)]
    assert any(b.skip_module is None for b in blocks) and any(b.skip_module is not None for b in blocks)
    assert any(isinstance(m, Multi_SymmetricConvNeXtBlock) and len(m.blocks) == 2 for m in net.modules())


def test_checkpoint_of_a_healpix_stepper_loads(gold):
    """load_stepper of a stepper state whose network is the HEALPixUNet and whose dataset_info carries HEALPixCoordinates
    (fme/core/coordinates.py:716-799: face / height / width): the face size is the image shape, the weights load strictly."""
    import datetime

    from ace_amd.checkpoint import load_stepper
    g = gold["unet"]["basic_maxpool"]
    case = g["case"]
    names = [f"v{i}" for i in range(max(case["n_in"], case["n_out"]))]
    state = {"stepper": {
        "config": {"step": {"type": "single_module", "config": {
            "builder": {"type": "HEALPixUNet", "config": case["config"]}, "in_names": names[: case["n_in"]],
            "out_names": names[: case["n_out"]],
            "normalization": {"network": {"means": {n: 0.0 for n in names}, "stds": {n: 1.0 for n in names}}},
            "ocean": None, "corrector": {"type": "atmosphere_corrector", "config": {}}}}},
        "dataset_info": {"horizontal_coordinates": {"face": torch.arange(12.0), "height": torch.arange(float(case["nside"])),
                                                    "width": torch.arange(float(case["nside"]))},
                         "timestep": datetime.timedelta(hours=6) // datetime.timedelta(microseconds=1)},
        "step": {"module": {**{f"module.{k}": v for k, v in g["state_dict"].items()}, "label_encoding": None}}}}
    loaded = load_stepper(state, device="cpu")
    assert loaded.dataset_info.img_shape == (case["nside"], case["nside"])
    net = loaded.stepper.modules[0]
    assert type(net).__name__ == "HEALPixUNet"
    for k, v in g["state_dict"].items():
        assert torch.equal(net.state_dict()[k], v)
    bad = {"stepper": {**state["stepper"], "dataset_info": {"horizontal_coordinates": {"face": torch.arange(11.0), "height": torch.arange(8.0),
                                                                                         "width": torch.arange(8.0)}}}}
    with pytest.raises(ValueError, match="12 faces"):
        load_stepper(bad, device="cpu")
