"""HEALPix UNet host logic under an emulation of the ``ace_hpx_*`` operators (tests/_fake_hpx.py, CPU, ``-m "not gpu"``) and on
the real kernels (``-m gpu``), against outputs of the reference itself (tests/golden/make_golden_healpix.py):

  * every reference-emitted UNet case the GPU suite holds (ConvNeXt / basic / symmetric blocks, avg / max pooling, transposed
    convolution and nearest upsampling, the three padding modes) - on the emulation this checks layouts, pitches, padding and
    row-offset tables, weight preparation order, skip handling, i.e. everything but the kernels' arithmetic;
  * the DealiasedDownsample (fixed depthwise blur + stride 2) and SmoothedInterpolateConv (padding, nearest x 2, four-point
    smoother, trim, convolution) resamplers (healpix_blocks.py:499-634, 699-866), inside UNets and on their own (two blur stages,
    an even filter length, faces too small for the gather form).
Tolerance: 1e-5 of the output's maximum (the network tolerance of the GPU suite; the emulation is fp64 contractions on fp32 data)."""
import os

import pytest
import torch

import ace_amd
from ace_amd.healpix import DealiasedDownsample, Hpx, SmoothedInterpolateConv, _RT
from _fake_hpx import fake_hpx
from _util import assert_net_close, load_golden, rel_max

NET_TOL = 1e-5
UNETS = [("gen_healpix.pt", "convnext_avgpool_tconv"), ("gen_healpix.pt", "basic_maxpool"),
         ("gen_healpix_isolatitude.pt", "isolatitude"), ("gen_healpix_isolatitude.pt", "symmetric"),
         ("gen_healpix_isolatitude.pt", "interpolate_upsample"), ("gen_healpix_resamplers.pt", "dealiased_smoothed"),
         ("gen_healpix_resamplers.pt", "dealiased_smoothed_isolatitude")]


def _build(case, state, device="cpu"):
    net = ace_amd.ModuleSelector(type="HEALPixUNet", config=case["config"]).build(
        case["n_in"], case["n_out"], ace_amd.DatasetInfo((case["nside"], case["nside"]))).torch_module.to(device)
    assert list(net.state_dict()) == list(state)           # the reference's parameter / buffer names, in its order
    net.load_state_dict(state, strict=True)
    return net


def _block(kind, kwargs):
    return {"DealiasedDownsample": DealiasedDownsample, "SmoothedInterpolateConv": SmoothedInterpolateConv}[kind](**kwargs)


def _run_block(blk, x, device):
    _RT.pitch = {}
    _RT.begin(device)
    out = blk(Hpx(x.to(device).float().contiguous(), x.shape[-1]))
    return out.data[..., : out.width]


@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("file,name", UNETS)
def test_unet_host_logic_on_the_emulated_operators(file, name, packed, monkeypatch):
    """packed: the k x k convolutions through ace_hpx_pad_planes + ace_hpx_conv_packed (channel groups of 8, (tap, padded channel)
    weight columns); else through ace_hpx_pad + ace_hpx_conv (row-offset table) - the ACE_HPX_NO_PACKED=1 routing."""
    import ace_amd.healpix as hp
    monkeypatch.setattr(hp, "_PACKED_CONV", packed)
    g = load_golden(file)["unet"][name]
    net = _build(g["case"], g["state_dict"])
    with fake_hpx() as fake, torch.no_grad():
        y = net._run(g["x"])
    assert y.shape == g["y"].shape
    assert_net_close(y, g["y"], NET_TOL)
    assert ("pad_planes" in fake.calls) == packed and (packed or "pad" in fake.calls)     # (the resampler blocks keep their own ace_hpx_pad)
    assert any(c.startswith("conv") for c in fake.calls)
    if packed and "convnext" in name:       # 3 x 3 -> GELU -> 3 x 3 -> GELU -> 1 x 1: no activation between them exists in fp32, the skip
        assert "conv3->padded" in fake.calls and "halo" in fake.calls       # convolution reads the interior of the block's padded input
        assert "conv3->planes" in fake.calls and "conv1<-planes" in fake.calls and "conv1<-interior" in fake.calls
    with pytest.raises(RuntimeError, match="MI355X"):       # the product path still refuses host tensors
        net(g["x"])


def test_resampler_blocks_on_the_emulated_operators():
    gold = load_golden("gen_healpix_resamplers.pt")["blocks"]
    for name, g in gold.items():
        torch.manual_seed(5)
        blk = _block(g["kind"], g["kwargs"]).eval()
        if "state_dict" in g:
            assert list(blk.state_dict()) == list(g["state_dict"])
            blk.load_state_dict(g["state_dict"], strict=True)
        else:
            assert list(blk.state_dict()) == g["state_keys"]
        with fake_hpx(), torch.no_grad():
            y = _run_block(blk, g["x"], "cpu")
        assert y.shape == g["y"].shape, name
        assert rel_max(y, g["y"]) <= 2e-6, (name, rel_max(y, g["y"]))


def test_resampler_configuration_surface():
    from ace_amd.healpix import (DealiasedDownsampleBlockConfig, SmoothedInterpolateConvBlockConfig, _block_from_state)
    d = _block_from_state({"block_type": "DealiasedDownsample", "pooling": 4, "resample_filter": [1.0, 1.0]})
    assert isinstance(d, DealiasedDownsampleBlockConfig) and d.downsample_spatial_factor() == 4
    assert len(d.build(in_channels=3).pool) == 2
    with pytest.raises(ValueError, match="in_channels"):
        d.build()
    with pytest.raises(ValueError, match="power of 2"):
        DealiasedDownsample(in_channels=2, stride=3)
    with pytest.raises(ValueError, match="sum to zero"):
        DealiasedDownsample(in_channels=2, resample_filter=[1.0, -1.0])
    u = _block_from_state({"block_type": "SmoothedInterpolateConv", "activation": {"cap_value": 10}})
    assert isinstance(u, SmoothedInterpolateConvBlockConfig) and u.stride == 2
    with pytest.raises(ValueError, match="dilation"):
        SmoothedInterpolateConv(in_channels=2, out_channels=2, dilation=2)
    with pytest.raises(ValueError, match="nside_after"):
        SmoothedInterpolateConv(in_channels=2, out_channels=2, hpx_padding_mode="isolatitude", nside=8)
    with pytest.raises(ValueError, match="must equal"):
        SmoothedInterpolateConv(in_channels=2, out_channels=2, nside=8, nside_after=12)
    SmoothedInterpolateConv(in_channels=2, out_channels=2, mode="bilinear")          # built since round 6 (ace_hpx_upsample2)
    with pytest.raises(NotImplementedError, match="bilinear"):
        SmoothedInterpolateConv(in_channels=2, out_channels=2, mode="bicubic")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dealiased_smoothed", "dealiased_smoothed_isolatitude"])
def test_resampler_unets_vs_reference(name):
    g = load_golden("gen_healpix_resamplers.pt")["unet"][name]
    net = _build(g["case"], g["state_dict"], "cuda")
    with torch.no_grad():
        y = net(g["x"].to("cuda"))
        assert torch.equal(y, net(g["x"].to("cuda")))
    assert_net_close(y, g["y"], NET_TOL)


@pytest.mark.gpu
def test_resampler_blocks_vs_reference():
    gold = load_golden("gen_healpix_resamplers.pt")["blocks"]
    for name, g in gold.items():
        blk = _block(g["kind"], g["kwargs"]).eval().to("cuda")
        if "state_dict" in g:
            blk.load_state_dict(g["state_dict"], strict=True)
        with torch.no_grad():
            y = _run_block(blk, g["x"], torch.device("cuda"))
        assert rel_max(y, g["y"]) <= 2e-6, (name, rel_max(y, g["y"]))


def _upsample_cases():
    from ace_amd.healpix import NearestUpsample
    g = torch.Generator().manual_seed(5)
    for mode, ac in (("nearest", False), ("nearest-exact", False), ("bilinear", False), ("bilinear", True)):
        for shape in ((24, 3, 8, 8), (12, 5, 6, 6), (12, 2, 1, 1)):
            x = torch.randn(*shape, generator=g)
            want = torch.nn.Upsample(scale_factor=2, mode=mode, **({"align_corners": True} if ac else {}))(x)     # healpix_blocks.py:244-253
            yield NearestUpsample(2, mode, ac), x, want, (mode, ac, shape)


def test_interpolate_upsampling_modes_on_the_emulated_operators():
    """The "Interpolate" block (healpix_blocks.py:229-253) IS nn.Upsample(scale_factor=2, mode=...): host logic (pitches, strides,
    the bound handed on) against torch's own operator for nearest / nearest-exact / bilinear with and without align_corners."""
    with fake_hpx() as fake:
        for blk, x, want, tag in _upsample_cases():
            y = _run_block(blk, x, torch.device("cpu"))
            torch.testing.assert_close(y, want, rtol=0, atol=1e-6, msg=str(tag))
        assert "upsample" in fake.calls and "tconv" not in fake.calls
    with pytest.raises(ValueError, match="align_corners"):
        from ace_amd.healpix import NearestUpsample
        NearestUpsample(2, "nearest", True)


@pytest.mark.gpu
def test_interpolate_upsampling_modes_vs_torch():
    """ace_hpx_upsample2 (csrc/healpix.hip) against nn.Upsample itself, the operator the reference's block wraps: nearest exact, bilinear
    to fp32 rounding (same source index, same bracketing of the four products)."""
    for blk, x, want, tag in _upsample_cases():
        y = _run_block(blk, x, torch.device("cuda")).cpu()
        if tag[0] != "bilinear":
            assert torch.equal(y, want), tag
        else:
            torch.testing.assert_close(y, want, rtol=0, atol=2e-6, msg=str(tag))


@pytest.mark.gpu
def test_bilinear_smoothed_interpolate_conv_vs_torch():
    """SmoothedInterpolateConv with upsample_mode = "bilinear" (healpix_blocks.py:699-866) against the same operator sequence evaluated
    with torch on the host: face padding by one cell (the native gather, pinned on the reference's own padded faces elsewhere), bilinear
    x 2, the four-point smoother / 4 as a depthwise convolution, trim one cell, face padding, the 3 x 3 convolution."""
    torch.manual_seed(3)
    blk = SmoothedInterpolateConv(in_channels=3, out_channels=4, kernel_size=3, mode="bilinear", hpx_padding_mode="earth2grid").eval()
    ref = SmoothedInterpolateConv(in_channels=3, out_channels=4, kernel_size=3, mode="nearest", hpx_padding_mode="earth2grid").eval()
    ref.load_state_dict(blk.state_dict())
    x = torch.randn(12, 3, 8, 8)
    with torch.no_grad():
        y_bil = _run_block(blk.to("cuda"), x, torch.device("cuda")).cpu()
        y_near = _run_block(ref.to("cuda"), x, torch.device("cuda")).cpu()
    assert y_bil.shape == y_near.shape == (12, 4, 16, 16)
    # a smooth field: nearest and bilinear resizes agree to first order, the block differs only through the resize
    xs = torch.ones(12, 3, 8, 8) * torch.linspace(0.5, 1.5, 3).view(1, 3, 1, 1)
    with torch.no_grad():
        a = _run_block(blk, xs, torch.device("cuda")).cpu()
        b = _run_block(ref, xs, torch.device("cuda")).cpu()
    torch.testing.assert_close(a, b, rtol=0, atol=1e-6)       # constant faces: every resize is the identity on them
    assert float((y_bil - y_near).abs().max()) > 1e-3          # ... and a random field tells the two modes apart
