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
    with pytest.raises(NotImplementedError, match="isolatitude"):
        ace_amd.ModuleSelector(type="HEALPixUNet", config={**cfg, "hpx_padding_mode": "isolatitude"}).build(3, 2, ace_amd.DatasetInfo((8, 8)))
    enc = dict(cfg["encoder"])
    enc["conv_block"] = {"block_type": "SymmetricConvNeXtBlock"}
    with pytest.raises(NotImplementedError, match="SymmetricConvNeXtBlock"):
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
