"""Size-independent properties at BASELINE.json's FULL sizes, where the fp64 oracle would take minutes to hours: the spherical
harmonic transform pair at embed width 384 on the 1-degree (configs[1]) and the 0.25-degree (configs[3]) grids - synthesis of a
band-limited field followed by analysis returns the coefficients, analysis is linear, channels are independent - and the
0.25-degree network at its real shape (embed 384, 8 layers): finite, bitwise repeatable, graph replay = eager launches, and the two
arithmetic modes agree within the per-step tolerance.  Through the same C-ABI entry points as the small-size parity tests
(ace_sht_forward / ace_sht_inverse, ace_sfno_forward[_graph]).  First run on an MI355X in round 4 (tools/full_size_properties.py,
which now only calls these)."""
from types import SimpleNamespace

import pytest
import torch

from _util import assert_net_close, rel_max

pytestmark = pytest.mark.gpu

GRIDS = {"one_degree": (180, 360), "quarter_degree": (721, 1440)}
WIDTH = 384


def _band_limited_coefficients(n, L, M, nlon, gen):
    """random coefficients of a real field: zero for l < m, real for m = 0, nothing on the longitude Nyquist column"""
    c = torch.complex(torch.randn(n, L, M, generator=gen), torch.randn(n, L, M, generator=gen))
    l = torch.arange(L)[:, None]
    m = torch.arange(M)[None, :]
    c = c * (l >= m)
    c[..., 0] = c[..., 0].real.to(torch.complex64)
    if (M - 1) * 2 == nlon:
        c[..., M - 1] = 0
    return c


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
@pytest.mark.parametrize("grid", list(GRIDS))
def test_sht_round_trip_and_linearity_at_full_width(grid, precision):
    """legendre-gauss quadrature with lmax = nlat is exact for products of two band-limited fields (degree <= 2 nlat - 1), so
    analysis(synthesis(c)) = c up to rounding; tolerance: twice the per-transform bound of the small-size oracle tests (5e-6) with
    a margin for the 721-term sums, in the max norm relative to the largest coefficient / value."""
    import ace_amd
    dev = torch.device("cuda")
    H, W = GRIDS[grid]
    L, M = H, W // 2 + 1
    fwd = ace_amd.RealSHT(H, W, L, M, "legendre-gauss", precision=precision)
    inv = ace_amd.InverseRealSHT(H, W, L, M, "legendre-gauss", precision=precision)
    gen = torch.Generator().manual_seed(11)
    c0 = _band_limited_coefficients(WIDTH, L, M, W, gen).to(dev)
    x = inv(c0)
    assert x.shape == (WIDTH, H, W) and bool(torch.isfinite(x).all())
    c1 = fwd(x)
    assert rel_max(c1, c0) <= 2e-5
    x1 = inv(c1)
    assert rel_max(x1, x) <= 2e-5
    # linearity of the analysis (a second, independent band-limited field)
    y = inv(_band_limited_coefficients(WIDTH, L, M, W, gen).to(dev))
    lhs = fwd(2.0 * x - 3.0 * y)
    rhs = 2.0 * c1 - 3.0 * fwd(y)
    assert rel_max(lhs, rhs) <= 2e-5
    # the channels do not talk to each other: transforming a slice equals slicing the transform
    assert rel_max(fwd(x[100:164]), c1[100:164]) <= 2e-6


def test_quarter_degree_network_at_its_real_shape():
    """configs[3] at embed 384 x 8 layers, B = 1: the compensated-fp16 and the exact-fp32 MFMA modes are independent kernels for
    every contraction"""
    from ace_amd.sfno import SphericalFourierNeuralOperatorNet
    dev = torch.device("cuda")
    H, W = GRIDS["quarter_degree"]
    params = SimpleNamespace(operator_type="dhconv", scale_factor=1, embed_dim=WIDTH, num_layers=8, data_grid="legendre-gauss")
    torch.manual_seed(0)
    net = SphericalFourierNeuralOperatorNet(params=params, in_chans=44, out_chans=50, img_shape=(H, W)).to(dev).eval()
    net.set_precision("f16x3")
    x = torch.randn(1, 44, H, W, device=dev)
    out = torch.empty(1, 50, H, W, device=dev)
    with torch.no_grad():
        a = net(x).clone()
        b = net(x).clone()
        net.forward_graph(x, out)
        net.forward_graph(x, out)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(a).all()) and float(a.abs().max()) > 0
        assert torch.equal(a, b) and torch.equal(out, a)
        net.set_precision("fp32")
        exact = net(x)
        assert_net_close(a, exact, 1e-5)      # whole tensor and channel by channel


def test_inverse_sht_whose_operand_passes_the_32_bit_row_offsets():
    """The register-resident Legendre kernels address their data operand with 32-bit byte offsets through a range-checked buffer
    descriptor; the inverse transform's row pitch is mmax * 2 * n floats, so from n ~ 16 k fields on the 1-degree grid one
    batch's rows no longer fit 4 GiB and the launch has to leave those kernels (round-4 ADVICE: the unfolded kernel had no size
    guard and wrapped - in-range, wrong rows, silently).  One call with n = 16 416 fields against the same fields in two halves
    that stay inside the kernels' range: the same result within the transform's bound."""
    import ace_amd
    dev = torch.device("cuda")
    H, W = GRIDS["one_degree"]
    L, M = H, W // 2 + 1
    n = 16416
    assert (L + 16) * (M * 2 * n) * 4 >= 4.0e9 > (L + 16) * (M * 2 * (n // 2)) * 4
    inv = ace_amd.InverseRealSHT(H, W, L, M, "legendre-gauss", precision="f16x3")
    gen = torch.Generator(device=dev).manual_seed(5)
    c = torch.complex(torch.randn(n, L, M, generator=gen, device=dev), torch.randn(n, L, M, generator=gen, device=dev))
    whole = inv(c)
    assert bool(torch.isfinite(whole).all())
    worst = 0.0
    for h in range(2):
        part = inv(c[h * (n // 2):(h + 1) * (n // 2)])
        ref = whole[h * (n // 2):(h + 1) * (n // 2)]
        worst = max(worst, float((part - ref).abs().max() / part.abs().max()))
        del part
    assert worst <= 1e-5, worst
