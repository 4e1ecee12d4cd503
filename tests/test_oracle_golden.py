"""The CPU oracle against every golden the reference holds for this path
 and against vectors emitted by the reference itself
(tests/golden/make_golden.py).  Tolerances are torch.testing.assert_close
defaults for fp32 (rtol 1.3e-6, atol 1e-5), the reference's own bar
(fme/core/testing/regression.py:8-16)."""

import dataclasses
import os

import pytest
import torch

import oracle
from oracle import stepper as ostep
from oracle.sfno import SFNOConfig, SFNOOracle, init_state


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


def checksum(t):
    return float(t.double().abs().sum())


def test_sht_regression(golden_dir):
    x = _load(golden_dir, "gen_sht_input.pt")["x"]
    g = _load(golden_dir, "ref_sht-regression.pt")["output"]
    out = oracle.RealSHT(9, 18)(x)  # default grid = lobatto, lmax = nlat-1
    assert out.shape == g.shape == (1, 8, 10)
    torch.testing.assert_close(out, g)


def test_inverse_sht_regression(golden_dir):
    x = _load(golden_dir, "gen_sht_input.pt")["x"]
    g = _load(golden_dir, "ref_inverse_sht-regression.pt")["output"]
    out = oracle.InverseRealSHT(9, 18)(oracle.RealSHT(9, 18)(x))
    torch.testing.assert_close(out, g)


@pytest.mark.parametrize("grid", ["equiangular", "legendre-gauss"])
@pytest.mark.parametrize("constant", [1.0, 0.42])
def test_constant_field(grid, constant):
    # fme/test_harmonics.py:10-25
    field = torch.full((6, 12), constant)
    coeffs = oracle.RealSHT(6, 12, grid=grid)(field).ravel()
    assert abs(coeffs[0]) > 1e-3
    assert torch.all(coeffs[1:].abs() < 1e-6)


@pytest.mark.parametrize("seed", [0, 1])
def test_roundtrip_idempotent(seed):
    # fme/test_harmonics.py:35-42
    torch.manual_seed(seed)
    f = torch.randn(6, 12)
    sht = oracle.RealSHT(6, 12, grid="legendre-gauss")
    isht = oracle.InverseRealSHT(6, 12, grid="legendre-gauss")
    p = isht(sht(f))
    assert torch.all(torch.isclose(p, isht(sht(p)), atol=1e-6))


def test_sht_180x360_vs_reference(golden_dir):
    d = _load(golden_dir, "gen_sht_180x360.pt")
    x = torch.randn(3, 180, 360, generator=torch.Generator().manual_seed(d["seed"]))
    assert checksum(x) == pytest.approx(d["x_checksum"], rel=1e-12)
    sht = oracle.RealSHT(180, 360, lmax=180, mmax=181, grid="legendre-gauss")
    isht = oracle.InverseRealSHT(180, 360, lmax=180, mmax=181, grid="legendre-gauss")
    c = sht(x)
    torch.testing.assert_close(c, d["coeffs"])
    torch.testing.assert_close(isht(c), d["roundtrip"])


def test_modulus_sfnonet_golden(golden_dir):
    d = _load(golden_dir, "gen_modulus_sfnonet_case.pt")
    g = _load(golden_dir, "ref_modulus_sfnonet_output.pt")
    net = SFNOOracle(SFNOConfig(**d["cfg"]), d["state"])
    y = net(d["x"])
    torch.testing.assert_close(y, g)
    torch.testing.assert_close(y, d["y"])


def test_stepper_predict_golden(golden_dir):
    d = _load(golden_dir, "gen_stepper_case.pt")
    g = _load(golden_dir, "ref_stepper_predict_regression.pt")
    net = SFNOOracle(SFNOConfig(**d["cfg"]), d["state"])
    names = ["a", "b", "c"]
    means = {k: torch.tensor(d["mean"]) for k in names}
    stds = {k: torch.tensor(d["std"]) for k in names}
    outs = ostep.predict(net, {"b": d["b"][:, :1]}, {"a": d["a"]}, 2, d["in_names"], d["out_names"], means, stds)
    ob = torch.stack([o["b"] for o in outs], 1)
    oc = torch.stack([o["c"] for o in outs], 1)
    torch.testing.assert_close(ob, g["output.b"])
    torch.testing.assert_close(oc, g["output.c"])
    torch.testing.assert_close(ob[:, -1:], g["next_state.b"])


@pytest.mark.parametrize("name", ["gen_sfno_dhconv_12x24.pt", "gen_sfno_dhconv_equiangular_9x18.pt",
                                  "gen_sfno_dhconv_180x360_c8.pt",
                                  # the "layer_norm" normalisation (sfnonet.py:584-592), make_golden_layer_norm.py
                                  "gen_sfno_layer_norm_12x24.pt", "gen_sfno_layer_norm_equiangular_9x18.pt"])
def test_dhconv_nets_vs_reference(golden_dir, name):
    d = _load(golden_dir, name)
    cfg = SFNOConfig(**{**d["cfg"], "img_shape": tuple(d["cfg"]["img_shape"])})
    state = init_state(cfg, seed=d["seed"])
    assert sum(checksum(v) for v in state.values()) == pytest.approx(d["state_checksum"], rel=1e-12)
    x = torch.randn(d["batch"], cfg.in_chans, *cfg.img_shape, generator=torch.Generator().manual_seed(d["seed"] + 1000))
    assert checksum(x) == pytest.approx(d["x_checksum"], rel=1e-12)
    y = SFNOOracle(cfg, state)(x)
    torch.testing.assert_close(y, d["y"])
    assert dataclasses.asdict(cfg)["operator_type"] == "dhconv"


@pytest.mark.parametrize("case", ["sf2_3blocks", "sf2_2blocks", "sf2_1block", "sf3_equiangular", "sf2_layer_norm", "sf2_diagonal_no_norm",
                                  "rff2", "rff2_sf2", "rff3_sf3_equiangular"])
def test_scale_factor_nets_vs_reference(golden_dir, case):
    """scale_factor != 1 (sfnonet.py:467-515): outputs of the REAL reference net (tests/golden/make_golden_scale_factor.py)."""
    d = _load(golden_dir, "gen_sfno_scale_factor.pt")[case]
    cfg = SFNOConfig(**{**d["cfg"], "img_shape": tuple(d["cfg"]["img_shape"])})
    state = init_state(cfg, seed=d["seed"])
    assert sum(checksum(v) for v in state.values()) == pytest.approx(d["state_checksum"], rel=1e-12)
    x = torch.randn(d["batch"], cfg.in_chans, *cfg.img_shape, generator=torch.Generator().manual_seed(d["seed"] + 1000))
    assert checksum(x) == pytest.approx(d["x_checksum"], rel=1e-12)
    torch.testing.assert_close(SFNOOracle(cfg, state)(x), d["y"])


def test_legendre_tables_against_scipy_spherical_harmonics():
    """An INDEPENDENT pin of the oracle's Legendre tables at the full sizes (the torch-harmonics arithmetic they restate is not
    in the reference tree, and the reference's own goldens hold it at 9 x 18 / 12 x 24 only): the orthonormal associated Legendre
    function with the Condon-Shortley phase is scipy's spherical harmonic at longitude 0, Y_l^m(theta, 0) - third-party
    arithmetic, different algorithm.  Every degree l < 180 with every third order m (and m = l) of the 1-degree table (180 Gauss-Legendre latitudes) and a
    sample of 1500 (l, m) of the 0.25-degree table (721 latitudes), all latitudes each; also the Gauss-Legendre nodes / weights
    against scipy.special.roots_legendre.  Agreement: 1e-12 absolute on values of order 1 at 180 (measured 1.5e-13), 5e-11 at 721 (measured 1e-11: 720 recursion steps)."""
    import numpy as np
    from scipy import special

    from oracle.legendre import legpoly
    from oracle.quadrature import legendre_gauss_weights
    rng = np.random.default_rng(0)
    for n, mmax, pairs, tol in ((180, 181, None, 1e-12), (721, 721, 1500, 5e-11)):
        x, w = legendre_gauss_weights(n)
        xs, ws = special.roots_legendre(n)
        assert np.abs(x - xs).max() <= 1e-14 and np.abs(w / ws - 1.0).max() <= 1e-8       # (the end-node weights of numpy's and scipy's rules differ by 1e-10 at n = 180, 4e-9 at n = 721: below fp32)
        theta = np.flip(np.arccos(x))
        tab = legpoly(mmax, n, np.cos(theta))                # [m][l][latitude]
        if pairs is None:      # every degree, every third order + the sectoral one
            lm = [(l, m) for l in range(n) for m in sorted(set(range(0, l + 1, 3)) | {l})]
        else:
            ls = rng.integers(0, n, size=pairs)
            lm = [(int(l), int(rng.integers(0, l + 1))) for l in ls] + [(n - 1, 0), (n - 1, 1), (n - 1, n - 1), (n - 1, n // 2)]
        worst = 0.0
        for l, m in lm:
            worst = max(worst, float(np.abs(tab[m, l] - special.sph_harm_y(l, m, theta, 0.0).real).max()))
        assert worst <= tol, (n, worst)
        assert not tab[np.triu_indices(min(mmax, n), 1)[1], np.triu_indices(min(mmax, n), 1)[0]].any()     # zero for l < m
