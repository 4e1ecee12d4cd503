"""Generate the golden fixtures under tests/golden/ (build container only).

Run:  python tests/golden/make_golden.py

Two kinds of fixture are written, both pure DATA (inputs / expected outputs):

1. ``ref_*.pt`` - byte copies of the golden tensors the reference's own tests
   hold for this path.
2. ``gen_*.pt`` - vectors produced HERE by the real reference modules, imported
   from /root/reference under stubs (oracle/ref_loader.py).  Inputs and weights
   are either stored, or regenerated in the tests from a CPU ``torch.Generator``
   seed via ``oracle.sfno.init_state`` (a checksum guards against RNG drift).

/root/reference does not exist on the GPU box; the tests only read the files
this script wrote.
"""

import os
import shutil
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from oracle.sfno import SFNOConfig, init_state  # noqa: E402

REF = ref_loader.REF

COPIES = {
    "fme/core/benchmark/testdata/sht-regression.pt": "ref_sht-regression.pt",
    "fme/core/benchmark/testdata/inverse_sht-regression.pt": "ref_inverse_sht-regression.pt",
    "fme/ace/models/modulus/testdata/test_sfnonet_output_is_unchanged.pt": "ref_modulus_sfnonet_output.pt",
    "fme/ace/stepper/testdata/stepper_predict_regression.pt": "ref_stepper_predict_regression.pt",
}


def checksum(t: torch.Tensor) -> float:
    return float(t.double().abs().sum())


def builder_params(**kw):
    """Field set of SphericalFourierNeuralOperatorBuilder (fme/ace/registry/sfno.py:21-42)."""
    d = dict(spectral_transform="sht", filter_type="linear", operator_type="diagonal", scale_factor=1,
             residual_filter_factor=1, embed_dim=256, num_layers=12, hard_thresholding_fraction=1.0,
             normalization_layer="instance_norm", use_mlp=True, activation_function="gelu",
             encoder_layers=1, pos_embed=True, big_skip=True, rank=1.0, factorization=None,
             separable=False, complex_network=True, complex_activation="real", spectral_layers=1,
             checkpointing=0, data_grid="legendre-gauss")
    d.update(kw)
    return types.SimpleNamespace(**d)


def ref_net_from_cfg(ns, cfg: SFNOConfig, state):
    params = builder_params(operator_type=cfg.operator_type, embed_dim=cfg.embed_dim, num_layers=cfg.num_layers,
                            data_grid=cfg.data_grid, scale_factor=cfg.scale_factor,
                            hard_thresholding_fraction=cfg.hard_thresholding_fraction,
                            normalization_layer=cfg.normalization_layer, use_mlp=cfg.use_mlp,
                            activation_function=cfg.activation_function, encoder_layers=cfg.encoder_layers,
                            pos_embed=cfg.pos_embed, big_skip=cfg.big_skip)
    net = ns.SFNO(params=params, in_chans=cfg.in_chans, out_chans=cfg.out_chans, img_shape=cfg.img_shape)
    missing, unexpected = net.load_state_dict(state, strict=True)
    assert not missing and not unexpected
    return net.eval()


def main():
    ns = ref_loader.load()
    for src, dst in COPIES.items():
        shutil.copyfile(os.path.join(REF, src), os.path.join(HERE, dst))
        os.chmod(os.path.join(HERE, dst), 0o644)

    # --- inputs of the SHT goldens: set_seed(0) => torch.manual_seed(3) (fme/core/rand.py:20-36)
    torch.manual_seed(3)
    x = torch.randn(1, 9, 18)
    torch.save({"x": x}, os.path.join(HERE, "gen_sht_input.pt"))

    # --- modulus net golden: weights + input of test_sfnonet.py:13-36 (seed 0), output from the reference
    torch.manual_seed(0)
    model = ns.SFNO(params=None, embed_dim=16, num_layers=2, img_shape=(9, 18), in_chans=2, out_chans=3)
    x = torch.randn(4, 2, 9, 18)
    with torch.no_grad():
        y = model(x)
    torch.save({"state": model.state_dict(), "x": x, "y": y,
                "cfg": dict(in_chans=2, out_chans=3, img_shape=(9, 18), embed_dim=16, num_layers=2,
                            operator_type="diagonal", data_grid="equiangular")},
               os.path.join(HERE, "gen_modulus_sfnonet_case.pt"))

    # --- stepper golden inputs: test_single_module.py:2293-2418 (module init draws first, then a, b, c)
    torch.manual_seed(0)
    m2 = ns.SFNO(params=builder_params(embed_dim=16, num_layers=2), in_chans=2, out_chans=2, img_shape=(9, 18))
    a, b, c = (torch.randn(3, 3, 9, 18) for _ in range(3))
    torch.save({"state": m2.state_dict(), "a": a, "b": b, "c": c,
                "cfg": dict(in_chans=2, out_chans=2, img_shape=(9, 18), embed_dim=16, num_layers=2,
                            operator_type="diagonal", data_grid="legendre-gauss"),
                "in_names": ["a", "b"], "out_names": ["b", "c"], "mean": 0.1, "std": 1.1},
               os.path.join(HERE, "gen_stepper_case.pt"))

    # --- SHT / iSHT at the 1-degree grid (legendre-gauss 180x360, L=180, M=181), 3 fields
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 180, 360, generator=g)
    sht = ns.RealSHT(180, 360, lmax=180, mmax=181, grid="legendre-gauss")
    isht = ns.InverseRealSHT(180, 360, lmax=180, mmax=181, grid="legendre-gauss")
    with torch.no_grad():
        c = sht(x)
        xr = isht(c)
    torch.save({"seed": 11, "x_checksum": checksum(x), "coeffs": c, "roundtrip": xr},
               os.path.join(HERE, "gen_sht_180x360.pt"))

    # --- whole nets with the ACE2 operator (dhconv) from seeded weights (regenerated in the tests)
    cases = {
        "gen_sfno_dhconv_12x24.pt": (SFNOConfig(in_chans=5, out_chans=7, img_shape=(12, 24), embed_dim=16,
                                                 num_layers=3, operator_type="dhconv"), 2, 21),
        "gen_sfno_dhconv_equiangular_9x18.pt": (SFNOConfig(in_chans=2, out_chans=3, img_shape=(9, 18), embed_dim=8,
                                                            num_layers=2, operator_type="dhconv",
                                                            data_grid="equiangular"), 3, 22),
        "gen_sfno_dhconv_180x360_c8.pt": (SFNOConfig(in_chans=3, out_chans=2, img_shape=(180, 360), embed_dim=8,
                                                      num_layers=2, operator_type="dhconv"), 1, 23),
    }
    for fname, (cfg, batch, seed) in cases.items():
        state = init_state(cfg, seed=seed)
        gx = torch.Generator().manual_seed(seed + 1000)
        x = torch.randn(batch, cfg.in_chans, *cfg.img_shape, generator=gx)
        net = ref_net_from_cfg(ns, cfg, state)
        with torch.no_grad():
            y = net(x)
        import dataclasses
        torch.save({"cfg": dataclasses.asdict(cfg), "seed": seed, "batch": batch,
                    "state_checksum": sum(checksum(v) for v in state.values()),
                    "x_checksum": checksum(x), "y": y}, os.path.join(HERE, fname))
        print(fname, tuple(y.shape), float(y.abs().max()))

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".pt"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
