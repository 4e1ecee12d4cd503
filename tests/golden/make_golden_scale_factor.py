"""Goldens of scale_factor != 1 and residual_filter_factor != 1 (fme/ace/models/modulus/sfnonet.py:467-515: the blocks between the first filter's inverse transform and
the last filter's forward transform work on the (H // sf) x (W // sf) Gauss-Legendre grid; the first and the last block change grids
and their residual is the spectrally round-tripped input, s2convolutions.py:165-172; residual_filter_factor band-limits the big skip's
input on the data grid, sfnonet.py:473-497, 715-716), produced HERE by the real reference net imported
from /root/reference under stubs (oracle/ref_loader.py).  Build container only:  python tests/golden/make_golden_scale_factor.py

Pure data: configurations, the seeds the tests regenerate weights / inputs from (oracle.sfno.init_state; checksums guard against RNG
drift) and the reference's outputs.
"""
import dataclasses
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import builder_params, checksum  # noqa: E402
from oracle import ref_loader  # noqa: E402
from oracle.sfno import SFNOConfig, init_state  # noqa: E402


def main():
    ns = ref_loader.load()
    base = dict(in_chans=5, out_chans=7, embed_dim=16, operator_type="dhconv")
    cases = {
        "sf2_3blocks": (SFNOConfig(img_shape=(12, 24), num_layers=3, scale_factor=2, **base), 2, 61),
        "sf2_2blocks": (SFNOConfig(img_shape=(12, 24), num_layers=2, scale_factor=2, **base), 2, 62),
        "sf2_1block": (SFNOConfig(img_shape=(12, 24), num_layers=1, scale_factor=2, **base), 1, 63),
        "sf3_equiangular": (SFNOConfig(img_shape=(9, 18), num_layers=3, scale_factor=3, data_grid="equiangular", **base), 3, 64),
        "sf2_layer_norm": (SFNOConfig(img_shape=(12, 24), num_layers=3, scale_factor=2, normalization_layer="layer_norm", **base), 2, 65),
        "sf2_diagonal_no_norm": (SFNOConfig(img_shape=(12, 24), num_layers=3, scale_factor=2, in_chans=5, out_chans=7, embed_dim=16,
                                            operator_type="diagonal", normalization_layer="none"), 2, 66),
        "rff2": (SFNOConfig(img_shape=(12, 24), num_layers=2, residual_filter_factor=2, **base), 2, 67),
        "rff2_sf2": (SFNOConfig(img_shape=(12, 24), num_layers=3, residual_filter_factor=2, scale_factor=2, **base), 2, 68),
        "rff3_sf3_equiangular": (SFNOConfig(img_shape=(9, 18), num_layers=2, residual_filter_factor=3, scale_factor=3, data_grid="equiangular", **base), 3, 69),
    }
    out = {}
    for name, (cfg, batch, seed) in cases.items():
        state = init_state(cfg, seed=seed)
        x = torch.randn(batch, cfg.in_chans, *cfg.img_shape, generator=torch.Generator().manual_seed(seed + 1000))
        params = builder_params(operator_type=cfg.operator_type, embed_dim=cfg.embed_dim, num_layers=cfg.num_layers, data_grid=cfg.data_grid,
                                scale_factor=cfg.scale_factor, residual_filter_factor=cfg.residual_filter_factor,
                                hard_thresholding_fraction=cfg.hard_thresholding_fraction, normalization_layer=cfg.normalization_layer,
                                use_mlp=cfg.use_mlp, activation_function=cfg.activation_function, encoder_layers=cfg.encoder_layers,
                                pos_embed=cfg.pos_embed, big_skip=cfg.big_skip)
        net = ns.SFNO(params=params, in_chans=cfg.in_chans, out_chans=cfg.out_chans, img_shape=cfg.img_shape,
                      residual_filter_factor=cfg.residual_filter_factor)
        net.load_state_dict(state, strict=True)
        net.eval()
        assert (cfg.residual_filter_factor == 1) == isinstance(net.residual_filter_down, torch.nn.Identity)
        with torch.no_grad():
            y = net(x)
        out[name] = {"cfg": dataclasses.asdict(cfg), "seed": seed, "batch": batch,
                     "state_checksum": sum(checksum(v) for v in state.values()), "x_checksum": checksum(x), "y": y}
        print(name, tuple(y.shape), float(y.abs().max()))
    torch.save(out, os.path.join(HERE, "gen_sfno_scale_factor.pt"))


if __name__ == "__main__":
    main()
