"""Golden vector at the HEADLINE shape (BASELINE.json configs[1]: ACE2-shape SFNO, embed 384, 8 layers, dhconv, 44 in / 50 out
channels, 180x360), emitted by the real reference network imported from /root/reference under stubs (build container
only; SURVEY.md 8(c)(iii): "SHA + sampled values").

Run:  python tests/golden/make_golden_headline.py        (~1.5 GB of weights, a few minutes on 8 cores)

The weights are regenerated in the test from the same seed (oracle.sfno.init_state, guarded by a checksum); the 13 MB
output is reduced to its SHA-256, per-channel statistics and 32768 sampled values (positions drawn from a seeded
generator).  B = 2 samples are emitted so that the batched path is pinned as well."""

import dataclasses
import hashlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import checksum, ref_net_from_cfg  # noqa: E402
from oracle import ref_loader  # noqa: E402
from oracle.sfno import SFNOConfig, init_state  # noqa: E402


def main():
    ns = ref_loader.load()
    cfg = SFNOConfig(in_chans=44, out_chans=50, img_shape=(180, 360), embed_dim=384, num_layers=8, operator_type="dhconv")
    seed, batch, nsample = 31, 2, 32768
    state = init_state(cfg, seed=seed)
    x = torch.randn(batch, cfg.in_chans, *cfg.img_shape, generator=torch.Generator().manual_seed(seed + 1000))
    net = ref_net_from_cfg(ns, cfg, state)
    torch.set_num_threads(os.cpu_count() or 1)
    with torch.no_grad():
        y = net(x)
    flat = y.reshape(-1)
    idx = torch.randint(0, flat.numel(), (nsample,), generator=torch.Generator().manual_seed(seed + 2000))
    out = {
        "cfg": dataclasses.asdict(cfg), "seed": seed, "batch": batch,
        "state_checksum": sum(checksum(v) for v in state.values()), "x_checksum": checksum(x),
        "y_sha256": hashlib.sha256(y.contiguous().numpy().tobytes()).hexdigest(),
        "y_absmax": float(y.abs().max()),
        "y_channel_mean": y.double().mean(dim=(0, 2, 3)).float(), "y_channel_std": y.double().std(dim=(0, 2, 3)).float(),
        "y_channel_absmax": y.abs().amax(dim=(0, 2, 3)),
        "sample_index": idx, "sample_value": flat[idx].clone(),
    }
    torch.save(out, os.path.join(HERE, "gen_sfno_headline_384x8.pt"))
    print("gen_sfno_headline_384x8.pt", os.path.getsize(os.path.join(HERE, "gen_sfno_headline_384x8.pt")), out["y_sha256"],
          out["y_absmax"])


if __name__ == "__main__":
    main()
