"""tests/golden/gen_masking.pt from the reference's own spatial masking (fme/core/spatial_masking.py, spatial_mask_provider.py,
name_and_prefix_matcher.py - pure torch; imported through oracle/ref_loader.load_corrector's stubs): mask lookup per name, the
matcher's decisions, input masking with a float fill / the normaliser means / exclusions, the provider's output masker."""
import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_loader  # noqa: E402


def main():
    ref_loader.load_corrector()
    sm = importlib.import_module("fme.core.spatial_masking")
    mp = importlib.import_module("fme.core.spatial_mask_provider")
    nm = importlib.import_module("fme.core.name_and_prefix_matcher")
    g = torch.Generator().manual_seed(0)
    H, W = 4, 6
    masks = {"mask_2d": (torch.rand(H, W, generator=g) > 0.4).float(), "mask_0": (torch.rand(H, W, generator=g) > 0.5).float(),
             "mask_1": (torch.rand(H, W, generator=g) > 0.5).float(), "mask_sst": (torch.rand(H, W, generator=g) > 0.3).float() * 0.9 + 0.05}
    provider = mp.SpatialMaskProvider(masks)
    names = ["sst", "thetao_0", "thetao_1", "thetao_2", "zos", "so_0"]
    lookup = {n: (provider.get_mask_tensor_for(n).clone() if provider.get_mask_tensor_for(n) is not None else None) for n in names}
    matcher_cases = []
    for cfg in (["thetao"], ["thetao_"], ["thetao_1"], ["zos", "so_"], None):
        m = nm.NameAndPrefixMatcher(cfg)
        matcher_cases.append((cfg, {n: m.match(n) for n in names + ["thetao", "so", "so_10"]}))
    data = {n: torch.randn(2, H, W, generator=g) for n in names}
    means = {n: torch.tensor(float(i + 1)) for i, n in enumerate(names)}
    cases = []
    for kw in (dict(mask_value=0, fill_value=0.0), dict(mask_value=1, fill_value=-3.5), dict(mask_value=0, fill_value="mean"),
               dict(mask_value=0, fill_value=7.0, exclude_names_and_prefixes=["thetao", "zos"]),
               dict(mask_value=1, fill_value="mean", exclude_names_and_prefixes=["thetao_1", "so_"])):
        masker = sm.StaticSpatialMaskingConfig(**kw).build(mask=provider, means=means)
        cases.append({"config": kw, "out": {k: v.clone() for k, v in masker(data).items()}})
    out_masked = provider.build_output_spatial_masker()(data)
    dst = os.path.join(HERE, "gen_masking.pt")
    torch.save({"masks": masks, "names": names, "lookup": lookup, "matcher": matcher_cases, "data": data, "means": means,
                "cases": cases, "output_masked": {k: v.clone() for k, v in out_masked.items()}}, dst)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
