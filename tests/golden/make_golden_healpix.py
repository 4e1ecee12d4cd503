"""Golden vectors for the HEALPix variant, emitted by the REAL reference modules imported under stubs
(oracle/ref_loader.load_healpix) - build container only.
  * padding: HEALPixPadding (fme/ace/models/healpix/healpix_paddings.py:239-611) applied to random faces for several
    (nside, padding) pairs, incl. padding == nside and the equatorial corner means;
  * HEALPixUNet cases (fme/ace/models/healpix/healpix_unet.py + blocks): the reference's own test configuration family
    (fme/ace/registry/test_hpx.py: ConvNeXt blocks with CappedGELU, AvgPool, TransposedConvUpsample, dilations 1 / 2 / 4) and a
    MaxPool / BasicConvBlock variant; each case stores the configuration as the plain dict the registry takes, the
    reference's seeded state_dict, the input and the reference output."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_loader  # noqa: E402

CAP = {"cap_value": 10}
CASES = {
    "convnext_avgpool_tconv": dict(
        nside=16, n_in=5, n_out=3, batch=2,
        config=dict(
            encoder=dict(conv_block=dict(block_type="ConvNeXtBlock", kernel_size=3, upscale_factor=4, activation=CAP),
                         down_sampling_block=dict(block_type="AvgPool", pooling=2), n_channels=[16, 8, 4], dilations=[1, 2, 4]),
            decoder=dict(conv_block=dict(block_type="ConvNeXtBlock", kernel_size=3, upscale_factor=4, activation=CAP),
                         up_sampling_block=dict(block_type="TransposedConvUpsample", stride=2, activation=CAP),
                         output_layer=dict(block_type="BasicConvBlock", kernel_size=1, n_layers=1),
                         n_channels=[4, 8, 16], dilations=[4, 2, 1]),
            hpx_padding_mode="karlbauer")),
    "basic_maxpool": dict(
        nside=8, n_in=3, n_out=2, batch=1,
        config=dict(
            encoder=dict(conv_block=dict(block_type="BasicConvBlock", kernel_size=3, n_layers=2, activation=CAP),
                         down_sampling_block=dict(block_type="MaxPool", pooling=2), n_channels=[12, 6], n_layers=[2, 1]),
            decoder=dict(conv_block=dict(block_type="BasicConvBlock", kernel_size=3, n_layers=1, activation=CAP),
                         up_sampling_block=dict(block_type="TransposedConvUpsample", stride=2),
                         output_layer=dict(block_type="BasicConvBlock", kernel_size=3, n_layers=1),
                         n_channels=[6, 12], n_layers=[1, 2]),
            hpx_padding_mode="karlbauer", nside=[8, 4])),
}


def build_reference(ref, case):
    b, a = ref.blocks, ref.activations

    def block(d):
        d = dict(d)
        cls = {"ConvNeXtBlock": b.ConvNeXtBlockConfig, "BasicConvBlock": b.BasicConvBlockConfig, "AvgPool": b.AvgPoolDownsamplingBlockConfig,
               "MaxPool": b.MaxPoolDownsamplingBlockConfig, "TransposedConvUpsample": b.TransposedConvUpsampleBlockConfig,
               "SymmetricConvNeXtBlock": b.SymmetricConvNeXtBlockConfig, "Interpolate": b.InterpolateUpsampleBlockConfig,
               "Multi_SymmetricConvNeXtBlock": b.MultiSymmetricConvNeXtBlockConfig,
               "DealiasedDownsample": b.DealiasedDownsampleBlockConfig,
               "SmoothedInterpolateConv": b.SmoothedInterpolateConvBlockConfig}[d.pop("block_type")]
        if d.get("activation") is not None:
            d["activation"] = a.CappedGELUConfig(**d["activation"])
        return cls(**d)

    cfg = case["config"]
    e, d = dict(cfg["encoder"]), dict(cfg["decoder"])
    enc = ref.encoder.UNetEncoderConfig(conv_block=block(e.pop("conv_block")), down_sampling_block=block(e.pop("down_sampling_block")), **e)
    dec = ref.decoder.UNetDecoderConfig(conv_block=block(d.pop("conv_block")), up_sampling_block=block(d.pop("up_sampling_block")),
                                        output_layer=block(d.pop("output_layer")), **d)
    nside = tuple(cfg["nside"]) if cfg.get("nside") is not None else None
    ctx = b.HEALPixBuildContext(hpx_padding_mode=cfg["hpx_padding_mode"], nside_levels=nside)
    torch.manual_seed(0)
    encoder = enc.build(input_channels=case["n_in"], ctx=ctx)
    decoder = dec.build(output_channels=case["n_out"], ctx=ctx)
    return ref.unet.HEALPixUNet(encoder=encoder, decoder=decoder, input_channels=case["n_in"], output_channels=case["n_out"], nside=nside)


ISO_CASE = dict(
    nside=8, n_in=3, n_out=2, batch=2,
    config=dict(
        encoder=dict(conv_block=dict(block_type="ConvNeXtBlock", kernel_size=3, upscale_factor=2, activation=CAP),
                     down_sampling_block=dict(block_type="AvgPool", pooling=2), n_channels=[8, 4], dilations=[1, 2]),
        decoder=dict(conv_block=dict(block_type="ConvNeXtBlock", kernel_size=3, upscale_factor=2, activation=CAP),
                     up_sampling_block=dict(block_type="TransposedConvUpsample", stride=2, activation=CAP),
                     output_layer=dict(block_type="BasicConvBlock", kernel_size=1, n_layers=1),
                     n_channels=[4, 8], dilations=[2, 1]),
        hpx_padding_mode="isolatitude", nside=[8, 4]))


SYM_CASES = {
    # channel schedule chosen so that the reference's own skip rule works (identity iff in_channels == latent_channels: then the
    # block must also keep the width) - encoder 4 -> 4 (identity), 4 -> 8 (1 x 1 skip), 8 -> 8; decoder 8 -> 8, cat 16 -> 4, cat 8 -> 4
    "symmetric": dict(
        nside=8, n_in=4, n_out=2, batch=1,
        config=dict(
            encoder=dict(conv_block=dict(block_type="SymmetricConvNeXtBlock", kernel_size=3, upscale_factor=2, activation=CAP),
                         down_sampling_block=dict(block_type="AvgPool", pooling=2), n_channels=[4, 8, 8], dilations=[1, 2, 1]),
            decoder=dict(conv_block=dict(block_type="Multi_SymmetricConvNeXtBlock", kernel_size=3, upscale_factor=2, n_layers=2, activation=CAP),
                         up_sampling_block=dict(block_type="TransposedConvUpsample", stride=2, activation=CAP),
                         output_layer=dict(block_type="BasicConvBlock", kernel_size=1, n_layers=1),
                         n_channels=[8, 8, 4], dilations=[1, 2, 1], n_layers=[2, 1, 2]),
            hpx_padding_mode="karlbauer")),
    # "Interpolate" upsampling block: nn.Upsample(scale_factor=2, mode="nearest") (healpix_blocks.py:229-253)
    "interpolate_upsample": dict(
        nside=8, n_in=3, n_out=2, batch=2,
        config=dict(
            encoder=dict(conv_block=dict(block_type="ConvNeXtBlock", kernel_size=3, upscale_factor=2, activation=CAP),
                         down_sampling_block=dict(block_type="MaxPool", pooling=2), n_channels=[8, 4], dilations=[1, 2]),
            decoder=dict(conv_block=dict(block_type="ConvNeXtBlock", kernel_size=3, upscale_factor=2, activation=CAP),
                         up_sampling_block=dict(block_type="Interpolate", stride=2, upsample_mode="nearest"),
                         output_layer=dict(block_type="BasicConvBlock", kernel_size=1, n_layers=1),
                         n_channels=[4, 8], dilations=[2, 1]),
            hpx_padding_mode="karlbauer")),
}


RESAMPLER_CASES = {
    # dealiased (blur + stride 2) downsampling and smoothed-interpolate + convolution upsampling (healpix_blocks.py:499-634, 699-866)
    "dealiased_smoothed": dict(
        nside=16, n_in=3, n_out=2, batch=2,
        config=dict(
            encoder=dict(conv_block=dict(block_type="ConvNeXtBlock", kernel_size=3, upscale_factor=2, activation=CAP),
                         down_sampling_block=dict(block_type="DealiasedDownsample", pooling=2), n_channels=[8, 6, 4], dilations=[1, 2, 1]),
            decoder=dict(conv_block=dict(block_type="ConvNeXtBlock", kernel_size=3, upscale_factor=2, activation=CAP),
                         up_sampling_block=dict(block_type="SmoothedInterpolateConv", stride=2, kernel_size=3, activation=CAP),
                         output_layer=dict(block_type="BasicConvBlock", kernel_size=1, n_layers=1),
                         n_channels=[4, 6, 8], dilations=[1, 2, 1]),
            hpx_padding_mode="karlbauer")),
    # the same resamplers with the isolatitude padding (nside per level), a 5-tap filter, stride-4 pooling is NOT used by the UNet
    # builder (factor 2 between levels), so the filter length is what varies; no activation on the upsampling convolution
    "dealiased_smoothed_isolatitude": dict(
        nside=16, n_in=2, n_out=2, batch=1,
        config=dict(
            encoder=dict(conv_block=dict(block_type="BasicConvBlock", kernel_size=3, n_layers=1, activation=CAP),
                         down_sampling_block=dict(block_type="DealiasedDownsample", pooling=2, resample_filter=[1.0, 3.0, 4.0, 3.0, 1.0]),
                         n_channels=[6, 4], n_layers=[1, 1]),
            decoder=dict(conv_block=dict(block_type="BasicConvBlock", kernel_size=3, n_layers=1, activation=CAP),
                         up_sampling_block=dict(block_type="SmoothedInterpolateConv", stride=2, kernel_size=3),
                         output_layer=dict(block_type="BasicConvBlock", kernel_size=1, n_layers=1),
                         n_channels=[4, 6], n_layers=[1, 1]),
            hpx_padding_mode="isolatitude", nside=[16, 8])),
}


def resamplers(ref):
    """the two resampler blocks inside UNets, and the blocks on their own (stride 4 = two blur stages; an even filter length) -
    own file, the other fixtures stay byte-identical"""
    out = {"unet": {}, "blocks": {}}
    g = torch.Generator().manual_seed(23)
    for name, case in RESAMPLER_CASES.items():
        model = build_reference(ref, case).eval()
        with torch.no_grad():
            for k, prm in model.named_parameters():
                if k.endswith("weight"):
                    prm.mul_(2.0)
        x = torch.randn(case["batch"], 12, case["n_in"], case["nside"], case["nside"], generator=g) * 2.0
        with torch.no_grad():
            y = model(x)
        out["unet"][name] = {"case": dict(case), "state_dict": {k: v.clone() for k, v in model.state_dict().items()}, "x": x, "y": y}
        print(name, tuple(x.shape), "->", tuple(y.shape), "max|y|", float(y.abs().max()))
    b = ref.blocks
    for name, kw, nside in [("stride4", dict(in_channels=5, resample_filter=[1.0, 2.0, 1.0], stride=4, hpx_padding_mode="karlbauer"), 16),
                            ("even_filter", dict(in_channels=3, resample_filter=[1.0, 3.0, 3.0, 1.0], stride=2, hpx_padding_mode="karlbauer"), 8),
                            ("tiny_faces", dict(in_channels=2, resample_filter=[1.0, 2.0, 1.0], stride=2, hpx_padding_mode="karlbauer"), 4)]:
        blk = b.DealiasedDownsample(**kw).eval()
        x = torch.randn(12, kw["in_channels"], nside, nside, generator=g)
        with torch.no_grad():
            y = blk(x)
        out["blocks"][name] = {"kind": "DealiasedDownsample", "kwargs": kw, "x": x, "y": y, "state_keys": list(blk.state_dict())}
        print(name, tuple(x.shape), "->", tuple(y.shape))
    torch.manual_seed(5)
    blk = b.SmoothedInterpolateConv(in_channels=4, out_channels=3, kernel_size=3, hpx_padding_mode="karlbauer").eval()
    x = torch.randn(24, 4, 8, 8, generator=g)
    with torch.no_grad():
        y = blk(x)
    out["blocks"]["smoothed"] = {"kind": "SmoothedInterpolateConv", "kwargs": dict(in_channels=4, out_channels=3, kernel_size=3, hpx_padding_mode="karlbauer"),
                                 "x": x, "y": y, "state_dict": {k: v.clone() for k, v in blk.state_dict().items()}}
    print("smoothed", tuple(x.shape), "->", tuple(y.shape))
    dst = os.path.join(HERE, "gen_healpix_resamplers.pt")
    torch.save(out, dst)
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB")


def isolatitude(ref):
    """isolatitude padding (healpix_paddings.py:613-1140): the reference's gather indices and padded outputs for several
    (nside, padding) pairs (padding <= nside / 2), and one UNet with hpx_padding_mode="isolatitude" - own file, gen_healpix.pt stays
    byte-identical"""
    out = {"padding": {}, "unet": {}}
    g = torch.Generator().manual_seed(11)
    for nside, p in [(4, 1), (4, 2), (8, 2), (8, 4), (16, 3), (6, 3)]:
        x = torch.randn(2 * 12, 3, nside, nside, generator=g)
        layer = ref.paddings.HEALPixPaddingIsolatitude(p, nside)
        idx, valid = ref.paddings.build_isolatitude_gather_index(p, nside)
        out["padding"][(nside, p)] = {"x": x, "padded": layer(x), "index": idx.clone(), "valid": valid.clone()}
    model = build_reference(ref, ISO_CASE).eval()
    with torch.no_grad():
        for k, prm in model.named_parameters():
            if k.endswith("weight"):
                prm.mul_(3.0)
    x = torch.randn(ISO_CASE["batch"], 12, ISO_CASE["n_in"], ISO_CASE["nside"], ISO_CASE["nside"], generator=g) * 2.0
    with torch.no_grad():
        y = model(x)
    out["unet"]["isolatitude"] = {"case": dict(ISO_CASE), "state_dict": {k: v.clone() for k, v in model.state_dict().items()}, "x": x, "y": y}
    print("isolatitude unet", tuple(x.shape), "->", tuple(y.shape), "max|y|", float(y.abs().max()))
    # symmetric ConvNeXt variants (healpix_blocks.py:1214-1402): residual added after the last activation; identity and 1 x 1 skips
    for name, case in SYM_CASES.items():
        model = build_reference(ref, case).eval()
        with torch.no_grad():
            for k, prm in model.named_parameters():
                if k.endswith("weight"):
                    prm.mul_(2.0)
        x = torch.randn(case["batch"], 12, case["n_in"], case["nside"], case["nside"], generator=g) * 2.0
        with torch.no_grad():
            y = model(x)
        out["unet"][name] = {"case": dict(case), "state_dict": {k: v.clone() for k, v in model.state_dict().items()}, "x": x, "y": y}
        print(name, tuple(x.shape), "->", tuple(y.shape), "max|y|", float(y.abs().max()))
    dst = os.path.join(HERE, "gen_healpix_isolatitude.pt")
    torch.save(out, dst)
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB")


def main():
    ref = ref_loader.load_healpix()
    if "--resamplers" in sys.argv:
        resamplers(ref)
        return
    isolatitude(ref)
    out = {"padding": {}, "unet": {}}
    g = torch.Generator().manual_seed(3)
    for nside, p in [(4, 1), (8, 2), (8, 4), (16, 3), (6, 6), (5, 2)]:
        x = torch.randn(2 * 12, 3, nside, nside, generator=g)
        out["padding"][(nside, p)] = {"x": x, "padded": ref.paddings.HEALPixPadding(p)(x)}
    for name, case in CASES.items():
        model = build_reference(ref, case).eval()
        with torch.no_grad():      # biases start at their default init; scale the weights up so that the capped GELU is exercised
            for k, prm in model.named_parameters():
                if k.endswith("weight"):
                    prm.mul_(3.0)
        x = torch.randn(case["batch"], 12, case["n_in"], case["nside"], case["nside"], generator=g) * 2.0
        with torch.no_grad():
            y = model(x)
        out["unet"][name] = {"case": {k: v for k, v in case.items()}, "state_dict": {k: v.clone() for k, v in model.state_dict().items()},
                             "x": x, "y": y}
        print(name, tuple(x.shape), "->", tuple(y.shape), "max|y|", float(y.abs().max()), "capped share",
              float((y.abs() >= 9.999).float().mean()))
    dst = os.path.join(HERE, "gen_healpix.pt")
    torch.save(out, dst)
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB")


if __name__ == "__main__":
    main()
