"""Parity at the HEADLINE shape (BASELINE.json configs[1]: ACE2-shape SFNO, embed 384, 8 layers, dhconv, 44/50 channels,
180x360): the whole network against values emitted by the real reference (tests/golden/make_golden_headline.py), the
fused MLP and the packed path at C = 384 / 128 against fp64, and a one-year (1460-step) hipGraph rollout."""
import hashlib
import os

import pytest
import torch

from _util import build_native_net, load_golden, rel_max

pytestmark = pytest.mark.gpu
NET_TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def headline():
    from oracle.sfno import SFNOConfig, init_state
    d = load_golden("gen_sfno_headline_384x8.pt")
    cfg = SFNOConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in d["cfg"].items()})
    state = init_state(cfg, seed=d["seed"])
    chk = sum(float(v.double().abs().sum()) for v in state.values())
    assert abs(chk - d["state_checksum"]) <= 1e-9 * abs(d["state_checksum"]), "weight generator drifted from the fixture"
    x = torch.randn(d["batch"], cfg.in_chans, *cfg.img_shape, generator=torch.Generator().manual_seed(d["seed"] + 1000))
    assert abs(float(x.double().abs().sum()) - d["x_checksum"]) <= 1e-9 * d["x_checksum"]
    return d, cfg, state, x


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_headline_network_vs_reference(dev, headline, precision):
    """C = 384 x 8 layers x 180x360, B = 2, against the reference's own output: 32768 sampled values and per-channel
    statistics, in max|err| / max|ref| (north_star tolerance 1e-5)."""
    d, cfg, state, x = headline
    net = build_native_net(cfg, state, dev, precision)
    with torch.no_grad():
        y = net(x.to(dev))
        y2 = net(x.to(dev))
    assert torch.equal(y, y2)                                    # deterministic
    assert torch.isfinite(y).all()
    ys = y.reshape(-1)[d["sample_index"].to(dev)].cpu()
    err = float((ys.double() - d["sample_value"].double()).abs().max() / d["y_absmax"])
    print(f"headline {precision}: sampled rel err vs reference {err:.3e}")
    assert err <= NET_TOL
    # ... and channel by channel (a max norm over the whole tensor does not see a channel of small magnitude): the sampled errors of
    # each output channel against THAT channel's own largest value in the reference output
    hw = y.shape[-1] * y.shape[-2]
    chan = (d["sample_index"] // hw) % y.shape[1]
    cerr = torch.zeros(y.shape[1], dtype=torch.float64).scatter_reduce(0, chan, (ys.double() - d["sample_value"].double()).abs(), "amax")
    worst = float((cerr / d["y_channel_absmax"].double().clamp_min(1e-300)).max())
    print(f"headline {precision}: worst per-channel rel err {worst:.3e}")
    assert worst <= 3 * NET_TOL
    mean = y.double().mean(dim=(0, 2, 3)).cpu()
    amax = y.abs().amax(dim=(0, 2, 3)).cpu()
    assert float((mean - d["y_channel_mean"].double()).abs().max()) <= NET_TOL * d["y_absmax"]
    assert float((amax - d["y_channel_absmax"]).abs().max()) <= 10 * NET_TOL * d["y_absmax"]
    # batched == per sample (the per-sample folded weights and statistics must not mix samples)
    with torch.no_grad():
        y0 = net(x[:1].to(dev))
    assert rel_max(y0, y[:1]) <= 2e-6
    # the graph path replays the same kernels
    out = torch.empty_like(y)
    xd = x.to(dev).contiguous()
    with torch.no_grad():
        net.forward_graph(xd, out)
        net.forward_graph(xd, out)
    torch.cuda.synchronize()
    assert torch.equal(out, y)


@pytest.mark.parametrize("B", [1, 2])
def test_dhconv_at_headline_shape_three_implementations(dev, B, monkeypatch):
    """The spectral filter contraction at the headline operand shape (rows = 181 B, K = N = 768, 180 degrees, triangular
    row skipping) on three independent kernels inside a one-block C = 384 network: dhconv_strip.hip (B = 1; B = 2 exceeds its
    192 spectral rows and runs the tile engine either way), the 128 x 128 f16x3 tile engine (ACE_NO_DHCONV_STRIP=1) and the
    exact-fp32 engine.  Everything else in the block is the same code in the two f16x3 runs, so their difference is the
    contraction's; against fp32 arithmetic both sit at the block's f16x3 level."""
    from oracle.sfno import SFNOConfig, init_state
    cfg = SFNOConfig(in_chans=8, out_chans=6, img_shape=(180, 360), embed_dim=384, num_layers=1, operator_type="dhconv")
    state = init_state(cfg, seed=5)
    x = (torch.randn(B, 8, 180, 360, generator=torch.Generator().manual_seed(6)) * 0.7 + 0.2).to(dev)
    outs = {}
    for name, prec, env in (("strip", "f16x3", None), ("tile", "f16x3", "1"), ("fp32", "fp32", None)):
        if env:
            monkeypatch.setenv("ACE_NO_DHCONV_STRIP", env)
        else:
            monkeypatch.delenv("ACE_NO_DHCONV_STRIP", raising=False)
        net = build_native_net(cfg, state, dev, prec)
        with torch.no_grad():
            outs[name] = net(x).clone()
        del net
    assert torch.isfinite(outs["fp32"]).all()
    assert rel_max(outs["strip"], outs["tile"]) <= 1e-6
    assert rel_max(outs["strip"], outs["fp32"]) <= 2e-6
    assert rel_max(outs["tile"], outs["fp32"]) <= 2e-6


@pytest.mark.parametrize("routing,reps", [("wl", 1000), ("ws", 200), ("tile", 200)])
def test_headline_bitwise_repeatability(dev, headline, routing, reps, monkeypatch):
    """1000 forwards of the headline network on the same input with the default routing (wl: fc1 on conv_wl.hip, inner skip
    and fc2 on conv_ws.hip) must be bitwise identical, and right; 200 more with all three on conv_ws.hip and 200 on the tile engine.  The
    register-resident kernels issue stores whose data registers are recycled a few instructions later; on gfx950 a load
    landing in such a register before the store has read it corrupts a few lanes, rarely and not reproducibly (r02: seen
    with 125 of 25 M plane entries wrong; tools/store_hazard.hip is the minimal reproducer) - a sampled accuracy check can
    miss that, bitwise repeatability cannot."""
    monkeypatch.setenv("ACE_CONV_WS", "none" if routing == "tile" else "all")
    monkeypatch.setenv("ACE_CONV_WL", "1" if routing == "wl" else "0")
    d, cfg, state, x = headline
    net = build_native_net(cfg, state, dev, "f16x3")
    xd = x[:1].to(dev).contiguous()
    with torch.no_grad():
        y0 = net(xd).clone()
        bad = 0
        for _ in range(reps):
            bad += int(not torch.equal(net(xd), y0))
    assert bad == 0, f"{bad} of {reps} forwards differ from the first"
    ys = y0.reshape(-1)
    assert torch.isfinite(ys).all()
    # batch 1 of the fixture input: compare with the sampled reference values that fall into sample 0
    idx = d["sample_index"]
    keep = idx < ys.numel()
    err = float((ys[idx[keep].to(dev)].cpu().double() - d["sample_value"][keep].double()).abs().max() / d["y_absmax"])
    assert err <= NET_TOL, err


@pytest.mark.parametrize("routing", ["ws", "tile", "mixed", "wl"])
@pytest.mark.parametrize("C,hw", [(384, (45, 90)), (128, (24, 48)), (256, (20, 40))])
def test_fused_mlp_shapes_vs_fp64(dev, C, hw, routing, monkeypatch):
    """the block's 1x1 convolutions at C in {128, 256, 384} inside 3-block nets, batch 3, ragged last workgroup, random
    norm gains - per-block taps against the fp64 oracle.  ws: all three on the weight-stationary conv_ws.hip (default);
    tile: all three on the 128 x 128 tile engine; mixed: inner skip and fc2 on conv_ws.hip, fc1 on the tile engine (the
    statistics partials and the folded operands of the two engines meet); wl: fc1 on conv_wl.hip (weights in LDS), the others
    on conv_ws.hip."""
    from oracle.sfno import SFNOConfig, SFNOOracle, init_state
    monkeypatch.setenv("ACE_CONV_WS", {"ws": "all", "tile": "none", "mixed": "skip,fc2", "wl": "all"}[routing])
    monkeypatch.setenv("ACE_CONV_WL", "1" if routing == "wl" else "0")
    cfg = SFNOConfig(in_chans=6, out_chans=5, img_shape=hw, embed_dim=C, num_layers=3, operator_type="dhconv")
    state = init_state(cfg, seed=17)
    g = torch.Generator().manual_seed(18)
    for k in state:
        if ".norm" in k and k.endswith("weight"):
            state[k] = 1.0 + 0.3 * torch.randn(state[k].shape, generator=g)
        if ".norm" in k and k.endswith("bias"):
            state[k] = 0.2 * torch.randn(state[k].shape, generator=g)
    x = torch.randn(3, 6, *hw, generator=g) * 0.8 + 0.5
    net = build_native_net(cfg, state, dev, "f16x3")
    with torch.no_grad():
        y, taps = net.forward_with_taps(x.to(dev))
    ref, rtaps = SFNOOracle(cfg, state, dtype=torch.float64).forward(x, return_blocks=True)
    for i, rt in enumerate(rtaps):
        assert rel_max(taps[i + 1], rt) <= NET_TOL, f"block {i}: {rel_max(taps[i + 1], rt)}"
    assert rel_max(y, ref) <= NET_TOL


@pytest.mark.parametrize("C,hw", [(384, (12, 24)), (128, (24, 48)), (256, (8, 16))])   # widths with a two-level FFT: the stream engages
def test_planes_only_residual_stream_vs_fp64(dev, C, hw, monkeypatch):
    """Default routing WITHOUT taps: from the first block on, fc2 (conv_ws modes 4 / 5) takes the outer-skip residual from the
    P-format planes of the block input and writes h' as planes only, and the longitude FFT reads those planes - no fp32 copy of
    the residual stream exists between the blocks.  Final output against the fp64 oracle, and against the same network with
    ACE_PLANES_STREAM=0 (fp32 residual stream, round 2's form): the two differ by the 22-bit rounding of h' only."""
    from oracle.sfno import SFNOConfig, SFNOOracle, init_state
    cfg = SFNOConfig(in_chans=6, out_chans=5, img_shape=hw, embed_dim=C, num_layers=4, operator_type="dhconv")
    state = init_state(cfg, seed=23)
    g = torch.Generator().manual_seed(24)
    for k in state:
        if ".norm" in k and k.endswith("weight"):
            state[k] = 1.0 + 0.3 * torch.randn(state[k].shape, generator=g)
        if ".norm" in k and k.endswith("bias"):
            state[k] = 0.2 * torch.randn(state[k].shape, generator=g)
    x = torch.randn(3, 6, *hw, generator=g) * 0.8 + 0.5
    ref = SFNOOracle(cfg, state, dtype=torch.float64).forward(x)
    outs = {}
    for stream in ("1", "0"):
        monkeypatch.setenv("ACE_PLANES_STREAM", stream)
        net = build_native_net(cfg, state, dev, "f16x3")
        with torch.no_grad():
            y = net(x.to(dev))
            assert torch.equal(y, net(x.to(dev)))
            assert rel_max(y[1:2], net(x[1:2].to(dev))) <= 2e-6      # batch 3 vs per sample (the range bounds span the batch: not bitwise)
        assert rel_max(y, ref) <= NET_TOL, (stream, rel_max(y, ref))
        outs[stream] = y
    assert rel_max(outs["1"], outs["0"]) <= 3e-6
    assert not torch.equal(outs["1"], outs["0"])                      # the switch does route differently


def test_weight_update_after_graph_capture(dev):
    """ace_sfno_set_weight drops captured graphs: scales baked into kernel arguments must follow a weight update."""
    from oracle.sfno import SFNOConfig, init_state
    cfg = SFNOConfig(in_chans=4, out_chans=3, img_shape=(24, 48), embed_dim=32, num_layers=2, operator_type="dhconv")
    state = init_state(cfg, seed=2)
    net = build_native_net(cfg, state, dev, "f16x3")
    x = torch.randn(1, 4, 24, 48, generator=torch.Generator().manual_seed(3)).to(dev)
    out = torch.empty(1, 3, 24, 48, device=dev)
    with torch.no_grad():
        net.forward_graph(x, out)
        big = {k: (v * 8.0 if k.endswith("mlp.fwd.0.weight") or k.endswith("inner_skip.weight") else v) for k, v in state.items()}
        net.load_state_dict(big, strict=True)
        net.forward_graph(x, out)
        torch.cuda.synchronize()
        eager = net(x)
    assert torch.equal(out, eager)


def test_one_year_graph_rollout(dev):
    """configs[1]: 1460 forward steps of the ACE2-shape network from a hipGraph-captured step (20 windows of 73 steps on
    static buffers): every value finite, the dynamic-range machinery healthy for the whole year (outputs of the last
    window keep their spread), two runs bit-identical; reports the drift of the state after 4 and 1460 steps."""
    import bench
    from ace_amd.rollout import RolloutEngine
    stepper, forcing, prog, diag = bench.build_stepper(dev, seed=0)
    T, nwin = 73, 20
    eng = RolloutEngine(stepper, batch=1, n_forward_steps=T, graph="step")
    g = torch.Generator().manual_seed(7)
    ic = {n: torch.randn(1, 1, *bench.IMG, generator=g).to(dev) for n in prog}
    fc = {n: torch.randn(1, T + 1, *bench.IMG, generator=g).to(dev) for n in forcing}

    def run():
        eng.load(ic, fc)
        stats = {}
        with torch.no_grad():
            for w in range(nwin):
                eng.run_window()
                if w == 0:
                    stats[4] = torch.stack([eng.out[n][0, 3] for n in prog]).clone()
                if w + 1 < nwin:
                    eng.continue_from_last()
        torch.cuda.synchronize()
        stats[T * nwin] = torch.stack([eng.out[n][0, -1] for n in prog]).clone()
        return stats

    a = run()
    b = run()
    x0 = torch.stack([ic[n][0, 0] for n in prog])
    for k in a:
        assert torch.isfinite(a[k]).all(), f"non-finite state after {k} steps"
        assert torch.equal(a[k], b[k]), f"replay not deterministic after {k} steps"
        drift = float((a[k] - x0).double().pow(2).mean().sqrt() / x0.double().pow(2).mean().sqrt())
        print(f"rollout: {k} steps, rms drift from the initial state {drift:.4f}, state rms {float(a[k].double().pow(2).mean().sqrt()):.4f}, "
              f"absmax {float(a[k].abs().max()):.3f}")
    last = a[T * nwin]
    assert float(last.std()) > 1e-6, "state collapsed: dynamic-range slots lost the signal"
    assert float(last.abs().max()) < 1e6, "state blew up"


def test_window_graph_keeps_corrector_state_across_replays(dev):
    """graph="window" with the ACE2-style corrector captured inside the graph: three one-step windows REPLAY the same
    captured graph; the dry-air reference mass seeded by the first must survive the replays (static buffer + device flag),
    so the three windows reproduce the reference stepper's continuous 3-step rollout (tests/golden/gen_checkpoint.pt)."""
    import ace_amd
    from _util import checkpoint_case, conditioning_floor
    from ace_amd.rollout import RolloutEngine
    g = checkpoint_case(load_golden("gen_checkpoint.pt"), "ace2_like")
    stepper = ace_amd.load_stepper(g["state"], device=dev).stepper
    ic = {k: v.to(dev) for k, v in g["ic"].items()}
    forcing = {k: v.to(dev) for k, v in g["forcing"].items()}
    eng = RolloutEngine(stepper, batch=2, n_forward_steps=1, graph="window")
    floor = conditioning_floor(g)
    state = ic
    for s, want_all in enumerate(g["steps"]):
        out, state = eng.predict(state, {k: v[:, s:s + 2] for k, v in forcing.items()})
        torch.cuda.synchronize()
        for k, want in want_all.items():
            err = float((out[k][:, 0].cpu() - want).abs().max()) / float(want.abs().max())
            assert err <= max(NET_TOL * (s + 1), 3.0 * floor[s][k]), (s, k, err)
    assert eng._window_graph is not None


def test_quarter_degree_mid_size_net(dev):
    """BASELINE configs[3] (0.25 degree, 721 x 1440, L = M = 721) at a mid channel count: C = 32, two blocks, both
    arithmetic modes against the fp64 oracle - the three-factor-free 40 x 36 longitude FFT, the tile-engine Legendre stages
    (K = 721 is beyond the register-resident strip kernels) and 1.04 M-pixel convolutions."""
    from oracle.sfno import SFNOConfig, SFNOOracle, init_state
    H, W = 721, 1440
    cfg = SFNOConfig(in_chans=4, out_chans=3, img_shape=(H, W), embed_dim=32, num_layers=2, operator_type="dhconv")
    st = init_state(cfg, seed=3)
    xin = torch.randn(1, 4, H, W, generator=torch.Generator().manual_seed(4))
    ref = SFNOOracle(cfg, st, dtype=torch.float64).forward(xin)
    for prec in ("f16x3", "fp32"):
        net = build_native_net(cfg, st, dev, prec)
        with torch.no_grad():
            out = net(xin.to(dev))
        err = rel_max(out, ref)
        print(f"0.25 degree C=32 x 2 blocks, {prec}: rel err vs fp64 {err:.3e}")
        assert err <= NET_TOL, prec


@pytest.mark.parametrize("n,c,L,Mm", [(1, 384, 180, 181), (2, 384, 180, 181), (3, 128, 24, 25), (5, 256, 40, 41)])
def test_dhconv_op_vs_fp64_einsum(dev, n, c, L, Mm):
    """The spectral filter contraction at the operator level through the C ABI (ace_dhconv_f16x3 -> dhconv_strip.hip) against
    the reference's einsum (fme/ace/models/modulus/contractions.py:183-195: "bixy,iox->boxy", complex) in fp64 - the headline
    operand shape (rows = 181 n, K = N = 768, 180 degrees) with one and with two samples (two 192-row chunks per degree), and
    two small ragged shapes.  Coefficients with m > l are zero, as every SHT output has them."""
    from ace_amd import _lib
    g = torch.Generator().manual_seed(100 + n + c)
    x = torch.randn(n, c, L, Mm, 2, generator=g)
    tri = (torch.arange(Mm)[None, :] <= torch.arange(L)[:, None]).float()           # (L, Mm): m <= l
    x = x * tri[None, None, :, :, None]
    w = torch.randn(c, c, L, 2, generator=g) * (1.0 / c)
    ref = torch.einsum("bixy,iox->boxy", torch.view_as_complex(x.double()), torch.view_as_complex(w.double()))
    xd, wd = x.to(dev).contiguous(), w.to(dev).contiguous()
    out = torch.empty_like(xd)
    _lib.check(_lib.lib().ace_dhconv_f16x3(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(out), n, c, L, Mm, _lib.current_stream()))
    got = torch.view_as_complex(out.cpu().double())
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= 2e-6, err
