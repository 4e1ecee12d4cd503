"""Non-asserting parity report (python tests/gpu_report.py > gpurun_out/report.txt): every operator and
network case, each error printed, so one failure does not hide the rest."""

import os
import sys
import time
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _util import build_native_net, load_golden, rel_l2, rel_max  # noqa: E402

dev = torch.device("cuda")


def section(name):
    print(f"\n== {name}", flush=True)


def guarded(fn):
    try:
        fn()
    except Exception:
        traceback.print_exc(file=sys.stdout)
    sys.stdout.flush()


def conv_cases():
    from ace_amd import _lib
    for (n, cin, cout, hw, act) in [(1, 2, 3, 162, 0), (2, 16, 16, 162, 1), (1, 7, 5, 13, 1), (3, 18, 16, 288, 2),
                                    (1, 44, 384, 64800, 1), (1, 384, 768, 8000, 1), (1, 428, 50, 4132, 3),
                                    (1, 64, 64, 256, 0), (1, 32, 128, 128, 0)]:
        g = torch.Generator().manual_seed(n * 1000 + cin)
        x = torch.randn(n, cin, hw, generator=g)
        w = torch.randn(cout, cin, generator=g) / cin**0.5
        b = torch.randn(cout, generator=g)
        ref = torch.nn.functional.conv1d(x.double(), w.double()[:, :, None], b.double())
        ref = [ref, torch.nn.functional.gelu(ref), torch.relu(ref), torch.nn.functional.silu(ref)][act]
        y = torch.full((n, cout, hw), float("nan"), device=dev)
        xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)  # keep alive until the kernel has run
        _lib.check(_lib.lib().ace_conv1x1(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y),
                                          n, cin, cout, hw, act, _lib.current_stream()))
        torch.cuda.synchronize()
        print(f"conv1x1 n={n} cin={cin} cout={cout} hw={hw} act={act}: relmax {rel_max(y, ref):.3e} nan={int(torch.isnan(y).sum())}")


def f16x3_cases():
    """compensated-fp16 conv vs fp64, next to the exact-fp32 engine on the same data."""
    from ace_amd import _lib
    L = _lib.lib()
    for (n, cin, cout, hw, act, xs) in [(1, 384, 768, 64800, 1, 1.0), (1, 768, 384, 64800, 0, 1.0), (2, 16, 16, 164, 1, 1.0),
                                        (1, 44, 384, 4096, 1, 1.0), (1, 428, 50, 4132, 0, 1.0), (1, 384, 384, 8000, 0, 30.0),
                                        (1, 384, 384, 8000, 0, 1e-3)]:
        g = torch.Generator().manual_seed(cin)
        x = (torch.randn(n, cin, hw, generator=g) * xs)
        w = torch.nn.init.trunc_normal_(torch.empty(cout, cin), std=0.02, generator=g)
        b = torch.randn(cout, generator=g) * 0.01
        ref = torch.nn.functional.conv1d(x.double(), w.double()[:, :, None], b.double())
        if act == 1:
            ref = torch.nn.functional.gelu(ref)
        xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
        y16 = torch.full((n, cout, hw), float("nan"), device=dev)
        y32 = torch.full((n, cout, hw), float("nan"), device=dev)
        st = _lib.current_stream()
        _lib.check(L.ace_conv1x1_f16x3(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y16), n, cin, cout, hw, act, st))
        _lib.check(L.ace_conv1x1(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y32), n, cin, cout, hw, act, st))
        torch.cuda.synchronize()
        print(f"conv n={n} cin={cin} cout={cout} hw={hw} act={act} xscale={xs}: f16x3 relmax {rel_max(y16, ref):.3e} "
              f"l2 {rel_l2(y16, ref):.3e} | fp32 relmax {rel_max(y32, ref):.3e} l2 {rel_l2(y32, ref):.3e} nan={int(torch.isnan(y16).sum())}")


def norm_cases():
    from ace_amd import _lib
    for (n, c, hw) in [(2, 16, 162), (3, 5, 77), (1, 384, 64800)]:
        g = torch.Generator().manual_seed(7)
        x = torch.randn(n, c, hw, generator=g) * 3 + 1.5
        gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
        ref = torch.nn.functional.instance_norm(x.double(), weight=gamma.double(), bias=beta.double(), eps=1e-6)
        y = torch.empty(n, c, hw, device=dev)
        xd, gd, bd = x.to(dev), gamma.to(dev), beta.to(dev)
        _lib.check(_lib.lib().ace_instance_norm(_lib.ptr(xd), _lib.ptr(gd), _lib.ptr(bd),
                                                1e-6, _lib.ptr(y), n, c, hw, _lib.current_stream()))
        print(f"instance_norm n={n} c={c} hw={hw}: relmax {rel_max(y, ref):.3e}")


def sht_cases():
    import ace_amd
    import oracle
    for (nlat, nlon, lmax, mmax, grid, n) in [(9, 18, None, None, "lobatto", 1), (9, 18, None, None, "equiangular", 5),
                                              (6, 12, None, None, "legendre-gauss", 3), (12, 24, 8, 9, "legendre-gauss", 4),
                                              (13, 27, None, None, "equiangular", 2), (45, 90, None, None, "legendre-gauss", 16),
                                              (64, 128, 40, 50, "legendre-gauss", 7), (180, 360, 180, 181, "legendre-gauss", 3)]:
        x = torch.randn(n, nlat, nlon, generator=torch.Generator().manual_seed(nlat))
        o_f = oracle.RealSHT(nlat, nlon, lmax, mmax, grid, dtype=torch.float64)
        o_i = oracle.InverseRealSHT(nlat, nlon, lmax, mmax, grid, dtype=torch.float64)
        o_f32 = oracle.RealSHT(nlat, nlon, lmax, mmax, grid)
        c_ref = o_f(x)
        c = ace_amd.RealSHT(nlat, nlon, lmax, mmax, grid)(x.to(dev))
        cz = torch.randn(n, o_i.lmax, o_i.mmax, dtype=torch.complex64, generator=torch.Generator().manual_seed(3))
        xi = ace_amd.InverseRealSHT(nlat, nlon, lmax, mmax, grid)(cz.to(dev))
        print(f"sht {grid} {nlat}x{nlon} L={o_i.lmax} M={o_i.mmax} n={n}: fwd relmax {rel_max(c, c_ref):.3e} "
              f"(cpu fp32 oracle {rel_max(o_f32(x), c_ref):.3e}) inv relmax {rel_max(xi, o_i(cz)):.3e}")
    x = load_golden("gen_sht_input.pt")["x"]
    g = load_golden("ref_sht-regression.pt")["output"]
    gi = load_golden("ref_inverse_sht-regression.pt")["output"]
    c = ace_amd.RealSHT(9, 18)(x.to(dev))
    print(f"sht golden: fwd {rel_max(c, g):.3e} inv {rel_max(ace_amd.InverseRealSHT(9, 18)(c), gi):.3e}")


def net_cases():
    from oracle.sfno import SFNOConfig, SFNOOracle, init_state
    d = load_golden("gen_modulus_sfnonet_case.pt")
    g = load_golden("ref_modulus_sfnonet_output.pt")
    net = build_native_net(SFNOConfig(**d["cfg"]), d["state"], dev)
    with torch.no_grad():
        y = net(d["x"].to(dev))
    print(f"modulus golden net (diagonal, equiangular): relmax {rel_max(y, g):.3e}")
    for name in ["gen_sfno_dhconv_12x24.pt", "gen_sfno_dhconv_equiangular_9x18.pt", "gen_sfno_dhconv_180x360_c8.pt"]:
        d = load_golden(name)
        cfg = SFNOConfig(**{**d["cfg"], "img_shape": tuple(d["cfg"]["img_shape"])})
        state = init_state(cfg, seed=d["seed"])
        x = torch.randn(d["batch"], cfg.in_chans, *cfg.img_shape, generator=torch.Generator().manual_seed(d["seed"] + 1000))
        net = build_native_net(cfg, state, dev)
        with torch.no_grad():
            y, taps = net.forward_with_taps(x.to(dev))
        ref, rt = SFNOOracle(cfg, state, dtype=torch.float64).forward(x, return_blocks=True)
        print(f"{name}: relmax vs reference {rel_max(y, d['y']):.3e}; vs fp64 oracle {rel_max(y, ref):.3e}; "
              f"taps {[f'{rel_max(taps[i + 1], rt[i]):.2e}' for i in range(len(rt))]}")


def ace2_shape():
    """ACE2 shape, teacher-forced per-block error against the fp64 and fp32 CPU oracles + stage times."""
    import ctypes
    from ace_amd import _lib
    from oracle.sfno import SFNOConfig, SFNOOracle, init_state
    cfg = SFNOConfig(in_chans=44, out_chans=50, img_shape=(180, 360), embed_dim=384, num_layers=8, operator_type="dhconv")
    t0 = time.time()
    state = init_state(cfg, seed=0)
    x = torch.randn(1, 44, 180, 360, generator=torch.Generator().manual_seed(1))
    net = build_native_net(cfg, state, dev)
    print(f"init {time.time() - t0:.1f}s", flush=True)
    with torch.no_grad():
        t0 = time.time()
        y, taps = net.forward_with_taps(x.to(dev))
        torch.cuda.synchronize()
        print(f"first forward (incl. table build + weight upload) {time.time() - t0:.1f}s", flush=True)
    L = _lib.lib()
    ns = L.ace_sfno_num_stages()
    ms = (ctypes.c_float * ns)()
    calls = (ctypes.c_int * ns)()
    out = torch.empty(1, 50, 180, 360, device=dev)
    xd = x.to(dev)
    for it in range(3):
        _lib.check(L.ace_sfno_forward_timed(net._native, _lib.ptr(xd), _lib.ptr(out), 1, _lib.current_stream(), ms, calls))
    tot = sum(ms)
    print(f"stage times (ms), total {tot:.3f}:")
    for i in range(ns):
        print(f"   {L.ace_sfno_stage_name(i).decode():32s} {ms[i]:8.3f}  x{calls[i]}")
    t0 = time.time()
    ref32, rt32 = SFNOOracle(cfg, state, dtype=torch.float32).forward(x, return_blocks=True)
    t32 = time.time() - t0
    print(f"cpu fp32 oracle forward {t32:.1f}s ({torch.get_num_threads()} threads)", flush=True)
    ref64, rt64 = SFNOOracle(cfg, state, dtype=torch.float64).forward(x, return_blocks=True)
    print(f"ACE2 shape: out relmax vs fp64 {rel_max(y, ref64):.3e} (cpu fp32 vs fp64 {rel_max(ref32, ref64):.3e}); "
          f"vs cpu fp32 {rel_max(y, ref32):.3e}; l2 vs fp64 {rel_l2(y, ref64):.3e}")
    print("   taps vs fp64:", [f"{rel_max(taps[i + 1], rt64[i]):.2e}" for i in range(8)])
    print("   cpu32 taps vs fp64:", [f"{rel_max(rt32[i], rt64[i]):.2e}" for i in range(8)])


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), torch.__version__)
    which = sys.argv[1:] or ["conv", "norm", "sht", "net", "ace2"]
    table = {"f16x3": f16x3_cases, "conv": conv_cases, "norm": norm_cases, "sht": sht_cases, "net": net_cases, "ace2": ace2_shape}
    for w in which:
        section(w)
        guarded(table[w])
