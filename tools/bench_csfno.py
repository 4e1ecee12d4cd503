#!/usr/bin/env python
"""Measurement of the NoiseConditionedSFNO row at the configuration ACE ships today
(configs/baselines/era5/ace-train-config-1-step-pretrain.yaml:93-108: embed 512, 8 layers, 32 isotropic noise channels,
affine_norms, normalize_big_skip) on the 1-degree grid with the 44-in / 50-out channel layout of bench.py:
rollout steps/s through the static-buffer engine (fresh noise every step), per-stage HIP-event times, and the parity of
one forward against the CPU oracle (same weights, same noise).  usage: python tools/bench_csfno.py [--steps K] [--no-oracle]"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ace_amd  # noqa: E402
from ace_amd import _lib  # noqa: E402
from ace_amd.rollout import RolloutEngine  # noqa: E402
from ace_amd.step import NormalizationConfig  # noqa: E402

IMG = (180, 360)
CFG = dict(embed_dim=512, noise_embed_dim=32, noise_type="isotropic", filter_type="linear", use_mlp=True, num_layers=8,
           operator_type="dhconv", affine_norms=True, normalize_big_skip=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--precision", default="f16x3")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    forcing = [f"forcing_{i}" for i in range(8)]
    prog = [f"prog_{i}" for i in range(36)]
    diag = [f"diag_{i}" for i in range(14)]
    names = forcing + prog + diag
    cfg = ace_amd.SingleModuleStepConfig(
        builder=ace_amd.ModuleSelector(type="NoiseConditionedSFNO", config=dict(CFG)),
        in_names=forcing + prog, out_names=prog + diag,
        normalization=NormalizationConfig(means={k: 0.1 for k in names}, stds={k: 1.1 for k in names}))
    torch.manual_seed(0)
    stepper = ace_amd.Stepper.from_config(cfg, ace_amd.DatasetInfo(IMG), device=dev)
    stepper.set_eval()
    net = stepper.modules[0]
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():      # the conditioning weights start at zero in the reference: make them matter
        for k, p in net.named_parameters():
            if "W_scale_2d" in k or "W_bias_2d" in k:
                p.copy_((0.1 * torch.randn(p.shape, generator=g)).to(dev))
    net.set_precision(args.precision)
    K, W = args.steps, args.warmup
    eng = RolloutEngine(stepper, batch=1, n_forward_steps=K, graph=None)
    ic = {n: torch.randn(1, 1, *IMG, generator=g).to(dev) for n in prog}
    fc = {n: torch.randn(1, K + 1, *IMG, generator=g).to(dev) for n in forcing}
    eng.load(ic, fc)
    with torch.no_grad():
        for s in range(W):
            eng._enqueue_step(s % K, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(K):
            eng._enqueue_step(s, False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        L = _lib.lib()
        ns = L.ace_sfno_num_stages()
        ms = (ctypes.c_float * ns)()
        calls = (ctypes.c_int * ns)()
        noise = net.draw_noise(1, dev)
        acc = [0.0] * ns
        for _ in range(3):
            _lib.check(L.ace_sfno_forward_conditioned_timed(net._native, eng.x.data_ptr(), noise.data_ptr(), eng.y.data_ptr(), 1,
                                                            _lib.current_stream(), ms, calls))
            for i in range(ns):
                acc[i] += ms[i] / 3
        stages = {L.ace_sfno_stage_name(i).decode(): {"ms_per_step": round(acc[i], 4), "launches": calls[i]} for i in range(ns)}
        parity = None
        if not args.no_oracle:
            from oracle.csfno import CSFNOConfig, CSFNOOracle
            x = eng.x.clone()
            y = net(x, noise=noise)
            ocfg = CSFNOConfig(in_chans=44, out_chans=50, img_shape=IMG, **{k: v for k, v in CFG.items()
                                                                            if k not in ("filter_type", "operator_type")})
            t1 = time.perf_counter()
            ref = CSFNOOracle(ocfg, {k: v.cpu() for k, v in net.state_dict().items()}, dtype=torch.float32).forward(
                x.cpu(), noise=noise.cpu())
            cpu_s = time.perf_counter() - t1
            parity = {"rel_err_vs_cpu_oracle_fp32": float((y.cpu() - ref).abs().max() / ref.abs().max()),
                      "cpu_oracle_seconds_per_forward": round(cpu_s, 2), "cpu_threads": torch.get_num_threads()}
    print(json.dumps({
        "metric": "rollout steps/sec, NoiseConditionedSFNO (ERA5 baseline configuration), one MI355X, B=1",
        "value": round(K / dt, 3), "unit": "steps/s", "ms_per_step": round(dt / K * 1e3, 3), "steps": K, "warmup": W,
        "precision": args.precision, "config": CFG, "noise": "fresh isotropic draw (torch RNG + native inverse SHT) every step",
        "stages": stages, "parity": parity}))


if __name__ == "__main__":
    main()
