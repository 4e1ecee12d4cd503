"""The reference's RNG contract for stochastic modules (fme/core/rand.py:39-104, fme/core/random_state.py,
fme/core/stepper_state.py:22-127, fme/ace/stepper/single_module.py:1063-1068) - CPU: the host mirror (ace_amd/rand.py,
StepperState), and the oracle network under a seeded generator against the rollout the REAL reference stepper produced
(tests/golden/gen_rng.pt, tests/golden/make_golden_rng.py)."""
import pytest
import torch

from _util import load_golden
from ace_amd import rand
from ace_amd.step import StepperState


def test_randn_routes_through_the_active_generator():
    shape = torch.Size([3, 5])
    want = torch.randn(shape, generator=torch.Generator().manual_seed(7))
    g = torch.Generator().manual_seed(7)
    torch.manual_seed(0)
    with rand.use_generator(g):
        assert rand.active_generator() is g
        got = rand.randn(shape, dtype=torch.float32, device="cpu")
        like = rand.randn_like(torch.empty(2, 2, dtype=torch.float64))
        with rand.use_generator(None):                      # None: a no-op, the outer generator stays active
            assert rand.active_generator() is g
        g2 = torch.Generator().manual_seed(1)
        with rand.use_generator(g2):                        # nested: restored on exit
            assert rand.active_generator() is g2
        assert rand.active_generator() is g
    assert rand.active_generator() is None
    assert torch.equal(got, want)
    assert like.dtype == torch.float64 and like.shape == (2, 2)
    # the generator advanced in place: the next draw continues the sequence
    g3 = torch.Generator().manual_seed(7)
    torch.randn(shape, generator=g3)
    torch.randn(2, 2, dtype=torch.float64, generator=g3)
    assert torch.equal(torch.randn(4, generator=g), torch.randn(4, generator=g3))
    # without a generator: the global RNG (seedable), on CPU under use_cpu_randn as well
    torch.manual_seed(3)
    a = rand.randn(shape)
    torch.manual_seed(3)
    with rand.use_cpu_randn():
        b = rand.randn(shape, device="cpu")
        c = rand.randn_like(torch.empty(1, 2))
    assert torch.equal(a, b) and c.shape == (1, 2) and rand.USE_CPU_RANDN is False
    assert 0 <= rand.alternate_seed(5) < 2**31 and rand.alternate_seed(5) == rand.alternate_seed(5) != rand.alternate_seed(6)


def test_random_state_and_stepper_state_round_trip():
    with pytest.raises(ValueError, match="CPU"):
        class FakeGen:
            device = torch.device("meta")
        rand.RandomState(generator=FakeGen())
    rs = rand.RandomState.from_seed(11)
    assert rs.to_device() is rs and rs.to_cpu() is rs and rs.pin_memory() is rs and rs.broadcast_ensemble(4) is rs
    assert rs.sample_dim_size() is None and rand.RandomState.per_sample_state_keys() == set()
    torch.randn(100, generator=rs.generator)                # advance
    from ace_amd.corrector import CorrectorState
    st = StepperState(corrector_state=CorrectorState(global_dry_air_mass=torch.ones(2, 1, 1)), random_state=rs)
    sd = st.to_state_dict()
    assert set(sd) == {"corrector_state.present", "corrector_state.global_dry_air_mass", "random_state.present",
                       "random_state.generator_state"}
    back = StepperState.from_state_dict(sd)
    assert torch.equal(back.corrector_state.global_dry_air_mass, torch.ones(2, 1, 1))
    assert torch.equal(torch.randn(8, generator=back.random_state.generator), torch.randn(8, generator=rs.generator))
    assert StepperState.from_state_dict({}).random_state is None and StepperState().to_state_dict() == {}
    empty = StepperState.from_state_dict({"corrector_state.present": torch.tensor(True)})     # present but empty
    assert empty.corrector_state is not None and empty.corrector_state.global_dry_air_mass is None


@pytest.mark.parametrize("case", ["isotropic", "gaussian_groups2"])
def test_oracle_rollout_under_a_seeded_generator_restates_the_reference(case):
    """oracle network (oracle/csfno.py) drawing its noise from the rollout's CPU generator, under the oracle stepper loop, against
    the REAL reference stepper's seeded 3-step rollout: same draws in the same order (isotropic: real parts, imaginary parts)."""
    from oracle import stepper as ostep
    from oracle.csfno import CSFNOConfig, CSFNOOracle
    g = load_golden("gen_rng.pt")[case]
    step = g["state"]["step"]
    weights = {k: v for k, v in step["module"].items() if isinstance(v, torch.Tensor)}
    cfg = CSFNOConfig(in_chans=len(g["in_names"]), out_chans=len(g["out_names"]), img_shape=tuple(g["ic"]["p0"].shape[-2:]), **g["kwargs"])
    net = CSFNOOracle(cfg, weights)
    gen = torch.Generator().manual_seed(g["seed"])
    names = sorted(set(g["in_names"] + g["out_names"]))
    means, stds = {n: torch.tensor(0.1) for n in names}, {n: torch.tensor(1.3) for n in names}
    torch.manual_seed(5)                                     # the global RNG must not matter
    outs = ostep.predict(lambda x: net.forward(x, generator=gen), g["ic"], g["forcing"], len(g["steps"]), g["in_names"], g["out_names"],
                         means, stds)
    for s, want_all in enumerate(g["steps"]):
        for k, want in want_all.items():
            err = float((outs[s][k] - want).abs().max() / want.abs().max())
            assert err <= 2e-6, (case, s, k, err)
    assert torch.equal(gen.get_state(), g["generator_state_after"])          # consumed exactly what the reference consumed
    assert torch.equal(torch.randn(4, generator=gen), g["next_draw"])
    assert g["steps_unseeded_differ"] > 1e-3


def test_rollout_engine_threads_the_random_state_on_the_host():
    """RolloutEngine.predict on the emulated C ABI (tests/_fake_sfno.py: the oracle network behind the real host logic): the
    PrognosticState's StepperState.random_state drives the gaussian conditioning noise of every step, the state rides out on the
    returned PrognosticState, and cutting the 3 steps into windows of 2 + 1 gives the same rollout (random_state.py: the sequence
    does not depend on the chunking) - all against the REAL reference stepper's seeded rollout."""
    import ace_amd
    from ace_amd.rollout import RolloutEngine
    from ace_amd.stepper import PrognosticState
    from _fake_sfno import fake_sfno
    g = load_golden("gen_rng.pt")["gaussian_groups2"]
    T = len(g["steps"])
    with fake_sfno():
        stepper = ace_amd.load_stepper(g["state"], device="cpu").stepper

        def run(windows):
            ic = PrognosticState({k: v.clone() for k, v in g["ic"].items()})
            ic.stepper_state = StepperState(random_state=rand.RandomState.from_seed(g["seed"]))
            outs, t0 = [], 0
            for n in windows:
                eng = RolloutEngine(stepper, batch=2, n_forward_steps=n, graph=None)
                forcing = {k: v[:, t0:t0 + n + 1] for k, v in g["forcing"].items()}
                out, ic = eng.predict(ic, forcing)
                outs.append({k: v.clone() for k, v in out.items()})
                t0 += n
            assert ic.stepper_state.random_state is not None
            return {k: torch.cat([o[k] for o in outs], dim=1) for k in outs[0]}, ic

        torch.manual_seed(1)
        whole, state = run([T])
        torch.manual_seed(2)
        parts, state2 = run([2, 1])
    for k in whole:
        assert torch.equal(whole[k], parts[k]), k
    for s, want_all in enumerate(g["steps"]):
        for k, want in want_all.items():
            err = float((whole[k][:, s] - want).abs().max() / want.abs().max())
            assert err <= 2e-6, (s, k, err)
    assert torch.equal(state.stepper_state.random_state.generator.get_state(), g["generator_state_after"])
    assert torch.equal(state2.stepper_state.random_state.generator.get_state(), g["generator_state_after"])
