"""N > 1 path on CPU: world_size-2 gloo processes exercising the Distributed facade, the member -> rank
partition and the ensemble-mean reduce (the only collective on the path), launched the way the driver
launches bench.py (torch.distributed.run, 127.0.0.1)."""

import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent(
    """
    import os, sys, torch
    sys.path.insert(0, os.environ["ACE_ROOT"])
    os.environ["FME_FORCE_CPU"] = "1"
    from ace_amd.distributed import AsyncEnsembleMean, Distributed, EnsembleMean
    d = Distributed.get_instance()
    assert d.world_size == 2 and d.is_distributed()
    # member g lives on rank g % world (fme/ace/data_loading/inference.py:296-298)
    assert d.local_members(5) == ([0, 2, 4] if d.rank == 0 else [1, 3])
    # each rank rolls its own member forward (no data-path collective): a deterministic stand-in state
    g = torch.Generator().manual_seed(100 + d.rank)
    state = torch.randn(3, 4, 8, generator=g)               # (C_out, H, W) of this rank's member
    mean = EnsembleMean(d)(state[None].clone())
    ref = (torch.randn(3, 4, 8, generator=torch.Generator().manual_seed(100))
           + torch.randn(3, 4, 8, generator=torch.Generator().manual_seed(101))) / 2
    torch.testing.assert_close(mean, ref)
    # the side-stream form bench.py and the rollout use (synchronous on CPU tensors): a list of per-name fields, then a tensor
    am = AsyncEnsembleMean((3, 4, 8), "cpu", d)
    am.submit([state[i] for i in range(3)])
    am.wait()
    torch.testing.assert_close(am.result(), ref)
    am.submit(2.0 * state)
    torch.testing.assert_close(am.result(), 2.0 * ref)
    assert am.last_allreduce_ms() is None
    t = torch.tensor([float(d.rank + 1)])
    assert d.reduce_max(t.clone()).item() == 2.0
    assert d.reduce_sum(t.clone()).item() == 3.0
    d.barrier()
    wrapped = d.wrap_module(torch.nn.Linear(2, 2))
    assert list(wrapped.state_dict())[0].startswith("module.")
    d.shutdown()
    sys.stdout.write("RANK_OK %d" % d.rank + chr(10))   # one write per rank: the two ranks share the pipe
    sys.stdout.flush()
    """
)


def test_gloo_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, ACE_ROOT=ROOT, FME_FORCE_CPU="1", OMP_NUM_THREADS="1")
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "RANK_OK 0" in res.stdout and "RANK_OK 1" in res.stdout


def test_single_process_is_a_noop():
    import torch
    from ace_amd.distributed import Distributed
    env_ws = os.environ.pop("WORLD_SIZE", None)
    try:
        Distributed.reset()
        d = Distributed.get_instance()
        assert not d.is_distributed() and d.local_members(3) == [0, 1, 2]
        t = torch.ones(2)
        assert torch.equal(d.reduce_mean(t.clone()), t)
    finally:
        Distributed.reset()
        if env_ws is not None:
            os.environ["WORLD_SIZE"] = env_ws
