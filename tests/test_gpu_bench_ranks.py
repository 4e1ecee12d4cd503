"""bench.py's N > 1 path on the DEVICE before a multi-GPU node runs it: two ranks under torch.distributed.run sharing this box's one
MI355X, collective backend gloo (RCCL refuses two ranks on one device; ACE_DIST_BACKEND=gloo, ace_amd/distributed.py).  Everything
but RCCL itself is the code of `bench.py --gpus 2`: one RolloutEngine per rank on device buffers, the two warm-up all-reduces and
the timed one through AsyncEnsembleMean's side stream and events, barrier / synchronise bracket, max over ranks, gather of the
per-rank records, ONE JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_sharing_one_device():
    env = dict(os.environ, ACE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 30500 + (os.getpid() % 1000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--precision", "f16x3"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    mg = d["multi_gpu"]
    assert mg["world_size"] == 2 and [r["rank"] for r in mg["ranks"]] == [0, 1] and mg["allreduce_bytes"] == 50 * 180 * 360 * 4
    assert all(r["backend"] == "gloo" and r["allreduce_ms"] is not None and r["allreduce_ms"] > 0 for r in mg["ranks"])
    assert abs(d["value"] - 2 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-2 * d["value"]      # whole-job aggregate over the slower rank's time
    assert d["cpu_baseline"] is None and d["roofline"] is not None                      # single-GPU legs skipped, rank 0's stage times kept
