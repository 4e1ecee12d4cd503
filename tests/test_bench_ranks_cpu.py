"""The N > 1 path of bench.py, executed: `bench.main` itself under `torch.distributed.run --nproc-per-node 2` with gloo and a
stand-in engine on host tensors - argument checks, the Distributed facade, the two warm-up all-reduces, the timed region with the
ensemble-mean submit on the (here synchronous) side path, the max over ranks, `all_gather_object` of the per-rank records, and the
one JSON line printed by rank 0.  The devices are the only thing the driver's `bench.py --gpus 8` adds.
Reference semantics: fme/core/distributed/torch_distributed.py:47-87, 130-132; fme/ace/aggregator/one_step/ensemble.py:93-112,299."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent(
    """
    import json, os, sys, time, torch
    sys.path.insert(0, os.environ["ACE_ROOT"])
    os.environ["FME_FORCE_CPU"] = "1"
    import bench

    class StubEngine:
        '''what bench.timed_window needs of a RolloutEngine: per-step enqueue on static (B, T, H, W) output buffers'''
        graph_mode = os.environ.get("STUB_GRAPH", "step")
        out_names = ["a", "b", "c"]

        def __init__(self, K, rank):
            self.K, self.rank, self.calls = K, rank, []
            self.out = {n: torch.zeros(1, K, 4, 8) for n in self.out_names}

        def _enqueue_step(self, s, use_library_graph):
            assert use_library_graph == (self.graph_mode == "step")
            self.calls.append(s)
            for i, n in enumerate(self.out_names):          # member `rank`, field i, step s
                self.out[n][0, s] = 100.0 * (self.rank + 1) + 10.0 * i + s
            time.sleep(0.002 * (1 + self.rank))              # rank 1 is the slower member: the line reports the MAX over ranks

        def run_window(self):
            for s in range(self.K):
                self._enqueue_step(s, False)

    engines = []

    def factory(K, rank):
        engines.append(StubEngine(K, rank))
        return engines[-1]

    K, W = 5, 2
    res = bench.main(["--gpus", "2", "--steps", str(K), "--warmup", str(W), "--graph", StubEngine.graph_mode], engine_factory=factory, device="cpu")
    rank = int(os.environ["RANK"])
    eng = engines[0]
    # warm-up W steps (or one window) untimed, then exactly K timed steps
    want = (list(range(K)) if eng.graph_mode == "window" else [s % K for s in range(W)]) + list(range(K))
    assert eng.calls == want, (eng.calls, want)
    if rank != 0:
        # ensemble mean of the final state over the two members, on every rank
        mean = res["ensemble_mean"]
        for i in range(3):
            assert torch.allclose(mean[i], torch.full((4, 8), 150.0 + 10.0 * i + (K - 1))), mean[i][0, 0]
        assert res["dt"] >= K * 0.004 * 0.9                  # the max over ranks, not this rank's own
    sys.stdout.flush()
    """
)


def _run(tmp_path, graph):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, ACE_ROOT=ROOT, FME_FORCE_CPU="1", OMP_NUM_THREADS="1", STUB_GRAPH=graph)
    port = 31500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout                        # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_rank_orchestration_world_size_2(tmp_path):
    d = _run(tmp_path, "step")
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "weak" and d["higher_is_better"] is True
    # whole-job aggregate: 2 members x 5 steps over the SLOWER rank's time (rank 1 sleeps 4 ms per step)
    assert d["ms_per_step"] >= 4.0 * 0.9 and abs(d["value"] - 2 * 5 / (d["ms_per_step"] * 5e-3)) < 1e-2 * d["value"]
    mg = d["multi_gpu"]
    assert mg["world_size"] == 2 and mg["allreduce_bytes"] == 3 * 4 * 8 * 4
    assert [r["rank"] for r in mg["ranks"]] == [0, 1] and all(r["backend"] == "gloo" and r["device"] == "cpu" for r in mg["ranks"])
    assert d["config"]["members"] == 2 and "all-reduce" in d["config"]["collective"]
    assert d["cpu_baseline"] is None and d["roofline"] is None      # single-GPU legs: not part of an N > 1 line


def test_bench_rank_orchestration_window_graph(tmp_path):
    d = _run(tmp_path, "window")
    assert d["n_gpus"] == 2 and d["config"]["graph"] == "window"
