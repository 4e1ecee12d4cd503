"""Inputs of the reference's OWN block-level regression goldens (fme/core/benchmark/testdata/csfno_block{,_8_groups}-regression.pt,
produced by fme/core/models/conditional_sfno/benchmark.py:100-119 under fme/core/benchmark/test_benchmark.py:44-57: set_seed(0) =
numpy seed 1, random seed 2, torch seed 3, fme/core/rand.py:20-32).  The held files carry the OUTPUT only; the block's random
parameters and inputs are RNG draws of that seeded run.  This script - build container only - repeats the run with the REAL reference
block imported under stubs (oracle/ref_loader.load_csfno), checks that it reproduces the held output (so the draws are the ones
the golden was made with) and stores state_dict + inputs next to a copy of the held output:  tests/golden/gen_csfno_block.pt."""
import importlib
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_loader  # noqa: E402

HELD = "/root/reference/fme/core/benchmark/testdata"


def main():
    ref_loader.load_csfno()
    bm = importlib.import_module("fme.core.models.conditional_sfno.benchmark")
    null_timer = importlib.import_module("fme.core.benchmark.timer").NullTimer
    out = {}
    for name, groups in (("csfno_block", 1), ("csfno_block_8_groups", 8)):
        held = torch.load(os.path.join(HELD, f"{name}-regression.pt"), map_location="cpu", weights_only=False)["output"]
        np.random.seed(1)
        random.seed(2)
        torch.manual_seed(3)
        bench = bm.get_block_benchmark(groups).new_for_regression()
        y = bench.run_instance(null_timer())["output"]
        torch.testing.assert_close(y, held)                 # the reference's own bar: these ARE the golden's draws
        ctx = bench.context
        out[name] = dict(groups=groups, state={k: v.detach().clone() for k, v in bench.block.state_dict().items()},
                         x=bench.x.clone(), noise=ctx.noise.clone(), labels=ctx.labels.clone(), embedding_pos=ctx.embedding_pos.clone(),
                         held_output=held.clone(), grid=bench.block.filter.filter.forward_transform.grid,
                         lmax=bench.block.filter.filter.forward_transform.lmax, mmax=bench.block.filter.filter.forward_transform.mmax)
        print(name, "reproduced; max |y - held| =", float((y - held).abs().max()))
    torch.save(out, os.path.join(HERE, "gen_csfno_block.pt"))


if __name__ == "__main__":
    main()
