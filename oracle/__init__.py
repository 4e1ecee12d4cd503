"""CPU oracle for the SFNO rollout hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It is a CPU restatement (numpy fp64 for the one-time tables, torch-CPU ops for
the per-step arithmetic) of the reference algorithm for the path named by
BASELINE.json's north_star; every function cites the reference file:line it
follows (paths relative to /root/reference).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the timed CPU baseline.  ``ace_amd`` never imports
it: the product path fails loudly when the HIP library is missing.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks this oracle
against the reference's own in-tree golden tensors (copied as data fixtures to
``tests/golden/ref_*.pt``) and against vectors emitted by the reference itself,
imported under stubs in the build container by ``tests/golden/make_golden.py``.

Third-party arithmetic that is NOT under /root/reference: torch-harmonics 0.8.0
(``pyproject.toml:41``) supplies the quadrature rules and the Legendre
recursion called at ``fme/sht_fix.py:50-51,87-107,169-189``.  It is restated
here from its published algorithm (see ``quadrature.py``/``legendre.py``) and
pinned by the reference's goldens at that boundary.
"""

from .quadrature import (  # noqa: F401
    clenshaw_curtiss_weights,
    legendre_gauss_weights,
    lobatto_weights,
)
from .legendre import precompute_legpoly  # noqa: F401
from .sht import InverseRealSHT, RealSHT  # noqa: F401
