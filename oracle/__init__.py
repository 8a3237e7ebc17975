"""CPU oracle for the training hot path of EndoscopyDepthEstimation-Pytorch.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.  The
product path (``endoscopydepthestimation-pytorch_amd``) never imports this
package and fails loudly when its HIP library is missing.

What it is: a plain-PyTorch (CPU, fp32) restatement of the reference algorithm
for the hot path, every function citing the reference ``file:line`` it follows.
It is written independently of the reference sources (closed-form math,
functional style), so it can travel to the GPU box where ``/root/reference``
does not exist.

Parity pin: the reference has no tests or golden vectors of its own
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, run in the build container by ``tests/golden/make_golden.py`` (imports
``/root/reference/{models,losses,scheduler,utils}.py`` under three process-local
shims) and committed as small ``.npz`` fixtures under ``tests/golden/``.
``tests/test_oracle_golden.py`` checks every oracle function against them.
"""

from . import geometry, losses, network, schedule, scatter, train_step  # noqa: F401
