import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    # The CPU oracle (plain PyTorch) is what most of the GPU suite's wall time goes to, and torch's default of one thread per logical CPU
    # is the slowest choice on the GPU box's 2 x 128-thread host: bench.py's cpu_baseline measured its training iteration at 1.49 s with
    # 32 threads against 12.4 s with 128 (batch 1).  At most 32 threads, as that leg uses; a container with fewer CPUs keeps its own count.
    try:
        import torch
        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    except Exception:
        pass
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "bench: timing prints without assertions on speed (run by hand with -m bench on a GPU box; not part of -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (its directory name has a hyphen, hence importlib)."""
    return importlib.import_module("endoscopydepthestimation-pytorch_amd")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def checkpoint_from_fixture(g):
    """The reference-written checkpoint of tests/golden/checkpoint_2x64x96.npz (make_golden.py checkpoint_case: the reference's
    DataParallel-wrapped model + torch.optim.SGD after two iterations, saved by ITS utils.save_model) back in the shape torch.load
    returns: {model: {'module.<key>': tensor}, optimizer: {state: {i: {momentum_buffer}}, param_groups: [..]}, epoch, step, validation}."""
    import numpy as np
    import torch
    model = {str(k): torch.from_numpy(np.array(g["model::" + str(k)])) for k in g["model_keys"]}
    index = [int(i) for i in g["opt_params"]]
    state = {i: {"momentum_buffer": torch.from_numpy(np.array(g["opt_state::%d" % i]))} for i in index}
    group = {"lr": float(g["opt_lr"]), "momentum": float(g["opt_momentum"]), "dampening": float(g["opt_dampening"]),
             "weight_decay": float(g["opt_weight_decay"]), "nesterov": bool(g["opt_nesterov"]), "params": index}
    return {"model": model, "optimizer": {"state": state, "param_groups": [group]}, "epoch": int(g["epoch"]), "step": int(g["step"]),
            "validation": float(g["validation"])}
