import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import torch
    from oracle.hostinfo import effective_cpus
    torch.set_num_threads(effective_cpus())      # respect the cgroup CPU quota (CPU oracle speed)


@pytest.fixture(scope="session")
def weights_raw():
    from pips_amd.weights import init_state_dict
    return init_state_dict(0, tamed=False)


@pytest.fixture(scope="session")
def weights_tamed():
    from pips_amd.weights import init_state_dict
    return init_state_dict(0, tamed=True)


@pytest.fixture(scope="session")
def arenas(weights_raw, weights_tamed):
    """Packed device arenas for both weight sets (GPU tests only)."""
    import torch
    from pips_amd import ops
    dev = torch.device("cuda:0")
    return {"raw": ops.pack_weights(weights_raw, dev), "tamed": ops.pack_weights(weights_tamed, dev)}
