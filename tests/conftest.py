import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# fp32 parity bound of BASELINE.json's north_star: 1e-3 relative
REL_TOL = 1e-3


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _fit_cpu_threads_to_the_quota()


def _fit_cpu_threads_to_the_quota():
    """The CPU oracle is the checker of every parity test.  torch sizes its thread pool by the host's logical CPUs (256 on the
    GPU box) while the container may run 16 of them at a time (cgroup cpu.max): hundreds of runnable threads under a 16-CPU
    quota are throttled, not parallel -- an oracle forward then takes several times longer.  Size the pool to the quota."""
    try:
        import torch
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, int(float(q) / float(per)))
            if n < torch.get_num_threads():
                torch.set_num_threads(n)
    except Exception:       # noqa: BLE001 -- no cgroup v2 / no torch: leave the defaults
        pass


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rel_err(a, b):
    """max |a-b| / max |b| (the relative measure the 1e-3 bound is stated in)."""
    import torch
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from diffwave_sashimi_amd import _lib
    _lib.load()  # fail loudly if libdws.so is missing: there is no fallback
    return torch.device("cuda:0")
