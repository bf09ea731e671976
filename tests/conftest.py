import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def port_lib():
    """Host build of the stepper core (oracle/_build/liboracle_port.so); built on demand with g++."""
    so = os.path.join(ROOT, "oracle", "_build", "liboracle_port.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"], stdout=subprocess.DEVNULL)
    from simlib import PortSim
    p = PortSim()
    v, t = p.procedural_mesh()
    p.set_mesh(v, t)
    p.mesh = (v, t)
    return p


@pytest.fixture(scope="session")
def ref_lib(port_lib):
    """The real reference (oracle/_ref/libref_oracle.so) if it has been built; tests that need it skip otherwise."""
    from simlib import RefSim, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref/libref_oracle.so not built (needs /root/reference; golden fixtures cover this case)")
    return RefSim(*port_lib.mesh)


def pytest_sessionfinish(session, exitstatus):
    """RLGPU_COUNTS_OUT=<path> (set by test_gpu_parity.py for its run of the suite against the cut-down contact layout): the stepper's process-wide
    counters of this session, as JSON."""
    out = os.environ.get("RLGPU_COUNTS_OUT")
    if not out:
        return
    import json
    from rlgymppo_cpp_amd.env import BatchedEnv
    env = BatchedEnv(4, 1)
    with open(out, "w") as f:
        json.dump({"big_layout_ticks": env.big_layout_ticks(), "lost_contacts": env.lost_contact_count(), "overflows": env.overflow_counts()}, f)
