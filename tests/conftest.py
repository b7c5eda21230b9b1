import os
import sys

# numpy / torch size their thread pools by the host's CPU count (256 on the GPU boxes), and idle pool threads spin after every parallel
# region: on a box with a cgroup CPU quota (16 there) that runs the container into the throttle -- every thread stopped for the rest
# of a 100 ms period -- which the rate-comparing GPU tests then see as noise.  Nothing in the tests needs the pools.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def scene_small():
    """640x360 stereo sequence (3 pairs), seeded."""
    from ergo_uvo_amd import synth
    sc = synth.Scene(123, 640)
    return [synth.stereo_pair(sc, k, 640, 360) for k in range(3)]


@pytest.fixture(scope="session")
def mono_small():
    """640x360 mono sequence: left views at frames 0, 4, 8 (0.24 m baseline per step so that depth/baseline
    stays below recoverPose's distance threshold of 50), seeded."""
    from ergo_uvo_amd import synth
    sc = synth.Scene(123, 640)
    return [synth.stereo_pair(sc, k, 640, 360)[0] for k in (0, 4, 8)]
