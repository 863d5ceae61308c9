"""The driver runs __graft_entry__.smoke() on the GPU box at the end of every round: the suite runs it too, so that a change of
launch shapes that smoke() asserts on is seen here first (round 5: the frontal shape changed what TEAM_AUTO_LATENCY means for a
140-variable sketch, and nothing but the driver would have noticed)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_the_drivers_smoke_check():
    import ezpz_amd

    if ezpz_amd.device_count() < 1:
        pytest.fail("GPU tests need a HIP device: the product path has no CPU fallback")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as entry

    entry.smoke()
