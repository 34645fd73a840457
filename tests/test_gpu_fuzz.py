"""A short, seeded run of each randomised-parity script of this directory (the long runs are done by hand on a GPU box:
python tests/fuzz_*.py --seconds N):
image front-ends (ORB, LSD incl. lsd_refine 1, LBD), the per-stage entry points (match, grid match, normal equations, optimizePose) and
the device-resident per-frame pipeline, the batched f2f + optimizePose path, images in -> poses out, and the C++ host mirror (the imagesStVO loop of stvo-pl_amd/app), each against the oracle.  Round 6: the first of them found a one-off in numOfPixels that the fixed
test images never met."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,cases", [("fuzz_frontends", 60), ("fuzz_entry_points", 400), ("fuzz_pipeline", 120), ("fuzz_handler", 30), ("fuzz_track_batched", 40), ("fuzz_images", 12)])
def test_randomised_parity_short_run(tool, cases):
    mod = __import__(tool)
    assert mod.main(["--seconds", "120", "--cases", str(cases), "--seed", "7"]) == 0
