"""include/stvo_reference_overloads.h meets a compiler: a -fsyntax-only build of tests/cpp/refcompile/overloads_tu.cpp against the
reference's REAL headers (/root/reference/include) with test-only declarations of the OpenCV / Eigen / line_descriptor names
they mention (tests/cpp/refcompile/standins/ — declarations only, never used for parity).  Checks that the header's guard opens,
that the overloads have exactly the signatures of include/matching.h:50-60, and that the optimizePose body type-checks against the
reference's StereoFrameHandler / StereoFrame / PointFeature / LineFeature / Config / auxiliar.h.  Needs /root/reference (this
container); skipped on the GPU box."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INC = "/root/reference/include"
HERE = os.path.join(ROOT, "tests", "cpp", "refcompile")


@pytest.mark.skipif(not os.path.isdir(REF_INC) or shutil.which("g++") is None, reason="needs the reference tree and g++")
def test_reference_overloads_header_compiles_against_the_reference_headers():
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-Wno-unused-variable", "-Wno-unused-function",
           "-I", os.path.join(HERE, "standins"), "-I", REF_INC, "-I", os.path.join(ROOT, "include"), os.path.join(HERE, "overloads_tu.cpp")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-4000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_reference_overloads_header_is_inert_without_the_reference_types():
    """In this repository's own image (no OpenCV, no Eigen) the header must preprocess to nothing and still be includable."""
    src = '#include "stvo_reference_overloads.h"\n#ifdef STVO_HAVE_REFERENCE_TYPES\n#error "guard opened without the reference"\n#endif\nint main() { return 0; }\n'
    p = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-x", "c++", "-I", os.path.join(ROOT, "include"), "-"], input=src, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
