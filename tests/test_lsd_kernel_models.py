"""CPU models of the two pieces of wave-level logic in stvo-pl_amd/csrc/lsd_kernels.hip that the GPU parity tests can only see from
the outside (tools/experiments/): the guess + verification of a sub-group's candidates (lsd_grow_kernel<true>: every region of a
synthetic scene grown by the lane model and by the sequential loop — identical lists, flags and angles), and the protocol of the
16-waves-per-image kernel (lsd_grow_waves_kernel: random interleavings of committer / speculator steps give the sequential regions;
without the validation they do not — the negative control shows the model can fail).  They link the oracle for fastAtan2 / sincos."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "tools", "experiments")
ORACLE = [os.path.join(ROOT, "oracle", f) for f in ("stvo_lsd_oracle.c", "stvo_orb_oracle.c")]


def build(tmp_path, name):
    exe = str(tmp_path / name)
    subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(EXP, name + ".c"), *ORACLE, "-lm"], check=True, capture_output=True)
    return exe


@pytest.mark.parametrize("noise", ["3", "12"])
def test_guess_and_verification_equals_the_sequential_growth(tmp_path, noise):
    r = subprocess.run([build(tmp_path, "lsd_resolve_model"), noise], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert " 0 differ; flags equal: 1" in r.stdout


def test_wave_protocol_commits_the_sequential_regions(tmp_path):
    exe = build(tmp_path, "lsd_waves_model")
    r = subprocess.run([exe, "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.count("identical") == 6 and "DIFFERENT" not in r.stdout, r.stdout
    r = subprocess.run([exe, "3", "no-validation"], capture_output=True, text=True, timeout=300)   # negative control
    assert r.returncode != 0 and "DIFFERENT" in r.stdout, r.stdout
