"""K1m's top-2 folds read matrix-core accumulators through inline asm, which the compiler's hazard recogniser does not cover (ADVICE round 5:
a wrong top-2 was observed in the reverse kernel for lack of wait states).  Every CPU test run re-derives, from the gfx950 assembly of
csrc/match_mfma.hip as the Makefile builds it, the smallest number of wait states between a v_mfma and an inline-asm read of its
destination over every control-flow path (tools/mfma_hazard_check.py; counted conservatively) and fails below the 12 an 8-pass XDL
write needs — a compiler, flag or schedule change cannot shrink the margin silently."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc (cross-compiles without a GPU)")
def test_inline_asm_folds_keep_their_distance_from_the_matrix_writes():
    import mfma_hazard_check as hz
    rep = hz.check()
    names = " ".join(rep)
    assert "hamming_knn2_mfma_kernel" in names and "hamming_knn2_mfma_reverse_kernel" in names
    for k, v in rep.items():
        assert v is None or v["wait_states"] >= hz.REQUIRED, (k, v)


def test_checker_sees_a_planted_hazard():
    """The walker itself: a read 3 instructions behind a v_mfma is reported at distance 3, one behind an s_nop fence is not."""
    import mfma_hazard_check as hz
    asm = """
_Z4testv:
	v_mfma_scale_f32_32x32x64_f8f6f4 v[32:47], v[0:3], v[4:7], v[32:47], v60, v60 op_sel_hi:[0,0,0] cbsz:4 blgp:4
	v_add_f32_e32 v1, v2, v3
	s_cbranch_scc1 .LBB0_2
	v_add_f32_e32 v1, v2, v3
	v_add_f32_e32 v1, v2, v3
.LBB0_2:
	v_min3_f32 v91, v41, v42, v43
	s_nop 7
	s_nop 7
	v_mfma_scale_f32_32x32x64_f8f6f4 v[48:63], v[0:3], v[4:7], v[48:63], v60, v60 op_sel_hi:[0,0,0] cbsz:4 blgp:4
	s_nop 7
	s_nop 7
	v_med3_f32 v92, v48, v49, v50
	s_endpgm
.Lfunc_end0:
"""
    (name, (instrs, labels)), = hz.parse(asm.split("\n")).items()
    w = hz.min_distance(instrs, labels, hz.ASM_READERS)
    assert w[0] == 2 and instrs[w[2]][0] == "v_min3_f32"       # the taken branch skips two instructions
