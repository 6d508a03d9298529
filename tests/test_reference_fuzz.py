"""Differential fuzz against the reference's own classes (build container only; scripts/fuzz_vs_reference.py): random
SEQUENCES of Network calls -- run with / without gradient tracking, in-place state writes, set_var, reset, detach,
fit_bptt windows -- compared call by call (outputs, state, trained parameters).  Tolerance 5e-5 (float32 engine vs the
reference's float64 run; measured worst: 3.6e-7 over 320 sequences)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/rectipy"),
                                reason="the reference is only present in the build container")


@pytest.mark.parametrize("seed", [0, 5])
def test_random_call_sequences_match_the_reference(seed, capsys):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_vs_reference
    assert fuzz_vs_reference.main(30, seed) == 0, capsys.readouterr().out[-2000:]
