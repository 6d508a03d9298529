"""HIP library vs CPU oracle through the front end on random shapes and call sequences (scripts/fuzz_hip_vs_oracle.py):
every node template, ragged sizes, run / reset / forward / fit_bptt / fit_rls / fit_ridge.  Forward quantities of a
directly driven population are bit-equal; trained quantities and anything behind a device-side torch product within 5e-5
(2e-3 for spiking populations behind an input projection).  450 cases over three seeds ran clean on the GPU box in r03;
the test keeps two short seeds."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4])
def test_random_front_end_sequences_hip_equals_oracle(seed, hip_device, capsys):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_hip_vs_oracle
    assert fuzz_hip_vs_oracle.main(30, seed, 1300) == 0, capsys.readouterr().out[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [5, 6])
def test_random_feedback_graphs_and_batched_runs_hip_equals_oracle(seed, hip_device, capsys):
    """scripts/fuzz_coupled_hip_vs_oracle.py: random FeedbackNetwork graphs (2..4 populations, tanh / sigmoid / QIF / LIF,
    all three edge read modes of rp_forward_coupled, shared and per-member launches by size), the same graphs trained through
    their cycles (fit_bptt: the fused adjoint on the GPU against autograd over the per-step walk) and random Network.run_batch
    calls (in-kernel sampling, all three batched forms), HIP against the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_coupled_hip_vs_oracle
    assert fuzz_coupled_hip_vs_oracle.main(25, seed, 1100) == 0, capsys.readouterr().out[-3000:]
