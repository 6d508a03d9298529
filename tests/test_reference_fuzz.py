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


@pytest.mark.parametrize("seed", [0, 3])
def test_random_chains_with_edge_variants_match_the_reference(seed, capsys):
    """inp --edge--> rnn --edge--> out with random plain / masked / delay / filter / delay + filter edges, two consecutive
    runs: the reference walks the graph step by step, this package evaluates the edges per window (fused plan)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_edges_vs_reference
    assert fuzz_edges_vs_reference.main(25, seed) == 0, capsys.readouterr().out[-2000:]


def test_edge_classes_match_the_reference_classes_step_by_step_and_per_window():
    """The four stateful / masked edge classes against the reference's own classes on random small cases -- delays
    with and WITHOUT a zero delay (LinearMemoryFilter steps its buffer twice per call in the reference, edges.py:143-146 ->
    LinearMemory.forward; fixture G27 only had delay sets containing 0), repeated delays, a window split in two."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import import_reference
    import_reference()
    from rectipy import edges as ref
    from rectipy_amd import edge_variants as amd
    rng = np.random.default_rng(0)
    for trial in range(150):
        n_in, n_out = int(rng.integers(1, 7)), int(rng.integers(1, 5))
        delays = rng.integers(0, int(rng.integers(1, 5)), n_in)
        F = rng.standard_normal((n_in, n_in)) * 0.3
        W = rng.standard_normal((n_out, n_in))
        mask = (rng.uniform(size=(n_out, n_in)) > 0.5).astype(np.float64)
        T = int(rng.integers(2, 12))
        X = torch.tensor(rng.standard_normal((T, n_in)))
        for name, kw in (("LinearMemoryFilter", dict(delays=delays, filter_weights=F)), ("LinearMemory", dict(delays=delays)),
                         ("LinearFilter", dict(filter_weights=F)), ("LinearMasked", dict(mask=mask))):
            mk = lambda mod: getattr(mod, name)(n_in, n_out, weights=W.copy(), dtype=torch.float64,
                                                **{k: np.array(v) for k, v in kw.items()})
            r, a, a2 = mk(ref), mk(amd), mk(amd)
            with torch.no_grad():
                want = torch.stack([r.forward(X[t]) for t in range(T)])
                step = torch.stack([a.forward(X[t]) for t in range(T)])
                win = torch.cat([a2.forward_window(X[:T // 2]), a2.forward_window(X[T // 2:])])
            assert float((want - step).abs().max()) < 1e-12 and float((want - win).abs().max()) < 1e-12, (name, delays)
            if hasattr(r, "buffer"):
                assert float((r.buffer - a.buffer).abs().max()) < 1e-12 and float((r.buffer - a2.buffer).abs().max()) < 1e-12


@pytest.mark.parametrize("seed", [0, 3])
def test_random_qif_call_sequences_match_the_reference(seed, capsys):
    """Spiking population (reference SpikeResetNet, nodes.py:382-392): run / reset / state write / surrogate-gradient
    fit_bptt sequences, tolerance 2e-3 on trajectories of s (measured worst 1.8e-6 over 40 sequences)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_qif_vs_reference
    assert fuzz_qif_vs_reference.main(15, seed) == 0, capsys.readouterr().out[-2000:]


@pytest.mark.parametrize("seed", [0, 2])
def test_random_readout_training_sequences_match_the_reference(seed, capsys):
    """fit_rls (array and epochs, random strides) / run / reset / test() sequences and a closing fit_ridge, on
    inp -> rnn -(rls)-> out; relative tolerance 2e-3 on weights, P, outputs, losses (measured worst 2.2e-4 over 150)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_training_vs_reference
    assert fuzz_training_vs_reference.main(20, seed) == 0, capsys.readouterr().out[-2000:]


@pytest.mark.parametrize("seed", [0, 2])
def test_random_population_chains_match_the_reference(seed, capsys):
    """Two or three rate populations of different sizes in a chain, with and without feedback edges (FeedbackNetwork):
    run / reset (all, one node) / forward() sequences; tolerance 5e-5 (measured worst 3.6e-7 over 220)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_chains_vs_reference
    assert fuzz_chains_vs_reference.main(30, seed) == 0, capsys.readouterr().out[-2000:]


@pytest.mark.parametrize("seed", [1, 2])
def test_random_trainable_chains_match_the_reference(seed, capsys):
    """The chains above with trainable populations / edges and fit_bptt windows (sgd, adam) mixed into the call
    sequences.  Seeds 1 and 2 hold the cases that found the None-vs-zero gradient difference: a window too short for an
    upstream population's parameters to reach its predictions leaves their grad None in the reference (Adam skips
    them), where a fused window's adjoint returned zeros (Adam's moments kept moving them)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_chains_vs_reference
    assert fuzz_chains_vs_reference.main(40, seed, tol=2e-3, trainable=True) == 0, capsys.readouterr().out[-2000:]


@pytest.mark.parametrize("seed", [0, 1])
def test_random_chains_with_spiking_populations_match_the_reference(seed, capsys):
    """Chains mixing tanh rate populations and QIF SpikeResetNet populations (thresholds low enough to spike inside every
    run), with and without feedback edges; s-trajectories and states to 2e-3 (measured worst 1.8e-6 over 150)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_chains_vs_reference
    assert fuzz_chains_vs_reference.main(20, seed, tol=2e-3, spiking=True) == 0, capsys.readouterr().out[-2000:]


@pytest.mark.parametrize("seed", [3, 6])
def test_random_trainable_chains_with_feedback_match_the_reference(seed, capsys):
    """Trainable chains WITH feedback edges (also trained): run / reset / forward / fit_bptt (sgd, adam) sequences against the
    reference's FeedbackNetwork -- the fused adjoint of coupled populations (one rp_backward_coupled call per window) where
    the plan allows it, the per-step autograd otherwise (a feedback edge from a population already stepped).  Found with it
    (r04): the reference's optimiser holds the nodes at a feedback edge twice (kept, see FeedbackNetwork.parameters); Adam
    and windows too short for a parameter to reach the predictions through the cycle (seed 3: grad None in the reference,
    zeros from a fused window -- FeedbackNetwork._unreached_parameters now follows the timing of feedback reads).
    Measured worst deviation over 360 sequences: 1.2e-6."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_chains_vs_reference
    assert fuzz_chains_vs_reference.main(30, seed, tol=2e-3, trainable=True, train_fb=True) == 0, capsys.readouterr().out[-2000:]


@pytest.mark.parametrize("seed", [2, 3])
def test_random_readout_training_sequences_on_the_blocked_sweep_match_the_reference(seed, capsys):
    """The read-out training sequences above with the BLOCKED symmetric RLS sweep forced at these sizes (the default from
    n_in = 8192: P streamed once per 8 updates, rp_rls.block): fit_rls on arrays / epochs with random strides, run, reset,
    test(), a closing fit_ridge, against the reference's float64 run.  Measured worst relative deviation over 180 sequences:
    1.7e-5 with beta = 1 (a pass per update: 8e-6); with beta < 1 the reference's recursion itself loses the definiteness of P
    (z = P x / beta without dividing P) and amplifies float32 rounding in either form -- worst per 30 sequences 5e-6 ... 1.3e-3
    for a pass per update, 3e-6 ... 6.6e-2 blocked; these two seeds: 3e-6 and 9e-6."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_training_vs_reference
    assert fuzz_training_vs_reference.main(20, seed, blocked=True) == 0, capsys.readouterr().out[-2000:]


@pytest.mark.parametrize("seed", [0, 1])
def test_random_spiking_chains_trained_through_feedback_match_the_reference(seed, capsys):
    """Chains mixing tanh and QIF populations with (trained) feedback edges and trainable population parameters: run / reset /
    forward sequences closed by ONE fit_bptt window with spikes inside (the reference's detach leaves the spike argument of a
    SpikeResetNet attached, so it allows one backward per network) -- surrogate gradients through the cycle, the lagging
    read of a spiking source included, by the fused coupled adjoint where the plan allows it.  Measured worst deviation
    over 75 sequences: 1.4e-6."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_chains_vs_reference
    assert fuzz_chains_vs_reference.main(20, seed, tol=2e-3, spiking=True, train_fb=True, spiking_fit=True) == 0, \
        capsys.readouterr().out[-2000:]
