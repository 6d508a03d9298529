"""The REFERENCE's own test-suite (rectipy_tests/test_nodes.py, test_edges.py, test_network.py) executed against THIS
package: `rectipy`, `rectipy.nodes`, `rectipy.edges`, ... are aliased to rectipy_amd (arithmetic on the CPU oracle backend:
the build container has no GPU), `pyrates` to an inert stand-in (only `clear_frontend_caches` is called by the tests).
Container-only: the test files are read from /root/reference at run time and never copied; skipped elsewhere.

Each reference test is run as it is.  Known, documented differences are listed in EXPECTED with the reason; everything
else must pass."""
import importlib.util
import os
import sys
import types

import pytest
import torch

REF_TESTS = "/root/reference/rectipy_tests"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_TESTS), reason="the reference is only present in the build container")

# reference test -> why it cannot pass unchanged ("Documented differences")
# test -> (exception type, the statement of the reference test where it stops, why).  Everything before it passes.
EXPECTED = {
    "test_2_1_ratenet_init": (AssertionError, "rnn4.y.dtype == torch.float64",
                              "the engine holds its state in float32 (DESIGN.md section 1); a float64 request warns once"),
    "test_2_3_forward": (NotImplementedError, "rnn3 = RateNet(func, args + (1.0,)",
                         "a callable that cannot be evaluated as given (one argument too many) is refused at "
                         "construction instead of at the first forward; likewise rnn4, whose input is routed into "
                         "tau's slot: not a member of the compiled leaky-integrator family"),
    "test_3_1_diffeq_nodes": (AttributeError, "NodeTemplate.from_yaml",
                              "PyRates template objects are outside the path; template paths (strings) are supported"),
    "test_3_3_edges": (AssertionError, 'get_edge("rnn", "readout_2").weights.dtype == torch.float64',
                       "Network(dtype=float64) warns and builds float32 edges for the float32 engine"),
}


def _alias_modules(oracle_backend):
    """Module objects named like the reference's, exposing this package's classes.  The only adaptation is the default
    device: the reference defaults to 'cpu', this package to 'cuda:0' (there is no GPU in the build container), so the
    classes that take a `device` are subclassed with that default; the arithmetic backend is the CPU oracle, installed
    as the package's default backend for the duration of a test."""
    import rectipy_amd
    from rectipy_amd import edges, network, nodes, observer, rls, utility

    def on_cpu(cls):
        import inspect
        sig = inspect.signature(cls.__init__)

        class Shim(cls):
            def __init__(self, *a, **kw):
                if "device" not in sig.bind_partial(self, *a, **kw).arguments:  # (also when passed by position)
                    kw["device"] = "cpu"
                super().__init__(*a, **kw)

            if hasattr(cls, "from_pyrates"):
                @classmethod
                def from_pyrates(klass, *a, **kw):
                    kw.setdefault("device", "cpu")
                    return super(Shim, klass).from_pyrates(*a, **kw)
        Shim.__name__ = cls.__name__
        Shim.__qualname__ = cls.__qualname__
        return Shim

    top = types.ModuleType("rectipy")
    for name in rectipy_amd.__all__:
        setattr(top, name, getattr(rectipy_amd, name))
    top.Network = on_cpu(rectipy_amd.Network)
    top.FeedbackNetwork = on_cpu(rectipy_amd.FeedbackNetwork)
    m_nodes = types.ModuleType("rectipy.nodes")
    for name in ("InstantNode", "Spike", "SpikeNet", "MultiSpikeResetNet", "RateNet", "SpikeResetNet"):
        setattr(m_nodes, name, getattr(nodes, name))
    m_nodes.RateNet = on_cpu(nodes.RateNet)
    m_nodes.SpikeResetNet = on_cpu(nodes.SpikeResetNet)
    m_edges = types.ModuleType("rectipy.edges")
    m_edges.Linear, m_edges.RLS = edges.Linear, rls.RLS
    pr = types.ModuleType("pyrates")
    pr.clear_frontend_caches = lambda *a, **k: None
    pr.clear = lambda *a, **k: None
    pr.NodeTemplate = type("NodeTemplate", (), {})
    pr.CircuitTemplate = type("CircuitTemplate", (), {})
    mods = {"rectipy": top, "rectipy.nodes": m_nodes, "rectipy.edges": m_edges, "rectipy.network": network,
            "rectipy.observer": observer, "rectipy.utility": utility, "pyrates": pr}
    top.nodes, top.edges, top.network, top.observer, top.utility = m_nodes, m_edges, network, observer, utility
    return mods


def _collect():
    if not os.path.isdir(REF_TESTS):
        return []
    items = []
    for fname in ("test_nodes.py", "test_edges.py", "test_network.py"):
        src = open(os.path.join(REF_TESTS, fname)).read()
        for line in src.splitlines():
            if line.startswith("def test_"):
                items.append((fname, line[4:line.index("(")]))
    return items


@pytest.mark.parametrize("fname,test", _collect())
def test_reference_test_passes_against_this_package(oracle_backend, fname, test):
    from rectipy_amd import engine
    mods = _alias_modules(oracle_backend)
    saved = {k: sys.modules.get(k) for k in mods}
    saved_backend = engine._DEFAULT_BACKEND
    sys.modules.update(mods)
    engine._DEFAULT_BACKEND = oracle_backend
    try:
        spec = importlib.util.spec_from_file_location(f"_ref_{fname[:-3]}", os.path.join(REF_TESTS, fname))
        module = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(module)
        fn = getattr(module, test)
        if test in EXPECTED:
            kind, where, why = EXPECTED[test]
            with pytest.raises(kind) as info:
                fn()
            ref_file = os.path.join(REF_TESTS, fname)
            lines = [tb.lineno for tb in info.traceback if str(tb.path) == ref_file]
            assert lines, "the failure did not come from the reference test's own statements"
            with open(ref_file) as f:
                stopped_at = f.readlines()[lines[-1]]  # (traceback entries are 0-based)
            assert where in stopped_at, f"stopped at an unexpected statement: {stopped_at.strip()}"
            pytest.xfail(why)
        fn()
    finally:
        engine._DEFAULT_BACKEND = saved_backend
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


# ---- the reference's documentation scripts (its user-facing gallery), unmodified ---------------------------------------
REF_DOCS = "/root/reference/documentation"
# script -> seconds it takes on the CPU oracle here.  The long ones run only with RECTIPY_AMD_SLOW_DOCS=1.
DOC_SCRIPTS = {"interfaces/simulation.py": 1, "interfaces/model_definition.py": 1, "models/leaky_integrator.py": 3,
               "models/lif.py": 1, "models/qif.py": 1,
               "interfaces/torch_integration.py": 62, "ridge_example.py": 100}
# Not run: scripts that the reference itself rejects today because they use an older API --
#   qif_example.py, interfaces/observer.py: `spike_def=` without `reset_var` -> the ValueError of network.py:292-294 (this
#     package raises the same message);  interfaces/train_test.py: integer class targets are cast to the network dtype
#     (network.py:695) before `nll` -> torch's "expected scalar type Long";  bptt_rate_neurons.py: a 3-D ndarray of epochs is
#     iterated as steps (network.py:693-701);  rls_example.py, rl_example.py: `Network.from_yaml` does not exist;
#   bptt_spiking_neurons_ffwd.py: `spike_def=`;  bptt_spiking_neurons_recurrent.py: current API, but 1000 epochs x 20000
#     Python-level steps (hours);  rnn_tryout.py: hard-wired to "cuda:0".


@pytest.mark.skipif(not os.path.isdir(REF_DOCS), reason="the reference is only present in the build container")
@pytest.mark.parametrize("script", list(DOC_SCRIPTS))
def test_reference_documentation_script_runs_unmodified(oracle_backend, script, tmp_path, monkeypatch):
    if DOC_SCRIPTS[script] > 5 and os.environ.get("RECTIPY_AMD_SLOW_DOCS", "0") != "1":
        pytest.skip(f"takes {DOC_SCRIPTS[script]} s on the CPU oracle; set RECTIPY_AMD_SLOW_DOCS=1")
    import runpy
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from rectipy_amd import engine
    monkeypatch.setattr(plt, "show", lambda *a, **k: None)
    monkeypatch.chdir(tmp_path)
    mods = _alias_modules(oracle_backend)
    saved = {k: sys.modules.get(k) for k in mods}
    saved_backend = engine._DEFAULT_BACKEND
    sys.modules.update(mods)
    engine._DEFAULT_BACKEND = oracle_backend
    try:
        runpy.run_path(os.path.join(REF_DOCS, script), run_name="__main__")
    finally:
        plt.close("all")
        engine._DEFAULT_BACKEND = saved_backend
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
