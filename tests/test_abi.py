"""The C-ABI library loads without a GPU, exports every symbol include/rectipy_hip.h declares and
agrees with the ctypes mirror on struct layouts.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from rectipy_amd import _abi, _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()
    return _lib.load()


def declared_symbols():
    """(product, measurement-only) entry points of include/rectipy_hip.h: the latter sit in its RP_MEASUREMENT section."""
    text = open(os.path.join(ROOT, "include", "rectipy_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    head, rest = text.split("#ifdef RP_MEASUREMENT", 1)
    measure, tail = rest.split("#endif", 1)
    names = lambda t: sorted(set(re.findall(r"\b(rp_[a-z0-9_]+)\s*\(", t)))
    return names(head + tail), names(measure)


def exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    # (functions only: the thread-local error text rp_err_buf is a data symbol shared by the library's translation units)
    return sorted(line.split()[-1] for line in out.splitlines() if line.split()[-1].startswith("rp_") and line.split()[-2] == "T")


def test_every_declared_symbol_is_exported(lib):
    syms, measure = declared_symbols()
    assert len(syms) >= 30 and len(measure) >= 10
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/rectipy_hip.h but not exported"
    assert sorted(_abi.EXPORTS) == syms, "python EXPORTS list is out of sync with the header"
    assert sorted(_abi.MEASUREMENT_EXPORTS) == measure, "python MEASUREMENT_EXPORTS list is out of sync with the header"
    # the product library exports exactly the product entry points: no knob, no sweep hook
    assert exported(_lib.lib_path(False)) == syms


def test_measurement_build_adds_the_knobs_and_nothing_to_the_product():
    """-DRP_MEASUREMENT: the product entry points plus the rp_set_* knobs; a second handle next to the product library."""
    syms, measure = declared_symbols()
    path = build.build(measurement=True)
    assert exported(path) == sorted(syms + measure)
    m = _lib.load(True)
    assert m is not _lib.load(False) and m.rp_abi_version() == _abi.RP_ABI_VERSION
    assert m.rp_set_tuning(2, 1, 1, 0) == _abi.RP_OK and m.rp_set_tuning(0, 0, 1, 0) == _abi.RP_OK
    assert not hasattr(_lib.load(False), "rp_set_tuning")
    assert os.path.getsize(path) > os.path.getsize(_lib.lib_path(False))  # (the sweep-only kernel instances)


def test_struct_layouts_and_version(lib):
    assert lib.rp_abi_version() == _abi.RP_ABI_VERSION
    assert lib.rp_sizeof_net() == C.sizeof(_abi.RpNet)
    assert lib.rp_sizeof_run() == C.sizeof(_abi.RpRun)
    assert lib.rp_sizeof_bwd() == C.sizeof(_abi.RpBwd)
    assert lib.rp_sizeof_member() == C.sizeof(_abi.RpMember)


def test_argument_validation_without_gpu(lib):
    """Bad arguments are rejected before any HIP call, with the documented status codes."""
    net = _abi.RpNet()
    run = _abi.RpRun()
    assert lib.rp_forward(C.byref(net), C.byref(run), None) == _abi.RP_ERR_SHAPE  # n = 0
    net.model = 99
    assert lib.rp_forward(C.byref(net), C.byref(run), None) == _abi.RP_ERR_ARG
    assert b"model" in lib.rp_last_error()
    assert lib.rp_outer_accum(0, 1, 1.0, None, 0, None, 0, None, 0, None) == _abi.RP_ERR_ARG
    # rp_tune_forward: arguments are checked, and shapes that do not stream from HBM return before any HIP call
    assert lib.rp_tune_forward(None, None) == _abi.RP_ERR_ARG
    small = _abi.RpNet()
    small.n = small.rows = 512
    small.ldw = 512
    small.W = 16  # (never dereferenced: 512 x 512 has nothing to measure)
    assert lib.rp_tune_forward(C.byref(small), None) == _abi.RP_OK
    small.ldw = 100
    assert lib.rp_tune_forward(C.byref(small), None) == _abi.RP_ERR_SHAPE
    # rp_install_forward_tuning: the configuration must name a compiled kernel instance; small shapes return before any HIP call
    assert lib.rp_install_forward_tuning(None, 2, 1, 683, 600.0) == _abi.RP_ERR_ARG
    assert lib.rp_install_forward_tuning(C.byref(small), 2, 1, 683, 600.0) == _abi.RP_ERR_SHAPE  # ldw < n
    small.ldw = 512
    assert lib.rp_install_forward_tuning(C.byref(small), 3, 1, 683, 600.0) == _abi.RP_ERR_ARG
    assert b"no step kernel instance" in lib.rp_last_error()
    assert lib.rp_install_forward_tuning(C.byref(small), 2, 1, -1, 600.0) == _abi.RP_ERR_ARG
    assert lib.rp_install_forward_tuning(C.byref(small), 2, 1, 683, 600.0) == _abi.RP_OK
    assert lib.rp_bwd_partial_rows(32768) == 24 and lib.rp_bwd_partial_rows(65536) == 12 and lib.rp_bwd_partial_rows(8192) == 64 and lib.rp_bwd_partial_rows(10) == 2
    with pytest.raises(ValueError):
        _lib.check(_abi.RP_ERR_ARG)
    with pytest.raises(RuntimeError):
        _lib.check(_abi.RP_ERR_SHAPE)


def test_product_refuses_cpu_tensors():
    """No CPU fallback: the HIP backend refuses to integrate on a CPU device."""
    import numpy as np
    from rectipy_amd import Network
    net = Network(dt=1e-2, device="cpu")
    with pytest.raises(_lib.HipLibraryError):
        net.add_diffeq_node("rnn", "neuron_model_templates.rate_neurons.leaky_integrator.tanh",
                            input_var="li_op/I_ext", output_var="li_op/v", weights=np.eye(4), source_var="tanh_op/r",
                            target_var="li_op/r_in")


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "rectipy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{f} imports the oracle"
                assert "rnn_oracle" not in text or f in ("rnn_forward.hip", "rnn_backward.hip", "rnn_edges.hip", "outer_gemm.hip", "rp_dot.h"), f  # (comments naming the mirror)


def test_python_constants_match_header():
    """Every `#define RP_* <int>` of include/rectipy_hip.h that _abi.py mirrors carries the same value."""
    import re

    from rectipy_amd import _abi
    text = open(os.path.join(ROOT, "include", "rectipy_hip.h")).read()
    checked = 0
    for name, val in re.findall(r"^#define\s+(RP_[A-Z0-9_]+)\s+\(?(-?\d+)\)?\s*(?:/\*.*)?$", text, flags=re.M):
        if hasattr(_abi, name):
            assert getattr(_abi, name) == int(val), name
            checked += 1
    assert checked >= 30
    assert _abi.RP_BWD_TRANSPOSE_MAX_N == 4096
