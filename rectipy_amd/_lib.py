"""Loader for the gfx950 C-ABI library.  There is NO fallback: if the library is missing
or was built against a different ABI, importing compute paths fails loudly."""
import ctypes as C
import os

from . import _abi

_LIBS = {}


class HipLibraryError(RuntimeError):
    pass


def measurement() -> bool:
    """RECTIPY_AMD_MEASUREMENT=1: the process uses the measurement build (benchmark-only knobs rp_set_*, sweep-only kernel
    instances, ablation kernels; `python -m rectipy_amd.build --measurement`).  For scripts/ only."""
    return os.environ.get("RECTIPY_AMD_MEASUREMENT", "0") == "1"


def lib_path(measure=None):
    measure = measurement() if measure is None else measure
    name = "librectipy_hip_measure.so" if measure else "librectipy_hip.so"
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "_C", name)


def load(measure=None):
    """Load (once) and return the ctypes handle of librectipy_hip.so (``measure=True``: of the measurement build, a
    separate handle with its own knobs -- tests use it to show that a knob never changes the bits)."""
    measure = measurement() if measure is None else bool(measure)
    if measure in _LIBS:
        return _LIBS[measure]
    path = lib_path(measure)
    if not os.path.exists(path):
        raise HipLibraryError(
            f"{path} not found. Build it with `python -m rectipy_amd.build{' --measurement' if measure else ''}` (needs "
            "hipcc, no GPU). rectipy_amd has no CPU fallback for the integration path.")
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise HipLibraryError(f"cannot load {path}: {e}") from e
    exports = _abi.EXPORTS + (_abi.MEASUREMENT_EXPORTS if measure else [])
    missing = [s for s in exports if not hasattr(lib, s)]
    if missing:
        raise HipLibraryError(f"{path} does not export {missing}; rebuild with `python -m rectipy_amd.build --force`")
    _abi.declare(lib)
    if lib.rp_abi_version() != _abi.RP_ABI_VERSION:
        raise HipLibraryError(f"ABI version mismatch: library {lib.rp_abi_version()}, python {_abi.RP_ABI_VERSION}")
    for name, struct in (("rp_sizeof_net", _abi.RpNet), ("rp_sizeof_run", _abi.RpRun), ("rp_sizeof_bwd", _abi.RpBwd),
                         ("rp_sizeof_member", _abi.RpMember), ("rp_sizeof_bwd_member", _abi.RpBwdMember),
                         ("rp_sizeof_comm", _abi.RpComm), ("rp_sizeof_edge_mem", _abi.RpEdgeMem),
                         ("rp_sizeof_batch", _abi.RpBatch), ("rp_sizeof_rls", _abi.RpRls)):
        if getattr(lib, name)() != C.sizeof(struct):
            raise HipLibraryError(f"{name}() = {getattr(lib, name)()} but ctypes mirror is {C.sizeof(struct)} bytes")
    _LIBS[measure] = lib
    return lib


def check(rc, lib=None):
    """Map a C status code to the exception the reference would raise at this seam."""
    if rc == _abi.RP_OK:
        return
    msg = (lib or load()).rp_last_error().decode("utf-8", "replace")
    if rc == _abi.RP_ERR_ARG:
        raise ValueError(msg)
    if rc == _abi.RP_ERR_SHAPE:
        raise RuntimeError(msg)  # torch raises RuntimeError on size mismatch (rectipy_tests/test_nodes.py:148-151)
    if rc == _abi.RP_ERR_ARCH:
        raise HipLibraryError(msg)
    if rc == _abi.RP_ERR_COMM:
        raise RuntimeError(f"RCCL failure: {msg}")
    raise RuntimeError(f"HIP failure: {msg}")
