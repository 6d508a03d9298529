"""When the streaming configuration of the dense step kernel is MEASURED (``rp_tune_forward``), and where measured
winners are kept between processes.

How W is walked (rows per wave, chunks in flight, resident workgroups) never changes the bits of a result, only the
speed, and the library's static rule is the measured winner at the benchmark shape (32768 x 32768: 2 rows x 1 chunk on
683 workgroups).  A measurement costs about ``PROBE_STEP_EQUIVALENTS`` steps of the shape being measured (r02 rocprof of
one bench process: ~600 ``step_kernel_tuning_probe`` launches plus their host-side waits = 0.83 s at 0.61 ms per step) and
buys 0-5 % on other shapes, so it is NOT paid by default:

* ``RECTIPY_AMD_AUTOTUNE=auto`` (default): measure only when ONE call brings at least ``AMORTISE x PROBE_STEP_EQUIVALENTS``
  steps (a run long enough that a few per cent repay the measurement);
* ``RECTIPY_AMD_AUTOTUNE=1`` / ``Population(..., tune=True)`` / ``Network(..., tune=True)``: measure at the first step;
* ``RECTIPY_AMD_AUTOTUNE=0`` / ``tune=False`` (``RECTIPY_AMD_NO_AUTOTUNE=1`` is kept as a synonym): static rule only.

Winners are persisted per (device name, rows, n, ldw) in a small JSON file (``RECTIPY_AMD_TUNE_CACHE``, default
``~/.cache/rectipy_amd/forward_tuning.json``; an empty value disables it) and installed with
``rp_install_forward_tuning``, so a second process pays nothing.  A failing measurement or an unwritable cache is never
an error of the caller's run: the static rule stays.
"""
import json
import os
import warnings
from typing import Optional

PROBE_STEP_EQUIVALENTS = 1400
AMORTISE = 32
_CACHE_VERSION = 1


def mode() -> str:
    if os.environ.get("RECTIPY_AMD_NO_AUTOTUNE", "0") == "1":
        return "off"
    val = os.environ.get("RECTIPY_AMD_AUTOTUNE", "auto").strip().lower()
    if val in ("0", "off", "false", "no"):
        return "off"
    if val in ("1", "on", "true", "yes", "always"):
        return "on"
    return "auto"


def cache_path() -> Optional[str]:
    path = os.environ.get("RECTIPY_AMD_TUNE_CACHE")
    if path is None:
        path = os.path.join(os.path.expanduser("~"), ".cache", "rectipy_amd", "forward_tuning.json")
    return path or None


def _key(device_name: str, rows: int, n: int, ldw: int, abi: int) -> str:
    return f"{device_name}|abi{abi}|{rows}x{n}|ldw{ldw}"


def load_cache() -> dict:
    path = cache_path()
    if not path:
        return {}
    try:
        with open(path) as fh:
            data = json.load(fh)
        if data.get("version") != _CACHE_VERSION or not isinstance(data.get("entries"), dict):
            return {}
        return data["entries"]
    except (OSError, ValueError, AttributeError):
        return {}


def store(key: str, entry: dict):
    path = cache_path()
    if not path:
        return
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        entries = load_cache()
        entries[key] = entry
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "w") as fh:
            json.dump({"version": _CACHE_VERSION, "entries": entries}, fh, indent=1, sort_keys=True)
        os.replace(tmp, path)  # atomic: a concurrent reader sees the old or the new file, never a torn one
    except OSError:
        pass  # an unwritable cache directory costs the next process its own measurement, nothing else


def should_measure(steps: int, requested: Optional[bool]) -> bool:
    """Policy for one call of `steps` steps of a population that has no measurement yet."""
    if requested is False:
        return False
    m = mode()
    if m == "off":
        return False
    if requested is True or m == "on":
        return True
    return steps >= AMORTISE * PROBE_STEP_EQUIVALENTS


def prepare(pop, net, steps: int) -> None:
    """Called by Population.forward before the library call while the population has no configuration beyond the static
    rule: installs a persisted winner for this shape if there is one, else measures when the policy says so."""
    backend = pop.backend
    if not hasattr(backend, "tune_forward") or pop.packed is not None:
        pop._tuned = True
        return
    requested = pop.tune
    if requested is False or mode() == "off":
        pop._tuned = True
        return
    if pop.rows < 8192 or 4 * pop.rows * pop.n <= (320 << 20):
        pop._tuned = True  # does not stream from HBM: rp_tune_forward has nothing to choose (its own rule), no cache entry either
        return
    key = _key(backend.device_name(pop.device), pop.rows, pop.n, int(net.ldw), backend.abi_version())
    if not pop._cache_checked:
        pop._cache_checked = True
        hit = load_cache().get(key)
        if hit is not None:
            try:
                backend.install_forward_tuning(net, int(hit["R"]), int(hit["U"]), int(hit["workgroups"]), float(hit["us"]),
                                               pop.device)
                pop._tuned = True
                return
            except (KeyError, TypeError, ValueError, RuntimeError):  # (library errors are RuntimeError / ValueError)
                pass  # a stale or foreign entry: ignore it
    if not should_measure(steps, requested):
        return  # asked again at the next call (a later, longer run may repay the measurement)
    pop._tuned = True
    try:
        backend.tune_forward(net, pop.device)
        got = backend.forward_tuning(net)
    except (RuntimeError, ValueError) as err:  # the measurement is an optimisation: never the caller's error
        warnings.warn(f"rp_tune_forward failed ({err}); keeping the static streaming rule")
        return
    if got["measured"]:
        store(key, {"R": got["rows_per_wave"], "U": got["chunks_in_flight"], "workgroups": got["workgroups"],
                    "us": got["us_per_step"]})
