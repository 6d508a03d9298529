"""When the step kernel's walk is measured (rectipy_amd/tuning.py): not by default, on request, or for a run long enough
to repay it; winners persist per (device name, shape) so that a second process installs them instead of measuring; a
failing measurement never fails the caller's run.  Host logic only -- a recording stand-in plays the library."""
import json
import types
import warnings

import pytest

from rectipy_amd import tuning


class FakeBackend:
    def __init__(self, fail=False):
        self.calls, self.fail, self.cfg = [], fail, None

    def tune_forward(self, net, device):
        self.calls.append("tune")
        if self.fail:
            raise RuntimeError("rp_tune_forward: timing failed")
        self.cfg = dict(rows_per_wave=2, chunks_in_flight=1, workgroups=683, measured=True, us_per_step=604.5)

    def forward_tuning(self, net):
        return self.cfg or dict(rows_per_wave=4, chunks_in_flight=2, workgroups=2048, measured=False, us_per_step=0.0)

    def install_forward_tuning(self, net, R, U, workgroups, us, device):
        self.calls.append(("install", R, U, workgroups))
        self.cfg = dict(rows_per_wave=R, chunks_in_flight=U, workgroups=workgroups, measured=True, us_per_step=us)

    def abi_version(self):
        return 9

    def device_name(self, device):
        return "AMD Instinct MI355X"


def population(backend, tune=None, rows=32768, n=32768):
    pop = types.SimpleNamespace(backend=backend, packed=None, device="cuda:0", rows=rows, n=n, tune=tune, _tuned=False,
                                _cache_checked=False)
    return pop, types.SimpleNamespace(ldw=n)


@pytest.fixture(autouse=True)
def clean_env(monkeypatch, tmp_path):
    monkeypatch.delenv("RECTIPY_AMD_AUTOTUNE", raising=False)
    monkeypatch.delenv("RECTIPY_AMD_NO_AUTOTUNE", raising=False)
    monkeypatch.setenv("RECTIPY_AMD_TUNE_CACHE", str(tmp_path / "cache" / "forward_tuning.json"))
    return tmp_path


def test_short_runs_keep_the_static_rule_and_long_runs_measure_once():
    be = FakeBackend()
    pop, net = population(be)
    tuning.prepare(pop, net, steps=100)          # a 100-step Network.run (61 ms of kernels) must not pay ~0.8 s
    tuning.prepare(pop, net, steps=10000)        # nor the BASELINE 10^4-step run: the static rule is the winner there
    assert be.calls == [] and not pop._tuned
    long_run = tuning.AMORTISE * tuning.PROBE_STEP_EQUIVALENTS
    tuning.prepare(pop, net, steps=long_run)     # a run that repays a few per cent
    assert be.calls == ["tune"] and pop._tuned


def test_request_and_environment_switches(monkeypatch):
    be = FakeBackend()
    pop, net = population(be, tune=True)
    tuning.prepare(pop, net, steps=1)
    assert be.calls == ["tune"]
    be = FakeBackend()
    pop, net = population(be, tune=False)
    tuning.prepare(pop, net, steps=10 ** 9)
    assert be.calls == [] and pop._tuned
    monkeypatch.setenv("RECTIPY_AMD_AUTOTUNE", "1")
    be = FakeBackend()
    pop, net = population(be, rows=16384)        # (another shape: the first one is in the cache by now)
    tuning.prepare(pop, net, steps=1)
    assert be.calls == ["tune"]
    for var, val in (("RECTIPY_AMD_AUTOTUNE", "0"), ("RECTIPY_AMD_NO_AUTOTUNE", "1")):
        monkeypatch.setenv(var, val)
        be = FakeBackend()
        pop, net = population(be, tune=None)
        tuning.prepare(pop, net, steps=10 ** 9)
        assert be.calls == [] and pop._tuned
        assert tuning.mode() == "off"


def test_winners_persist_and_a_second_process_installs_them(clean_env):
    be = FakeBackend()
    pop, net = population(be, tune=True)
    tuning.prepare(pop, net, steps=5)
    data = json.load(open(clean_env / "cache" / "forward_tuning.json"))
    (key, entry), = data["entries"].items()
    assert "MI355X" in key and "32768x32768" in key and "abi9" in key
    assert entry == {"R": 2, "U": 1, "workgroups": 683, "us": 604.5}
    be2 = FakeBackend()                          # "second process": nothing measured yet, short run
    pop2, net2 = population(be2)
    tuning.prepare(pop2, net2, steps=5)
    assert be2.calls == [("install", 2, 1, 683)] and pop2._tuned
    be3 = FakeBackend()                          # another shape: no entry, short run -> static rule, asked again later
    pop3, net3 = population(be3, rows=16384)
    tuning.prepare(pop3, net3, steps=5)
    assert be3.calls == [] and not pop3._tuned


def test_a_failing_measurement_is_a_warning_not_an_error(clean_env):
    be = FakeBackend(fail=True)
    pop, net = population(be, tune=True)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        tuning.prepare(pop, net, steps=5)
    assert be.calls == ["tune"] and pop._tuned
    assert any("static streaming rule" in str(w.message) for w in caught)
    assert not (clean_env / "cache" / "forward_tuning.json").exists()


def test_corrupt_or_unwritable_cache_is_ignored(clean_env, monkeypatch):
    path = clean_env / "cache" / "forward_tuning.json"
    path.parent.mkdir()
    path.write_text("{ not json")
    assert tuning.load_cache() == {}
    be = FakeBackend()
    pop, net = population(be, tune=True)
    tuning.prepare(pop, net, steps=5)            # overwrites the corrupt file with a valid one
    assert json.load(open(path))["version"] == 1
    monkeypatch.setenv("RECTIPY_AMD_TUNE_CACHE", "")   # disabled
    assert tuning.cache_path() is None and tuning.load_cache() == {}
    tuning.store("k", {"R": 2})                  # no-op, no exception
    monkeypatch.setenv("RECTIPY_AMD_TUNE_CACHE", "/proc/definitely/not/writable/x.json")
    tuning.store("k", {"R": 2})                  # swallowed


def test_packed_and_oracle_populations_have_nothing_to_measure():
    be = FakeBackend()
    pop, net = population(be, tune=True)
    pop.packed = object()                        # the sparse step kernel has its own walk
    tuning.prepare(pop, net, steps=5)
    assert be.calls == [] and pop._tuned
    pop = types.SimpleNamespace(backend=object(), packed=None, tune=True, _tuned=False, _cache_checked=False)
    tuning.prepare(pop, None, steps=5)           # a backend without tune_forward (the CPU oracle in tests)
    assert pop._tuned
