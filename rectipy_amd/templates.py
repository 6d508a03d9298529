"""Closed table of neuron-model templates for the HIP integration path.

The reference hands a YAML template to the third-party PyRates code generator
(rectipy/nodes.py:232-262), which returns a torch function for N vectorised neurons.
PyRates is not a dependency of this engine: the hot path runs a fixed set of vector
fields compiled into the gfx950 kernels, and this table maps the reference's template
paths and ``<operator>/<variable>`` names onto them.  Equations, variable names and
default values are those of the reference's YAML files:

  neuron_model_templates/rate_neurons/leaky_integrator.yaml:8-36  (li_op, sigmoid_op, tanh_op)
  neuron_model_templates/spiking_neurons/qif.yaml:8-35            (qif_op, qif_sfa_op)
  neuron_model_templates/spiking_neurons/lif.yaml:10-23           (lif_op)
  neuron_model_templates/spiking_neurons/ik.yaml:8-68             (ik_op, iku_op, ik_biexp_op)

Anything else (user YAML, PyRates template objects) raises -- there is
no silent fallback to a generic interpreter.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Tuple


@dataclass(frozen=True)
class Template:
    name: str                       # e.g. "tanh"
    model: str                      # engine model id (rectipy_amd.engine.MODEL_IDS)
    state: Tuple[Tuple[str, float], ...]     # state variables in state-vector order, with defaults
    params: Tuple[Tuple[str, float], ...]    # per-neuron constants with YAML defaults
    input_var: str                  # extrinsic input the kernels add (I_ext)
    target_var: str                 # variable the recurrent weights project to
    source_var: str                 # variable the recurrent weights read
    derived: Tuple[str, ...] = ()   # non-state outputs computed from the state (r = act(v))
    spike_var: str = ""             # spiking models: input that receives spike/dt
    reset_var: str = ""             # spiking models: state variable that is reset
    alt_input_var: str = ""         # LIF: extrinsic input into the synaptic equation (s_ext)
    engine_names: Dict[str, str] = field(default_factory=dict)  # "<op>/<var>" -> engine parameter name


_LI_PARAMS = (("li_op/tau", 10.0), ("li_op/k", 1.0), ("li_op/eta", 0.0))

_FILES: Dict[str, Dict[str, Template]] = {
    "leaky_integrator": {
        "tanh": Template(
            name="tanh", model="li_tanh", state=(("li_op/v", 0.0),), params=_LI_PARAMS,
            input_var="li_op/I_ext", target_var="li_op/r_in", source_var="tanh_op/r", derived=("tanh_op/r",),
            engine_names={"li_op/tau": "tau", "li_op/k": "k", "li_op/eta": "eta"}),
        "sigmoid": Template(
            name="sigmoid", model="li_sigmoid", state=(("li_op/v", 0.0),),
            params=_LI_PARAMS + (("sigmoid_op/r_max", 1.0), ("sigmoid_op/s", 1.0), ("sigmoid_op/v0", 0.0)),
            input_var="li_op/I_ext", target_var="li_op/r_in", source_var="sigmoid_op/r", derived=("sigmoid_op/r",),
            engine_names={"li_op/tau": "tau", "li_op/k": "k", "li_op/eta": "eta", "sigmoid_op/r_max": "r_max",
                          "sigmoid_op/s": "s", "sigmoid_op/v0": "v0"}),
    },
    "qif": {
        "qif": Template(
            name="qif", model="qif", state=(("qif_op/v", -2.0), ("qif_op/s", 0.0)),
            params=(("qif_op/tau", 1.0), ("qif_op/k", 1.0), ("qif_op/tau_s", 1.0), ("qif_op/eta", -5.0)),
            input_var="qif_op/I_ext", target_var="qif_op/s_in", source_var="qif_op/s",
            spike_var="qif_op/spike", reset_var="qif_op/v",
            engine_names={"qif_op/tau": "tau", "qif_op/k": "k", "qif_op/tau_s": "tau_s", "qif_op/eta": "eta"}),
    },
}

_FILES["qif"]["qif_sfa"] = Template(
    # qif.yaml:25-35: qif_op with eta -> eta - x and x' = -x/tau_x + alpha*spike; state order v, s, x
    name="qif_sfa", model="qif_sfa", state=(("qif_sfa_op/v", -2.0), ("qif_sfa_op/s", 0.0), ("qif_sfa_op/x", 0.0)),
    params=(("qif_sfa_op/tau", 1.0), ("qif_sfa_op/k", 1.0), ("qif_sfa_op/tau_s", 1.0), ("qif_sfa_op/eta", -5.0),
            ("qif_sfa_op/alpha", 1.0), ("qif_sfa_op/tau_x", 10.0)),
    input_var="qif_sfa_op/I_ext", target_var="qif_sfa_op/s_in", source_var="qif_sfa_op/s",
    spike_var="qif_sfa_op/spike", reset_var="qif_sfa_op/v",
    engine_names={"qif_sfa_op/tau": "tau", "qif_sfa_op/k": "k", "qif_sfa_op/tau_s": "tau_s", "qif_sfa_op/eta": "eta",
                  "qif_sfa_op/alpha": "alpha", "qif_sfa_op/tau_x": "tau_x"})
_FILES["lif"] = {"lif": Template(
    # lif.yaml:10-23 (li_op with k*r_in -> k*s_in, plus s' = -s/tau_s + spike + s_ext); state order v, s
    name="lif", model="lif", state=(("lif_op/v", 0.0), ("lif_op/s", 0.0)),
    params=(("lif_op/tau", 10.0), ("lif_op/k", 1.0), ("lif_op/eta", 0.0), ("lif_op/tau_s", 0.5)),
    input_var="lif_op/I_ext", target_var="lif_op/s_in", source_var="lif_op/s", spike_var="lif_op/spike",
    reset_var="lif_op/v", alt_input_var="lif_op/s_ext",
    engine_names={"lif_op/tau": "tau", "lif_op/k": "k", "lif_op/eta": "eta", "lif_op/tau_s": "tau_s"})}
_FILES["ik"] = {"ik": Template(
    # ik.yaml:8-31; state order v, u, s
    name="ik", model="ik", state=(("ik_op/v", -60.0), ("ik_op/u", 0.0), ("ik_op/s", 0.0)),
    params=(("ik_op/C", 100.0), ("ik_op/k", 0.7), ("ik_op/v_r", -60.0), ("ik_op/v_theta", -40.0), ("ik_op/eta", 0.0),
            ("ik_op/g", 1.0), ("ik_op/E_r", 0.0), ("ik_op/b", -2.0), ("ik_op/tau_u", 33.33), ("ik_op/kappa", 10.0),
            ("ik_op/tau_s", 6.0)),
    input_var="ik_op/I_ext", target_var="ik_op/s_in", source_var="ik_op/s", spike_var="ik_op/spike",
    reset_var="ik_op/v",
    engine_names={"ik_op/C": "C", "ik_op/k": "k", "ik_op/v_r": "v_r", "ik_op/v_theta": "v_theta", "ik_op/eta": "eta",
                  "ik_op/g": "g", "ik_op/E_r": "E_r", "ik_op/b": "b", "ik_op/tau_u": "tau_u", "ik_op/kappa": "kappa",
                  "ik_op/tau_s": "tau_s"})}

_IK_PARAMS = (("C", 100.0), ("k", 0.7), ("v_r", -60.0), ("v_theta", -40.0), ("eta", 0.0), ("g", 1.0), ("E_r", 0.0),
              ("b", -2.0), ("tau_u", 33.33), ("kappa", 10.0))
_FILES["ik"]["iku"] = Template(
    # ik.yaml:34-40: ik_op with b*(mean(v)-v_r) and kappa*mean(spike) in the recovery equation; state order v, u, s
    name="iku", model="iku", state=(("iku_op/v", -60.0), ("iku_op/u", 0.0), ("iku_op/s", 0.0)),
    params=tuple((f"iku_op/{q}", d) for q, d in _IK_PARAMS) + (("iku_op/tau_s", 6.0),),
    input_var="iku_op/I_ext", target_var="iku_op/s_in", source_var="iku_op/s", spike_var="iku_op/spike",
    reset_var="iku_op/v",
    engine_names={**{f"iku_op/{q}": q for q, _ in _IK_PARAMS}, "iku_op/tau_s": "tau_s"})
_FILES["ik"]["ik_biexp"] = Template(
    # ik.yaml:42-68: global recovery variable + bi-exponential synapse; state order v, u, s, x
    name="ik_biexp", model="ik_biexp",
    state=(("ik_biexp_op/v", -60.0), ("ik_biexp_op/u", 0.0), ("ik_biexp_op/s", 0.0), ("ik_biexp_op/x", 0.0)),
    params=tuple((f"ik_biexp_op/{q}", d) for q, d in _IK_PARAMS) + (("ik_biexp_op/tau_r", 2.0), ("ik_biexp_op/tau_d", 6.0)),
    input_var="ik_biexp_op/I_ext", target_var="ik_biexp_op/s_in", source_var="ik_biexp_op/s",
    spike_var="ik_biexp_op/spike", reset_var="ik_biexp_op/v",
    engine_names={**{f"ik_biexp_op/{q}": q for q, _ in _IK_PARAMS}, "ik_biexp_op/tau_r": "tau_r",
                  "ik_biexp_op/tau_d": "tau_d"})

# operator templates of the reference's YAML files that no node template uses
_KNOWN_UNSUPPORTED = {"qif": ("qif_reset",)}


def resolve(path) -> Template:
    """Map a template path such as
    ``neuron_model_templates.rate_neurons.leaky_integrator.tanh`` to a :class:`Template`.

    Error behaviour follows what the reference surfaces from PyRates
    (rectipy_tests/test_network.py:94-101): unknown file -> FileNotFoundError, unknown
    template inside a known file -> AttributeError.
    """
    if not isinstance(path, str):
        raise NotImplementedError(
            "rectipy_amd accepts template paths of its built-in models only; PyRates NodeTemplate/"
            "CircuitTemplate objects need the PyRates code generator, which this engine does not use.")
    parts = path.replace("/", ".").split(".")
    if len(parts) < 2:
        raise FileNotFoundError(f"'{path}' is not of the form <package>.<file>.<template>")
    fname, tname = parts[-2], parts[-1]
    if fname in _FILES and tname in _FILES[fname]:
        return _FILES[fname][tname]
    if fname in _KNOWN_UNSUPPORTED and tname in _KNOWN_UNSUPPORTED[fname]:
        raise NotImplementedError(f"template '{fname}.{tname}' has no gfx950 kernel yet (see DESIGN.md, 'next')")
    if fname in _FILES or fname in _KNOWN_UNSUPPORTED:
        raise AttributeError(f"template file '{fname}' has no template named '{tname}'")
    raise FileNotFoundError(f"no built-in template file '{fname}' (from '{path}')")


def available() -> List[str]:
    return [f"{f}.{t}" for f, d in _FILES.items() for t in d]
