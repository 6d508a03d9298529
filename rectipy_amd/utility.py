"""Small host-side helpers with the behaviour of their namesakes in the reference
(rectipy/utility.py:10-56): keyword peeling and the ``<op>/<var>`` name mangling that
``Network.add_diffeq_node(op=...)`` relies on."""
from typing import Optional


def retrieve_from_dict(keys: list, data: dict) -> dict:
    """Pop `keys` out of `data` into a new dict (missing keys are skipped)."""
    return {k: data.pop(k) for k in keys if k in data}


def add_op_name(op: str, var: Optional[str], new_var_names: dict) -> Optional[str]:
    """Qualify a bare variable name with its operator and remember the mapping bare -> qualified.

    ``None`` and ``"weights"`` pass through; an already qualified ``"<op>/<var>"`` is kept and registered
    under its bare name."""
    if var is None or var == "weights":
        return var
    if "/" in var:
        bare = var.split("/")[-1]
        new_var_names[bare] = var
        return var
    new_var_names[var] = f"{op}/{var}"
    return new_var_names[var]
