"""Node classes of the MI355X engine, mirroring the node protocol of the reference
(rectipy/nodes.py): ``forward(x)``, ``y``, ``n_in``, ``n_out``, ``parameters()``,
``detach()``, ``reset()``, ``set_param()``, ``__getitem__``, ``parameter_names``,
``variable_names``.

``RateNet`` / ``SpikeResetNet`` here do not wrap a Python vector-field callable: they own a
:class:`rectipy_amd.engine.Population` whose Euler step, nonlinearity, spike/reset and
adjoint run in hand-written gfx950 kernels.  A whole window of T steps is one library call
(``run_window``); ``forward(x)`` is the T = 1 case kept for drop-in per-step loops.
State and arithmetic are float32 (the reference defaults the node state to float64,
rectipy/nodes.py:59,120 -- documented difference).
"""
import warnings
from typing import Iterator, List, Optional, Union

import numpy as np
import torch

from . import _abi
from .engine import Population
from .templates import Template, resolve

_warned_f64 = False


def _warn_f64(dtype):
    global _warned_f64
    if dtype is torch.float64 and not _warned_f64:
        warnings.warn("rectipy_amd integrates in float32 on the GPU; a float64 node state was requested "
                      "(the reference's default) and is held as float32.", stacklevel=3)
        _warned_f64 = True


class InstantNode:
    """Memory-less activation node (reference rectipy/nodes.py:14-51)."""

    _funcs = {
        "tanh": lambda kw: torch.nn.Tanh(),
        "sigmoid": lambda kw: torch.nn.Sigmoid(),
        "identity": lambda kw: torch.nn.Identity(),
        "softmax": lambda kw: torch.nn.Softmax(dim=kw.get("dim", 0)),
        "softmin": lambda kw: torch.nn.Softmin(dim=kw.get("dim", 0)),
        "log_softmax": lambda kw: torch.nn.LogSoftmax(dim=kw.get("dim", 0)),
    }
    elementwise = ("tanh", "sigmoid", "identity")
    # every activation can be applied to a whole [T, n] window: the element-wise ones as they are, the normalising ones
    # (built with dim=0 for the reference's one-step vectors) along the neuron axis of each row
    batched = ("tanh", "sigmoid", "identity", "softmax", "softmin", "log_softmax")

    def __init__(self, n: int, func: str, **kwargs):
        if func not in self._funcs:
            raise ValueError(f"Invalid keyword argument `func`: {func} is not a valid option "
                             f"(choose from {sorted(self._funcs)}).")
        self.n_in = n
        self.n_out = n
        self.name = func
        self.func = self._funcs[func](kwargs)
        self._rowwise = {"softmax": torch.softmax, "softmin": torch.nn.functional.softmin,
                         "log_softmax": torch.log_softmax}.get(func) if kwargs.get("dim", 0) == 0 else None

    def __getitem__(self, item):
        return None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._rowwise is not None and x.dim() == 2:  # a window of steps: normalise every step's vector
            return self._rowwise(x, dim=1)
        return self.func(x)

    def parameters(self, **kwargs):
        return self.func.parameters(**kwargs)


class _RateWindow(torch.autograd.Function):
    """T Euler steps of a leaky-integrator population as ONE autograd node.

    forward  = rp_forward with the pre-update v_t, r_t (and W r_t) rows captured,
    backward = rp_rate_backward (adjoint sweep) + rp_outer_accum (MFMA dW GEMM).
    Replaces autograd over the Python-unrolled loop (reference rectipy/network.py:1028-1040).
    """

    ACT_PARAMS = ("r_max", "s", "v0")  # the sigmoid's own parameters (leaky_integrator.yaml:22), extra autograd inputs

    @staticmethod
    def forward(ctx, node, steps, x, v_in, W, tau, k, eta, *act):
        pop: Population = node._pop
        N, dev = pop.rows, pop.device
        need_acc = k.requires_grad
        traj_v = torch.empty(steps, N, dtype=torch.float32, device=dev)
        traj_r = torch.empty(steps, N, dtype=torch.float32, device=dev)
        traj_acc = torch.empty(steps, N, dtype=torch.float32, device=dev) if need_acc else None
        pop.forward(steps, x=x, save_v=traj_v, save_src=traj_r, save_acc=traj_acc)
        ctx.node = node
        ctx.steps = steps
        ctx.x_width = None if x is None else x.shape[1]
        ctx.saved = (traj_v, traj_r, traj_acc)
        ctx.W = W   # (the leaf itself: a plain-SGD fit applies its update in the gradient GEMM, see backward)
        node._last_window = (traj_v, traj_r)
        out = traj_r if node._out_code == _abi.RP_OUT_SOURCE else traj_v
        return out.clone(), pop.v.clone()

    @staticmethod
    def backward(ctx, g_traj, g_vend):
        node = ctx.node
        pop: Population = node._pop
        N, dev, T = pop.rows, pop.device, ctx.steps
        traj_v, traj_r, traj_acc = ctx.saved
        f32 = dict(dtype=torch.float32, device=dev)
        lam = torch.zeros(N, **f32) if g_vend is None else g_vend.detach().to(**f32).clone().contiguous()
        g_out = None if g_traj is None else g_traj.detach().to(**f32).contiguous()
        need = ctx.needs_input_grad  # (node, steps, x, v_in, W, tau, k, eta, *act)
        u_traj = torch.empty(T, N, **f32)
        g_tau = torch.zeros(N, **f32) if need[5] else None
        g_k = torch.zeros(N, **f32) if need[6] else None
        g_eta = torch.zeros(N, **f32) if need[7] else None
        g_x = torch.empty(T, N, **f32) if need[2] else None
        g_act = {name: torch.zeros(N, **f32) for i, name in enumerate(_RateWindow.ACT_PARAMS[:len(need) - 8])
                 if need[8 + i]}
        keep = pop.backward(traj_v, traj_r, g_out, node._out_code, lam, u_traj, traj_acc=traj_acc,
                            grad_tau=g_tau, grad_k=g_k, grad_eta=g_eta, grad_x=g_x, grad_params=g_act)
        g_W = None
        if need[4]:
            fused = getattr(node, "_fused_sgd", None)   # id(leaf) -> learning rate (a plain-SGD fit_bptt: Network._fused_sgd_plan)
            lr_total = fused.get(id(ctx.W)) if fused else None
            if lr_total is not None and ctx.W.is_contiguous() and ctx.W.dtype == torch.float32 and pop.packed is None \
                    and tuple(ctx.W.shape) == (N, N):
                # W <- W - lr (dt u^T r) in the GEMM's epilogue: one read and one write of W instead of a stored gradient matrix
                # and the optimiser's three passes over it; the gradient stays None and the optimiser skips W
                pop.outer_accum(-lr_total * pop.dt, u_traj, traj_r, ctx.W.detach())
                torch.autograd.graph.increment_version(ctx.W)   # (an in-place write torch did not see)
            else:
                g_W = pop.outer_product(pop.dt, u_traj, traj_r)
        if g_x is not None and ctx.x_width == 1:
            g_x = g_x.sum(dim=1, keepdim=True)
        del keep
        return (None, None, g_x, (lam if need[3] else None), g_W, g_tau, g_k, g_eta) + \
            tuple(g_act.get(name) for name in _RateWindow.ACT_PARAMS[:len(need) - 8])


class RateNet:
    """Rate-neuron population integrated on the GPU (reference ``RateNet``, rectipy/nodes.py:54-290)."""

    state_vars = ["y"]
    _spiking = False

    # ---- construction ----------------------------------------------------------------
    def __init__(self, template: Template, weights, node_vars: dict, input_var: str, output_var: str = None,
                 dt: float = 1e-3, dtype: torch.dtype = torch.float64, train_params: list = None,
                 device: str = "cuda:0", backend=None, **kwargs):
        """Two calling conventions:

        * ``RateNet(template, weights, node_vars, input_var, output_var, ...)`` -- what ``from_pyrates`` /
          ``Network.add_diffeq_node`` use (``template``: a :class:`rectipy_amd.templates.Template`);
        * the reference's own constructor ``RateNet(rnn_func, rnn_args, var_map, param_map, dt=, dtype=, train_params=,
          device=)`` (rectipy/nodes.py:58-90; rectipy_tests/test_nodes.py:40-79): ``rnn_func`` is recognised as one of
          the compiled vector fields (:mod:`rectipy_amd.field_match`) or names one (template path / ``"li_tanh"``);
          ``rnn_args[0]`` is the initial state, ``param_map["in"]`` the input slot, ``var_map["out"]`` the output slice.
        """
        ref_style = None
        if not isinstance(template, Template):
            ref_style = self._from_reference_args(template, weights, node_vars, input_var, train_params)
            template, weights, node_vars, input_var, output_var, train_params = ref_style[:6]
        _warn_f64(dtype)
        self._template = template
        self.dt = dt
        self.device = device
        state_names = [s for s, _ in template.state]

        # initial state and parameters: YAML defaults overridden by node_vars ("all/<op>/<var>": value)
        values = dict(template.params)
        values.update(dict(template.state))
        for key, val in (node_vars or {}).items():
            parts = key.split("/")
            if len(parts) == 3 and parts[0] != "all":
                raise NotImplementedError(f"node_vars key '{key}': only 'all/<op>/<var>' (optionally with per-neuron "
                                          "vectors as values) is supported")
            name = "/".join(parts[-2:])
            if name not in values:
                raise KeyError(f"Variable {name} was not found on template '{template.name}'.")
            values[name] = val
        if weights is None:
            raise ValueError("the dense-recurrence engine needs the `weights` matrix of the population")
        eng_params = {template.engine_names[p]: values[p] for p, _ in template.params}
        kw = {}
        # which state variable lives where in the engine: v (reset / first state), s (the source), aux (x or u)
        self._aux_name = self._aux2_name = None
        if self._spiking:
            aux = [s for s in state_names if s not in (template.reset_var, template.source_var)]
            self._aux_name = aux[0] if aux else None
            self._aux2_name = aux[1] if len(aux) > 1 else None
            kw = dict(v0=values[template.reset_var], s0=values[template.source_var],
                      theta=kwargs.pop("spike_threshold", 1e2), v_reset=kwargs.pop("spike_reset", -1e2))
            if self._aux_name:
                kw["aux0"] = values[self._aux_name]
            if self._aux2_name:
                kw["aux2_0"] = values[self._aux2_name]
        else:
            kw = dict(v0=values[state_names[0]])
        self._sparse_mode = kwargs.pop("sparse_weights", "auto")  # True | False | "auto": packed storage of a fixed W
        shard = kwargs.pop("shard", None)  # (row0, n): this node holds rows [row0, row0+rows) of an n x n matrix
        if shard is not None:
            kw.update(row0=int(shard[0]), n=int(shard[1]))
        if "tune" in kwargs:  # True: measure the step kernel's walk for this shape at the first step (tuning.py)
            kw["tune"] = kwargs.pop("tune")
        self._pop = Population(template.model, weights, eng_params, dt, device, backend=backend, **kw)
        N = self._pop.rows
        self.n_in = N
        self.n_out = N

        # name maps (the reference builds them from PyRates' key lists, nodes.py:150-159)
        self._param_map = {"weights": "weights"}
        for p, _ in template.params:
            self._param_map[p] = template.engine_names[p]
        for key, var in (kwargs.pop("param_mapping", None) or {}).items():
            if var in self._param_map:
                self._param_map[key] = self._param_map[var]
        self._var_map = {name: (i * N, (i + 1) * N) for i, name in enumerate(state_names)}
        self._x_target = _abi.RP_XT_MEMBRANE
        if template.alt_input_var and input_var == template.alt_input_var:
            self._x_target = _abi.RP_XT_SYNAPSE  # lif.yaml:16: s_ext enters the synaptic equation
        elif input_var != template.input_var:
            if input_var in self._param_map or input_var in self._var_map or input_var == template.target_var:
                raise NotImplementedError(f"input_var='{input_var}': the kernels add extrinsic input to "
                                          f"'{template.input_var}' only")
            raise KeyError(f"Variable {input_var} was not found on template '{template.name}'.")
        if output_var in self._var_map:
            if output_var in (self._aux_name, self._aux2_name):
                raise NotImplementedError(f"output_var='{output_var}': the kernels emit v or the synaptic variable")
            first = template.reset_var if self._spiking else state_names[0]
            self._out_code = _abi.RP_OUT_STATE0 if output_var == first else _abi.RP_OUT_SOURCE
        elif output_var in template.derived:
            self._out_code = _abi.RP_OUT_SOURCE
        else:
            raise KeyError(f"Variable {output_var} was not found on template '{template.name}'.")
        self._out_name = output_var
        self._var_map["out"] = self._var_map.get(output_var, (0, N))
        for key, var in (kwargs.pop("var_mapping", None) or {}).items():
            if var in self._var_map:
                self._var_map[key] = self._var_map[var]
            elif var not in template.derived:
                raise KeyError(f"Variable {var} was not found on template '{template.name}'.")

        # trainable parameters
        self._train_names: List[str] = []
        for p in (train_params or []):
            if p not in self._param_map:
                raise KeyError(f"Parameter {p} was not found on the node.")
            self._train_names.append(self._param_map[p])
        self._check_trainable()
        for name in self._train_names:
            self._tensor(name).requires_grad_(True)
        self._y_grad: Optional[torch.Tensor] = None  # autograd handle of the state (same values as pop.v)
        self._y_handed = None  # (handle, its version when it was handed out): detects in-place writes by the holder
        self._v_exposed = False  # pop.v was handed out through `y`: the next step continues on a copy
        self._y_requires_grad = len(self._train_names) >= 1  # nodes.py:89-90
        self._out_slice = None
        if ref_style is not None:
            self._apply_reference_maps(*ref_style[6:])
        self._maybe_pack_weights()

    # ---- the reference's constructor convention (seam 2) ----------------------------------------------------------
    @staticmethod
    def _from_reference_args(rnn_func, rnn_args, var_map, param_map, train_params):
        """(template, weights, node_vars, input_var, output_var, train_params, var_map, roles, n_in) for
        ``RateNet(rnn_func, rnn_args, var_map, param_map, ...)``."""
        from . import field_match
        if not isinstance(rnn_args, (tuple, list)) or not isinstance(var_map, dict) or not isinstance(param_map, dict):
            raise TypeError("expected RateNet(rnn_func, rnn_args: tuple, var_map: dict, param_map: dict, ...)")
        tpl = resolve("neuron_model_templates.rate_neurons.leaky_integrator.tanh")
        y0 = rnn_args[0]
        n = int(y0.shape[0])
        if callable(rnn_func):
            W, vals, roles = field_match.recognise_rate_field(rnn_func, tuple(rnn_args), param_map)
        else:
            named = resolve(rnn_func) if isinstance(rnn_func, str) and "." in rnn_func else None
            if (named is not None and named.model != "li_tanh") or (named is None and rnn_func != "li_tanh"):
                raise NotImplementedError("the reference-style constructor covers the tanh leaky-integrator field; use "
                                          "from_pyrates for the other templates")
            args = list(rnn_args[1:])
            W = args[param_map["weights"]]
            roles, vals = {}, {q: d for (_, d), q in zip(tpl.params, ("tau", "k", "eta"))}
            for name, idx in param_map.items():
                short = name.split("/")[-1]
                if short in ("tau", "k", "eta"):
                    roles[name], vals[short] = short, args[idx]
        node_vars = {f"all/li_op/{q}": (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v)
                     for q, v in vals.items()}
        node_vars["all/li_op/v"] = y0.detach().cpu().numpy() if isinstance(y0, torch.Tensor) else np.asarray(y0)
        train = []
        for p in (train_params or []):
            if p == "weights":
                train.append("weights")
            elif p in roles:
                train.append(f"li_op/{roles[p]}")
            else:
                raise KeyError(f"Parameter {p} was not found on the node.")
        in_arg = rnn_args[1 + int(param_map["in"])]
        n_in = int(in_arg.shape[0]) if hasattr(in_arg, "shape") and len(in_arg.shape) else 1
        weights = W.detach().cpu().numpy() if isinstance(W, torch.Tensor) else np.asarray(W)
        return tpl, weights, node_vars, tpl.input_var, "li_op/v", train, dict(var_map), roles, n_in

    def _apply_reference_maps(self, var_map: dict, roles: dict, n_in: int):
        """Names and slices as the caller's ``var_map`` / ``param_map`` define them (nodes.py:64-69, 92-99)."""
        N = self._pop.rows
        self._param_map = {"weights": "weights", **roles}
        self._var_map = {k: (int(v[0]), int(v[-1])) if isinstance(v, (tuple, list)) else (int(v), int(v) + 1)
                         for k, v in var_map.items()}
        if "out" not in self._var_map:
            raise KeyError("var_map needs the output slice ('out')")
        lo, hi = self._var_map["out"]
        if not (0 <= lo < hi <= N):
            raise RuntimeError(f"output slice {(lo, hi)} outside the state vector of length {N}")
        self.n_out, self.n_in = hi - lo, n_in
        if (lo, hi) != (0, N):
            self._out_slice = (lo, hi)
            self._fusable = False  # Network: step by step (the fused windows emit all N outputs)

    def _maybe_pack_weights(self):
        """A coupling matrix that is not trained and mostly zeros (`utility.random_connectivity`, p = 0.1 in the
        reference's examples) is streamed in its packed form: same bits out, a fraction of the bytes in (sparse.py).
        "auto" packs when the packed stream is measurably faster (scripts/bench_sparse.py, bench_sparse_large.py): when
        it is at most 0.7x the dense bytes (the source vector is gathered from LDS at every n: whole below n = 32768,
        window by window above)."""
        pop = self._pop
        mode = self._sparse_mode
        if mode is False or "weights" in self._train_names or pop.n % 4 != 0 or pop.device.type != "cuda":
            if mode is True and pop.device.type == "cuda":
                raise ValueError("sparse_weights=True needs an untrained matrix with n % 4 == 0")
            return
        if mode == "auto" and pop.n < 4096:
            return  # launch-bound sizes: the dense step is as fast (n = 2048: 4.8 vs 5.1 us)
        from . import sparse
        if mode == "auto" and sparse.density(pop.W) > 0.5:
            return
        packed = pop.pack_weights()
        limit = 0.7
        if mode == "auto" and packed.stored_bytes > limit * 4.0 * pop.rows * pop.n:
            pop.pack_weights(False)

    def _check_trainable(self):
        ok = {"weights", "tau", "k", "eta"} | (set(_RateWindow.ACT_PARAMS) if self._pop.model == "li_sigmoid" else set())
        bad = [n for n in self._train_names if n not in ok]
        if bad:
            raise NotImplementedError(f"no adjoint kernel for parameters {bad}; trainable: {sorted(ok)}")

    @classmethod
    def from_pyrates(cls, node, input_var: str, output_var: str, weights: np.ndarray = None,
                     source_var: str = None, target_var: str = None, train_params: list = None, **kwargs):
        """Build a population from a template path -- the signature of the reference's
        ``RateNet.from_pyrates`` (rectipy/nodes.py:112-164).  ``node`` must name one of the built-in
        templates (:mod:`rectipy_amd.templates`); PyRates itself is not used."""
        template = resolve(node)
        dt = kwargs.pop("dt", 1e-3)
        dtype = kwargs.pop("dtype", torch.float64)
        device = kwargs.pop("device", "cuda:0")
        for ignored in ("clear", "verbose", "file_name", "float_precision"):
            kwargs.pop(ignored, None)
        for unsupported in ("edge_attr", "edge_template"):
            if kwargs.pop(unsupported, None) is not None:
                raise NotImplementedError(f"'{unsupported}' (edge operators on recurrent connections) is not supported")
        if weights is None:
            n = kwargs.pop("N", None)
            if n is None:
                raise ValueError("either `weights` or `N` must be given")
            weights = np.zeros((n, n), dtype=np.float32)
        else:
            kwargs.pop("N", None)
            if source_var is None or target_var is None:
                raise ValueError("If synaptic weights are passed (`weights`), please provide the names of the source "
                                 "and target variable that should be connected via `weights`.")
            known = {s for s, _ in template.state} | set(template.derived) | {p for p, _ in template.params} | \
                {template.input_var, template.target_var}
            for v in (source_var, target_var):
                if v not in known:
                    raise KeyError(f"Variable {v} was not found on template '{template.name}'.")
            if source_var != template.source_var or target_var != template.target_var:
                raise NotImplementedError(
                    f"recurrent coupling {source_var} -> {target_var} has no kernel; the '{template.name}' kernel "
                    f"couples {template.source_var} -> {template.target_var}")
        node_vars = kwargs.pop("node_vars", None)
        return cls(template, weights, node_vars, input_var, output_var, dt=dt, dtype=dtype,
                   train_params=train_params, device=device, **kwargs)

    # ---- name access -------------------------------------------------------------------
    def _tensor(self, engine_name: str) -> torch.Tensor:
        return self._pop.W if engine_name == "weights" else self._pop.params[engine_name]

    def __getitem__(self, item):
        if item in self._param_map:
            return self._tensor(self._param_map[item])
        if item == "out" and not self._spiking and self._out_code == _abi.RP_OUT_SOURCE:
            r = self._pop.source_local.clone()   # output variable r = act(v) of the current state
            v = self._pop.v if self._y_grad is None else self._y_grad
            par = self._pop.params
            sigmoid = self._pop.model == "li_sigmoid"
            if torch.is_grad_enabled() and (v.requires_grad or (sigmoid and any(par[q].requires_grad for q in _RateWindow.ACT_PARAMS))):
                # a feedback edge reads this inside a recorded graph: the kernel's value, the gradient of act(v)
                g = par["r_max"] / (1.0 + torch.exp(par["s"] * (par["v0"] - v))) if sigmoid else torch.tanh(v)
                r = g + (r - g).detach()
            return r
        if item in self._template.derived:
            # r = act(v) lives in a ping-pong buffer the kernels overwrite two steps later; the reference computes a
            # fresh tensor per step, so hand out a copy that a per-step recorder may keep (observer.py:100-101)
            return self._pop.source_local.clone()
        lo, hi = self._var_map[item]  # KeyError for unknown names, as the reference (test_nodes.py:78-79)
        return self.y[lo:hi]

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    @property
    def parameter_names(self) -> list:
        return list(self._param_map.keys())

    @property
    def variable_names(self) -> list:
        return list(self._var_map.keys())

    @property
    def train_params(self) -> list:
        return [self._tensor(n) for n in self._train_names]

    def parameters(self, recurse: bool = True) -> Iterator:
        for p in self.train_params:
            yield p

    def set_param(self, param: str, val: Union[torch.Tensor, float]):
        if param not in self._param_map:
            raise KeyError(f"Parameter {param} was not found on the node.")
        name = self._param_map[param]
        old = self._tensor(name)
        with torch.no_grad():
            if name == "weights":
                new = torch.as_tensor(val, dtype=torch.float32, device=old.device)
                if tuple(new.shape) != tuple(old.shape):
                    raise RuntimeError(f"weights shape {tuple(new.shape)} != {tuple(old.shape)}")
                old.copy_(new)
            else:
                old.copy_(self._pop._vec(val))

    # ---- state --------------------------------------------------------------------------
    @property
    def y(self) -> torch.Tensor:
        """State vector: the live device state (the reference also hands out the live tensor, network.py:96), or the
        autograd handle of the current window when one is being recorded.

        The reference REBINDS ``self.y`` to a new tensor at every step (nodes.py:169), so a tensor obtained from
        ``y`` / ``node[var]`` keeps its values afterwards -- its Observer stores such references per step
        (observer.py:100-101).  The kernels update the state in place; to keep the reference's behaviour the node
        remembers that the live tensor was handed out and, at the next step, continues on a fresh copy
        (:meth:`_before_step`): the handed-out tensor then holds the state it showed when it was read, and writes made
        into it before the next step (``net.state[...][:] = ...``) are part of the state, as in the reference.
        Fused windows never read ``y`` and stay zero-copy.

        While a gradient is being tracked the tensor handed out is the autograd handle (same values as the device
        state, carrying the history of the window that produced it).  Its version counter is remembered: an in-place
        write by the holder (``net.state["rnn"][:] = v`` -- allowed by torch once the handle is a non-leaf, or under
        ``no_grad``, exactly as for the reference's ``y``) is copied into the device state before the next step, and the
        next window continues from the written handle, so autograd sees the write the way it does in the reference
        (network.py:88-98, nodes.py:166-170)."""
        if self._y_grad is None and self._y_requires_grad:
            self._y_grad = self._pop.v.detach().clone().requires_grad_(True)
        if self._y_grad is None:
            self._v_exposed = True
            return self._pop.v
        if self._y_handed is None or self._y_handed[0] is not self._y_grad:
            # the version at the FIRST hand-out of this handle (the engine never writes into it, so that is its version
            # since it was bound): a later read of `y` must not forget a write made after an earlier one
            self._y_handed = (self._y_grad, self._y_grad._version)
        return self._y_grad

    def _before_step(self) -> bool:
        """Bring the device state up to date with what a holder of :attr:`y` did to it; True if the state changed
        hands (the source vector r = act(v) was recomputed).  (i) Copy-on-step for the live tensor; (ii) a write into
        the autograd handle of the state is copied into the device state."""
        pop = self._pop
        synced = False
        if self._v_exposed:
            pop.v = pop.v.clone()        # the old tensor stays with whoever holds it
            pop.refresh()                # the holder may have written into it: r = act(v) again (bit-identical otherwise)
            self._v_exposed = False
            synced = True
        handed, self._y_handed = self._y_handed, None
        if handed is not None and handed[0] is self._y_grad and handed[0]._version != handed[1]:
            pop.v = handed[0].detach().to(dtype=torch.float32).clone().contiguous()
            pop.refresh()
            synced = True
        return synced

    def set_state(self, var: str, val):
        """Write a state variable (``Network.set_var`` on a state name): goes to the device state itself, also when
        ``y`` currently is the autograd handle of a window (whose history the write cuts, as an in-place write to the
        reference's ``y`` under no_grad would)."""
        lo, hi = self._var_map[var]
        self._before_step()  # a write the holder made into the autograd handle comes first (it may cover other entries)
        with torch.no_grad():
            self._pop.v[lo:hi] = torch.as_tensor(val, dtype=torch.float32).to(self._pop.device)
        self._y_grad = None
        self._pop.refresh()

    def detach(self, requires_grad: bool = False, detach_params: bool = False):
        """Cut the autograd history of the state (reference nodes.py:176-196)."""
        self._before_step()  # values the holder wrote into the handle stay (the reference detaches the SAME tensor's data)
        self._y_grad = None
        self._y_requires_grad = requires_grad
        if detach_params:
            for name in self._train_names:
                self._tensor(name).requires_grad_(requires_grad)

    def reset(self, y: Union[np.ndarray, torch.Tensor] = None, idx: np.ndarray = None):
        """Set the state (cloned, reference nodes.py:198-211); ``idx`` resets a subset."""
        pop = self._pop
        if idx is not None:
            self._before_step()  # a subset reset keeps the other entries: including what the holder wrote into the handle
        with torch.no_grad():
            if y is None:
                new = torch.zeros_like(pop.v)
            else:
                new = torch.as_tensor(y).detach().to(dtype=torch.float32, device=pop.device).reshape(-1).clone()
            # the state is REBOUND to a new tensor (the reference does the same, nodes.py:202-211), so a tensor handed
            # out through `y` before the reset keeps its values
            if idx is None:
                if new.numel() != pop.v.numel():
                    raise RuntimeError(f"state has {pop.v.numel()} entries, got {new.numel()}")
                pop.v = new.contiguous()
            else:
                index = torch.as_tensor(np.asarray(idx), dtype=torch.long, device=pop.device)
                if index.numel() != new.numel() or (index.numel() and int(index.max()) >= pop.v.numel()):
                    raise RuntimeError("index/value size mismatch in reset")
                v_new = pop.v.clone()
                v_new[index] = new
                pop.v = v_new
        self._y_grad = None
        self._v_exposed = False
        pop.refresh()
        pop.reset_sampling()

    # ---- integration --------------------------------------------------------------------
    def _as_input(self, x, steps=None) -> Optional[torch.Tensor]:
        if x is None:
            return None
        if not isinstance(x, torch.Tensor):
            if isinstance(x, np.ndarray):
                raise RuntimeError("node inputs must be torch tensors (the reference raises here as well)")
            x = torch.tensor([float(x)], dtype=torch.float32)
        x = x.to(device=self._pop.device, dtype=torch.float32)
        if steps is None:  # single step: 0-d or 1-d
            x = x.reshape(1, -1)
        if x.dim() != 2:
            raise RuntimeError(f"inputs must be [steps, m], got {tuple(x.shape)}")
        if x.shape[1] not in (1, self._pop.rows):
            raise RuntimeError(f"The size of the input ({x.shape[1]}) must match the number of neurons "
                               f"({self._pop.rows}) or be 1")
        return x

    def _wants_grad(self, x) -> bool:
        if not torch.is_grad_enabled():
            return False
        if any(self._tensor(n).requires_grad for n in self._train_names):
            return True
        if x is not None and x.requires_grad:
            return True
        return self._y_grad is not None and self._y_grad.requires_grad

    def _grad_window(self, steps: int, x: Optional[torch.Tensor]) -> torch.Tensor:
        """`steps` Euler steps recorded as one autograd node; returns the [steps, N] output trajectory."""
        self._before_step()
        pop = self._pop
        v_in = self._y_grad
        if v_in is None:
            v_in = pop.v.detach().clone().requires_grad_(self._y_requires_grad)
        act = [pop.params[q] for q in _RateWindow.ACT_PARAMS] if pop.model == "li_sigmoid" else []
        traj, v_end = _RateWindow.apply(self, steps, x, v_in, pop.W, pop.params["tau"], pop.params["k"],
                                        pop.params["eta"], *act)
        self._y_grad = v_end
        return traj

    def forward(self, x, edges: Optional[list] = None) -> torch.Tensor:
        """One Euler step; returns the PRE-update output slice (reference nodes.py:166-170).  ``edges``: dense edges
        ``[(W_e [N, n_e], source vector [n_e])]`` whose products join the input inside the step kernel (no gradient
        passes through them; Population.forward)."""
        x2 = self._as_input(x)
        if self._wants_grad(x2):
            if edges:
                raise RuntimeError("in-kernel edges carry no gradient; add their products to the input instead")
            out = self._grad_window(1, x2)[0]
        else:
            self._before_step()
            out = torch.empty(1, self._pop.rows, dtype=torch.float32, device=self._pop.device)
            self._pop.forward(1, x=x2, traj=out, out_var=self._out_code, edges=edges)
            self._ungraphed_step()
            out = out[0]
        return out if self._out_slice is None else out[self._out_slice[0]:self._out_slice[1]]

    # ---- member of a coupled window (FeedbackNetwork._coupled_window, rp_forward_coupled) -------------------------------
    def _coupled_begin(self):
        self._before_step()
        self._ungraphed_step()

    def _coupled_kwargs(self) -> dict:
        return dict(out_var=self._out_code)

    def _coupled_end(self, bufs):
        pass

    def run_window(self, steps: int, x: Optional[torch.Tensor] = None, w_in: Optional[torch.Tensor] = None,
                   **kw):
        """Fused multi-step integration without autograd (see Population.forward for kwargs)."""
        self._before_step()
        self._ungraphed_step()
        return self._pop.forward(steps, x=x, w_in=w_in, out_var=self._out_code, **kw)

    def _ungraphed_step(self):
        """The state after a step that recorded no graph has no history; computed under ``no_grad`` it does not
        require grad either (the reference's ``y = y_old + dt * f(...)`` under ``Network.run(enable_grad=False)``), so
        ``net.state[...][:] = v`` afterwards is a plain write."""
        self._y_grad = None
        if not torch.is_grad_enabled():
            self._y_requires_grad = False


class Spike(torch.autograd.Function):
    """The reference's spike non-linearity as a stand-alone torch function (rectipy/nodes.py:468-481), for user code that
    builds its own nodes: forward = heaviside(x) with value `center` (1) at x == 0, backward = the surrogate gradient
    g / (1 + slope |x|)^2.  The compiled spiking populations apply the same two formulas inside their kernels (per node:
    ``spike_slope``); `slope` / `center` here are class attributes, as in the reference."""

    slope = 10.0
    center = 1.0

    @staticmethod
    def forward(ctx, x: torch.Tensor) -> torch.Tensor:
        ctx.save_for_backward(x)
        return torch.where(x == 0, torch.as_tensor(Spike.center, dtype=x.dtype, device=x.device), (x > 0).to(x.dtype))

    @staticmethod
    def backward(ctx, grad_output: torch.Tensor) -> torch.Tensor:
        x, = ctx.saved_tensors
        den = 1.0 + Spike.slope * x.abs()
        return grad_output / (den * den)


class SpikeNet(RateNet):
    """The reference's in-equation-reset spiking node (rectipy/nodes.py:292-330).  Its ``forward`` reads an attribute it
    never sets (``_y_spike``, nodes.py:324), i.e. it cannot run in the reference either; the name exists for imports
    (rectipy_tests/test_network.py:5) and construction fails with a clear message."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("SpikeNet (reset=False: the reset written into the equations) is not supported; the "
                                  "reference's own implementation of it is broken (rectipy/nodes.py:323-330). Use "
                                  "SpikeResetNet (reset=True).")


class SpikeResetNet(RateNet):
    """Spiking population with threshold + reset (reference ``SpikeResetNet``, rectipy/nodes.py:333-401).

    The state vector is cat(v, s).  As in the reference, ``y`` / ``__getitem__`` expose the state
    BEFORE the most recent step (nodes.py:387-392 never advances ``self.y``), while the current state
    lives in the population (``v``, ``s``).
    """

    state_vars = ["_y_start", "_y_spike", "_y_stop"]
    _spiking = True

    def __init__(self, template: Template, weights, node_vars, input_var, output_var, spike_var: str = None,
                 reset_var: str = None, **kwargs):
        if not template.spike_var:
            raise ValueError(f"template '{template.name}' is not a spiking model")
        def same(name, full):  # accept '<op>/<var>' and the bare variable name
            return name == full or name == full.split("/")[-1]

        if spike_var is not None and not same(spike_var, template.spike_var):
            raise KeyError(f"Variable {spike_var} was not found on template '{template.name}' as a spike input.")
        if reset_var is not None and not same(reset_var, template.reset_var):
            raise NotImplementedError(f"reset_var='{reset_var}': the kernel resets '{template.reset_var}'")
        kwargs.pop("spike_center", None)  # Spike.center is 1.0 in every use of the reference (x == 0 spikes)
        self._spike_slope = kwargs.pop("spike_slope", None)
        super().__init__(template, weights, node_vars, input_var, output_var, **kwargs)
        N = self._pop.rows
        self._var_map["reset_var"] = self._var_map[template.reset_var]
        self._param_map["spike_var"] = "spike"
        self._y_prev = None  # pre-update state of the last step
        self.spike_events: Optional[torch.Tensor] = None
        self._spike_count: Optional[torch.Tensor] = None
        # surrogate slope of Spike.backward; the reference's default is 100 / |theta - v_reset| (nodes.py:346)
        if self._spike_slope is None:
            self._spike_slope = 100.0 / abs(self._pop.theta - self._pop.v_reset)

    @classmethod
    def from_pyrates(cls, node, input_var: str, output_var: str, weights: np.ndarray = None,
                     source_var: str = None, target_var: str = None, spike_var: str = "spike",
                     reset_var: str = "v", train_params: list = None, **kwargs):
        if isinstance(spike_var, (list, tuple)):  # several spike/reset pairs: one node, several populations
            return MultiSpikeResetNet.from_pyrates(node, input_var, output_var, weights=weights, source_var=source_var,
                                                   target_var=target_var, spike_var=list(spike_var),
                                                   reset_var=list(reset_var), train_params=train_params, **kwargs)
        kwargs["spike_var"] = spike_var
        kwargs["reset_var"] = reset_var
        return super().from_pyrates(node, input_var, output_var, weights=weights, source_var=source_var,
                                    target_var=target_var, train_params=train_params, **kwargs)

    def _check_trainable(self):
        from .spiking_autograd import TRAINABLE
        model = self._template.model
        ok = ({"weights"} | set(TRAINABLE[model])) if model in TRAINABLE else set()
        bad = [n for n in self._train_names if n not in ok]
        if bad:
            raise NotImplementedError(f"no adjoint kernel for parameters {bad} of template '{self._template.name}'; "
                                      f"trainable: {sorted(ok)}")

    def _tensor(self, engine_name: str) -> torch.Tensor:
        if engine_name == "spike":
            raise KeyError("the spike input is internal to the fused kernel")
        return super()._tensor(engine_name)

    def _pack(self, v, s, aux=None, aux2=None) -> torch.Tensor:
        """cat of the engine's (v, s, aux, aux2) vectors in the template's state-vector order."""
        parts = {self._template.reset_var: v, self._template.source_var: s}
        if self._aux_name:
            parts[self._aux_name] = aux
        if self._aux2_name:
            parts[self._aux2_name] = aux2
        return torch.cat([parts[name] for name, _ in self._template.state])

    def _state_bufs(self, rows: int = 1):
        """Scratch rows that receive the pre-update (v, s, aux, aux2) of a step."""
        pop = self._pop
        mk = lambda: torch.empty(rows, pop.rows, dtype=torch.float32, device=pop.device)
        return mk(), mk(), (mk() if self._aux_name else None), (mk() if self._aux2_name else None)

    @property
    def y(self) -> torch.Tensor:
        if self._y_prev is not None:
            return self._y_prev
        return self.current_state

    @property
    def current_state(self) -> torch.Tensor:
        """State after the last step (what the reference keeps in _y_start/_y_spike/_y_stop), in template order."""
        return self._pack(self._pop.v, self._pop.source_local, self._pop.aux, self._pop.aux2)

    def enable_spike_recording(self, capacity: int):
        """Allocate a device buffer for (step, neuron) spike events."""
        dev = self._pop.device
        self.spike_events = torch.zeros(capacity, 2, dtype=torch.int32, device=dev)
        self._spike_count = torch.zeros(1, dtype=torch.int32, device=dev)

    def spikes(self) -> np.ndarray:
        """Recorded events sorted by (step, neuron); raises if the buffer overflowed."""
        if self._spike_count is None:
            raise RuntimeError("call enable_spike_recording() first")
        count = int(self._spike_count.item())
        if count > self.spike_events.shape[0]:
            raise RuntimeError(f"spike buffer overflow: {count} events, capacity {self.spike_events.shape[0]}")
        ev = self.spike_events[:count].cpu().numpy()
        return ev[np.lexsort((ev[:, 1], ev[:, 0]))]

    def set_state(self, var: str, val):
        """As in the reference, `y` of a spiking node is the lagging pre-update copy (nodes.py:387-392): writing a
        state variable through ``Network.set_var`` lands in that copy and leaves the integration state (the
        reference's _y_start/_y_spike/_y_stop) untouched.  Use :meth:`reset` to set the state."""
        lo, hi = self._var_map[var]
        if self._y_prev is None:
            self._y_prev = self.current_state
        elif self._y_prev.requires_grad:  # (the handle of a recorded step: leave it to its graph)
            self._y_prev = self._y_prev.detach().clone()
        with torch.no_grad():
            self._y_prev[lo:hi] = torch.as_tensor(val, dtype=torch.float32).to(self._pop.device)

    def reset(self, y=None, idx=None):
        pop = self._pop
        N = pop.rows
        total = len(self._template.state) * N
        with torch.no_grad():
            cur = self.y.clone()
            if y is None:
                new = torch.zeros(total, dtype=torch.float32, device=pop.device)
            else:
                new = torch.as_tensor(y).detach().to(dtype=torch.float32, device=pop.device).reshape(-1).clone()
            if idx is None:
                if new.numel() != total:
                    raise RuntimeError(f"state has {total} entries, got {new.numel()}")
                cur = new
            else:
                index = torch.as_tensor(np.asarray(idx), dtype=torch.long, device=pop.device)
                cur[index] = new
            for name in (n for n, _ in self._template.state):
                lo, hi = self._var_map[name]
                if name == self._template.reset_var:
                    pop.v.copy_(cur[lo:hi])
                elif name == self._template.source_var:
                    pop.set_source_local(cur[lo:hi])
                elif name == self._aux_name:
                    pop.aux.copy_(cur[lo:hi])
                else:
                    pop.aux2.copy_(cur[lo:hi])
        self._y_prev = None
        self._y_grad = None
        pop.reset_sampling()

    def _grad_window(self, steps: int, x: Optional[torch.Tensor]) -> torch.Tensor:
        """`steps` spiking steps recorded as one autograd node (surrogate-gradient BPTT)."""
        from .spiking_autograd import TRAINABLE, SpikeWindow
        pop = self._pop
        if pop.model not in TRAINABLE:
            raise NotImplementedError(f"no adjoint kernel for the '{pop.model}' template")
        y_in = self._y_grad
        if y_in is None:
            y_in = self.current_state.detach().requires_grad_(self._y_requires_grad)
        params = [pop.params[n] for n in TRAINABLE[pop.model]]
        traj, y_end = SpikeWindow.apply(self, steps, x, y_in, pop.W, *params)
        self._y_grad = y_end
        tv, ts, ta, tb = self._last_window
        # `y` exposes the pre-update state of the last step.  After a single recorded step that is the handle the step
        # started from: a feedback edge reading `node["out"]` then carries the gradient into this population's earlier
        # steps, as the reference's `self.y = torch.cat(...)` does (nodes.py:387; pinned by G37)
        self._y_prev = y_in if steps == 1 else \
            self._pack(tv[-1], ts[-1], None if ta is None else ta[-1], None if tb is None else tb[-1])
        return traj

    def detach(self, requires_grad: bool = False, detach_params: bool = False):
        super().detach(requires_grad=requires_grad, detach_params=detach_params)
        if self._y_prev is not None:
            self._y_prev = self._y_prev.detach()

    def forward(self, x, edges: Optional[list] = None) -> torch.Tensor:
        x2 = self._as_input(x)
        if self._wants_grad(x2):
            if edges:
                raise RuntimeError("in-kernel edges carry no gradient; add their products to the input instead")
            return self._grad_window(1, x2)[0]
        self._y_grad = None
        pop = self._pop
        N = pop.rows
        out = torch.empty(1, N, dtype=torch.float32, device=pop.device)
        sv, ss, sa, sb = self._state_bufs()
        pop.forward(1, x=x2, traj=out, save_v=sv, save_src=ss, save_aux=sa, save_aux2=sb, out_var=self._out_code,
                    x_target=self._x_target, spike_events=self.spike_events, spike_count=self._spike_count, edges=edges)
        self._y_prev = self._pack(sv[0], ss[0], None if sa is None else sa[0], None if sb is None else sb[0])
        return out[0]

    def _coupled_begin(self):
        self._y_grad = None

    def _coupled_kwargs(self) -> dict:
        return dict(out_var=self._out_code, x_target=self._x_target, spike_events=self.spike_events,
                    spike_count=self._spike_count)

    def _coupled_end(self, bufs):
        """`bufs`: the pre-update (v, s, aux, aux2) rows of the window's last step -- what `y` exposes from now on."""
        sv, ss, sa, sb = bufs
        self._y_prev = self._pack(sv[0], ss[0], None if sa is None else sa[0], None if sb is None else sb[0])

    def run_window(self, steps: int, x=None, w_in=None, **kw):
        """Fused window; the last step is issued separately so that ``y`` can expose its pre-update state."""
        self._y_grad = None
        pop = self._pop
        N = pop.rows
        kw.setdefault("spike_events", self.spike_events)
        kw.setdefault("spike_count", self._spike_count)
        kw.setdefault("x_target", self._x_target)
        step0 = kw.pop("step0", 0)
        idx = kw.pop("sample_idx", 0)

        def rows(t, a, b):
            return None if t is None else t[a:b]

        traj = kw.pop("traj", None)
        if steps > 1:
            idx = pop.forward(steps - 1, x=rows(x, 0, steps - 1), w_in=w_in, out_var=self._out_code, step0=step0,
                              sample_idx=idx, traj=rows(traj, 0, steps - 1), **kw)
        sv, ss, sa, sb = self._state_bufs()
        idx = pop.forward(1, x=rows(x, steps - 1, steps), w_in=w_in, out_var=self._out_code,
                          step0=step0 + steps - 1, sample_idx=idx, traj=rows(traj, steps - 1, steps),
                          save_v=sv, save_src=ss, save_aux=sa, save_aux2=sb, **kw)
        self._y_prev = self._pack(sv[0], ss[0], None if sa is None else sa[0], None if sb is None else sb[0])
        return idx


class MultiSpikeResetNet(SpikeResetNet):
    """Several spiking populations in ONE node, each with its own spike / reset variable pair (reference
    ``MultiSpikeResetNet``, rectipy/nodes.py:404-465, which PyRates builds from a multi-population circuit).

    Here the populations share one compiled template; they differ in their per-neuron parameters and are coupled
    through the blocks of one ``N x N`` matrix (N = sum of the population sizes), so the whole node is still ONE fused
    step kernel launch per Euler step.  ``populations`` maps population labels to sizes; ``spike_var[i]`` /
    ``reset_var[i]`` name population i's pair as ``"<label>/<op>/<var>"``; ``node_vars`` keys are
    ``"all/<op>/<var>"`` (scalar or length N) or ``"<label>/<op>/<var>"`` (scalar or length n_label).

    What the reference's class does differently from ``SpikeResetNet`` is kept:
      * ``forward`` returns the POST-update output slice and ``y`` is the current state (nodes.py:457-465);
      * spikes are detected on a private copy of every reset variable (``_y_reset``) that starts at ZERO, is refreshed
        after every step and is NOT touched by ``reset()`` (nodes.py:428-431, 451-462) -- the first step therefore never
        spikes, and the first step after a ``reset`` detects on the values from before it (``rp_run.spike_src``);
      * the reset is a masked assignment (same values as the blend of SpikeResetNet).
    The exposed state vector is population-major: [vars of population 0 in template order, vars of population 1, ...].
    Networks evaluate such a node step by step (its output timing differs from the fused windows' pre-update rule).
    """

    state_vars = ["_y_reset"]
    _fusable = False  # Network: no fused window (post-update output timing)

    def __init__(self, template: Template, weights, node_vars, input_var, output_var, spike_var: list, reset_var: list,
                 populations: dict, **kwargs):
        if not isinstance(populations, dict) or not populations:
            raise ValueError("`populations` must map population labels to sizes, e.g. {'exc': 800, 'inh': 200}")
        if not template.spike_var:
            raise ValueError(f"template '{template.name}' is not a spiking model")
        labels = list(populations)
        sizes = [int(populations[q]) for q in labels]
        N = sum(sizes)
        offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
        if len(spike_var) != len(labels) or len(reset_var) != len(labels):
            raise ValueError("one spike variable and one reset variable per population are required")
        for i, lab in enumerate(labels):
            if spike_var[i] not in (f"{lab}/{template.spike_var}", template.spike_var):
                raise KeyError(f"Variable {spike_var[i]} was not found on population '{lab}' as a spike input.")
            if reset_var[i] not in (f"{lab}/{template.reset_var}", template.reset_var):
                raise NotImplementedError(f"reset_var='{reset_var[i]}': the kernel resets '{template.reset_var}'")

        def split(name):  # "[label/]op/var" -> (label or None, "op/var")
            parts = name.split("/")
            if len(parts) == 3 and parts[0] in labels:
                return parts[0], "/".join(parts[1:])
            return None, "/".join(parts[-2:])

        # per-neuron vectors of the whole node from population-wise values
        full = {}
        known = dict(template.params)
        known.update(dict(template.state))
        for key, val in (node_vars or {}).items():
            parts = key.split("/")
            scope, name = (parts[0], "/".join(parts[1:])) if len(parts) == 3 else ("all", key)
            if name not in known:
                raise KeyError(f"Variable {name} was not found on template '{template.name}'.")
            vec = full.setdefault(name, np.full(N, known[name], dtype=np.float32))
            arr = val.detach().cpu().numpy() if isinstance(val, torch.Tensor) else np.asarray(val, dtype=np.float32)
            if scope == "all":
                vec[:] = arr
            elif scope in labels:
                p = labels.index(scope)
                vec[offsets[p]:offsets[p + 1]] = arr
            else:
                raise KeyError(f"node_vars key '{key}': unknown population '{scope}' (populations: {labels})")
        in_label, in_name = split(input_var)
        out_label, out_name = split(output_var)
        super().__init__(template, weights, {f"all/{k}": v for k, v in full.items()}, in_name, out_name, **kwargs)
        if self._pop.rows != N:
            raise RuntimeError(f"weights are {self._pop.rows} x {self._pop.n}, the populations hold {N} neurons")
        if self._pop.model in ("iku", "ik_biexp"):
            raise NotImplementedError("population-mean templates cannot be split into several spike/reset pairs")
        self._labels, self._sizes, self._offsets = labels, sizes, offsets
        dev = self._pop.device
        state_names = [q for q, _ in template.state]
        # exposed layout: population-major; internal layout (engine, autograd): variable-major over all N neurons
        perm, var_map, pos = [], {}, 0
        for p, lab in enumerate(labels):
            for j, name in enumerate(state_names):
                perm.append(torch.arange(j * N + offsets[p], j * N + offsets[p + 1]))
                var_map[f"{lab}/{name}"] = (pos, pos + sizes[p])
                pos += sizes[p]
        self._perm = torch.cat(perm).to(dev)
        self._inv_perm = torch.empty_like(self._perm)
        self._inv_perm[self._perm] = torch.arange(self._perm.numel(), device=dev)
        for i, lab in enumerate(labels):
            var_map[f"spike_reset_{i}"] = var_map[f"{lab}/{template.reset_var}"]
            self._param_map[f"spike_var_{i}"] = "spike"
        self._param_map.pop("spike_var", None)
        # output: one population's variable (a slice, as the reference) or the variable of every neuron
        if out_label is not None:
            self._out_index = torch.arange(*var_map[f"{out_label}/{out_name}"], device=dev)
            var_map["out"] = var_map[f"{out_label}/{out_name}"]
        else:
            self._out_index = torch.cat([torch.arange(*var_map[f"{lab}/{out_name}"]) for lab in labels]).to(dev)
        self.n_out = int(self._out_index.numel())
        # input: into one population (zeros elsewhere) or into every neuron
        self._in_slice = None
        if in_label is not None:
            p = labels.index(in_label)
            self._in_slice = (int(offsets[p]), int(offsets[p + 1]))
            self.n_in = sizes[p]
        self._internal_var_map = {name: (j * N, (j + 1) * N) for j, name in enumerate(state_names)}  # autograd layout
        self._var_map = var_map
        self._detector = torch.zeros(N, dtype=torch.float32, device=dev)  # the reference's _y_reset copies

    def __getitem__(self, item):
        if item == "out":
            return self.y[self._out_index]
        return super().__getitem__(item)

    @classmethod
    def from_pyrates(cls, node, input_var: str, output_var: str, weights: np.ndarray = None, source_var: str = None,
                     target_var: str = None, spike_var: list = None, reset_var: list = None, train_params: list = None,
                     **kwargs):
        kwargs["spike_var"], kwargs["reset_var"] = spike_var, reset_var
        if "populations" not in kwargs:
            raise ValueError("several spike/reset pairs need `populations={label: size, ...}` (PyRates circuits are "
                             "not parsed; the populations share the node template)")
        strip = lambda v: None if v is None else "/".join(v.split("/")[-2:])
        return RateNet.from_pyrates.__func__(cls, node, input_var, output_var, weights=weights, source_var=strip(source_var),
                                             target_var=strip(target_var), train_params=train_params, **kwargs)

    # ---- state ------------------------------------------------------------------------------------------------
    def _internal_state(self) -> torch.Tensor:
        return self.current_state if self._y_grad is None else self._y_grad

    @property
    def y(self) -> torch.Tensor:
        """Current state (post-update), population-major."""
        return self._internal_state()[self._perm]

    def set_state(self, var: str, val):
        lo, hi = self._var_map[var]
        cur = self.y.detach().clone()
        cur[lo:hi] = torch.as_tensor(val, dtype=torch.float32).to(cur.device)
        self.reset(cur)

    def reset(self, y=None, idx=None):
        total = self._perm.numel()
        dev = self._pop.device
        with torch.no_grad():
            if y is None:
                new = torch.zeros(total, dtype=torch.float32, device=dev)
            else:
                new = torch.as_tensor(y).detach().to(dtype=torch.float32, device=dev).reshape(-1).clone()
            if idx is not None:
                cur = self.y.detach().clone()
                cur[torch.as_tensor(np.asarray(idx), dtype=torch.long, device=dev)] = new
                new = cur
            if new.numel() != total:
                raise RuntimeError(f"state has {total} entries, got {new.numel()}")
            internal = torch.empty_like(new)
            internal[self._perm] = new  # exposed = internal[perm]
            pop, N = self._pop, self._pop.rows
            for j, (name, _) in enumerate(self._template.state):
                chunk = internal[j * N:(j + 1) * N]
                if name == self._template.reset_var:
                    pop.v.copy_(chunk)
                elif name == self._template.source_var:
                    pop.set_source_local(chunk)
                elif name == self._aux_name:
                    pop.aux.copy_(chunk)
                else:
                    pop.aux2.copy_(chunk)
        self._y_prev = None
        self._y_grad = None
        self._pop.reset_sampling()  # the spike detector keeps its old values (the reference has no reset override)

    # ---- integration ------------------------------------------------------------------------------------------
    def _embed(self, x2: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        if x2 is None or self._in_slice is None:
            return x2
        lo, hi = self._in_slice
        if x2.shape[1] not in (1, hi - lo):
            raise RuntimeError(f"The size of the input ({x2.shape[1]}) must match the population ({hi - lo}) or be 1")
        full = torch.zeros(x2.shape[0], self._pop.rows, dtype=torch.float32, device=x2.device)
        full[:, lo:hi] = x2
        return full

    def _as_input(self, x, steps=None):
        if self._in_slice is None:
            return super()._as_input(x, steps)
        if x is None:
            return None
        if not isinstance(x, torch.Tensor):
            if isinstance(x, np.ndarray):
                raise RuntimeError("node inputs must be torch tensors (the reference raises here as well)")
            x = torch.tensor([float(x)], dtype=torch.float32)
        x = x.to(device=self._pop.device, dtype=torch.float32)
        if steps is None:
            x = x.reshape(1, -1)
        if x.dim() != 2:
            raise RuntimeError(f"inputs must be [steps, m], got {tuple(x.shape)}")
        return self._embed(x)  # checks the width against the addressed population and places it among zeros

    def forward(self, x) -> torch.Tensor:
        """One step; returns the POST-update output slice (reference nodes.py:449-465)."""
        x2 = self._as_input(x)
        pop = self._pop
        if self._wants_grad(x2):
            if not torch.equal(self._detector >= pop.theta, pop.v >= pop.theta):
                raise NotImplementedError("BPTT through a step whose spike detector differs from the state (the first "
                                          "step after construction / reset with neurons above threshold)")
            self._grad_window(1, x2)
        else:
            self._y_grad = None
            pop.forward(1, x=x2, out_var=self._out_code, x_target=self._x_target, spike_events=self.spike_events,
                        spike_count=self._spike_count, spike_src=self._detector)
        self._detector = pop.v.detach().clone()
        return self.y[self._out_index]

    def run_window(self, steps: int, x=None, w_in=None, traj=None, step0: int = 0, **kw):
        """`steps` steps in one library call; ``traj`` receives the POST-update output of every step."""
        if w_in is not None or any(kw.get(q) is not None for q in ("samples", "rec")):
            raise NotImplementedError("MultiSpikeResetNet windows deliver `traj` only (post-update output timing)")
        self._y_grad = None
        pop = self._pop
        x2 = self._embed(x)
        mk = lambda: torch.empty(steps, pop.rows, dtype=torch.float32, device=pop.device)
        saves = dict(save_v=mk(), save_src=mk(), save_aux=mk() if self._aux_name else None)
        pop.forward(steps, x=x2, out_var=self._out_code, step0=step0, x_target=self._x_target,
                    spike_events=self.spike_events, spike_count=self._spike_count, spike_src=self._detector, **saves)
        self._detector = pop.v.detach().clone()
        if traj is not None:
            # post-update state of step t = pre-update state of step t+1; the last row is the current state
            pre = self._pack_rows(saves["save_v"], saves["save_src"], saves["save_aux"])  # [steps, total] internal layout
            post = torch.cat((pre[1:], self.current_state.reshape(1, -1)))
            traj.copy_(post[:, self._perm][:, self._out_index])
        return 0

    def _pack_rows(self, v, s, aux):
        parts = {self._template.reset_var: v, self._template.source_var: s}
        if self._aux_name:
            parts[self._aux_name] = aux
        return torch.cat([parts[name] for name, _ in self._template.state], dim=1)
