"""``Network``: the user-facing graph of nodes and edges with ``run`` / ``fit_bptt`` / ``test``,
API-compatible with the reference's ``rectipy.Network`` (rectipy/network.py:16-1193) for the
integration + BPTT path.

What is different underneath: the reference walks the graph in Python once per Euler step
(network.py:588-599, 962-981).  Here a chain ``[input nodes/edges] -> one diff-eq population ->
[readout edges/nodes]`` is executed as ONE fused window on the GPU: the population's T steps are a
single C-ABI call (rectipy_amd.engine), the input projection is folded into the step kernel, and
the readout is applied to the sampled trajectory afterwards.  Topologies that do not match fall
back to a per-step walk in which every diff-eq node still steps through the HIP kernels.
"""
from time import perf_counter
from typing import Callable, Iterator, List, Optional, Union

import numpy as np
import torch
from networkx import DiGraph, topological_sort

from .edges import Linear
from .nodes import InstantNode, RateNet, SpikeResetNet
from .observer import Observer
from .utility import add_op_name, retrieve_from_dict

_OPTIMIZERS = {"sgd": torch.optim.SGD, "adam": torch.optim.Adam, "adamw": torch.optim.AdamW,
               "adagrad": torch.optim.Adagrad, "adadelta": torch.optim.Adadelta, "adamax": torch.optim.Adamax,
               "rmsprop": torch.optim.RMSprop, "rprop": torch.optim.Rprop}
_LOSSES = {"mse": torch.nn.MSELoss, "l1": torch.nn.L1Loss, "nll": torch.nn.NLLLoss, "ce": torch.nn.CrossEntropyLoss,
           "kld": torch.nn.KLDivLoss, "hinge": torch.nn.HingeEmbeddingLoss}

_WINDOW_BYTES = 2 << 30  # cap for [T, N] staging buffers of one fused call


class PerformanceWarning(UserWarning):
    """A graph is walked step by step from Python (one kernel launch group per node and Euler step, per-step autograd in
    training) instead of running as fused windows inside the library: same results, several times slower.  The message names
    the node or edge that forced it."""


class Network:
    """Graph of diff-eq populations, activation nodes and linear edges (reference network.py:16-44)."""

    def __init__(self, dt: float, device: str = "cuda:0", dtype: torch.dtype = torch.float32, backend=None,
                 tune: Optional[bool] = None):
        self.graph = DiGraph()
        self.device = device
        if dtype is not torch.float32:
            from .nodes import _warn_f64
            _warn_f64(torch.float64)  # one warning: the engine, its inputs and its edges are float32
        self.dtype = torch.float32
        self.dt = dt
        self._backend = backend  # test seam, see rectipy_amd.engine
        self._tune = tune        # True: populations measure their step kernel's walk at the first step (tuning.py)
        self._var_map = {}
        self._in_node = None
        self._out_node = None
        self._order: List[str] = []
        self._train_edge = ()
        self._compiled_shape = None  # (nodes, edges) of the graph at the last successful compile()
        self._input_device = None

    # ---- container protocol -------------------------------------------------------------
    def __getitem__(self, item):
        if isinstance(item, tuple):
            return self.graph[item[0]][item[1]]
        return self.graph.nodes[item]

    def __iter__(self):
        for n in self.graph.nodes:
            yield self[n]

    def __len__(self):
        return len(self.graph.nodes)

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    @property
    def n_out(self) -> int:
        return self[self._out_node]["n_out"] if self._out_node is not None else 0

    @property
    def n_in(self) -> int:
        return self[self._in_node]["n_in"] if self._in_node is not None else 0

    @property
    def nodes(self):
        return self.graph.nodes

    @property
    def state(self) -> dict:
        """State vectors of all diff-eq nodes (live tensors, as the reference: network.py:88-98)."""
        return {n: self.get_node(n).y for n in self.nodes if hasattr(self.get_node(n), "y")}

    def get_node(self, node: str):
        return self[node]["node"]

    def get_edge(self, source: str, target: str) -> Linear:
        return self[source, target]["edge"]

    def get_var(self, node: str, var: str):
        try:
            return self.get_node(node)[self._relabel_var(var)]
        except KeyError:
            return self[node][var]

    def set_var(self, node: str, var: str, val):
        try:
            n = self.get_node(node)
            try:
                n.set_param(self._relabel_var(var), val)
            except KeyError:
                if hasattr(n, "set_state"):   # compiled population: write the device state itself
                    n.set_state(self._relabel_var(var), val)
                else:
                    with torch.no_grad():
                        n[self._relabel_var(var)][:] = torch.as_tensor(val)
        except KeyError:
            raise KeyError(f"Variable {var} was not found on node {node}.")

    # ---- construction ---------------------------------------------------------------------
    def add_node(self, label: str, node, node_type: str, op: str = None, **node_attrs) -> None:
        """Add any object that follows the node protocol (reference network.py:178-211)."""
        if op:
            for p in getattr(node, "parameter_names", []):
                add_op_name(op, p, self._var_map)
            for v in getattr(node, "variable_names", []):
                add_op_name(op, v, self._var_map)
        self.graph.add_node(label, node=node, node_type=node_type, n_out=node.n_out, n_in=node.n_in, eval=True,
                            out=None, **node_attrs)

    def add_diffeq_node(self, label: str, node: str, input_var: str, output_var: str, weights: np.ndarray = None,
                        source_var: str = None, target_var: str = None, spike_var: Union[str, list] = None,
                        reset_var: Union[str, list] = None, reset: bool = True, op: str = None,
                        train_params: list = None, **kwargs) -> RateNet:
        """Add a population defined by a built-in template (signature of reference network.py:213-216).

        ``op`` shorthand: bare variable names get ``<op>/`` prefixed, as the reference does
        (network.py:262-285)."""
        names = {"svar": source_var, "tvar": target_var, "in_ext": input_var, "out": output_var, "spike": spike_var,
                 "reset": reset_var}
        record_vars = kwargs.pop("record_vars", None)
        self._var_map = {}
        if op is not None:
            for key, var in list(names.items()):
                names[key] = [add_op_name(op, v, self._var_map) for v in var] if isinstance(var, list) \
                    else add_op_name(op, var, self._var_map)
            if record_vars:
                record_vars = [add_op_name(op, v, self._var_map) for v in record_vars]
            if train_params:
                train_params = [add_op_name(op, p, self._var_map) for p in train_params]
            if "node_vars" in kwargs:
                kwargs["node_vars"] = {(k if "/" in k else f"all/{op}/{k}"): v for k, v in kwargs["node_vars"].items()}
        if record_vars:
            kwargs["var_mapping"] = {v: v for v in record_vars}
        kwargs.setdefault("device", self.device)
        kwargs.setdefault("dt", self.dt)
        if self._backend is not None:
            kwargs.setdefault("backend", self._backend)
        if self._tune is not None:
            kwargs.setdefault("tune", self._tune)
        common = dict(weights=weights, source_var=names["svar"], target_var=names["tvar"], train_params=train_params)
        if spike_var is None:
            new = RateNet.from_pyrates(node, names["in_ext"], names["out"], **common, **kwargs)
        elif reset_var is None:
            raise ValueError("To define a reservoir with a spiking neural network layer, please provide the "
                             "name of the variable that should be reset after a spike occurred (`reset_var`).")
        elif not reset:
            raise NotImplementedError("reset=False (in-equation reset, the reference's SpikeNet) is not supported; "
                                      "the reference's own implementation of it is broken (nodes.py:323-330)")
        else:
            new = SpikeResetNet.from_pyrates(node, names["in_ext"], names["out"], spike_var=names["spike"],
                                             reset_var=names["reset"], **common, **kwargs)
        self.add_node(label, node=new, node_type="diff_eq", op=op)
        return new

    def add_func_node(self, label: str, n: int, activation_function: str, **kwargs) -> InstantNode:
        kwargs.pop("node_type", None)
        node = InstantNode(n, activation_function, **kwargs)
        self.add_node(label, node=node, node_type="func_instant")
        return node

    def add_edge(self, source: str, target: str, weights=None, train: Optional[str] = None,
                 edge_attrs: dict = None, **kwargs) -> Linear:
        """Linear projection source -> target (reference network.py:340-401); `train` in {None, 'gd'}."""
        from .edge_variants import LinearFilter, LinearMasked, LinearMemory, LinearMemoryFilter
        if "mask" in kwargs:  # same selection rule as the reference (network.py:372-380)
            LinEdge = LinearMasked
        elif "delays" in kwargs:
            LinEdge = LinearMemoryFilter if "filter_weights" in kwargs else LinearMemory
        elif "filter_weights" in kwargs:
            LinEdge = LinearFilter
        else:
            LinEdge = Linear
        kwargs.update(n_in=self[source]["n_out"], n_out=self[target]["n_in"], weights=weights, dtype=self.dtype)
        if train is None:
            edge = LinEdge(**kwargs, detach=True)
        elif train == "gd":
            edge = LinEdge(**kwargs, detach=False)
        elif train == "rls":
            from .rls import RLS
            kwargs.pop("delta", None)  # accepted and ignored by the reference's examples (documentation/rls_example.py:36)
            edge = RLS(**kwargs, backend=self._backend)
            self._train_edge = (source, target)
        else:
            raise ValueError("Invalid option for keyword argument `train`. Valid options: None, 'gd', 'rls'.")
        edge_device = self.device if self._backend is None else "cpu"
        if self._backend is not None and hasattr(edge, "memory_stage"):
            edge._backend = self._backend   # (test seam: the edge's memory stage on the injected backend)
        self.graph.add_edge(source, target, edge=edge.to(edge_device), trainable=train is not None, n_in=edge.n_in,
                            n_out=edge.n_out, **(edge_attrs or {}))
        return edge

    def pop_node(self, node: str):
        data = self.get_node(node)
        self.graph.remove_node(node)
        return data

    def pop_edge(self, source: str, target: str) -> Linear:
        edge = self.get_edge(source, target)
        self.graph.remove_edge(source, target)
        return edge

    def clear(self):
        for node in list(self.nodes):
            self.pop_node(node)

    def compile(self):
        """Find the unique input / output node and an evaluation order (reference network.py:439-460)."""
        shape = (tuple(self.graph.nodes), tuple(self.graph.edges))
        if shape == self._compiled_shape:  # same nodes and edges as at the last successful compile: nothing to redo
            return
        self._compiled_shape = None
        in_nodes = [n for n in self.graph.nodes if self.graph.in_degree(n) == 0]
        if len(in_nodes) != 1:
            raise ValueError(f"Unable to identify the input node of the Network. Nodes that have no input edges: "
                             f"{in_nodes}. Make sure that exactly one such node exists in the network.")
        out_nodes = [n for n in self.graph.nodes if self.graph.out_degree(n) == 0]
        if len(out_nodes) != 1:
            raise ValueError(f"Unable to identify the output node of the Network. Nodes that have no outgoing edges: "
                             f"{out_nodes}. Make sure that exactly one such node exists in the network.")
        self._in_node, self._out_node = in_nodes[0], out_nodes[0]
        self._order = list(topological_sort(self.graph))
        # the reference's backward view of the graph (network.py:455-460, 953-960): node -> its source nodes, for every
        # node on a path to the output node that has sources
        self._bwd_graph = {n: list(self.graph.predecessors(n)) for n in self._order if self.graph.in_degree(n) > 0}
        self._compiled_shape = shape

    # ---- per-step evaluation (drop-in for user loops) ----------------------------------------
    def forward(self, x) -> torch.Tensor:
        """One step through the whole graph (reference network.py:462-478)."""
        if not self._order:
            self.compile()
        if isinstance(x, torch.Tensor) and x.dtype != self.dtype:
            x = x.to(self.dtype)  # a float64 input to the float32 engine (Network(dtype=float64) warns once)
        for n in self._order:
            preds = list(self.graph.predecessors(n))
            if preds:
                inp = None
                for p in preds:
                    contrib = self.get_edge(p, n).forward(self[p]["out"])
                    inp = contrib if inp is None else inp + contrib
            else:
                inp = x
            self[n]["out"] = self.get_node(n).forward(inp)
        return self[self._out_node]["out"]

    def parameters(self, recurse: bool = True) -> Iterator:
        for n in self.graph.nodes:
            for p in self.get_node(n).parameters(recurse=recurse):
                yield p
        for s, t in self.graph.edges:
            for p in self.get_edge(s, t).parameters():
                yield p

    def detach(self, requires_grad: bool = True, detach_params: bool = False) -> None:
        for n in self.nodes:
            node = self.get_node(n)
            if hasattr(node, "y"):
                node.detach(requires_grad=requires_grad, detach_params=detach_params)
        # the memory of the edges (delay buffer, filter state) becomes a constant of the next window too -- an extension: the
        # reference (network.py:511-514) leaves it attached, and its second truncated window fails in autograd
        for s_, t_ in self.graph.edges:
            edge = self.get_edge(s_, t_)
            for name in ("buffer", "y"):
                val = getattr(edge, name, None)
                if isinstance(val, torch.Tensor) and val.grad_fn is not None:
                    setattr(edge, name, val.detach())

    def reset(self, state: dict = None):
        for n in self.nodes:
            node = self.get_node(n)
            if hasattr(node, "y"):
                if state and n in state:
                    node.reset(y=state[n])
                else:
                    node.reset()

    # ---- fused-window planning ------------------------------------------------------------------
    def _chain_plan(self):
        """(pre, rnn_label, post) when the graph is a simple chain around ONE diff-eq node whose other
        members are element-wise activation nodes and plain Linear edges; else None."""
        order = self._order
        if getattr(self, "_force_stepwise", False):  # test seam: always walk the graph step by step
            return None
        if any(self.graph.in_degree(n) > 1 or self.graph.out_degree(n) > 1 for n in order):
            return None
        diffeq = [n for n in order if self[n]["node_type"] == "diff_eq"]
        if len(diffeq) != 1 or not isinstance(self.get_node(diffeq[0]), RateNet) \
                or not getattr(self.get_node(diffeq[0]), "_fusable", True):
            return None
        for n in order:
            node = self.get_node(n)
            if n != diffeq[0] and not (isinstance(node, InstantNode) and node.name in InstantNode.batched):
                return None
        for s, t in self.graph.edges:
            if not self._windowable(self.get_edge(s, t)):
                return None
        i = order.index(diffeq[0])
        return order[:i], diffeq[0], order[i + 1:]

    @staticmethod
    def _windowable(edge) -> bool:
        """Edges that can be evaluated for a whole window of their source's outputs: plain / RLS projections, and the
        masked, delay and filter variants (causal maps of the source's window on an acyclic edge,
        edge_variants.forward_window).  Foreign edge classes are walked step by step."""
        from .edge_variants import _Stateful
        from .rls import RLS
        return type(edge) in (Linear, RLS) or isinstance(edge, _Stateful)

    def _memoryless_linear(self, labels: List[str], entry: Optional[str]) -> bool:
        """True when the chain `entry -> labels...` is a LINEAR, MEMORYLESS map: identity nodes joined by plain / RLS /
        masked projections.  Only then does it commute with the buffer mean of `run` (so that it may be applied to the
        sampled means instead of every step); an edge with memory must see every step."""
        from .edge_variants import LinearMasked
        from .rls import RLS
        prev = entry
        for lab in labels:
            if self.get_node(lab).name != "identity":
                return False
            if prev is not None and type(self.get_edge(prev, lab)) not in (Linear, RLS, LinearMasked):
                return False
            prev = lab
        return True

    def _edge_window(self, edge, X: torch.Tensor) -> torch.Tensor:
        """Outputs of `edge` for the window X [T, n_in] of its source's per-step outputs."""
        fw = getattr(edge, "forward_window", None)
        return edge.forward(X) if fw is None else fw(X)

    def _apply_chain(self, labels: List[str], X: Optional[torch.Tensor], entry: Optional[str], exit_to: Optional[str]):
        """Apply activation nodes + the edges between them to a whole [T, m] batch.  `entry` is the node
        feeding labels[0] (None: X is the external input); `exit_to` the node after labels[-1]."""
        prev = entry
        for lab in labels:
            if prev is not None:
                X = self._edge_window(self.get_edge(prev, lab), X)
            X = self.get_node(lab).forward(X)
            prev = lab
        if exit_to is not None and prev is not None:
            X = self._edge_window(self.get_edge(prev, exit_to), X)
        return X

    def _to_input(self, inputs) -> torch.Tensor:
        dev = self.device if self._backend is None else "cpu"
        if isinstance(inputs, torch.Tensor) and inputs.dtype == torch.float32 and inputs.dim() == 2:
            if self._input_device is None:
                self._input_device = torch.empty(0, device=dev).device  # (with the index torch resolves `dev` to)
            if inputs.device == self._input_device:
                return inputs  # already what the engine takes: nothing to convert
        if isinstance(inputs, np.ndarray):
            inputs = torch.tensor(inputs, device=dev, dtype=torch.float32)
        inputs = inputs.to(device=dev, dtype=torch.float32)
        if inputs.dim() == 1:
            inputs = inputs.reshape(-1, 1)
        return inputs

    @staticmethod
    def _sample_steps(steps: int, cutoff: int, sampling_steps: int, start: int = 0) -> List[int]:
        return [t for t in range(start, start + steps) if t >= cutoff and t % sampling_steps == 0]

    @staticmethod
    def _seq_mean(block: torch.Tensor) -> torch.Tensor:
        """Mean over dim -2 of [..., m, width] the way the step kernel forms a sample (`acc += out` in step order, then
        `acc / count`): sequential float32 additions and a TRUE division.  ``torch.mean`` reduces in a device-dependent
        order and divides through a rounded reciprocal on the GPU, so sampled outputs would differ in the last bit between
        a gradient-tracking run and the fused no-grad run, and between devices."""
        m = block.shape[-2]
        acc = block[..., 0, :]
        for u in range(1, m):
            acc = acc + block[..., u, :]
        return acc if m == 1 else torch.div(acc, torch.full((), float(m), dtype=acc.dtype, device=acc.device))

    @staticmethod
    def _buffer_means(per_step: torch.Tensor, sample_at: List[int], cutoff: int, offset: int = 0) -> torch.Tensor:
        """Row j = mean of the outputs buffered since the previous sample (reference network.py:590-597): per_step[t - offset]
        for t from the previous sample + 1 (`cutoff` for the first one) to sample_at[j].  All windows but the first have
        the same length, so they are averaged together: `sampling_steps` element-wise additions in all."""
        if not sample_at:
            return per_step[:0]
        first = Network._seq_mean(per_step[cutoff - offset:sample_at[0] + 1 - offset]).unsqueeze(0)
        if len(sample_at) == 1:
            return first
        m = sample_at[1] - sample_at[0]
        rest = per_step[sample_at[0] + 1 - offset:sample_at[-1] + 1 - offset]
        return torch.cat((first, Network._seq_mean(rest.reshape(len(sample_at) - 1, m, *per_step.shape[1:]))))

    # ---- simulation ---------------------------------------------------------------------------
    def run(self, inputs: Union[np.ndarray, torch.Tensor], sampling_steps: int = 1, cutoff: int = 0,
            verbose: bool = True, enable_grad: bool = True, **kwargs) -> Observer:
        """Integrate the input-driven network (signature and recording semantics of reference
        network.py:542-601): steps below `cutoff` are computed but not buffered; a sample is taken when
        ``step >= cutoff and step % sampling_steps == 0`` and stores the MEAN of the outputs buffered
        since the previous sample; recorded state variables are the values right after that step."""
        inputs = self._to_input(inputs)
        steps = inputs.shape[0]
        truncate_steps = kwargs.pop("truncate_steps", steps)
        self.compile()
        if "obs" in kwargs:
            obs = kwargs.pop("obs")
        else:
            obs = Observer(dt=self.dt, record_loss=kwargs.pop("record_loss", False), **kwargs)
        rec_vars = list(obs.recorded_state_variables)
        plan = self._chain_plan()
        if plan is not None and all(v[0] == plan[1] for v in rec_vars):
            rnn = self.get_node(plan[1])
            slots = self._record_slots(rnn, rec_vars)
            if slots is not None:
                with (torch.enable_grad() if enable_grad else torch.no_grad()):
                    self._run_fused(plan, inputs, sampling_steps, cutoff, truncate_steps, obs, slots, verbose)
                return obs
        rec_slots = self._dag_plan(rec_vars) if plan is None else None
        if rec_slots is not None:
            with (torch.enable_grad() if enable_grad else torch.no_grad()):
                self._run_windows(inputs, sampling_steps, cutoff, truncate_steps, obs, rec_vars, rec_slots, verbose)
            return obs
        coupled = self._coupled_plan(rec_vars) if plan is None else None
        if coupled is not None:
            with (torch.enable_grad() if enable_grad else torch.no_grad()):
                if not self._coupled_wants_grad(inputs):  # populations joined by dense edges: one library call per window
                    with torch.no_grad():
                        self._run_coupled(coupled, inputs, sampling_steps, cutoff, obs, rec_vars, verbose)
                    return obs
                grad_plan = None if rec_vars else self._coupled_grad_plan()
                if grad_plan is not None:  # ... and the graph of one autograd node per window (rp_backward_coupled)
                    self._run_coupled_grad(grad_plan, inputs, sampling_steps, cutoff, truncate_steps, obs, verbose)
                    return obs
        if coupled is not None:  # (the forward plan exists: it is the adjoint of the coupled call that is missing)
            self._warn_stepwise("Network.run with a gradient", "coupled_grad") if not rec_vars else \
                self._warn_stepwise("Network.run with a gradient and recorded variables")
        elif plan is not None:
            self._why_not = dict(getattr(self, "_why_not", {}), rec="a recorded variable is not recordable inside the step kernel")
            self._warn_stepwise("Network.run", "rec")
        else:
            self._warn_stepwise("Network.run", "dag", "coupled")
        self._run_stepwise(inputs, sampling_steps, cutoff, truncate_steps, obs, rec_vars, enable_grad, verbose)
        return obs

    def _coupled_plan(self, rec_vars):
        """Graphs with cycles (feedback edges) exist in :class:`rectipy_amd.feedback.FeedbackNetwork` only."""
        return None

    def _coupled_grad_plan(self):
        return None

    # ---- B trajectories through one pass over W -------------------------------------------------------
    def run_batch(self, inputs, sampling_steps: int = 1, cutoff: int = 0, y0=None, verbose: bool = False,
                  mfma: Optional[bool] = None, exact: bool = False) -> List[Observer]:
        """Integrate B independent trajectories of the SAME network -- ``inputs``: [B, T, m] -- and return one
        :class:`Observer` per trajectory, each as :meth:`run` would have produced it for ``inputs[b]`` from the same
        initial state (``y0``: [B, N] initial states of the population; default: its current state for every b).

        Not part of the reference's API (it has no batch axis: B trials are B passes of its time loop, each streaming W
        again).  Here every Euler step streams W ONCE for all B trajectories (``rp_forward_batched``), in groups of at most
        32: up to 4 trajectories on the vector ALUs (each BIT-IDENTICAL to an unbatched `run`), 5..32 on the matrix cores
        (equal to unbatched runs up to fp32 rounding: their own summation order).  ``exact=True`` keeps
        groups of up to 8 on the vector ALUs, i.e. bit-identical to `run`, at 0.86 instead of 0.63 ms per step for 8
        trajectories at N = 32768; ``mfma=True`` puts smaller groups on the matrix cores as well.

        Sampling happens inside the step kernels exactly as in `run` (running fp32 sum of the buffered outputs in step
        order / count, per trajectory): memory is O(B x samples x N), never O(B x T x N).  Only behind a read-out that does
        not commute with the buffer mean (a non-linear activation, an edge with memory) are the per-step outputs staged,
        one window of steps at a time.  Chain topologies around one RATE population; the network's own state is left
        untouched."""
        from .engine import BatchedPopulation
        self.compile()
        plan = self._chain_plan()
        if plan is None:
            raise NotImplementedError("run_batch needs a chain [inputs] -> one population -> [read-out]")
        pre, label, post = plan
        rnn: RateNet = self.get_node(label)
        pop = rnn._pop
        if isinstance(rnn, SpikeResetNet):
            raise NotImplementedError("run_batch covers the rate templates")
        from .edge_variants import LinearMasked, _Stateful
        if any(isinstance(self.get_edge(s_, t_), _Stateful) and not isinstance(self.get_edge(s_, t_), LinearMasked)
               for s_, t_ in self.graph.edges):
            raise NotImplementedError("run_batch: edges with memory (delay / filter) hold ONE state, not one per trajectory")
        dev = self.device if self._backend is None else "cpu"
        X = torch.as_tensor(inputs).to(device=dev, dtype=torch.float32)
        if X.dim() != 3:
            raise RuntimeError(f"inputs must be [B, T, m], got {tuple(X.shape)}")
        B, T = int(X.shape[0]), int(X.shape[1])
        sample_at = self._sample_steps(T, cutoff, sampling_steps)
        S = len(sample_at)
        N = pop.rows
        f32 = dict(dtype=torch.float32, device=pop.device)
        start = pop.v.detach().clone().expand(B, N) if y0 is None else \
            torch.as_tensor(y0).to(device=dev, dtype=torch.float32).reshape(B, N)
        w_in = None
        if pre:  # input nodes / edges act on every (b, t) row alike
            if len(pre) == 1 and self.get_node(pre[0]).name == "identity" and X.shape[2] <= 64 \
                    and type(self.get_edge(pre[0], label)) is Linear:
                w_in = self.get_edge(pre[0], label).weights  # the projection is folded into the step kernel (RP_X_PROJ)
            else:
                X = self._apply_chain(pre, X.reshape(B * T, -1), None, label).reshape(B, T, -1)
        if w_in is None:
            self._fit_width(X.reshape(B * T, -1), N)
        # sampling as `run` does it for this topology (_run_fused): behind identity read-out nodes the buffer mean is the
        # step kernel's, taken on the population's output and read out afterwards; a non-linear read-out is applied to
        # every step and averaged afterwards
        linear_post = self._memoryless_linear(post, label)
        width = X.shape[2]
        outputs, finals = [None] * B, torch.empty(B, N, **f32)
        for b0 in range(0, B, BatchedPopulation.SLOTS):  # groups of at most 32 trajectories
            nb = min(BatchedPopulation.SLOTS, B - b0)
            on_valu = nb <= (8 if exact else 4) and not mfma
            size = next(q for q in (2, 4, 8) if q >= nb) if on_valu else nb
            bp = BatchedPopulation(pop, size, mfma=not on_valu)
            bp.v.zero_()
            bp.v[:nb] = start[b0:b0 + nb]
            bp.refresh()
            bp.reset_sampling()
            chunk = max(1, min(T, _WINDOW_BYTES // (4 * size * max(width, N if not linear_post else 1))))
            samples = torch.empty(size, S, N, **f32) if linear_post else None
            per_step = None if linear_post else [[] for _ in range(nb)]
            idx = 0
            for a in range(0, T, chunk):
                e = min(a + chunk, T)
                xb = torch.zeros(e - a, size, width, **f32)
                xb[:, :nb] = X[b0:b0 + nb, a:e].transpose(0, 1)
                if linear_post:
                    idx = bp.forward(e - a, x=xb, w_in=w_in, out_var=rnn._out_code, step0=a, cutoff=cutoff,
                                     sampling_steps=sampling_steps, samples=samples, sample_idx=idx)
                else:
                    traj = torch.empty(e - a, size, N, **f32)
                    bp.forward(e - a, x=xb, w_in=w_in, traj=traj, out_var=rnn._out_code)
                    for b in range(nb):
                        per_step[b].append(self._apply_chain(post, traj[:, b], label, None))
                if verbose:
                    print(f"Progress: trajectories {b0}..{b0 + nb - 1}, {e}/{T} integration steps finished.")
            finals[b0:b0 + nb] = bp.v[:nb]
            for b in range(nb):
                if linear_post:  # read out per trajectory: the same [S, N] product `run` computes
                    outputs[b0 + b] = self._apply_chain(post, samples[b], label, None) if post else samples[b]
                else:
                    outputs[b0 + b] = self._buffer_means(torch.cat(per_step[b]), sample_at, cutoff)
        observers = []
        for b in range(B):
            obs = Observer(dt=self.dt, record_loss=False)
            obs.record_block(sample_at, outputs[b], 0.0, [])
            obs.save("state", finals[b].clone())
            observers.append(obs)
        return observers

    # ---- several populations without cycles: one fused window per population ---------------------
    def _dag_plan(self, rec_vars) -> Optional[List[int]]:
        """Record slots of `rec_vars` when the (acyclic) graph can be evaluated window by window -- every node a
        compiled population or an element-wise activation, every edge a plain Linear/RLS projection -- else None.
        Inside one step a node only sees its predecessors' outputs of the SAME step (forward(), reference
        network.py:462-478), so node n's whole window follows from its predecessors' whole windows."""
        def no(why):
            self._why_not = dict(getattr(self, "_why_not", {}), dag=why)
            return None

        if getattr(self, "_force_stepwise", False):
            return no(None)
        if not any(isinstance(self.get_node(n), RateNet) for n in self._order):
            return no("the graph holds no compiled population")
        for n in self._order:
            node = self.get_node(n)
            if not isinstance(node, RateNet) and not (isinstance(node, InstantNode) and node.name in InstantNode.batched):
                return no(f"node '{n}' ({type(node).__name__}) is neither a compiled population nor a built-in activation node")
            if not getattr(node, "_fusable", True):
                return no(f"population '{n}' cannot be integrated as a fused window")
        for s_, t_ in self.graph.edges:
            if not self._windowable(self.get_edge(s_, t_)):
                return no(f"edge '{s_}' -> '{t_}' of class {type(self.get_edge(s_, t_)).__name__} cannot be evaluated for a whole window")
        slots = []
        for label, var in rec_vars:
            node = self.get_node(label)
            one = self._record_slots(node, [(label, var)]) if isinstance(node, RateNet) else None
            if one is None:
                return no(f"recorded variable '{var}' of node '{label}' is not recordable inside the step kernel")
            slots.append(one[0])
        no(None)
        return slots

    def _warn_stepwise(self, what: str, *plans: str):
        """One PerformanceWarning per network and cause: `what` runs on the per-step Python walk because none of the fused
        plans took the graph (`plans`: the keys of `_why_not` that were tried)."""
        if getattr(self, "_force_stepwise", False) or getattr(self, "_force_stepwise_adjoint", False):
            return  # (test seam: the walk was asked for)
        why = [w for w in (getattr(self, "_why_not", {}).get(k) for k in plans) if w]
        msg = (f"{what} walks the graph step by step from Python instead of running fused windows inside the library "
               f"(several times slower, same results)" + (": " + "; ".join(dict.fromkeys(why)) if why else "") + ".")
        seen = self.__dict__.setdefault("_warned", set())
        if msg not in seen:
            seen.add(msg)
            import warnings
            warnings.warn(msg, PerformanceWarning, stacklevel=3)

    def _window_outputs(self, inputs_ab: torch.Tensor, a: int, grad: bool, rec: dict, cutoff: int, sampling_steps: int):
        """Per-step output of every node over the window of steps [a, a + len): nodes in topological order, each
        population as one fused window.  `rec`: label -> {slot: buffer, "idx": next sample row} (no-grad recording in
        the step kernel) or label -> {slot: list} (grad mode: rows derived from the captured window)."""
        steps = inputs_ab.shape[0]
        outs = {}
        for n in self._order:
            preds = list(self.graph.predecessors(n))
            if preds:
                X = None
                for p_ in preds:
                    contrib = self._edge_window(self.get_edge(p_, n), outs[p_])
                    X = contrib if X is None else X + contrib
            else:
                X = inputs_ab
            node = self.get_node(n)
            if not isinstance(node, RateNet):
                outs[n] = node.forward(X)
                continue
            pop = node._pop
            X = self._fit_width(X, pop.rows)
            r = rec.get(n)
            needs = grad and (X.requires_grad or node._wants_grad(None) or any(q.requires_grad for q in node.parameters()))
            if needs or (grad and r):  # (a recorded population goes through the capturing window in a gradient run)
                outs[n] = node._grad_window(steps, X)
                if r:
                    window = tuple(node._last_window) + (None, None)
                    for slot, rows in r.items():
                        src = window[slot]
                        if isinstance(node, SpikeResetNet):
                            rows.append(src)
                        else:
                            last = pop.v if slot == 0 else pop.source_local
                            rows.append(torch.cat((src[1:], last.detach().reshape(1, -1))))
            else:
                traj = torch.empty(steps, pop.rows, dtype=torch.float32, device=pop.device)
                kw = {}
                if r and not grad:
                    kw = dict(rec={k_: v_ for k_, v_ in r.items() if k_ != "idx"}, cutoff=cutoff,
                              sampling_steps=sampling_steps, sample_idx=r["idx"])
                idx = node.run_window(steps, x=X.detach(), traj=traj, step0=a, **kw)
                if r:
                    r["idx"] = idx
                outs[n] = traj
        return outs

    def _run_windows(self, inputs, sampling_steps, cutoff, truncate_steps, obs: Observer, rec_vars, rec_slots, verbose):
        T = inputs.shape[0]
        sample_at = self._sample_steps(T, cutoff, sampling_steps)
        S = len(sample_at)
        grad = torch.is_grad_enabled() and (inputs.requires_grad or any(p_.requires_grad for p_ in self.parameters()) or
                                            any(self.get_node(n)._wants_grad(None) for n in self._order
                                                if isinstance(self.get_node(n), RateNet)))
        widest = max(self.get_node(n)._pop.rows for n in self._order if isinstance(self.get_node(n), RateNet))
        chunk = truncate_steps if grad else max(1, min(T, _WINDOW_BYTES // (4 * widest * max(1, len(self._order)))))
        rec = {}
        for (label, _), slot in zip(rec_vars, rec_slots):
            pop = self.get_node(label)._pop
            entry = rec.setdefault(label, {} if grad else {"idx": 0})
            if slot not in entry:
                entry[slot] = [] if grad else torch.empty(S, pop.rows, dtype=torch.float32, device=pop.device)
            if not grad:
                pop.reset_sampling()
        pending = []
        for a in range(0, T, chunk):
            b = min(a + chunk, T)
            outs = self._window_outputs(inputs[a:b], a, grad, rec, cutoff, sampling_steps)
            pending.append(outs[self._out_node])
            if grad and truncate_steps < T and b - a == truncate_steps:
                self.detach()
            if verbose:
                print(f"Progress: {b}/{T} integration steps finished.")
        per_step = torch.cat(pending)
        outputs = self._buffer_means(per_step, sample_at, cutoff)  # mean over the steps buffered since the previous sample
        rec_out = []
        for (label, _), slot in zip(rec_vars, rec_slots):
            rec_out.append(torch.cat(rec[label][slot])[sample_at] if grad else rec[label][slot])
        obs.record_block(sample_at, outputs, 0.0, rec_out)

    def _record_slots(self, rnn: RateNet, rec_vars) -> Optional[List[int]]:
        """Kernel record slot of each requested variable of the population (0: first state variable,
        1: source r / s), or None if one of them is not recordable in-kernel."""
        slots = []
        state_names = [s for s, _ in rnn._template.state]
        for _, var in rec_vars:
            name = self._relabel_var(var)
            first = rnn._template.reset_var if rnn._template.spike_var else state_names[0]
            if name == first:
                slots.append(0)
            elif name == rnn._template.source_var:
                slots.append(1)
            elif name == getattr(rnn, "_aux_name", None):
                slots.append(2)
            elif name == getattr(rnn, "_aux2_name", None):
                slots.append(3)
            else:
                return None
        return slots

    def _wants_grad(self, plan, inputs) -> bool:
        if not torch.is_grad_enabled():
            return False
        if inputs.requires_grad or any(p.requires_grad for p in self.parameters()):
            return True
        return self.get_node(plan[1])._wants_grad(None)

    def _run_fused(self, plan, inputs, sampling_steps, cutoff, truncate_steps, obs: Observer, slots, verbose):
        pre, label, post = plan
        rnn: RateNet = self.get_node(label)
        pop = rnn._pop
        N, dev, T = pop.rows, pop.device, inputs.shape[0]
        sample_at = self._sample_steps(T, cutoff, sampling_steps)
        S = len(sample_at)
        exit_to = post[0] if post else None
        linear_post = self._memoryless_linear(post, label)
        grad = self._wants_grad(plan, inputs)
        spiking = isinstance(rnn, SpikeResetNet)

        if grad:
            # autograd path: the population's window is one autograd node, everything around it is torch
            outs, recs = [], [[] for _ in slots]
            for a in range(0, T, truncate_steps):
                b = min(a + truncate_steps, T)
                X = self._apply_chain(pre, inputs[a:b], None, label) if pre else inputs[a:b]
                traj = rnn._grad_window(b - a, self._fit_width(X, N))
                outs.append(self._apply_chain(post, traj, label, None) if post else traj)
                if slots:
                    window = tuple(rnn._last_window) + (None, None)  # (v, source, aux, aux2) of the window
                    for j, slot in enumerate(slots):
                        src = window[slot]
                        if spiking:  # SpikeResetNet exposes the pre-update state (nodes.py:387-392)
                            recs[j].append(src)
                        else:  # post-update values: row t+1, last row = current state
                            last = pop.v if slot == 0 else pop.source_local
                            recs[j].append(torch.cat((src[1:], last.detach().reshape(1, -1))))
                if truncate_steps < T and b - a == truncate_steps:
                    self.detach()
            per_step = torch.cat(outs)
            outputs = self._buffer_means(per_step, sample_at, cutoff)  # (network.py:595)
            rec_out = [torch.cat(r)[sample_at] for r in recs]
            obs.record_block(sample_at, outputs, 0.0, rec_out)
            return

        # no-grad path: sampling happens inside the step kernel
        w_in, X, x_chunked = None, None, False
        if pre:
            if len(pre) == 1 and self.get_node(pre[0]).name == "identity" and inputs.shape[1] <= 64 \
                    and type(self.get_edge(pre[0], label)) is Linear:
                w_in, X = self.get_edge(pre[0], label).weights, inputs  # projection folded into the kernel
            else:
                x_chunked = True
        else:
            X = self._fit_width(inputs, N)
        pop.reset_sampling()
        f32 = dict(dtype=torch.float32, device=dev)
        rec_bufs = {slot: torch.empty(S, N, **f32) for slot in set(slots)}
        chunk = max(1, min(T, _WINDOW_BYTES // (4 * N)))
        if linear_post:
            samples = torch.empty(S, N, **f32)
            idx = 0
            for a in range(0, T, chunk):
                b = min(a + chunk, T)
                Xc = self._fit_width(self._apply_chain(pre, inputs[a:b], None, label), N) if x_chunked else \
                    (None if X is None else X[a:b])
                idx = rnn.run_window(b - a, x=Xc, w_in=w_in, step0=a, cutoff=cutoff, sampling_steps=sampling_steps,
                                     samples=samples, rec=rec_bufs, sample_idx=idx)
                if verbose:
                    print(f"Progress: {b}/{T} integration steps finished.")
            outputs = self._apply_chain(post, samples, label, None) if post else samples
        else:
            # a non-linear readout does not commute with the buffer mean: read out every step, then average
            rows, pending, start, idx = [], [], cutoff, 0
            for a in range(0, T, chunk):
                b = min(a + chunk, T)
                Xc = self._fit_width(self._apply_chain(pre, inputs[a:b], None, label), N) if x_chunked else \
                    (None if X is None else X[a:b])
                traj = torch.empty(b - a, N, **f32)
                idx = rnn.run_window(b - a, x=Xc, w_in=w_in, step0=a, cutoff=cutoff, sampling_steps=sampling_steps,
                                     traj=traj, rec=rec_bufs, sample_idx=idx)
                pending.append(self._apply_chain(post, traj, label, None))
            per_step = torch.cat(pending)
            outputs = self._buffer_means(per_step, sample_at, cutoff)
        obs.record_block(sample_at, outputs, 0.0, [rec_bufs[s] for s in slots])

    @staticmethod
    def _fit_width(X: Optional[torch.Tensor], n: int) -> Optional[torch.Tensor]:
        if X is not None and X.shape[1] not in (1, n):
            raise RuntimeError(f"The size of the input ({X.shape[1]}) must match the number of neurons ({n}) or be 1")
        return X

    def _run_stepwise(self, inputs, sampling_steps, cutoff, truncate_steps, obs, rec_vars, enable_grad, verbose):
        """Generic topologies: the reference's loop (network.py:586-599), one graph walk per step."""
        steps = inputs.shape[0]
        buffer = []
        with (torch.enable_grad() if enable_grad else torch.no_grad()):
            for step in range(steps):
                output = self.forward(inputs[step, :])
                if step >= cutoff:
                    buffer.append(output)
                    if step % sampling_steps == 0:
                        if verbose:
                            print(f"Progress: {step}/{steps} integration steps finished.")
                        obs.record(step, self._seq_mean(torch.stack(buffer, dim=0)), 0.0,
                                   [self.get_var(v[0], v[1]) for v in rec_vars])
                        buffer = []
                if truncate_steps < steps and step % truncate_steps == truncate_steps - 1:
                    self.detach()

    # ---- training -------------------------------------------------------------------------------
    def fit_bptt(self, inputs: Union[np.ndarray, list], targets: Union[np.ndarray, list], optimizer: str = "sgd",
                 optimizer_kwargs: dict = None, loss: str = "mse", loss_kwargs: dict = None, lr: float = 1e-3,
                 sampling_steps: int = 1, update_steps: int = 100, verbose: bool = True, **kwargs) -> Observer:
        """Backpropagation through time (signature of reference network.py:603-707).

        array inputs: windows of `update_steps` steps -- forward window, loss on the window's predictions,
        ``zero_grad / backward / step``, detach (network.py:1028-1040).  list inputs: one update per epoch
        on the sampled outputs of ``run`` and a state reset afterwards (network.py:983-1014)."""
        self.compile()
        loss_fn = self._get_loss_function(loss, loss_kwargs)
        opt = self._get_optimizer(optimizer, lr, self.parameters(), optimizer_kwargs)
        step_kwargs = retrieve_from_dict(["closure"], kwargs)
        error_kwargs = retrieve_from_dict(["retain_graph"], kwargs)
        obs = Observer(dt=self.dt, **retrieve_from_dict(["record_output", "record_loss", "record_vars"], kwargs))
        t0 = perf_counter()
        if isinstance(inputs, list):
            if len(inputs) != len(targets):
                raise ValueError("Wrong dimensions of input and target output. Please make sure that `inputs` and "
                                 "`targets` agree in the first dimension (epochs).")
            obs = self._bptt_epochs(inputs, targets, loss_fn, opt, obs, error_kwargs, step_kwargs, sampling_steps,
                                    verbose, **kwargs)
        else:
            inp = self._to_input(inputs)
            tgt = torch.as_tensor(targets).to(device=inp.device, dtype=torch.float32)
            if inp.shape[0] != tgt.shape[0]:
                raise ValueError("Wrong dimensions of input and target output. Please make sure that `inputs` and "
                                 "`targets` agree in the first dimension.")
            obs = self._bptt(inp, tgt, loss_fn, opt, obs, error_kwargs, step_kwargs, sampling_steps, update_steps,
                             verbose)
        print(f"Finished optimization after {perf_counter() - t0} s.")
        return obs

    @staticmethod
    def _bptt_step(predictions, targets, optimizer, loss, error_kwargs, step_kwargs, lazy: bool = False, unreached=()):
        """One optimisation step (reference network.py:1123-1130).  ``lazy``: return the loss as a 0-dim device tensor
        instead of a float, so that the window loop does not synchronise with the GPU once per window (the Observer
        converts on read).  ``unreached``: parameters no prediction of this window depends on (_unreached_parameters):
        their gradient is None, as the reference's step-by-step graph leaves it, not the zeros a fused window returns."""
        error = loss(predictions, targets)
        optimizer.zero_grad()
        error.backward(**error_kwargs)
        for p in unreached:
            p.grad = None
        optimizer.step(**step_kwargs)
        return error.detach() if lazy else error.item()

    @staticmethod
    def _fused_sgd_plan(optimizer, step_kwargs) -> Optional[dict]:
        """id(parameter) -> its learning rate summed over its occurrences in the optimiser's groups, when the optimiser is PLAIN
        stochastic gradient descent (torch.optim.SGD without momentum, dampening, weight decay, Nesterov, maximize; no closure):
        p <- p - lr g per occurrence.  A coupled window then applies the update of a trained matrix in the gradient GEMM's epilogue
        (W += -lr * dW: one read and one write of W) instead of storing dW and leaving three more passes over the matrix to the
        optimiser -- which skips a parameter whose gradient is None.  The same update up to the rounding of the last bit (the
        occurrences of a repeated parameter, network.py:1308-1322, are one addition instead of one after the other; lr * dW is
        rounded once inside the GEMM instead of dW and lr * dW in turn).  Not observable from outside fit_bptt: the call's last
        complete window takes the optimiser's route, so every parameter carries that window's gradient afterwards, as in the
        reference.  RECTIPY_AMD_FUSED_SGD=0 keeps the optimiser's passes in every window."""
        import os
        if os.environ.get("RECTIPY_AMD_FUSED_SGD", "1") == "0" or type(optimizer) is not torch.optim.SGD or step_kwargs:
            return None
        plan = {}
        for g in optimizer.param_groups:
            if g.get("momentum", 0) != 0 or g.get("dampening", 0) != 0 or g.get("weight_decay", 0) != 0 or g.get("nesterov") \
                    or g.get("maximize"):
                return None
            for p in g["params"]:
                plan[id(p)] = plan.get(id(p), 0.0) + float(g["lr"])
        return plan

    def _unreached_parameters(self, steps: int) -> list:
        """Trainable parameters that cannot influence any prediction of a window of `steps` steps that starts from
        states without an older graph.  A population hands on its state from BEFORE the step (nodes.py:166-170), so every
        population between a quantity and the output node delays its influence by one step: a population's own
        parameters and the weights of an edge into it first show in the output `c` steps later, c = the populations on
        the shortest path from it (inclusive) to the output.  The reference's autograd graph simply has no path to such a
        parameter and leaves ``grad`` None, which a stateful optimiser (Adam's moments, momentum, weight decay) treats
        differently from a zero gradient; the fused window's adjoint returns zeros, so the step sets them back to None.
        Edges with memory delay further and are left alone (zeros), as are windows whose start state carries a graph."""
        from .edge_variants import LinearMasked, _Stateful
        # (once per fit call and window length: the graph and the trainable flags do not change inside fit_bptt's window loop --
        # only whether a start state carries a graph does, which is looked at every time)
        memo = getattr(self, "_unreached_memo", None)
        for n in (memo["nodes"] if memo is not None else self.graph.nodes):
            node = n if memo is not None else self.get_node(n)
            handle = getattr(node, "_y_grad", None)
            if handle is not None and handle.grad_fn is not None:
                return []
        if memo is not None and steps in memo["by_steps"]:
            return memo["by_steps"][steps]
        if any(isinstance(self.get_edge(s, t), _Stateful) and not isinstance(self.get_edge(s, t), LinearMasked)
               for s, t in self.graph.edges):
            if memo is not None:
                memo["by_steps"][steps] = []
            return []
        count = {}
        for n in reversed(self._order):
            nxt = [count[m] for m in self.graph.successors(n) if m in count]
            if n != self._out_node and not nxt:
                continue  # not on a path to the output node
            count[n] = int(self[n]["node_type"] == "diff_eq") + (min(nxt) if nxt and n != self._out_node else 0)
        out = []
        for n, c in count.items():
            if steps >= c + 1:
                continue
            if self[n]["node_type"] == "diff_eq":
                out.extend(p for p in self.get_node(n).parameters() if p.requires_grad)
            for m in self.graph.predecessors(n):
                out.extend(p for p in self.get_edge(m, n).parameters() if p.requires_grad)
        if memo is not None:
            memo["by_steps"][steps] = out
        return out

    def _bptt(self, inp, target, loss, optimizer, obs: Observer, error_kwargs, step_kwargs, sampling_steps,
              optim_steps, verbose) -> Observer:
        plan = self._chain_plan()
        rec_vars = list(obs.recorded_state_variables)
        steps = inp.shape[0]
        # recorded variables of THE population of a chain come out of the window the autograd node stores anyway
        slots = self._record_slots(self.get_node(plan[1]), rec_vars) \
            if rec_vars and plan is not None and all(v[0] == plan[1] for v in rec_vars) else None
        windows = plan is None and not rec_vars and self._dag_plan([]) is not None  # several populations, no cycles
        # populations joined by dense edges, cycles included (FeedbackNetwork): forward and adjoint one call per window each
        coupled = self._coupled_grad_plan() if plan is None and not windows and not rec_vars else None
        if (plan is None and not windows and coupled is None) or (rec_vars and slots is None):
            if rec_vars:
                self._why_not = dict(getattr(self, "_why_not", {}), rec="fit_bptt records state variables (record_vars)")
                self._warn_stepwise("Network.fit_bptt", "rec")
            else:
                self._warn_stepwise("Network.fit_bptt", "dag", "coupled_grad")
            return self._bptt_stepwise(inp, target, loss, optimizer, obs, error_kwargs, step_kwargs, sampling_steps,
                                       optim_steps, verbose)
        self._unreached_memo = dict(nodes=[self.get_node(n) for n in self.graph.nodes],
                                    diffeq=[self.get_node(n) for n in self.graph.nodes if self[n]["node_type"] == "diff_eq"], by_steps={})
        try:
            return self._bptt_windows(inp, target, loss, optimizer, obs, error_kwargs, step_kwargs, sampling_steps, optim_steps,
                                      verbose, plan, windows, coupled, slots or [])
        finally:
            self._unreached_memo = None
            self._set_fused_sgd(None)   # (an exception inside a window must not leave the plan to a later backward call)

    def _set_fused_sgd(self, plan: Optional[dict]) -> None:
        """Hand the plain-SGD plan (or None) to the window nodes that apply it: the coupled window reads it from the network, a
        rate population's own window from the node."""
        self._fused_sgd = plan
        memo = getattr(self, "_unreached_memo", None)   # (inside fit_bptt's window loop: the population nodes, looked up once)
        for node in (memo["diffeq"] if memo is not None else
                     [self.get_node(n) for n in self.graph.nodes if self[n]["node_type"] == "diff_eq"]):
            node._fused_sgd = plan

    def _bptt_windows(self, inp, target, loss, optimizer, obs, error_kwargs, step_kwargs, sampling_steps, optim_steps, verbose,
                      plan, windows, coupled, slots=()) -> Observer:
        steps, error = inp.shape[0], 0.0
        if not windows and coupled is None:
            pre, label, post = plan
            rnn: RateNet = self.get_node(label)
            N = rnn._pop.rows
        for a in range(0, steps, optim_steps):
            b = min(a + optim_steps, steps)
            unreached = self._unreached_parameters(b - a)
            if unreached and b - a == optim_steps and {id(p) for p in unreached} == {id(p) for p in self.parameters() if p.requires_grad} \
                    and not any(getattr(self.get_node(n), "_y_requires_grad", False) or
                                getattr(getattr(self.get_node(n), "_y_grad", None), "requires_grad", False)
                                for n in self.graph.nodes):
                # the window's loss depends on nothing trainable and on no tracked state: the reference's backward()
                # fails with exactly this message (network.py:1128)
                raise RuntimeError("element 0 of tensors does not require grad and does not have a grad_fn")
            # (a complete window of a coupled fit under plain SGD: the trained matrices are updated inside the backward call --
            # except in the fit's LAST complete window, which leaves its gradient on every parameter as the reference does)
            self._set_fused_sgd(self._fused_sgd_plan(optimizer, step_kwargs) if not error_kwargs and b + optim_steps <= steps
                                else None)
            with torch.enable_grad():
                if windows:
                    preds = self._window_outputs(inp[a:b], a, True, {}, 0, 1)[self._out_node]
                elif coupled is not None:
                    preds = self._coupled_grad_outputs(coupled, inp[a:b], a, fit=True)
                else:
                    X = self._apply_chain(pre, inp[a:b], None, label) if pre else inp[a:b]
                    # the adjoint sweep runs only if a gradient has to pass INTO the population: its own trainable
                    # parameters or something trainable upstream.  (The reference also tracks the state after
                    # `detach(requires_grad=True)`, network.py:496; inside fit_bptt that leaf is unreachable for the
                    # caller, so a read-out-only fit skips the sweep: same parameter updates at the forward's speed.)
                    into_rnn = X.requires_grad or any(p.requires_grad for p in rnn.parameters()) or bool(slots)
                    traj = rnn._grad_window(b - a, self._fit_width(X, N)) if into_rnn else \
                        self._nograd_traj(rnn, b - a, self._fit_width(X, N))
                    preds = self._apply_chain(post, traj, label, None) if post else traj
                    rec_rows = []
                    if slots:   # (as `run` with a gradient: the window the autograd node stored)
                        window = tuple(rnn._last_window) + (None, None)  # (v, source, aux, aux2)
                        for slot in slots:
                            src = window[slot]
                            if isinstance(rnn, SpikeResetNet):  # exposes the pre-update state (nodes.py:387-392)
                                rec_rows.append(src)
                            else:  # post-update values: row t + 1, last row = the state now
                                last = rnn._pop.v if slot == 0 else rnn._pop.source_local
                                rec_rows.append(torch.cat((src[1:], last.detach().reshape(1, -1))))
                prev_error = error
                if b - a == optim_steps:  # only complete windows trigger an update (network.py:1035)
                    error = self._bptt_step(preds, target[a:b], optimizer, loss, error_kwargs, step_kwargs,
                                            lazy=not verbose, unreached=unreached)
                    self._set_fused_sgd(None)
                    self.detach()
            sample_at = [t for t in range(a, b) if t % sampling_steps == 0]
            if sample_at:
                # the loss recorded at a step is that of the last COMPLETED window at that step (network.py:1043-1046)
                losses = [error if (t == b - 1 and b - a == optim_steps) else prev_error for t in sample_at]
                # (a strided view: indexing with a Python list builds an index tensor on the host and copies it to the device --
                # a blocking copy that waits for the whole window's kernels, once per window)
                obs.record_block(sample_at, preds.detach()[sample_at[0] - a::sampling_steps].clone(), losses,
                                 [r.detach()[sample_at[0] - a::sampling_steps].clone() for r in rec_rows] if slots else [])
                if verbose:
                    print(f"Progress: {sample_at[-1]}/{steps} training steps finished. Current loss: {error}.")
        return obs

    @staticmethod
    def _nograd_traj(rnn: RateNet, steps: int, X):
        traj = torch.empty(steps, rnn._pop.rows, dtype=torch.float32, device=rnn._pop.device)
        rnn.run_window(steps, x=X, traj=traj)
        return traj

    def _bptt_stepwise(self, inp, target, loss, optimizer, obs, error_kwargs, step_kwargs, sampling_steps,
                       optim_steps, verbose) -> Observer:
        rec_vars = list(obs.recorded_state_variables)
        steps, error, predictions, old_step = inp.shape[0], 0.0, [], 0
        for step in range(steps):
            if not predictions:
                unreached = self._unreached_parameters(optim_steps)  # (a one-step node returns zeros too)
            pred = self.forward(inp[step, :])
            predictions.append(pred)
            if step % optim_steps == optim_steps - 1:
                error = self._bptt_step(torch.stack(predictions), target[old_step:step + 1], optimizer, loss,
                                        error_kwargs, step_kwargs, unreached=unreached)
                self.detach()
                old_step = step + 1
                predictions.clear()
            if step % sampling_steps == 0:
                if verbose:
                    print(f"Progress: {step}/{steps} training steps finished. Current loss: {error}.")
                obs.record(step, pred.detach(), error, [self.get_var(v[0], v[1]) for v in rec_vars])
        return obs

    def _bptt_epochs(self, inp: list, target: list, loss, optimizer, obs, error_kwargs, step_kwargs,
                     sampling_steps: int = 1, verbose: bool = False, **kwargs) -> Observer:
        y0 = {k: v.detach().clone() for k, v in self.state.items()}
        epoch_losses = []
        for epoch in range(len(inp)):
            obs = self.run(inp[epoch], verbose=False, sampling_steps=sampling_steps, enable_grad=True, **kwargs)
            tgt = torch.as_tensor(target[epoch]).to(device=obs.stacked("out").device, dtype=torch.float32)
            epoch_loss = self._bptt_step(obs.stacked("out"), tgt, optimizer, loss, error_kwargs, step_kwargs)
            epoch_losses.append(epoch_loss)
            self.reset(y0)
            if verbose:
                print(f"Progress: {epoch + 1}/{len(inp)} training epochs finished.\nEpoch loss: {epoch_loss}.\n")
        obs.save("epoch_loss", epoch_losses)
        obs.save("epochs", np.arange(len(inp)))
        return obs

    def test(self, inputs: np.ndarray, targets: np.ndarray, loss: str = "mse", loss_kwargs: dict = None,
             sampling_steps: int = 100, verbose: bool = True, **kwargs) -> tuple:
        """Run with frozen parameters and evaluate the loss (reference network.py:898-944)."""
        loss_fn = self._get_loss_function(loss, loss_kwargs)
        obs = self.run(inputs=inputs, sampling_steps=sampling_steps, verbose=verbose, enable_grad=False, **kwargs)
        output = obs.stacked("out")
        tgt = torch.as_tensor(targets).to(device=output.device, dtype=output.dtype)
        return obs, loss_fn(output, tgt).item()

    def fit_ridge(self, inputs: np.ndarray, targets: np.ndarray, sampling_steps: int = 100, alpha: float = 1e-4,
                  verbose: bool = True, add_readout_node: bool = True, **kwargs) -> Observer:
        """Ridge-regression readout on the collected states (reference network.py:709-784).

        Accuracy: the states are collected in float32 (the engine's arithmetic; about 1e-6 off a float64 run of the same
        network), the Gram matrix and the solve are float64 (`_ridge_solve`).  Against the reference's float64 run the
        read-out weights agree to about 2e-3 of max |w_out| in well-conditioned fits (fixture G26: 400 samples for 64
        neurons, alpha = 0.1) and STAY within 5e-4 when there are only as many samples as neurons and alpha is small
        (fixture G39: 60 samples, 64 neurons, alpha = 1e-3, cond(X^T X + alpha I) up to 4e7 -- measured 4e-5 ... 1.3e-4;
        the reference's own float32 run is off by 7e-2 ... 1 there): the state rounding of 1e-6 is amplified by the
        inverse of the smallest regularised singular value, so a fit with alpha near zero on nearly collinear states has no
        float32-state answer at all -- raise alpha or the sample count."""
        if np.shape(inputs)[0] != np.shape(targets)[0]:
            raise ValueError("Wrong dimensions of input and target output. Please make sure that `inputs` and "
                             "`targets` agree in the first dimension.")
        self.compile()
        t0 = perf_counter()
        obs = self.run(inputs=inputs, sampling_steps=sampling_steps, verbose=verbose, enable_grad=False, **kwargs)
        print(f"Finished network state collection after {perf_counter() - t0} s.")
        t0 = perf_counter()
        X = obs.stacked("out")
        Y = torch.as_tensor(targets).to(device=X.device, dtype=X.dtype)
        if Y.shape[0] != X.shape[0]:
            Y = Y[obs["steps"]]
        w_out = self._ridge_solve(X, Y, alpha)
        y = X @ w_out
        print(f"Finished fitting of read-out weights after {perf_counter() - t0} s.")
        if add_readout_node:
            out_node = self._out_node
            self.add_func_node("readout", n=w_out.shape[1], activation_function="identity")
            self.add_edge(out_node, target="readout", weights=w_out.T)
        obs.save("y", y)
        obs.save("w_out", w_out)
        return obs

    @staticmethod
    def _ridge_solve(X: torch.Tensor, Y: torch.Tensor, alpha: float) -> torch.Tensor:
        """(X^T X + alpha I)^-1 X^T Y (reference network.py:765-768).  With fewer samples than neurons the identical
        dual form X^T (X X^T + alpha I)^-1 Y is solved instead: an S x S system in place of an N x N one.  Gram matrix
        and solve run in float64 (the Gram matrix of thousands of states is too ill-conditioned for float32); the
        weights come back in the dtype of X."""
        S, N = X.shape
        Xd, Yd = X.double(), Y.double()
        if S < N:
            eye = torch.eye(S, device=X.device, dtype=torch.float64)
            w = Xd.T @ torch.linalg.solve(Xd @ Xd.T + alpha * eye, Yd)
        else:
            eye = torch.eye(N, device=X.device, dtype=torch.float64)
            w = torch.linalg.solve(Xd.T @ Xd + alpha * eye, Xd.T @ Yd)
        return w.to(X.dtype)

    def fit_rls(self, inputs: Union[list, np.ndarray], targets: Union[list, np.ndarray], update_steps: int = 1,
                sampling_steps: int = 100, verbose: bool = True, **kwargs) -> Observer:
        """Recursive-least-squares training of the edge added with ``train="rls"`` (signature of reference
        network.py:786-856).  array inputs: one pass, an update every `update_steps` steps (network.py:1105-1113);
        list inputs: one pass per epoch with a state reset in between (network.py:1062-1081)."""
        self.compile()
        if not self._train_edge or not self.graph.has_edge(*self._train_edge):
            raise ValueError("fit_rls needs an edge added with train='rls'")
        obs = Observer(dt=self.dt, **retrieve_from_dict(["record_output", "record_loss", "record_vars"], kwargs))
        t0 = perf_counter()
        if isinstance(inputs, list):
            if len(inputs) != len(targets):
                raise ValueError("Wrong dimensions of input and target output. Please make sure that `inputs` and "
                                 "`targets` agree in the first dimension (epochs).")
            y0 = {k: v.detach().clone() for k, v in self.state.items()}
            epoch_losses = []
            for epoch in range(len(inputs)):
                inp = self._to_input(inputs[epoch])
                tgt = torch.as_tensor(targets[epoch]).to(device=inp.device, dtype=torch.float32)
                _, loss = self._rls_pass(inp, tgt, update_steps, [])
                epoch_losses.append(float(loss[-1]) if loss.numel() else 0.0)
                self.reset(y0)
                if verbose:
                    print(f"Progress: {epoch + 1}/{len(inputs)} training epochs finished.\nEpoch loss: {epoch_losses[-1]}.\n")
            obs.save("epoch_loss", epoch_losses)
            obs.save("epochs", np.arange(len(inputs)))
        else:
            if np.shape(inputs)[0] != np.shape(targets)[0]:
                raise ValueError("Wrong dimensions of input and target output. Please make sure that `inputs` and "
                                 "`targets` agree in the first dimension.")
            inp = self._to_input(inputs)
            tgt = torch.as_tensor(targets).to(device=inp.device, dtype=torch.float32)
            rec_vars = list(obs.recorded_state_variables)
            pred, loss = self._rls_pass(inp, tgt, update_steps, rec_vars, obs=obs, sampling_steps=sampling_steps,
                                        verbose=verbose)
            if pred is not None:  # fused pass: record the sampled steps in one block
                sample_at = [t for t in range(inp.shape[0]) if t % sampling_steps == 0]
                obs.record_block(sample_at, pred[sample_at], [float(v) for v in loss[sample_at].cpu()], [])
        print(f"Finished optimization after {perf_counter() - t0} s.")
        return obs

    def _rls_pass(self, inp, tgt, optim_steps, rec_vars, obs=None, sampling_steps=1, verbose=False):
        """One pass over `inp`.  Fused when the graph is a chain ... -> population -> ... -(RLS)-> output node with an
        identity output: the population integrates the whole window first (the readout does not feed back), then the
        RLS sweep runs over the stored edge inputs.  Returns (pred [T, k], loss [T]); (None, loss) after a stepwise pass."""
        from .rls import RLS
        src, dst = self._train_edge
        edge: RLS = self.get_edge(src, dst)
        order = self._order
        simple = all(self.graph.in_degree(n) <= 1 and self.graph.out_degree(n) <= 1 for n in order)
        diffeq = [n for n in order if self[n]["node_type"] == "diff_eq"]
        fused = (simple and len(diffeq) == 1 and isinstance(self.get_node(diffeq[0]), RateNet) and not rec_vars
                 and dst == self._out_node and getattr(self.get_node(dst), "name", None) == "identity"
                 and order.index(src) >= order.index(diffeq[0])
                 and all(isinstance(self.get_node(n), InstantNode) and self.get_node(n).name in InstantNode.batched
                         for n in order if n != diffeq[0])
                 and all(type(self.get_edge(s, t)) is Linear or (s, t) == (src, dst) for s, t in self.graph.edges))
        if fused:
            label = diffeq[0]
            rnn: RateNet = self.get_node(label)
            i = order.index(label)
            pre, mid = order[:i], order[i + 1:order.index(src) + 1]
            with torch.no_grad():
                X = self._apply_chain(pre, inp, None, label) if pre else inp
                traj = self._nograd_traj(rnn, inp.shape[0], self._fit_width(X, rnn._pop.rows))
                Xe = self._apply_chain(mid, traj, label, None) if mid else traj
                return edge.sweep(Xe, tgt, optim_steps)
        if (not rec_vars and dst == self._out_node and getattr(self.get_node(dst), "name", None) == "identity"
                and self._chain_plan() is None and self._dag_plan([]) is not None):
            # several populations without cycles: every node's whole window first (the read-out does not feed back),
            # then the RLS sweep over the stored inputs of the trained edge
            with torch.no_grad():
                outs = self._window_outputs(inp, 0, False, {}, 0, 1)
                return edge.sweep(outs[src], tgt, optim_steps)
        coupled = None
        if not rec_vars and dst == self._out_node and getattr(self.get_node(dst), "name", None) == "identity" \
                and self._chain_plan() is None:
            coupled = self._coupled_plan([])
        if coupled is not None and dst in coupled["post"] and src in coupled["pops"] + coupled["post"]:
            # populations joined by dense edges, cycles included (FeedbackNetwork): the read-out does not feed back, so all
            # populations integrate the whole input first (one rp_forward_coupled call per window), then the RLS sweep
            with torch.no_grad():
                return edge.sweep(self._coupled_node_outputs(coupled, inp, src), tgt, optim_steps)
        self._warn_stepwise("Network.fit_rls", "dag", "coupled")
        steps, loss, losses = inp.shape[0], 0.0, []
        with torch.no_grad():
            for step in range(steps):
                pred = self.forward(inp[step, :])
                if step % optim_steps == 0:
                    edge.update(self[src]["out"], tgt[step, :], self[dst]["out"])
                    loss = edge.loss
                losses.append(float(loss))
                if obs is not None and step % sampling_steps == 0:
                    if verbose:
                        print(f"Progress: {step}/{steps} training steps finished. Current loss: {loss}.")
                    obs.record(step, pred, loss, [self.get_var(v[0], v[1]) for v in rec_vars])
        return None, torch.tensor(losses)

    def fit_eprop(self, *args, **kwargs):
        raise NotImplementedError("Method is currently not implemented")  # as in the reference (network.py:896)

    # ---- helpers ------------------------------------------------------------------------------
    def _relabel_var(self, var: str) -> str:
        return self._var_map.get(var, var)

    @staticmethod
    def _get_optimizer(optimizer: str, lr: float, model_params: Iterator, optimizer_kwargs: dict = None):
        if optimizer not in _OPTIMIZERS:
            raise ValueError(f"Invalid optimizer choice '{optimizer}'. Valid options: {sorted(_OPTIMIZERS)}.")
        params = list(model_params)
        kwargs = dict(optimizer_kwargs or {})
        cls = _OPTIMIZERS[optimizer]
        if len({id(p) for p in params}) < len(params) and "foreach" not in kwargs and "fused" not in kwargs:
            # the same tensor twice in the list (FeedbackNetwork.parameters follows the reference, which repeats the nodes at
            # a feedback edge): torch then steps it twice per update.  Its multi-tensor path -- the default on a GPU -- would
            # run both steps inside ONE kernel, racing on the tensor; one parameter after the other is what the reference's
            # CPU runs do, and is deterministic.
            import inspect
            if "foreach" in inspect.signature(cls.__init__).parameters:
                kwargs["foreach"] = False
        return cls(params, lr=lr, **kwargs)

    @staticmethod
    def _get_loss_function(loss: str, loss_kwargs: dict = None) -> Callable:
        if loss not in _LOSSES:
            raise ValueError(f"Invalid loss function choice '{loss}'. Valid options: {sorted(_LOSSES)}.")
        return _LOSSES[loss](**(loss_kwargs or {}))
