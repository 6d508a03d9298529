"""``FeedbackNetwork``: a Network whose edges may be flagged ``feedback=True`` (reference
rectipy/network.py:1196-1357).  Feed-forward edges define the evaluation order and the input / output nodes; feedback
edges are added to a node's input without taking part in the ordering, so cycles through them are allowed.

What a feedback edge delivers follows the reference (`_edge_bwd`, network.py:1354-1357): ``source_node["out"]`` read at
the moment the target is evaluated -- the output slice of the source's state vector `y` as it stands then, NOT the
value the source returned from `forward`.  For a rate population that has not been stepped yet in this step (the usual
case: the source sits downstream of the target) this is its current state, i.e. what it is about to emit; for a
spiking population `y` lags one step (nodes.py:387-392), so it is the previous step's output; a source that was already
stepped in this step delivers its post-update state.  Pinned by fixture G28 (tests/golden/make_golden_topologies.py).
The reference cannot feed back from an activation node (``InstantNode["out"]`` is None -> TypeError); here such an edge
delivers the node's output of the previous step (zero before the first step) -- an extension.
Such networks are evaluated step by step; every diff-eq node still advances through the HIP step kernel."""
from typing import Optional

import torch
from networkx import DiGraph, topological_sort

from .edges import Linear
from .network import Network


class FeedbackNetwork(Network):

    def add_edge(self, source: str, target: str, weights=None, train: Optional[str] = None, feedback: bool = False,
                 edge_attrs: dict = None, **kwargs) -> Linear:
        kwargs.pop("dtype", None)
        attrs = dict(edge_attrs or {})
        attrs["feedback"] = bool(feedback)
        return super().add_edge(source, target, weights=weights, train=train, edge_attrs=attrs, **kwargs)

    def _ffwd_graph(self) -> DiGraph:
        g = DiGraph()
        g.add_nodes_from(self.graph.nodes)
        g.add_edges_from((s, t) for s, t, d in self.graph.edges(data=True) if not d.get("feedback"))
        return g

    def compile(self):
        g = self._ffwd_graph()
        in_nodes = [n for n in g.nodes if g.in_degree(n) == 0]
        out_nodes = [n for n in g.nodes if g.out_degree(n) == 0]
        if len(in_nodes) != 1:
            raise ValueError(f"Unable to identify the input node of the Network. Nodes that have no (feed-forward) "
                             f"input edges: {in_nodes}.")
        if len(out_nodes) != 1:
            raise ValueError(f"Unable to identify the output node of the Network. Nodes that have no (feed-forward) "
                             f"outgoing edges: {out_nodes}.")
        self._in_node, self._out_node = in_nodes[0], out_nodes[0]
        self._order = list(topological_sort(g))
        self._prev_out = {}

    def _chain_plan(self):
        if any(d.get("feedback") for _, _, d in self.graph.edges(data=True)):
            return None
        return super()._chain_plan()

    def _dag_plan(self, rec_vars):
        if any(d.get("feedback") for _, _, d in self.graph.edges(data=True)):
            return None  # a feedback edge needs the source's output of the previous step: step by step
        return super()._dag_plan(rec_vars)

    def _unreached_parameters(self, steps: int) -> list:
        if any(d.get("feedback") for _, _, d in self.graph.edges(data=True)):
            return []  # feedback edges open further paths to the output; the delay count of the base class does not hold
        return super()._unreached_parameters(steps)

    def forward(self, x) -> torch.Tensor:
        if not self._order:
            self.compile()
        new_out = {}
        for n in self._order:
            inp = None
            has_ffwd = False
            for p in self.graph.predecessors(n):
                fb = self.graph[p][n].get("feedback")
                if fb:
                    node_p = self.get_node(p)
                    if hasattr(node_p, "y"):
                        src = node_p["out"]  # the reference's `self.get_node(source)["out"]` (network.py:1355)
                    else:
                        src = self._prev_out.get(p)
                        if src is None:
                            continue  # nothing has been emitted yet: zero feedback
                else:
                    has_ffwd = True
                    src = new_out[p]
                contrib = self.get_edge(p, n).forward(src)
                inp = contrib if inp is None else inp + contrib
            if not has_ffwd:
                inp = x if inp is None else x + inp
            new_out[n] = self.get_node(n).forward(inp)
            self[n]["out"] = new_out[n]
        self._prev_out = {k: v.detach() for k, v in new_out.items()}
        return new_out[self._out_node]

    def reset(self, state: dict = None):
        super().reset(state)
        self._prev_out = {}
