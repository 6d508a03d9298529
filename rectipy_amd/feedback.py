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

How such a network is evaluated (r04): populations joined DIRECTLY by plain dense edges -- feed-forward or feedback, cycles
included -- are integrated as ONE library call per window (``rp_forward_coupled``, include/rectipy_hip.h): every Euler step
steps the populations in evaluation order, one launch each, and every edge matrix is streamed by the target's step kernel
next to its recurrent matrix (`_coupled_plan` / `_run_coupled`).  Activation nodes upstream of every population (input
layers) are evaluated for the whole window in advance, those downstream of every population (read-outs) afterwards.
Gradient-tracking runs of such a graph (`fit_bptt`, `run(enable_grad=True)`) record ONE autograd node per window over all
populations (`_CoupledWindow`: forward `rp_forward_coupled` with the windows stored, backward `rp_backward_coupled`).
Everything else -- an activation node BETWEEN populations, edges with memory inside the cycle, and with a gradient a
population-mean spiking model -- is walked step by step
(`forward`), where a plain dense edge into a population is still evaluated inside that population's step kernel whenever
no gradient has to pass through it."""
from typing import List, Optional

import torch
from networkx import DiGraph, ancestors, descendants, topological_sort

from . import _abi
from .edges import Linear
from .network import _WINDOW_BYTES, Network
from .nodes import InstantNode, RateNet, SpikeResetNet, _RateWindow
from .observer import Observer


class _CoupledWindow(torch.autograd.Function):
    """T Euler steps of SEVERAL populations joined by dense edges -- cycles through feedback edges included -- as ONE
    autograd node: forward = rp_forward_coupled with every member's window stored, backward = rp_backward_coupled (one
    library call for the whole sweep: per step the input gradients go back through the edge matrices into the sources'
    output gradients, then every member's own adjoint step; surrogate gradient for spiking members) + the MFMA GEMMs of
    the weight gradients.  Replaces autograd over the per-step graph walk (reference rectipy/network.py:1028-1040 with
    1322-1357).

    Members (r05): populations AND element-wise activation nodes between them (their adjoint step is gx = g_out * act'(out));
    edges between members may be plain, masked (the kernels stream weights * mask; the gradient is the plain one times the
    mask) or carry a delay / filter memory (the gradient passes through the memory: rp_bwd_member.edge_mem_kind).

    Inputs after (net, plan, steps, a): per member x | None and, for a population, its state handle (v of a rate population,
    the packed state of a spiking one), W and its per-neuron parameters (`_names`); then the edge matrices in plan order.
    Outputs: per member its output trajectory [steps, N] and, for a population, its final state.  What a feedback edge from a
    spiking population (or an activation node) reads at the window's FIRST step -- that member's output one step before the
    window -- is a constant of the window."""

    @staticmethod
    def _names(pop) -> tuple:
        from .spiking_autograd import TRAINABLE
        if pop.model == "instant":
            return ()
        if pop.spiking:
            return TRAINABLE[pop.model]
        return ("tau", "k", "eta") + (_RateWindow.ACT_PARAMS if pop.model == "li_sigmoid" else ())

    @staticmethod
    def _layout(net, plan):
        """Per member (index of its x among the inputs, names of its parameters, is a population); index of the first edge."""
        out, i = [], 4
        for p in plan["members"]:
            if p in plan["pops"]:
                names = _CoupledWindow._names(net.get_node(p)._pop)
                out.append((i, names, True))
                i += 3 + len(names)
            else:
                out.append((i, (), False))
                i += 1
        return out, i

    @staticmethod
    def forward(ctx, net, plan, steps, a, *flat):
        members = plan["members"]
        layout, _ = _CoupledWindow._layout(net, plan)
        stored, xin, widths, lag0 = {}, {}, {}, {}
        for p, (i, names, is_pop) in zip(members, layout):
            pop = net._member_pop(p)
            x = flat[i - 4]
            f32 = dict(dtype=torch.float32, device=pop.device)
            buf = lambda pop=pop, f32=f32: torch.empty(steps, pop.rows, **f32)
            if is_pop:
                req = {name: flat[i - 4 + 3 + j].requires_grad for j, name in enumerate(names)}
                need_acc = req.get("k", False) or pop.model in ("ik", "iku", "ik_biexp")
                stored[p] = dict(save_v=buf(), save_src=buf(), save_acc=buf() if need_acc else None)
                if pop.spiking:
                    stored[p].update(save_aux=buf() if pop.aux is not None else None, save_aux2=buf() if pop.aux2 is not None else None,
                                     save_in=buf() if plan["edges"][p] else None)  # (the input the adjoint of tau / C needs)
            else:
                stored[p] = dict(traj=buf())   # an activation node: its outputs are all its adjoint needs
            xin[p] = None if x is None else x.detach()
            widths[p] = None if x is None else x.shape[1]
        for p in {s_ for q in members for s_, mode in plan["edges"][q] if net._bwd_mode(plan, s_, mode) == _abi.RP_EDGE_LAG}:
            lag0[p] = net._exposed_output(plan, p).detach().clone()
        # the windows of what the edges with memory hand to their projections (u_t): their adjoint and weight gradients need them
        saved_u = {}
        for q in members:
            for s_, _ in plan["edges"][q]:
                edge = net.get_edge(s_, q)
                if getattr(edge, "_stage_kind", None):
                    n_s = net._member_pop(s_).rows
                    saved_u[(s_, q)] = torch.empty(steps, (n_s + 3) // 4 * 4, dtype=torch.float32, device=net._member_pop(q).device)
        # (autograd runs this method without gradient mode: what `detach(requires_grad=...)` asked of the states must survive
        # the bookkeeping of an un-recorded step)
        pops = plan["pops"]
        asked = {p: net.get_node(p)._y_requires_grad for p in pops}
        net._coupled_steps(plan, xin, steps, a, set(), {}, None, 0, 1, stored=stored, saved_u=saved_u)
        for p in pops:
            net.get_node(p)._y_requires_grad = asked[p]
        outs = []
        order = {p: i for i, p in enumerate(members)}
        # a feedback edge from a rate population that was already stepped read its output AFTER the step: for the window's
        # last step that is the final state -- kept for the adjoint (act'(r_final)) and the edge's weight gradient
        final = {}
        for p in members:
            for s_, mode in plan["edges"][p]:
                if mode == _abi.RP_EDGE_LIVE and s_ in pops and order[s_] < order[p] and s_ not in final:
                    src = net.get_node(s_)._pop
                    final[s_] = (src.v.clone(), src.source.clone())
        for p, (_, _, is_pop) in zip(members, layout):
            st = stored[p]
            if not is_pop:
                outs.append(st["traj"].clone())
                continue
            node = net.get_node(p)
            pop = node._pop
            out = st["save_src"] if node._out_code == _abi.RP_OUT_SOURCE else st["save_v"]
            if pop.spiking:
                node._last_window = (st["save_v"], st["save_src"], st["save_aux"], st["save_aux2"])
                outs += [out.clone(), node._pack(pop.v, pop.source_local, pop.aux, pop.aux2)]
            else:
                node._last_window = (st["save_v"], st["save_src"])
                outs += [out.clone(), pop.v.clone()]
        ctx.net, ctx.plan, ctx.steps, ctx.stored, ctx.widths, ctx.xin, ctx.lag0, ctx.final = net, plan, steps, stored, widths, xin, lag0, final
        ctx.saved_u = saved_u
        ctx.flat = flat   # (the leaves themselves: a plain-SGD fit applies a matrix's update in the gradient GEMM, see backward)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g):
        from .edge_variants import LinearMasked
        from .engine import backward_coupled
        net, plan, T, stored = ctx.net, ctx.plan, ctx.steps, ctx.stored
        members, pops = plan["members"], plan["pops"]
        index = {p: i for i, p in enumerate(members)}
        need = ctx.needs_input_grad
        layout, first_edge = _CoupledWindow._layout(net, plan)
        read = {s_ for p in members for s_, _ in plan["edges"][p]}
        fused = getattr(net, "_fused_sgd", None)   # id(leaf) -> summed learning rate (a plain-SGD fit_bptt: Network._fused_sgd_plan)
        specs, res = [], []
        gpos = 0   # position in g: a population has two outputs (trajectory, final state), an activation node one
        for m, (p, (i, names, is_pop)) in enumerate(zip(members, layout)):
            pop = net._member_pop(p)
            N = pop.rows
            f32 = dict(dtype=torch.float32, device=pop.device)
            g_traj = g[gpos]
            g_end = g[gpos + 1] if is_pop else None
            gpos += 2 if is_pop else 1
            st = stored[p]
            R = dict(gx=torch.empty(T, N, **f32) if need[i] or plan["edges"][p] or not is_pop else None)
            # the sweep adds what comes back through the edges into the output gradient of their sources: a private copy
            if p in read or not is_pop:
                R["g_out"] = torch.zeros(T, N, **f32) if g_traj is None else g_traj.detach().to(**f32).clone().contiguous()
            else:
                R["g_out"] = None if g_traj is None else g_traj.detach().to(**f32).contiguous()
            if not is_pop:
                kw = dict(instant=True, traj_r=st["traj"], grad_out=R["g_out"], grad_x=R["gx"])
            else:
                node = net.get_node(p)
                R.update(u=torch.empty(T, N, **f32), par={name: torch.zeros(N, **f32) for j, name in enumerate(names) if need[i + 3 + j]})
                kw = dict(traj_v=st["save_v"], traj_r=st["save_src"], grad_out=R["g_out"], grad_var=node._out_code, u_traj=R["u"],
                          traj_acc=st["save_acc"], grad_x=R["gx"])
                if pop.spiking:
                    tpl = node._template

                    def piece(name, g_end=g_end, node=node, N=N, f32=f32):
                        if g_end is None or name is None:
                            return torch.zeros(N, **f32)
                        lo, hi = getattr(node, "_internal_var_map", node._var_map)[name]
                        return g_end.detach().to(**f32)[lo:hi].clone().contiguous()

                    R["lam"], R["lam_s"] = piece(tpl.reset_var), piece(tpl.source_var)
                    R["lam_x"] = piece(node._aux_name) if node._aux_name else None
                    R["lam_x2"] = piece(node._aux2_name) if node._aux2_name else None
                    kw.update(lam=R["lam"], lam_s=R["lam_s"], slope=node._spike_slope, x_target=node._x_target,
                              x=st["save_in"] if st["save_in"] is not None else ctx.xin[p], traj_aux=st["save_aux"],
                              lam_x=R["lam_x"], traj_aux2=st["save_aux2"], lam_x2=R["lam_x2"], grad_params=R["par"])
                else:
                    R["lam"] = torch.zeros(N, **f32) if g_end is None else g_end.detach().to(**f32).clone().contiguous()
                    kw.update(lam=R["lam"], grad_tau=R["par"].get("tau"), grad_k=R["par"].get("k"), grad_eta=R["par"].get("eta"),
                              grad_params={k_: v_ for k_, v_ in R["par"].items() if k_ in _RateWindow.ACT_PARAMS})
            res.append(R)
            edges = []
            for s_, mode in plan["edges"][p]:
                edge = net.get_edge(s_, p)
                edges.append(dict(W=net._edge_matrix(edge).detach(), source=index[s_], mode=net._bwd_mode(plan, s_, mode),
                                  memory=getattr(edge, "_stage_kind", None), edge=edge))
            specs.append(dict(pop=pop, bwd=kw, final_src=ctx.final[p][1] if p in ctx.final else None, edges=edges))
        packed_t = net._packed_edges(plan, transposed=True)   # fixed sparse edges: W_e^T gx by packed row dots
        for m, p in enumerate(members):
            for spec, (s_, _) in zip(specs[m]["edges"], plan["edges"][p]):
                spec["packed_t"] = packed_t.get((s_, p)) if packed_t else None
        keep = backward_coupled(specs)
        grads = [None] * len(need)
        for m, (p, (i, names, is_pop)) in enumerate(zip(members, layout)):
            R = res[m]
            if need[i]:
                grads[i] = R["gx"].sum(dim=1, keepdim=True) if ctx.widths[p] == 1 else R["gx"]
            if not is_pop:
                continue
            node = net.get_node(p)
            pop = node._pop
            if need[i + 1]:
                grads[i + 1] = node._pack(R["lam"], R["lam_s"], R["lam_x"], R["lam_x2"]) if pop.spiking else R["lam"]
            if need[i + 2]:
                Wt = ctx.flat[i - 4 + 2]
                lr_total = fused.get(id(Wt)) if fused else None
                if lr_total is not None and Wt.is_contiguous() and Wt.dtype == torch.float32 and pop.packed is None:
                    # plain SGD: W <- W - lr (dt u^T r) inside the GEMM's epilogue (read + write of W) instead of a stored
                    # gradient matrix and the optimiser's three passes over it; the gradient stays None: the optimiser skips W
                    pop.outer_accum(-lr_total * pop.dt, R["u"], stored[p]["save_src"], Wt.detach())
                    torch.autograd.graph.increment_version(Wt)   # (an in-place write torch did not see: caches keyed on W._version)
                else:
                    grads[i + 2] = pop.outer_product(pop.dt, R["u"], stored[p]["save_src"])
            for j, name in enumerate(names):
                grads[i + 3 + j] = R["par"].get(name)
        e = first_edge
        for m, p in enumerate(members):
            pop = net._member_pop(p)
            for s_, mode in plan["edges"][p]:
                if need[e]:  # dW_e = sum_t gx_p(t) (x) (what the edge's projection read at step t)
                    edge = net.get_edge(s_, p)
                    src_pop = net._member_pop(s_)
                    if (s_, p) in ctx.saved_u:   # an edge with memory: the u_t the forward call saved
                        out_s = ctx.saved_u[(s_, p)]
                    else:
                        if s_ in pops:
                            src = net.get_node(s_)
                            out_s = stored[s_]["save_src"] if src._out_code == _abi.RP_OUT_SOURCE else stored[s_]["save_v"]
                        else:
                            out_s = stored[s_]["traj"]
                        bmode = net._bwd_mode(plan, s_, mode)
                        if bmode == _abi.RP_EDGE_LAG:  # the source's output one step earlier; before the window: its value then
                            out_s = torch.cat([ctx.lag0[s_][None, :], out_s[:-1]])
                        elif bmode == _abi.RP_EDGE_LIVE and index[s_] < m:  # ... one step LATER (a post-update read); the last: the final one
                            src = net.get_node(s_)
                            last = ctx.final[s_][1] if src._out_code == _abi.RP_OUT_SOURCE else ctx.final[s_][0]
                            out_s = torch.cat([out_s[1:], last[None, :]])
                    gx = res[m]["gx"]
                    We = ctx.flat[e - 4]
                    lr_total = fused.get(id(We)) if fused else None
                    if lr_total is not None and not isinstance(edge, LinearMasked) and We.dtype == torch.float32 and We.stride(1) == 1 \
                            and tuple(We.shape) == (pop.rows, src_pop.rows):
                        pop.backend.outer_accum_rect(pop.rows, src_pop.rows, T, -lr_total, gx.data_ptr(), gx.stride(0) if T > 1 else pop.rows,
                                                     out_s.data_ptr(), out_s.stride(0) if T > 1 else src_pop.rows, We.data_ptr(),
                                                     We.stride(0), pop.device)   # (the update itself: plain SGD, see above)
                        torch.autograd.graph.increment_version(We)
                    else:
                        gW = torch.empty(pop.rows, src_pop.rows, dtype=torch.float32, device=pop.device)
                        pop.backend.outer_store_rect(pop.rows, src_pop.rows, T, 1.0, gx.data_ptr(), gx.stride(0) if T > 1 else pop.rows,
                                                     out_s.data_ptr(), out_s.stride(0) if T > 1 else src_pop.rows, gW.data_ptr(),
                                                     gW.stride(0), pop.device)
                        if isinstance(edge, LinearMasked):
                            gW = gW * edge.mask
                        grads[e] = gW
                e += 1
        del keep
        return tuple(grads)


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

    def parameters(self, recurse: bool = True):
        """The reference yields the parameters of the feed-forward graph and then those of the feedback graph
        (network.py:1308-1322, 946-952): the nodes at either end of a feedback edge come TWICE, so an optimiser built from
        this list holds them twice and applies two steps per update to them (torch warns about the duplicates; with SGD
        their effective learning rate doubles).  Kept as it is: a training run then ends where the reference's does
        (fixture G37, tests/test_coupled.py)."""
        fb = [(s, t) for s, t, d in self.graph.edges(data=True) if d.get("feedback")]
        for n in self.graph.nodes:
            yield from self.get_node(n).parameters(recurse=recurse)
        for s, t, d in self.graph.edges(data=True):
            if not d.get("feedback"):
                yield from self.get_edge(s, t).parameters()
        ends = {n for e in fb for n in e}
        for n in self.graph.nodes:
            if n in ends:
                yield from self.get_node(n).parameters(recurse=recurse)
        for s, t in fb:
            yield from self.get_edge(s, t).parameters()

    def _chain_plan(self):
        if any(d.get("feedback") for _, _, d in self.graph.edges(data=True)):
            return None
        return super()._chain_plan()

    def _dag_plan(self, rec_vars):
        if any(d.get("feedback") for _, _, d in self.graph.edges(data=True)):
            return None  # a feedback edge needs the source's output of the previous step: step by step
        return super()._dag_plan(rec_vars)

    def _unreached_parameters(self, steps: int) -> list:
        """As the base class, with the timing of feedback reads: the parameters that cannot influence any prediction of a
        window of `steps` steps that starts from states without an older graph -- the reference's autograd leaves their
        ``grad`` None (a stateful optimiser then skips them), a fused window's adjoint returns zeros.

        V_n(t) = what node n emits at step t (a population: its state BEFORE the step).  V_p(t + 1) of a population depends
        on its inputs of step t: V_s(t) through a feed-forward edge or a feedback edge from a rate population stepped later
        (delay 1), V_s(t + 1) through a feedback edge from a rate population already stepped (delay 0), V_s(t - 1) through a
        feedback edge from a spiking population not yet stepped (its exposed state lags, nodes.py:387-392: delay 2; already
        stepped: 1) or from an activation node (its output of the step before: 2).  An activation node emits a function of
        its inputs of the same step (delay 0).  With D(n) the smallest total delay from V_n to the output node's V, a
        population's parameters and the weights of an edge into it act from step a on V(a + 1), i.e. on the prediction of
        step a + 1 + D: reached iff steps >= D + 2 (an edge into an activation node: D + 1)."""
        if not any(d.get("feedback") for _, _, d in self.graph.edges(data=True)):
            return super()._unreached_parameters(steps)
        from .edge_variants import LinearMasked, _Stateful
        if any(isinstance(self.get_edge(s, t), _Stateful) and not isinstance(self.get_edge(s, t), LinearMasked)
               for s, t in self.graph.edges):
            return []
        for n in self.graph.nodes:
            handle = getattr(self.get_node(n), "_y_grad", None)
            if handle is not None and handle.grad_fn is not None:
                return []
        order = {n: i for i, n in enumerate(self._order)}
        is_pop = {n: isinstance(self.get_node(n), RateNet) for n in self._order}
        delay = {}
        for s_, t_, d in self.graph.edges(data=True):
            if not is_pop[t_]:
                w = 1 if d.get("feedback") else 0   # (a feedback edge into an activation node: the step before)
            elif not d.get("feedback"):
                w = 1
            elif not is_pop[s_]:
                w = 2
            else:
                stepped = order[s_] < order[t_]
                w = (1 if stepped else 2) if isinstance(self.get_node(s_), SpikeResetNet) else (0 if stepped else 1)
            delay[(s_, t_)] = w
        inf = float("inf")
        dist = {n: inf for n in self._order}
        dist[self._out_node] = 0
        for _ in range(len(self._order)):  # Bellman-Ford on a handful of nodes
            for (s_, t_), w in delay.items():
                if dist[t_] + w < dist[s_]:
                    dist[s_] = dist[t_] + w
        out = []
        for n in self._order:
            if is_pop[n] and steps < dist[n] + 2:
                out.extend(p for p in self.get_node(n).parameters() if p.requires_grad)
        for (s_, t_) in delay:
            if steps < dist[t_] + (2 if is_pop[t_] else 1):
                out.extend(p for p in self.get_edge(s_, t_).parameters() if p.requires_grad)
        return out

    def forward(self, x) -> torch.Tensor:
        if not self._order:
            self.compile()
        if isinstance(x, torch.Tensor) and x.dtype != self.dtype:
            x = x.to(self.dtype)
        new_out = {}
        for n in self._order:
            node = self.get_node(n)
            has_ffwd = False
            feeds = []  # (edge, what it reads) in predecessor order
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
                feeds.append((self.get_edge(p, n), src))
            # plain dense edges into a population are evaluated by its step kernel -- unless a gradient has to pass
            # through the node's input (then every product is a torch op, as before)
            tracked = torch.is_grad_enabled() and (
                (not has_ffwd and isinstance(x, torch.Tensor) and x.requires_grad) or
                any(src.requires_grad or any(q.requires_grad for q in edge.parameters()) for edge, src in feeds) or
                (isinstance(node, RateNet) and node._wants_grad(None)))
            inp, in_kernel = None, []
            for edge, src in feeds:
                if not tracked and len(in_kernel) < _abi.RP_MAX_EDGES and self._edge_in_kernel(edge, node, src):
                    in_kernel.append((edge.weights, src.detach()))
                    continue
                contrib = edge.forward(src)
                inp = contrib if inp is None else inp + contrib
            if not has_ffwd:
                inp = x if inp is None else x + inp
            new_out[n] = node.forward(inp, edges=in_kernel) if in_kernel else node.forward(inp)
            self[n]["out"] = new_out[n]
        # (with their history: like the reference's node["out"], network.py:1349-1355, the gradient passes through a feedback
        # edge from an activation node too -- until `detach` ends the window)
        self._prev_out = dict(new_out)
        return new_out[self._out_node]

    @staticmethod
    def _edge_in_kernel(edge, node, src) -> bool:
        """A plain dense edge into a compiled population is evaluated by the population's step kernel (rp_run.edge_*)
        whenever no gradient has to pass through it: no product launch, no [n] temporary, W_e streamed next to W."""
        if type(edge) is not Linear or not isinstance(node, RateNet) or not getattr(node, "_fusable", True):
            return False
        W = edge.weights
        pop = node._pop
        return (W.dtype == torch.float32 and W.device == pop.device and W.stride(1) == 1 and W.shape[0] == pop.rows
                and src.dtype == torch.float32 and src.device == pop.device and src.dim() == 1 and src.is_contiguous()
                and src.shape[0] == W.shape[1] and (W.shape[1] % 4 or W.stride(0) % 4 or src.data_ptr() % 16 == 0))

    # ---- populations joined by dense edges: one library call per window ----------------------------------------------
    def _coupled_plan(self, rec_vars) -> Optional[dict]:
        """How to evaluate the graph with rp_forward_coupled, or None (then `_why_not["coupled"]` says why, and `run` /
        `fit_*` warn before they walk the graph step by step).  Populations (compiled, fusable) may be joined in any pattern,
        feedback edges included, by plain / masked projections and by the edges with memory (delay, filter: their memory is
        a stage the library advances once per step, rp_run.edge_mem); element-wise activation nodes BETWEEN populations are
        members of the call themselves (RP_MODEL_INSTANT); other activation nodes must sit entirely upstream of the
        populations ("pre": evaluated for the whole window in advance) or entirely downstream ("post": afterwards)."""
        plan, why = self._coupled_plan_why(rec_vars)
        self._why_not = dict(getattr(self, "_why_not", {}), coupled=why)
        return plan

    def _coupled_plan_why(self, rec_vars):
        from .edge_variants import LinearMasked, _Stateful
        from .rls import RLS
        if getattr(self, "_force_stepwise", False) or not self._order:
            return None, None
        g = self.graph
        pops = [n for n in self._order if isinstance(self.get_node(n), RateNet)]
        if not pops:
            return None, None
        for n in self._order:
            node = self.get_node(n)
            if isinstance(node, RateNet):
                if not getattr(node, "_fusable", True):
                    return None, f"population '{n}' cannot be integrated as a fused window"
                if node._pop.rows != node._pop.n:
                    return None, f"population '{n}' is a row shard"
            elif not (isinstance(node, InstantNode) and node.name in InstantNode.batched):
                return None, f"node '{n}' ({type(node).__name__}) is neither a compiled population nor a built-in activation node"
        below = set(pops).union(*(descendants(g, p) for p in pops))   # a population or downstream of one
        above = set(pops).union(*(ancestors(g, p) for p in pops))     # a population or upstream of one
        pre = [n for n in self._order if n not in below]
        post = [n for n in self._order if n in below and n not in above]
        between = [n for n in self._order if n in below and n in above and n not in pops]
        for n in between:  # an activation node between populations: a member of the call (out = act(node input), in-kernel)
            if self.get_node(n).name not in InstantNode.elementwise:
                return None, (f"activation node '{n}' ({self.get_node(n).name}) lies between populations and has no in-kernel form "
                              f"(element-wise {', '.join(InstantNode.elementwise)} do)")
        members = [n for n in self._order if n in pops or n in between]
        if len(members) > _abi.RP_MAX_MEMBERS:
            return None, f"{len(members)} coupled populations / activation nodes (at most {_abi.RP_MAX_MEMBERS} per library call)"
        pos = {n: i for i, n in enumerate(self._order)}
        edges = {m: [] for m in members}
        for s_, t_, d in g.edges(data=True):
            edge, fb = self.get_edge(s_, t_), bool(d.get("feedback"))
            if s_ in edges and t_ in edges:
                stateful = isinstance(edge, _Stateful) and not isinstance(edge, LinearMasked)
                if not (type(edge) in (Linear, RLS) or isinstance(edge, LinearMasked) or (stateful and edge._stage_kind)):
                    return None, f"edge '{s_}' -> '{t_}' of class {type(edge).__name__} has no in-kernel form"
                if edge.weights.stride(1) != 1 or edge.weights.dtype != torch.float32:
                    return None, f"edge '{s_}' -> '{t_}': its weights are not a float32 row-major matrix"
                if stateful and edge.memory_stage() is None:
                    return None, (f"edge '{s_}' -> '{t_}' ({type(edge).__name__}): its memory cannot be held by the library "
                                  f"(float32 tensors on the GPU; a delay buffer whose rows are identical)")
                src_node = self.get_node(s_)
                earlier = pos[s_] < pos[t_]
                if s_ in between:
                    # the library reads an activation node's output "as it stands": this step's once the node was stepped,
                    # else the previous step's -- the latter is what a feedback edge delivers (`_prev_out`, an extension:
                    # the reference raises there), the former what a feed-forward edge does
                    if fb and earlier:
                        return None, f"feedback edge '{s_}' -> '{t_}' from an activation node that is evaluated earlier in the step"
                    mode = _abi.RP_EDGE_LIVE if fb else _abi.RP_EDGE_OUT
                elif s_ == t_ and not isinstance(src_node, SpikeResetNet) and src_node._out_code == _abi.RP_OUT_STATE0:
                    return None, f"population '{s_}' reads its own state through an edge"  # (while it is stepped)
                elif not fb:
                    mode = _abi.RP_EDGE_OUT            # what the source's step of this Euler step returned
                elif isinstance(src_node, SpikeResetNet):
                    mode = _abi.RP_EDGE_OUT if earlier else _abi.RP_EDGE_LAG  # its exposed state lags (nodes.py:387-392)
                else:
                    mode = _abi.RP_EDGE_LIVE           # `source["out"]` as it stands (network.py:1354-1357)
                edges[t_].append((s_, mode))
            elif fb:
                return None, f"feedback edge '{s_}' -> '{t_}' from / into an activation node outside the coupled populations"
            elif not self._windowable(edge):
                return None, f"edge '{s_}' -> '{t_}' of class {type(edge).__name__} cannot be evaluated for a whole window"
        slots = []
        for label, var in rec_vars:
            node = self.get_node(label)
            one = self._record_slots(node, [(label, var)]) if isinstance(node, RateNet) else None
            if one is None:
                return None, f"recorded variable '{var}' of node '{label}' is not recordable inside the step kernel"
            slots.append(one[0])
        return dict(pre=pre, pops=pops, post=post, between=between, members=members, edges=edges, slots=slots), None

    def _coupled_grad_plan(self) -> Optional[dict]:
        """The coupled plan if its adjoint runs as one call too (rp_backward_coupled): feed-forward edges, feedback edges from
        rate populations (stepped later: the usual direction; stepped earlier: a post-update read) and from spiking populations
        (their exposed state lags one step), the population-mean spiking models included (their adjoint means are one more
        reduction launch per member and step).  Activation nodes between populations and edges with memory / masks between
        populations keep the per-step autograd (`_why_not["coupled_grad"]`)."""
        plan, why = self._coupled_grad_plan_why()
        self._why_not = dict(getattr(self, "_why_not", {}), coupled_grad=why)
        return plan

    def _coupled_grad_plan_why(self):
        from .spiking_autograd import TRAINABLE
        plan = self._coupled_plan([])
        if plan is None:
            return None, self._why_not.get("coupled")
        if getattr(self, "_force_stepwise_adjoint", False):
            return None, None
        from .edge_variants import LinearMasked
        pos = {n: i for i, n in enumerate(plan["members"])}
        for p in plan["pops"]:
            node = self.get_node(p)
            if node._out_slice is not None:
                return None, f"population '{p}' exposes a slice of its output variable"
            if node._pop.spiking and node._pop.model not in TRAINABLE:
                return None, f"population '{p}' ({node._pop.model}) has no surrogate-gradient adjoint"
        for p in plan["members"]:
            for s_, mode in plan["edges"][p]:
                edge = self.get_edge(s_, p)
                kind = getattr(edge, "_stage_kind", None)
                if not (type(edge) is Linear or isinstance(edge, LinearMasked) or kind in ("delay", "filter")):
                    return None, (f"edge '{s_}' -> '{p}' of class {type(edge).__name__} between populations: "
                                  f"its adjoint is not part of rp_backward_coupled")
                if kind and mode == _abi.RP_EDGE_LIVE and pos[s_] < pos[p]:
                    return None, f"edge '{s_}' -> '{p}' ({type(edge).__name__}) is a feedback edge from a population already stepped"
                if any(q.requires_grad for q in (getattr(edge, "filter", None),) if isinstance(q, torch.Tensor)):
                    return None, f"edge '{s_}' -> '{p}': a trained filter matrix"
        return plan, None

    def _bwd_mode(self, plan: dict, source: str, mode: int) -> int:
        """The read mode of an edge as rp_backward_coupled wants it: a feedback edge from an activation node delivered the
        node's output of the step BEFORE (the library read it "as it stands" before the node's step) -- RP_EDGE_LAG."""
        if source in plan.get("between", ()) and mode == _abi.RP_EDGE_LIVE:
            return _abi.RP_EDGE_LAG
        return mode

    def _exposed_output(self, plan: dict, label: str) -> torch.Tensor:
        """What a lagging read of `label` sees at a window's first step: a spiking population's exposed `y` slice; an
        activation node's previous output (zeros before its first step)."""
        if label in plan["pops"]:
            return self.get_node(label)["out"]
        prev = self._prev_out.get(label)
        pop = self._member_pop(label)
        return torch.zeros(pop.rows, dtype=torch.float32, device=pop.device) if prev is None else prev.reshape(-1)

    def _coupled_grad_outputs(self, plan: dict, inputs_ab: torch.Tensor, a: int, fit: bool = False) -> torch.Tensor:
        """The output node's per-step outputs over the window [a, a + len), with the graph of ONE autograd node for all
        populations (`_CoupledWindow`) between the torch ops of the input layers and the read-outs.  ``fit`` (fit_bptt): when
        nothing trainable lies in or upstream of the populations -- a read-out-only fit -- the populations integrate without
        a graph (the only other thing a gradient could reach is the state leaf of the window, which no caller of fit_bptt
        sees): same parameter updates at the forward's speed, packed sparse matrices included."""
        pops = plan["pops"]
        self._consolidate(plan)
        if fit and not inputs_ab.requires_grad:
            upstream = list(plan["pre"]) + list(pops)
            trainable = any(q.requires_grad for n in upstream for q in self.get_node(n).parameters()) or \
                any(q.requires_grad for s_, t_ in self.graph.edges if t_ in upstream for q in self.get_edge(s_, t_).parameters())
            carried = any(getattr(self.get_node(p), "_y_grad", None) is not None and self.get_node(p)._y_grad.grad_fn is not None
                          for p in pops)
            if not trainable and not carried:
                feeders = {p for p in pops if p == self._out_node or any(t_ in plan["post"] for t_ in self.graph.successors(p))}
                with torch.no_grad():
                    outs = dict(self._coupled_window(plan, inputs_ab, a, feeders, {}, None, 0, 1))
                for n in plan["post"]:
                    X = None
                    for p_ in self.graph.predecessors(n):
                        if p_ in outs:
                            contrib = self._edge_window(self.get_edge(p_, n), outs[p_])
                            X = contrib if X is None else X + contrib
                    outs[n] = self.get_node(n).forward(X)
                for n in self._order:
                    self[n]["out"] = None
                self._prev_out = {k: v for k, v in self._prev_out.items() if k in plan["between"]}
                return outs[self._out_node]
        xin = self._coupled_inputs(plan, inputs_ab)
        flat = []
        for p in plan["members"]:
            if p not in pops:   # an activation node between populations: a member without state or parameters
                flat.append(xin[p])
                continue
            node: RateNet = self.get_node(p)
            pop = node._pop
            state_in = node._y_grad
            if pop.spiking:
                if state_in is None:
                    state_in = node.current_state.detach().requires_grad_(node._y_requires_grad)
            else:
                node._before_step()
                state_in = node._y_grad
                if state_in is None:
                    state_in = pop.v.detach().clone().requires_grad_(node._y_requires_grad)
            flat += [xin[p], state_in, pop.W] + [pop.params[q] for q in _CoupledWindow._names(pop)]
        for p in plan["members"]:
            flat += [self.get_edge(s_, p).weights for s_, _ in plan["edges"][p]]
        res = _CoupledWindow.apply(self, plan, inputs_ab.shape[0], a, *flat)
        outs, i = {}, 0
        for p in plan["members"]:
            outs[p] = res[i]
            if p not in pops:
                i += 1
                continue
            node = self.get_node(p)
            node._y_grad = res[i + 1]
            i += 2
            if node._pop.spiking:  # `y` exposes the pre-update state of the window's last step (nodes.py:387-392)
                tv, ts, ta, tb = node._last_window
                node._y_prev = node._pack(tv[-1], ts[-1], None if ta is None else ta[-1], None if tb is None else tb[-1])
        for n in plan["post"]:
            X = None
            for p_ in self.graph.predecessors(n):
                contrib = self._edge_window(self.get_edge(p_, n), outs[p_])
                X = contrib if X is None else X + contrib
            outs[n] = self.get_node(n).forward(X)
        for n in self._order:
            self[n]["out"] = None
        self._prev_out = {k: v for k, v in self._prev_out.items() if k in plan["between"]}
        return outs[self._out_node]

    def _run_coupled_grad(self, plan: dict, inputs, sampling_steps, cutoff, truncate_steps, obs: Observer, verbose):
        """`run` with a gradient: windows of `truncate_steps` steps, each ONE autograd node over all populations."""
        T = inputs.shape[0]
        sample_at = self._sample_steps(T, cutoff, sampling_steps)
        pending = []
        for a in range(0, T, truncate_steps):
            b = min(a + truncate_steps, T)
            pending.append(self._coupled_grad_outputs(plan, inputs[a:b], a))
            if truncate_steps < T and b - a == truncate_steps:
                self.detach()
            if verbose:
                print(f"Progress: {b}/{T} integration steps finished.")
        obs.record_block(sample_at, self._buffer_means(torch.cat(pending), sample_at, cutoff), 0.0, [])

    def _coupled_wants_grad(self, inputs) -> bool:
        if not torch.is_grad_enabled():
            return False
        return bool(inputs.requires_grad or any(p.requires_grad for p in self.parameters()) or
                    any(self.get_node(n)._wants_grad(None) for n in self._order if isinstance(self.get_node(n), RateNet)))

    def _coupled_inputs(self, plan: dict, inputs_ab: torch.Tensor) -> dict:
        """What enters every population over the window from OUTSIDE the coupled populations: the external input and
        the activation nodes upstream, evaluated for the whole window at once (torch ops: a gradient passes)."""
        outs, xin = {}, {}
        for n in plan["pre"]:
            preds = list(self.graph.predecessors(n))
            X = inputs_ab if not preds else None
            for p_ in preds:
                contrib = self._edge_window(self.get_edge(p_, n), outs[p_])
                X = contrib if X is None else X + contrib
            outs[n] = self.get_node(n).forward(X)
        for p in plan["members"]:
            X = inputs_ab if p == self._in_node else None
            for p_ in self.graph.predecessors(p):
                if p_ in outs and not self.graph[p_][p].get("feedback"):
                    contrib = self._edge_window(self.get_edge(p_, p), outs[p_])
                    X = contrib if X is None else X + contrib
            xin[p] = None if X is None else self._fit_width(X, self._member_pop(p).rows)
        return xin

    def _member_pop(self, label: str):
        """The engine object of a member of the coupled call: a population's, or -- for an activation node between
        populations -- an InstantPopulation created on first use (its `source` is the node's last output)."""
        node = self.get_node(label)
        if isinstance(node, RateNet):
            return node._pop
        ipop = getattr(node, "_ipop", None)
        if ipop is None:
            from .engine import InstantPopulation
            dev = self.device if self._backend is None else "cpu"
            ipop = node._ipop = InstantPopulation(node.n_in, node.name, dev, backend=self._backend)
        return ipop

    @staticmethod
    def _edge_matrix(edge) -> torch.Tensor:
        """The matrix the step kernel streams for an edge between members: its weights (a masked edge: weights * mask)."""
        from .edge_variants import LinearMasked
        if isinstance(edge, LinearMasked):
            return (edge.weights.detach() * edge.mask).contiguous()
        return edge.weights

    def _coupled_window(self, plan: dict, inputs_ab: torch.Tensor, a: int, want_traj: set, rec: dict, sampler: Optional[dict],
                        cutoff: int, sampling_steps: int) -> dict:
        """Steps [a, a + len) of every population in one (spiking members: two) library calls.  `want_traj`: populations
        whose per-step output is needed afterwards; `rec`: label -> {slot: buffer, "idx": next row}; `sampler`: the
        population whose output is sampled in-kernel ({"label", "samples", "idx"}).  Returns label -> [steps, N] trajectory."""
        xin = {p: None if X is None else X.detach() for p, X in self._coupled_inputs(plan, inputs_ab).items()}
        return self._coupled_steps(plan, xin, inputs_ab.shape[0], a, want_traj, rec, sampler, cutoff, sampling_steps)

    def _coupled_steps(self, plan: dict, xin: dict, steps: int, a: int, want_traj: set, rec: dict, sampler: Optional[dict],
                       cutoff: int, sampling_steps: int, stored: Optional[dict] = None, saved_u: Optional[dict] = None) -> dict:
        """The library call(s) of `_coupled_window`; `xin`: label -> [steps, rows | 1] input (or None) without a graph;
        `stored`: label -> keyword arguments capturing the window for the adjoint (save_v / save_src / save_acc)."""
        from .engine import forward_coupled
        pops: List[str] = plan["pops"]
        members: List[str] = plan["members"]
        index = {p: i for i, p in enumerate(members)}
        spiking = any(isinstance(self.get_node(p), SpikeResetNet) for p in pops)
        lag_read = {s_ for p in members for s_, mode in plan["edges"][p] if mode == _abi.RP_EDGE_LAG}
        # every population that is read through its state variable gets the history ring: RP_EDGE_OUT edges need it, and
        # with it on the RP_EDGE_LIVE sources too the library may step all members in ONE launch (rp_forward_coupled)
        hist_read = lag_read | {s_ for p in members for s_, mode in plan["edges"][p]
                                if s_ in pops and self.get_node(s_)._out_code == _abi.RP_OUT_STATE0}
        packed_edges = self._packed_edges(plan)
        trajs = {}
        for p in members:
            node = self.get_node(p)
            pop = self._member_pop(p)
            if p in want_traj:
                trajs[p] = torch.empty(steps, pop.rows, dtype=torch.float32, device=pop.device)
            if p not in pops:  # an activation node: what a feedback edge reads before the node's first step of the window
                prev = self._prev_out.get(p)
                pop.source.zero_() if prev is None else pop.source.copy_(prev.detach().reshape(-1))
                continue
            node._coupled_begin()
            if p in hist_read:
                if getattr(node, "_hist", None) is None:
                    node._hist = [torch.zeros(pop.rows, dtype=torch.float32, device=pop.device) for _ in (0, 1, 2)]
                if p in lag_read:  # what a feedback edge reads at the window's first step: the node's exposed output
                    node._hist[(a + 2) % 3].copy_(node["out"].detach())
        # the matrices the members' step kernels stream, and the memory stages of the edges that have one (loaded with the
        # edges' current state; written back after the last call)
        edge_args, stages = {}, []
        for p in members:
            row = []
            for s_, mode in plan["edges"][p]:
                edge = self.get_edge(s_, p)
                mem = edge.memory_stage() if getattr(edge, "_stage_kind", None) else None
                if mem is not None:
                    stages.append((edge, mem))
                    mem.save = saved_u.get((s_, p)) if saved_u else None   # (the window of u_t, for the edge's adjoint)
                row.append((self._edge_matrix(edge), index[s_], mode, packed_edges.get((s_, p)), mem))
            edge_args[p] = row

        def call(t0: int, t1: int, last: bool):
            specs, saves = [], {}
            for _, mem in stages:
                mem.save_row0 = t0
            for p in members:
                node = self.get_node(p)
                pop = self._member_pop(p)
                kw = node._coupled_kwargs() if p in pops else dict(out_var=_abi.RP_OUT_SOURCE)
                kw.update(steps=t1 - t0, step0=a + t0, x=None if xin[p] is None else xin[p][t0:t1])
                if p in trajs:
                    kw["traj"] = trajs[p][t0:t1]
                r = rec.get(p)
                if r or (sampler and sampler["label"] == p):
                    kw.update(cutoff=cutoff, sampling_steps=sampling_steps)
                    if r:
                        kw.update(rec={k_: v_ for k_, v_ in r.items() if k_ != "idx"}, sample_idx=r["idx"])
                    if sampler and sampler["label"] == p:
                        kw.update(samples=sampler["samples"], sample_idx=sampler["idx"])
                if p in hist_read:
                    kw["hist"] = node._hist
                if stored and p in stored:
                    kw.update({k_: v_[t0:t1] for k_, v_ in stored[p].items() if v_ is not None})
                if last and isinstance(node, SpikeResetNet) and not (stored and p in stored):
                    saves[p] = node._state_bufs()
                    kw.update(save_v=saves[p][0], save_src=saves[p][1], save_aux=saves[p][2], save_aux2=saves[p][3])
                specs.append(dict(pop=pop, run=kw, edges=edge_args[p]))
            idx = forward_coupled(specs)
            for p, i in zip(members, idx):
                if p in rec:
                    rec[p]["idx"] = i
                if sampler and sampler["label"] == p:
                    sampler["idx"] = i
            for p, bufs in saves.items():
                self.get_node(p)._coupled_end(bufs)

        if stored:  # (a stored window holds the pre-update state of its last step already)
            spiking = any(isinstance(self.get_node(p), SpikeResetNet) and p not in stored for p in pops)
        if spiking and steps > 1:  # the last step on its own: it leaves the pre-update state a spiking node exposes as `y`
            call(0, steps - 1, False)
            call(steps - 1, steps, True)
        else:
            call(0, steps, spiking)
        for edge, mem in stages:
            edge.stage_done(mem)
            mem.save = None
        for p in plan["between"]:  # what a feedback edge from the activation node reads next: its last output
            self._prev_out[p] = self._member_pop(p).source.clone()
        return trajs

    PACKED_EDGE_MAX_DENSITY = 0.5   # (the rules of a population's own matrix, nodes.py: _maybe_pack_weights)
    PACKED_EDGE_MIN_ROWS = 4096

    def _packed_edges(self, plan: dict, transposed: bool = False) -> dict:
        """(source, target) -> packed form of the edge matrix, for every member whose edges are ALL fixed and sparse enough
        (sparse long-range coupling between populations, whatever their own W): their stored nonzeros are streamed by packed
        mat-vec launches instead of the dense matrices (rp_run.edge_sp; the bits of the dense products).  Packed once per edge
        and weight version.  Launch-bound sizes keep the in-kernel dense edges (one launch per member, or one for all).
        ``transposed``: the packed form of W_e^T instead, for the adjoint's products W_e^T gx (rp_bwd_member.edge_spt)."""
        from . import sparse
        out = {}
        for p in plan["pops"]:
            node = self.get_node(p)
            pop = node._pop
            if not plan["edges"][p] or pop.device.type != "cuda":
                continue
            if pop.packed is not None:
                pop._sync_packed()
            forced = getattr(node, "_sparse_mode", "auto") is True
            if getattr(node, "_sparse_mode", "auto") is False or (not forced and pop.rows < self.PACKED_EDGE_MIN_ROWS):
                continue
            found = {}
            attr = "_packed_cache_t" if transposed else "_packed_cache"
            for s_, _ in plan["edges"][p]:
                edge = self.get_edge(s_, p)
                W = edge.weights
                if type(edge) is not Linear or s_ in plan.get("between", ()):
                    break
                if W.requires_grad or W.shape[1] % 4 or W.shape[0] % 4 or not W.is_contiguous() or W.device != pop.device:
                    break
                cache = getattr(edge, attr, None)
                if cache is None or cache[0] != (W._version, W.data_ptr()):
                    if sparse.density(W) > self.PACKED_EDGE_MAX_DENSITY:
                        cache = ((W._version, W.data_ptr()), None)
                    else:
                        M = W.detach().T.contiguous() if transposed else W.detach()   # [rows x columns] of the product's matrix
                        rows_pop = self.get_node(s_)._pop if transposed else pop
                        if M.shape[1] > sparse.WINDOW_COLUMNS:
                            pk = sparse.pack_rows_windowed(M)
                        elif rows_pop._block_rows() > 0:
                            pk = sparse.pack_rows_blocked(M, block_rows=rows_pop._block_rows())
                        else:
                            pk = sparse.pack_rows(M)
                        if pk.stored_bytes > 0.7 * 4.0 * W.shape[0] * W.shape[1]:   # (the rule of a population's own matrix)
                            pk = None
                        cache = ((W._version, W.data_ptr()), pk)
                    setattr(edge, attr, cache)
                if cache[1] is None:
                    break
                found[(s_, p)] = cache[1]
            else:
                out.update(found)
        return out

    ARENA_MIN_BYTES = 256 << 20

    def _consolidate(self, plan: dict):
        """Move the matrices a coupled step streams -- every population's W and every edge matrix between populations --
        into ONE device allocation (the tensors the nodes / edges hold become views of it; same values, still live).
        Measured on MI355X (scripts/probes/coupled_alloc2.py; two populations of 16384 neurons, four 1 GiB matrices): out
        of one allocation the step takes 603-619 us on every box and trial -- the single-population kernel's time for the
        same bytes -- while four separate 1 GiB allocations give 611-619 us or 653-667 us depending on where the driver
        happened to place them (same kernel, same data; which matrices share a launch does not matter).  A trained matrix
        keeps its tensor object (an optimiser holds on to it): its `.data` becomes the view."""
        from .edge_variants import LinearMasked
        mats = []
        for p in plan.get("members", plan["pops"]):
            if p in plan["pops"]:
                mats.append((self.get_node(p)._pop, "W"))
            for s_, _ in plan["edges"][p]:
                edge = self.get_edge(s_, p)
                if isinstance(edge, LinearMasked):
                    continue  # (the kernel streams weights * mask, formed per window)
                mats.append((edge, "weights"))
                if getattr(edge, "_stage_kind", None) in ("filter", "delay_filter") and edge.n_in > _abi.RP_EMEM_NARROW_MAX:
                    mats.append((edge, "filter"))  # a wide edge's filter matrix is streamed every step like a recurrent one
        tensors = [getattr(o, a) for o, a in mats]
        if not tensors or tensors[0].device.type != "cuda" or any(not t.is_contiguous() or t.dtype != torch.float32 or
                                                                  (t.requires_grad and not t.is_leaf) for t in tensors):
            return
        total = sum(t.numel() for t in tensors)
        if 4 * total < self.ARENA_MIN_BYTES:
            return
        arena = getattr(self, "_arena", None)
        if arena is not None and all(t.untyped_storage().data_ptr() == arena.untyped_storage().data_ptr() for t in tensors):
            return
        pad = 64  # floats: every matrix starts on a 256-byte boundary
        offs, off = [], 0
        for t in tensors:
            offs.append(off)
            off += (t.numel() + pad - 1) // pad * pad
        try:
            arena = torch.empty(off, dtype=torch.float32, device=tensors[0].device)
        except torch.cuda.OutOfMemoryError:
            return  # no room for a second copy of the matrices: they stay where they are (placement costs 5-8 %, not a run)
        for (obj, attr), t, o in zip(mats, tensors, offs):
            view = arena[o:o + t.numel()].view(t.shape)
            view.copy_(t.detach())
            # a packed form built from W stays valid only if it was valid before the move (an in-place edit of W since
            # packing must still trigger the re-pack: Population._sync_packed compares version counters)
            valid = attr == "W" and obj.packed is not None and obj._packed_version == t._version
            if t.requires_grad:
                t.data = view
            else:
                setattr(obj, attr, view)
            if attr == "W":
                obj._net_cache = None
                if obj.packed is not None:
                    obj._packed_version = obj.W._version if valid else -1
        self._arena = arena

    def _run_coupled(self, plan: dict, inputs, sampling_steps, cutoff, obs: Observer, rec_vars, verbose):
        self._consolidate(plan)
        T = inputs.shape[0]
        sample_at = self._sample_steps(T, cutoff, sampling_steps)
        S = len(sample_at)
        pops, post = plan["pops"], plan["post"]
        out_node = self._out_node
        rec = {}
        for (label, _), slot in zip(rec_vars, plan["slots"]):
            pop = self.get_node(label)._pop
            entry = rec.setdefault(label, {"idx": 0})
            if slot not in entry:
                entry[slot] = torch.empty(S, pop.rows, dtype=torch.float32, device=pop.device)
        members = plan["members"]
        for p in members:
            self._member_pop(p).reset_sampling()
        # the output: sampled in-kernel when the output node is a population, or a linear memory-less read-out of ONE
        # population (it commutes with the buffer mean); otherwise per-step trajectories -> read-out -> buffer means
        feeders = [p for p in members if any(t_ in post for t_ in self.graph.successors(p))]
        sampler = None
        if out_node in members:  # (a population, or an activation node inside the cycle: a member of the call itself)
            src_pop, chain = out_node, []
        elif len(feeders) == 1 and all(self.graph.in_degree(n) == 1 for n in post) and \
                self._memoryless_linear(post, feeders[0]) and list(topological_sort(self.graph.subgraph(post))) == post:
            src_pop, chain = feeders[0], post
        else:
            src_pop, chain = None, None
        if src_pop is not None:
            pop = self._member_pop(src_pop)
            sampler = dict(label=src_pop, samples=torch.empty(S, pop.rows, dtype=torch.float32, device=pop.device), idx=0)
        want_traj = set() if sampler else set(feeders)
        widest = max(self._member_pop(p).rows for p in members)
        chunk = max(1, min(T, _WINDOW_BYTES // (4 * widest * max(1, len(want_traj) + len(plan["pre"]) + 1))))
        pending = []
        for a in range(0, T, chunk):
            b = min(a + chunk, T)
            trajs = self._coupled_window(plan, inputs[a:b], a, want_traj, rec, sampler, cutoff, sampling_steps)
            if sampler is None:
                outs = dict(trajs)
                for n in post:
                    X = None
                    for p_ in self.graph.predecessors(n):
                        contrib = self._edge_window(self.get_edge(p_, n), outs[p_])
                        X = contrib if X is None else X + contrib
                    outs[n] = self.get_node(n).forward(X)
                pending.append(outs[out_node])
            if verbose:
                print(f"Progress: {b}/{T} integration steps finished.")
        if sampler is not None:
            outputs = self._apply_chain(chain, sampler["samples"], src_pop, None) if chain else sampler["samples"]
        else:
            outputs = self._buffer_means(torch.cat(pending), sample_at, cutoff)
        obs.record_block(sample_at, outputs, 0.0, [rec[label][slot] for (label, _), slot in zip(rec_vars, plan["slots"])])
        for n in self._order:  # what the per-step walk leaves in `net[n]["out"]`: not kept for windows
            self[n]["out"] = None
        self._prev_out = {k: v for k, v in self._prev_out.items() if k in plan["between"]}

    def _coupled_node_outputs(self, plan: dict, inputs: torch.Tensor, label: str) -> torch.Tensor:
        """Per-step outputs [T, n] of the node `label` -- a population or a node downstream of all populations -- over the
        whole input, the populations integrated window by window with one rp_forward_coupled call each (no gradient)."""
        self._consolidate(plan)
        T = inputs.shape[0]
        pops, post = plan["members"], plan["post"]
        if label in pops:
            want, chain = {label}, []
        else:
            chain = [n for n in post if n == label or label in descendants(self.graph, n)]
            want = {p for p in pops if any(t_ in chain for t_ in self.graph.successors(p))}
        for p in pops:
            self._member_pop(p).reset_sampling()
        widest = max(self._member_pop(p).rows for p in pops)
        chunk = max(1, min(T, _WINDOW_BYTES // (4 * widest * max(1, len(want) + len(plan["pre"]) + 1))))
        pieces = []
        for a in range(0, T, chunk):
            b = min(a + chunk, T)
            outs = dict(self._coupled_window(plan, inputs[a:b], a, want, {}, None, 0, 1))
            for n in chain:
                X = None
                for p_ in self.graph.predecessors(n):
                    if p_ in outs:
                        contrib = self._edge_window(self.get_edge(p_, n), outs[p_])
                        X = contrib if X is None else X + contrib
                outs[n] = self.get_node(n).forward(X)
            pieces.append(outs[label])
        for n in self._order:
            self[n]["out"] = None
        self._prev_out = {k: v for k, v in self._prev_out.items() if k in plan["between"]}
        return pieces[0] if len(pieces) == 1 else torch.cat(pieces)

    def reset(self, state: dict = None):
        super().reset(state)
        self._prev_out = {}

    def detach(self, requires_grad: bool = True, detach_params: bool = False) -> None:
        super().detach(requires_grad=requires_grad, detach_params=detach_params)
        self._prev_out = {k: v.detach() for k, v in self._prev_out.items()}   # (a window ends: the fed-back outputs become constants)
