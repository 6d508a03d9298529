"""Autograd node for a window of spiking steps (QIF, QIF_SFA, LIF, IK): forward = ``rp_forward`` with the
pre-update states captured, backward = ``rp_qif_backward`` (surrogate-gradient adjoint: Spike.backward of reference
rectipy/nodes.py:479-481, no gradient through the reset mask, nodes.py:384) + ``rp_outer_accum`` for the weight
gradient.  Replaces autograd through the Python-unrolled ``SpikeResetNet.forward`` loop
(reference rectipy/network.py:1028-1040)."""
import torch

from . import _abi

# per-neuron parameters with a gradient kernel, by engine model
TRAINABLE = {
    "qif": ("tau", "k", "eta", "tau_s"),
    "lif": ("tau", "k", "eta", "tau_s"),
    "qif_sfa": ("tau", "k", "eta", "tau_s", "alpha", "tau_x"),
    "ik": ("C", "k", "eta", "tau_s", "g", "b", "tau_u", "kappa", "v_r", "v_theta", "E_r"),
    "iku": ("C", "k", "eta", "tau_s", "g", "b", "tau_u", "kappa", "v_r", "v_theta", "E_r"),
    "ik_biexp": ("C", "k", "eta", "tau_d", "g", "b", "tau_u", "kappa", "v_r", "v_theta", "E_r", "tau_r"),
}


class SpikeWindow(torch.autograd.Function):

    @staticmethod
    def forward(ctx, node, steps, x, y_in, W, *params):
        pop = node._pop
        N, dev = pop.rows, pop.device
        f32 = dict(dtype=torch.float32, device=dev)
        names = TRAINABLE[pop.model]
        req = {n: p.requires_grad for n, p in zip(names, params)}
        need_acc = pop.model in ("ik", "iku", "ik_biexp") or req.get("k", False)
        traj_v = torch.empty(steps, N, **f32)
        traj_s = torch.empty(steps, N, **f32)
        traj_aux = torch.empty(steps, N, **f32) if pop.aux is not None else None
        traj_aux2 = torch.empty(steps, N, **f32) if pop.aux2 is not None else None
        traj_acc = torch.empty(steps, N, **f32) if need_acc else None
        pop.forward(steps, x=x, save_v=traj_v, save_src=traj_s, save_aux=traj_aux, save_aux2=traj_aux2,
                    save_acc=traj_acc, x_target=node._x_target, spike_events=node.spike_events,
                    spike_count=node._spike_count)
        ctx.node, ctx.steps, ctx.names = node, steps, names
        ctx.x = None if x is None else x.detach()
        ctx.saved = (traj_v, traj_s, traj_aux, traj_acc, traj_aux2)
        node._last_window = (traj_v, traj_s, traj_aux, traj_aux2)
        out = traj_s if node._out_code == _abi.RP_OUT_SOURCE else traj_v
        return out.clone(), node._pack(pop.v, pop.source_local, pop.aux, pop.aux2)

    @staticmethod
    def backward(ctx, g_traj, g_yend):
        node = ctx.node
        pop = node._pop
        N, dev, T = pop.rows, pop.device, ctx.steps
        traj_v, traj_s, traj_aux, traj_acc, traj_aux2 = ctx.saved
        f32 = dict(dtype=torch.float32, device=dev)
        tpl = node._template

        def piece(name):
            if g_yend is None or name is None:
                return torch.zeros(N, **f32)
            lo, hi = getattr(node, "_internal_var_map", node._var_map)[name]  # layout of y_end (template order)
            return g_yend.detach().to(**f32)[lo:hi].clone().contiguous()

        lam_v, lam_s = piece(tpl.reset_var), piece(tpl.source_var)
        lam_x = piece(node._aux_name) if node._aux_name else None
        lam_x2 = piece(node._aux2_name) if node._aux2_name else None
        g_out = None if g_traj is None else g_traj.detach().to(**f32).contiguous()
        need = ctx.needs_input_grad  # (node, steps, x, y_in, W, *params)
        grads = {name: torch.zeros(N, **f32) for i, name in enumerate(ctx.names) if need[5 + i]}
        u_traj = torch.empty(T, N, **f32)
        g_x = torch.empty(T, N, **f32) if need[2] else None
        keep = pop.backward(traj_v, traj_s, g_out, node._out_code, lam_v, u_traj, traj_acc=traj_acc, grad_x=g_x,
                            lam_s=lam_s, slope=node._spike_slope, x=ctx.x, x_target=node._x_target,
                            traj_aux=traj_aux, lam_x=lam_x, grad_params=grads, traj_aux2=traj_aux2, lam_x2=lam_x2)
        g_W = None
        if need[4]:
            g_W = pop.outer_product(pop.dt, u_traj, traj_s)
        if g_x is not None and ctx.x is not None and ctx.x.shape[1] == 1:
            g_x = g_x.sum(dim=1, keepdim=True)
        del keep
        g_y = node._pack(lam_v, lam_s, lam_x, lam_x2) if need[3] else None
        return (None, None, g_x, g_y, g_W) + tuple(grads.get(name) for name in ctx.names)
