"""ctypes mirror of ``include/rectipy_hip.h`` (structs and constants only, no library loading).

Field order and types must match the header exactly; ``tests/test_abi.py`` checks the
struct sizes against the values the compiled library reports.
"""
import ctypes as C

RP_ABI_VERSION = 19
RP_BATCH_VALU, RP_BATCH_MFMA, RP_BATCH_MFMA16, RP_BATCH_SLOTS = 0, 1, 2, 32
RP_COMM_RCCL, RP_COMM_DIRECT = 0, 1

RP_OK = 0
RP_ERR_ARG = -1
RP_ERR_SHAPE = -2
RP_ERR_ARCH = -3
RP_ERR_HIP = -4
RP_ERR_COMM = -5
RP_BWD_TRANSPOSE_MAX_N = 4096

RP_MODEL_LI_TANH = 0
RP_MODEL_LI_SIGMOID = 1
RP_MODEL_QIF = 2
RP_MODEL_QIF_SFA = 3
RP_MODEL_LIF = 4
RP_MODEL_IK = 5
RP_MODEL_IKU = 6
RP_MODEL_IK_BIEXP = 7
RP_MODEL_INSTANT = 8
RP_ACT_IDENTITY, RP_ACT_TANH, RP_ACT_SIGMOID = 0, 1, 2
RP_EMEM_NONE, RP_EMEM_DELAY, RP_EMEM_FILTER, RP_EMEM_DELAY_FILTER = 0, 1, 2, 3
RP_EMEM_NARROW_MAX = 64

RP_MAX_PARAMS = 12
RP_MAX_STATE = 4
RP_MAX_EDGES = 8
RP_MAX_MEMBERS = 64
RP_EDGE_LIVE, RP_EDGE_OUT, RP_EDGE_LAG = 0, 1, 2

RP_P_TAU, RP_P_K, RP_P_ETA = 0, 1, 2
RP_P_RMAX, RP_P_SLOPE, RP_P_V0 = 3, 4, 5
RP_P_TAU_S = 3
RP_P_ALPHA, RP_P_TAU_X = 4, 5
RP_P_IK_C, RP_P_V_R, RP_P_V_THETA, RP_P_G, RP_P_E_R, RP_P_B, RP_P_TAU_U, RP_P_KAPPA = 0, 4, 5, 6, 7, 8, 9, 10
RP_P_TAU_R = 11
RP_Y_V, RP_Y_AUX, RP_Y_AUX2 = 0, 1, 2
RP_REC_V, RP_REC_SOURCE, RP_REC_AUX, RP_REC_AUX2 = 0, 1, 2, 3
RP_XT_MEMBRANE, RP_XT_SYNAPSE = 0, 1

RP_X_NONE, RP_X_FULL, RP_X_BCAST, RP_X_PROJ = 0, 1, 2, 3
RP_OUT_STATE0, RP_OUT_SOURCE = 0, 1

fp = C.c_void_p  # device (HIP library) or host (oracle) float*/int32* pointers travel as integers


class RpNet(C.Structure):
    _fields_ = [
        ("model", C.c_int32),
        ("n", C.c_int32),
        ("rows", C.c_int32),
        ("row0", C.c_int32),
        ("ldw", C.c_int64),
        ("W", fp),
        ("p", fp * RP_MAX_PARAMS),
        ("y", fp * RP_MAX_STATE),
        ("src", fp * 2),
        ("cur", C.c_int32),
        ("dt", C.c_float),
        ("theta", C.c_float),
        ("v_reset", C.c_float),
        ("scratch", fp),
        ("sp_vals", fp),
        ("sp_cidx", fp),
        ("sp_rowoff", fp),
        ("sp_row_ids", fp),
        ("sp_batch_class", fp),
        ("sp_phase_window", fp),
        ("sp_phases", C.c_int32),
        ("sp_batches", C.c_int32),
        ("sp_block_rows", C.c_int32),
        ("act", C.c_int32),
    ]


class RpPacked(C.Structure):
    _fields_ = [
        ("vals", fp), ("cidx", fp), ("rowoff", fp), ("row_ids", fp), ("batch_class", fp), ("phase_window", fp),
        ("phases", C.c_int32), ("batches", C.c_int32), ("block_rows", C.c_int32),
    ]


class RpEdgeMem(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("n", C.c_int32), ("depth", C.c_int32),
        ("col_chan", fp), ("col_chan_host", fp), ("F", fp), ("ldf", C.c_int64),
        ("state", fp * 2), ("cur", C.c_int32), ("u", fp), ("save_u", fp), ("ld_save", C.c_int64),
    ]


class RpRun(C.Structure):
    _fields_ = [
        ("steps", C.c_int32),
        ("step0", C.c_int64),
        ("x_mode", C.c_int32),
        ("m", C.c_int32),
        ("x", fp),
        ("ldx", C.c_int64),
        ("w_in", fp),
        ("ldwin", C.c_int64),
        ("out_var", C.c_int32),
        ("x_target", C.c_int32),
        ("cutoff", C.c_int32),
        ("sampling_steps", C.c_int32),
        ("max_samples", C.c_int32),
        ("traj", fp),
        ("ldtraj", C.c_int64),
        ("save_state", fp * RP_MAX_STATE),
        ("save_src", fp),
        ("save_acc", fp),
        ("ldsave", C.c_int64),
        ("acc", fp),
        ("acc_count", C.c_int32),
        ("sample_idx", C.c_int32),
        ("samples", fp),
        ("ldsamples", C.c_int64),
        ("rec", fp * RP_MAX_STATE),
        ("spike_events", fp),
        ("spike_capacity", C.c_int32),
        ("spike_count", fp),
        ("save_src_full", fp),
        ("ldfull", C.c_int64),
        ("spike_src", fp),
        ("n_edges", C.c_int32),
        ("edge_w", fp * RP_MAX_EDGES),
        ("edge_ld", C.c_int64 * RP_MAX_EDGES),
        ("edge_n", C.c_int32 * RP_MAX_EDGES),
        ("edge_src", fp * RP_MAX_EDGES),
        ("edge_src_ld", C.c_int64 * RP_MAX_EDGES),
        ("hist", fp * 3),
        ("save_in", fp),
        ("edge_sp", RpPacked * RP_MAX_EDGES),
        ("edge_sum", fp),
        ("edge_mem", RpEdgeMem * RP_MAX_EDGES),
    ]


class RpMember(C.Structure):
    _fields_ = [
        ("net", C.POINTER(RpNet)),
        ("run", C.POINTER(RpRun)),
        ("edge_source", C.c_int32 * RP_MAX_EDGES),
        ("edge_mode", C.c_int32 * RP_MAX_EDGES),
    ]


class RpBwdMember(C.Structure):
    _fields_ = [
        ("net", C.c_void_p),     # const rp_net*
        ("bwd", C.c_void_p),     # const rp_bwd*
        ("n_edges", C.c_int32),
        ("edge_w", fp * RP_MAX_EDGES),
        ("edge_ld", C.c_int64 * RP_MAX_EDGES),
        ("edge_source", C.c_int32 * RP_MAX_EDGES),
        ("edge_mode", C.c_int32 * RP_MAX_EDGES),
        ("edge_work", fp),
        ("z", fp),
        ("edge_spt", RpPacked * RP_MAX_EDGES),
        ("final_src", fp),
        ("edge_mem_kind", C.c_int32 * RP_MAX_EDGES),
        ("edge_F", fp * RP_MAX_EDGES),
        ("edge_ldf", C.c_int64 * RP_MAX_EDGES),
        ("edge_lam", fp * RP_MAX_EDGES),
        ("edge_fwork", fp * RP_MAX_EDGES),
        ("edge_delay", C.c_int32 * RP_MAX_EDGES),
        ("edge_chan", C.c_int32 * RP_MAX_EDGES),
    ]


class RpBwd(C.Structure):
    _fields_ = [
        ("steps", C.c_int32),
        ("traj_v", fp),
        ("traj_r", fp),
        ("traj_acc", fp),
        ("ld", C.c_int64),
        ("grad_var", C.c_int32),
        ("grad_out", fp),
        ("ldg", C.c_int64),
        ("lam", fp),
        ("u_traj", fp),
        ("partials", fp),
        ("grad_tau", fp),
        ("grad_k", fp),
        ("grad_eta", fp),
        ("grad_x", fp),
        ("ldgx", C.c_int64),
        ("lam2", fp),
        ("grad_tau_s", fp),
        ("slope", C.c_float),
        ("x_mode", C.c_int32),
        ("x", fp),
        ("ldx", C.c_int64),
        ("x_target", C.c_int32),
        ("traj_aux", fp),
        ("lam3", fp),
        ("grad_p", fp * RP_MAX_PARAMS),
        ("wt", fp),
        ("ldwt", C.c_int64),
        ("traj_aux2", fp),
        ("lam4", fp),
        ("means", fp),
        ("spt_vals", fp),
        ("spt_cidx", fp),
        ("spt_rowoff", fp),
        ("spt_row_ids", fp),
        ("spt_batch_class", fp),
        ("spt_phase_window", fp),
        ("spt_phases", C.c_int32),
        ("spt_batches", C.c_int32),
        ("spt_block_rows", C.c_int32),
    ]


class RpBatch(C.Structure):
    _fields_ = [
        ("batch", C.c_int32),
        ("steps", C.c_int32),
        ("v", fp),
        ("ldv", C.c_int64),
        ("src", fp * 2),
        ("ldsrc", C.c_int64),
        ("cur", C.c_int32),
        ("x_mode", C.c_int32),
        ("x", fp),
        ("ldx", C.c_int64),
        ("out_var", C.c_int32),
        ("traj", fp),
        ("ldtraj", C.c_int64),
        ("layout", C.c_int32),
        ("step0", C.c_int64),
        ("cutoff", C.c_int32),
        ("sampling_steps", C.c_int32),
        ("max_samples", C.c_int32),
        ("acc", fp),
        ("ldacc", C.c_int64),
        ("acc_count", C.c_int32),
        ("sample_idx", C.c_int32),
        ("samples", fp),
        ("ldsamples", C.c_int64),
        ("rec", fp * 2),
        ("m", C.c_int32),
        ("w_in", fp),
        ("ldwin", C.c_int64),
    ]


class RpRls(C.Structure):
    _fields_ = [
        ("n_in", C.c_int32),
        ("n_out", C.c_int32),
        ("ldp", C.c_int64),
        ("ldw", C.c_int64),
        ("P", fp),
        ("W", fp),
        ("beta_inv", C.c_float),
        ("z", fp),
        ("zn", fp),
        ("scratch", fp),
        ("partials", fp),
        ("colparts", fp),
        ("ldc", C.c_int64),
        ("sym_rows", C.c_int32),
        ("defer_mirror", C.c_int32),
        ("block", C.c_int32),
        ("blk", fp),
    ]


class RpComm(C.Structure):
    _fields_ = [
        ("comm", C.c_void_p),
        ("all_gather", C.c_void_p),
        ("reduce_scatter", C.c_void_p),
        ("rank", C.c_int32),
        ("world", C.c_int32),
        ("work", fp),
        ("mode", C.c_int32),
        ("seq", C.c_uint32),
        ("peer_src", C.POINTER(C.c_void_p)),
        ("peer_flags", C.POINTER(C.c_void_p)),
        ("flags", C.c_void_p),
        ("peer_inbox", C.POINTER(C.c_void_p)),
        ("inbox", C.c_void_p),
    ]


def declare(lib):
    """Attach argtypes/restypes for the `rp_*` entry points of include/rectipy_hip.h that `lib` exports."""
    P = C.POINTER

    def sig(name, restype, argtypes):
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = restype
            fn.argtypes = argtypes
        return fn

    sig("rp_abi_version", C.c_int, [])
    sig("rp_last_error", C.c_char_p, [])
    sig("rp_async_status", C.c_int, [])
    sig("rp_device_check", C.c_int, [C.c_int])
    sig("rp_device_name", C.c_int, [C.c_int, C.c_char_p, C.c_int32])
    sig("rp_sizeof_net", C.c_int, [])
    sig("rp_sizeof_run", C.c_int, [])
    sig("rp_sizeof_bwd", C.c_int, [])
    sig("rp_sizeof_member", C.c_int, [])
    sig("rp_forward_coupled", C.c_int, [P(RpMember), C.c_int32, C.c_void_p])
    sig("rp_sizeof_bwd_member", C.c_int, [])
    sig("rp_backward_coupled", C.c_int, [P(RpBwdMember), C.c_int32, C.c_void_p])
    sig("rp_refresh_source", C.c_int, [P(RpNet), C.c_void_p])
    sig("rp_forward", C.c_int, [P(RpNet), P(RpRun), C.c_void_p])
    sig("rp_bwd_partial_rows", C.c_int32, [C.c_int32])
    sig("rp_rate_backward", C.c_int, [P(RpNet), P(RpBwd), C.c_void_p])
    sig("rp_qif_backward", C.c_int, [P(RpNet), P(RpBwd), C.c_void_p])
    sig("rp_outer_accum", C.c_int,
        [C.c_int32, C.c_int32, C.c_float, fp, C.c_int64, fp, C.c_int64, fp, C.c_int64, C.c_void_p])
    sig("rp_outer_accum_rect", C.c_int,
        [C.c_int32, C.c_int32, C.c_int32, C.c_float, fp, C.c_int64, fp, C.c_int64, fp, C.c_int64, C.c_void_p])
    sig("rp_outer_store_rect", C.c_int,
        [C.c_int32, C.c_int32, C.c_int32, C.c_float, fp, C.c_int64, fp, C.c_int64, fp, C.c_int64, C.c_void_p])
    sig("rp_bwd_shard_partial_rows", C.c_int32, [C.c_int32, C.c_int32])
    sig("rp_bwd_prepare", C.c_int, [P(RpNet), P(RpBwd), C.c_void_p])
    sig("rp_bwd_partial", C.c_int, [P(RpNet), P(RpBwd), C.c_int32, fp, C.c_void_p])
    sig("rp_bwd_apply", C.c_int, [P(RpNet), P(RpBwd), C.c_int32, fp, C.c_void_p])
    sig("rp_sizeof_comm", C.c_int, [])
    sig("rp_forward_sharded", C.c_int, [P(RpNet), P(RpRun), P(RpComm), C.c_void_p])
    sig("rp_backward_sharded", C.c_int, [P(RpNet), P(RpBwd), P(RpComm), C.c_void_p])
    sig("rp_ipc_alloc", C.c_int, [C.c_int64, C.POINTER(C.c_void_p), C.c_char_p])
    sig("rp_ipc_open", C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)])
    sig("rp_ipc_close", C.c_int, [C.c_void_p])
    sig("rp_ipc_free", C.c_int, [C.c_void_p])
    sig("rp_sizeof_batch", C.c_int, [])
    sig("rp_sizeof_rls", C.c_int, [])
    sig("rp_forward_batched", C.c_int, [P(RpNet), P(RpBatch), C.c_void_p])
    sig("rp_set_batched_grid", C.c_int, [C.c_int])
    sig("rp_set_batched_tuning", C.c_int, [C.c_int, C.c_int])
    sig("rp_set_batched_mfma_tuning", C.c_int, [C.c_int, C.c_int, C.c_int])
    sig("rp_set_rls_sym", C.c_int, [C.c_int])
    sig("rp_set_gemm_dma", C.c_int, [C.c_int])
    sig("rp_rls_sweep", C.c_int, [P(RpRls), C.c_int32, C.c_int32, fp, C.c_int64, fp, C.c_int64, fp, C.c_int64, fp,
                                  C.c_void_p])
    sig("rp_rls_mirror", C.c_int, [P(RpRls), C.c_void_p])
    sig("rp_sizeof_edge_mem", C.c_int, [])
    sig("rp_edge_window", C.c_int, [P(RpEdgeMem), C.c_int32, fp, C.c_int64, fp, C.c_int64, C.c_void_p])
    sig("rp_stream_probe", C.c_int, [fp, C.c_int64, fp, C.c_void_p])
    sig("rp_tune_forward", C.c_int, [C.POINTER(RpNet), C.c_void_p])
    sig("rp_forward_tuning", C.c_int, [C.POINTER(RpNet), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                       C.POINTER(C.c_float), C.POINTER(C.c_int32)])
    sig("rp_install_forward_tuning", C.c_int, [C.POINTER(RpNet), C.c_int32, C.c_int32, C.c_int32, C.c_float])
    sig("rp_clear_forward_tuning", C.c_int, [])
    sig("rp_set_tuning", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int])
    sig("rp_set_sparse_tuning", C.c_int, [C.c_int, C.c_int])
    sig("rp_set_blocked_tuning", C.c_int, [C.c_int, C.c_int])
    sig("rp_set_bwd_grid", C.c_int, [C.c_int])
    sig("rp_set_bwd_tuning", C.c_int, [C.c_int, C.c_int])
    sig("rp_set_bwd_chunks", C.c_int, [C.c_int])
    sig("rp_set_rls_tile", C.c_int, [C.c_int, C.c_int])
    sig("rp_set_rls_blk", C.c_int, [C.c_int, C.c_int])
    sig("rp_set_rls_tuning", C.c_int, [C.c_int, C.c_int, C.c_int])
    return lib


EXPORTS = [
    "rp_abi_version", "rp_last_error", "rp_async_status", "rp_device_check", "rp_device_name", "rp_sizeof_net", "rp_sizeof_run", "rp_sizeof_bwd",
    "rp_sizeof_member", "rp_forward_coupled", "rp_sizeof_bwd_member", "rp_backward_coupled",
    "rp_refresh_source", "rp_forward", "rp_bwd_partial_rows", "rp_rate_backward", "rp_qif_backward", "rp_outer_accum",
    "rp_outer_accum_rect", "rp_outer_store_rect", "rp_bwd_shard_partial_rows", "rp_bwd_prepare", "rp_bwd_partial",
    "rp_bwd_apply", "rp_sizeof_comm", "rp_forward_sharded", "rp_backward_sharded", "rp_ipc_alloc", "rp_ipc_open",
    "rp_ipc_close", "rp_ipc_free", "rp_sizeof_batch", "rp_sizeof_rls", "rp_forward_batched", "rp_rls_sweep", "rp_rls_mirror",
    "rp_sizeof_edge_mem", "rp_edge_window", "rp_stream_probe", "rp_tune_forward", "rp_forward_tuning", "rp_install_forward_tuning", "rp_clear_forward_tuning",
]
# the measurement build (-DRP_MEASUREMENT) adds the knobs of include/rectipy_hip.h's RP_MEASUREMENT section
MEASUREMENT_EXPORTS = [
    "rp_set_batched_tuning", "rp_set_batched_mfma_tuning", "rp_set_batched_grid", "rp_set_tuning", "rp_set_sparse_tuning",
    "rp_set_blocked_tuning", "rp_set_bwd_tuning", "rp_set_bwd_grid", "rp_set_bwd_chunks", "rp_set_rls_tuning",
    "rp_set_rls_sym", "rp_set_gemm_dma", "rp_set_rls_tile", "rp_set_rls_blk",
]
