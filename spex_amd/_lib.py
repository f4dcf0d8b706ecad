"""ctypes binding of libspexhip.so (the C ABI declared in include/spex_hip.h).

There is no CPU fallback: if the library is missing, or a call fails, this module raises.  Build it with
`python -c "import __graft_entry__ as g; g.build()"` (or `make -C spex_amd/csrc`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libspexhip.so")

c_i32, c_i64, c_f32, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/spex_hip.h one-to-one (tests/test_abi.py parses the header and checks)
SIGNATURES = {
    "spex_version": (ctypes.c_int, []),
    "spex_last_error": (ctypes.c_char_p, []),
    "spex_graph_create": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, ctypes.POINTER(c_vp)]),
    "spex_graph_destroy": (ctypes.c_int, [c_vp]),
    "spex_graph_pack_digest": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_vp]),
    "spex_graph_pack_hub_table": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_vp, c_i64, c_vp]),
    "spex_graph_info": (ctypes.c_int, [c_vp, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), ctypes.POINTER(c_i64),
                                       ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    "spex_graph_set_edge_mask": (ctypes.c_int, [c_vp, ctypes.c_int, c_vp, c_f32, ctypes.c_uint64]),
    "spex_spmm_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_f32, c_i32, c_vp]),
    "spex_spmm_rowlist_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_i32, c_i64, c_vp, c_vp, c_vp, c_f32, c_i32,
                                             c_vp]),
    "spex_spmm_owned_rows_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "spex_propagate_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "spex_propagate_bwd_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "spex_score_bce_f32": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i64, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp,
                                          c_vp, c_vp, c_vp, c_f32, c_vp]),
    "spex_score_bce_slots_f32": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i64, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp,
                                                c_vp, c_f32, c_vp, c_i32, c_vp]),
    "spex_bpr_sgd_step_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_i32,
                                             c_f32, c_f32, c_vp, c_vp]),
    "spex_bpr_loss_f32": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp,
                                         c_f32, c_vp]),
    "spex_bpr_grouped_workspace_bytes": (c_i64, [c_i64, c_i64, c_i64]),
    "spex_bpr_sgd_step_grouped_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_i32,
                                                     c_f32, c_f32, c_vp, c_vp, c_i64, c_vp]),
    "spex_bpr_loss_grouped_f32": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp,
                                                 c_f32, c_vp, c_i64, c_vp]),
    "spex_gather_owned_rows_f32": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_vp, c_vp]),
    "spex_scatter_add_owned_rows_f32": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_vp, c_i32, c_vp]),
    "spex_adam_step_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp]),
    "spex_ngcf_layer_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_f32,
                                           c_vp]),
    "spex_ngcf_message_mask": (ctypes.c_int, [c_vp]),
    "spex_ngcf_layer_fwd_rows_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32,
                                                    ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, c_i32, c_vp, c_i32, c_i64, c_vp, c_i32,
                                                    c_i64, c_vp]),
    "spex_ngcf_layer_fwd_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_i32, c_i32, c_f32,
                                               c_f32, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, c_i32, c_vp]),
    "spex_ngcf_layer_bwd_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32,
                                               c_f32, c_f32, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, c_i32, c_vp,
                                               c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "spex_ngcf_layer_bwd_rows_parts": (c_i32, [c_i32]),
    "spex_ngcf_layer_bwd_rows_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32,
                                                    c_f32, c_f32, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, c_i32, c_vp,
                                                    c_i32, c_i64, c_vp, c_i32, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "spex_spmm_push_batch_f32": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i64, c_vp, c_i32, c_i64, c_vp, c_i32, c_vp, c_i32, c_f32, c_vp,
                                                c_i32, c_vp]),
    "spex_ngcf_fwd_score_bwd_rows_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_i32, c_f32, c_f32,
                                                        ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, c_i32, c_vp, c_vp, c_i32, c_i64,
                                                        c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "spex_ngcf_score_bwd_rows_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_i32, c_f32, c_f32,
                                                    ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, c_i32, c_vp, c_vp, c_i32, c_i64,
                                                    c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "spex_gated_batch_fwd_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_vp, c_vp,
                                                c_vp, c_vp, c_i32, c_vp]),
    "spex_gated_batch_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32, c_vp, c_vp,
                                            c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "spex_lightgcn_batch_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32, c_vp, c_vp, c_vp,
                                               c_vp, c_i32, c_vp]),
    "spex_lightgcn_batch_slots_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp,
                                                     c_i32, c_vp]),
    "spex_reduce_slots_f32": (ctypes.c_int, [c_vp, c_i32, c_i64, c_vp, c_i32, c_i64, c_i32, c_vp, c_i32, c_f32, c_vp, c_i32, c_i32, c_vp]),
    "spex_expert_gate_rows_bwd_parts": (c_i32, [c_i32]),
    "spex_expert_gate_rows_bwd_det_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_i32, c_i64, c_i64, c_i64, c_i32,
                                                         c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "spex_step_events_release": (ctypes.c_int, [c_vp, c_vp]),
    "spex_adam_step_sum_f32": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i64, c_vp, c_vp, c_i64, c_i32, c_f32, c_f32, c_f32, c_f32, c_vp]),
    "spex_unique_rows_i32": (ctypes.c_int, [c_vp, c_i32, c_i64, c_vp, c_i32, c_i64, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "spex_spmm_push_rows_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_f32, c_vp, c_i32, c_vp]),
    "spex_expert_gate_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "spex_expert_gate_bwd_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "spex_expert_gate_bwd_parts": (c_i32, [c_i32]),
    "spex_expert_gate_bwd_det_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "spex_expert_gate_rows_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_i32, c_i64, c_i64, c_i64, c_i32, c_vp,
                                                 c_vp]),
    "spex_expert_gate_rows_bwd_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_i32, c_i64, c_i64, c_i64, c_i32,
                                                     c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "spex_sample_negatives": (ctypes.c_int, [c_vp, c_vp, c_i32, c_vp, c_i64, c_i32, c_i32, ctypes.c_uint64, c_vp, c_vp]),
    "spex_graph_set_values": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp]),
    "spex_sddmm_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "spex_edge_softmax_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "spex_edge_softmax_bwd_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "spex_attn_fuse_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp]),
    "spex_attn_fuse_bwd_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32, c_f32, c_f32, c_vp,
                                              c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "spex_path_attention_f32": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp,
                                               c_vp]),
    "spex_path_attention_bwd_f32": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp,
                                                   c_vp, c_vp, c_vp, c_vp]),
    "spex_trust_param_count": (ctypes.c_int64, [c_i32, c_i32]),
    "spex_trust_workspace_floats": (ctypes.c_int64, [c_i32, c_i32, c_i32, c_i32, c_i64]),
    "spex_trust_head_fwd_f32": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "spex_trust_head_train_f32": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp,
                                                 c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "spex_lightgcn_step_bpr_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_vp,
                                                   c_vp]),
    "spex_lightgcn_step_bce_f32": (ctypes.c_int, [ctypes.c_void_p, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "spex_lightgcn_epoch_bce_f32": (ctypes.c_int, [ctypes.c_void_p, c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_f32, ctypes.c_uint32, c_vp, c_vp,
                                                   c_vp]),
    "spex_ngcf_step_bce_f32": (ctypes.c_int, [ctypes.c_void_p, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "spex_ngcf_epoch_bce_f32": (ctypes.c_int, [ctypes.c_void_p, c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp]),
    "spex_dual_task_step_f32": (ctypes.c_int, [ctypes.c_void_p, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "spex_dual_task_epoch_f32": (ctypes.c_int, [ctypes.c_void_p, c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_f32,
                                                ctypes.c_uint32, c_vp]),
    "spex_dual_task_step_join": (ctypes.c_int, [ctypes.c_void_p, c_vp]),
    "spex_comm_unique_id": (ctypes.c_int, [c_vp]),
    "spex_comm_create": (ctypes.c_int, [c_i32, c_i32, c_vp, ctypes.POINTER(c_vp)]),
    "spex_comm_destroy": (ctypes.c_int, [c_vp]),
    "spex_comm_info": (ctypes.c_int, [c_vp, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    "spex_comm_allgather_rows_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp]),
    "spex_comm_allreduce_sum_f32": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp]),
    "spex_partitioned_propagate_f32": (ctypes.c_int, [c_vp, c_vp]),
    "spex_partitioned_step_bce_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "spex_ngcf_deep_step_bce_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "spex_partitioned_dual_task_step_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "spex_timer_create": (ctypes.c_int, [c_i32, c_i32, ctypes.POINTER(c_vp)]),
    "spex_timer_destroy": (ctypes.c_int, [c_vp]),
    "spex_timer_attach": (ctypes.c_int, [c_vp, c_vp]),
    "spex_timer_read": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, ctypes.POINTER(c_i32), ctypes.c_int]),
}



class LightGCNStepDesc(ctypes.Structure):
    """spex_lightgcn_step_t (include/spex_hip.h)."""
    _fields_ = ([(n, c_vp) for n in ("graph", "graph_t", "E0", "m", "v", "light_out", "ws_fwd", "lo_batch", "g_out", "ws_bwd",
                                     "grad_E0", "grad_slots")]
                + [(n, c_i32) for n in ("slot_capacity", "n_user_rows", "L", "d")]
                + [(n, c_f32) for n in ("lr", "beta1", "beta2", "eps")] + [("t", c_i32), ("flags", c_i32)])


STEP_DETERMINISTIC = 1          # spex_hip.h: SPEX_STEP_DETERMINISTIC
STEP_FIXED_TASK_WEIGHTS = 2     # spex_hip.h: SPEX_STEP_FIXED_TASK_WEIGHTS
STEP_PIPELINED = 4              # spex_hip.h: SPEX_STEP_PIPELINED


class NGCFStepDesc(ctypes.Structure):
    """spex_ngcf_step_t (include/spex_hip.h)."""
    _fields_ = ([(n, c_vp) for n in ("graph", "E0", "mE", "vE", "W", "mW", "vW", "all_emb", "side", "g_slots", "g_side_c", "g_ego_c",
                                     "gW_parts", "grad")]
                + [(n, c_i32) for n in ("slot_capacity", "n_user_rows", "pad_row")] + [("slope", c_f32), ("p_drop", c_f32)]
                + [("seed", ctypes.c_uint64), ("dropout_step", c_i32), ("t", c_i32)]
                + [(n, c_f32) for n in ("lr", "beta1", "beta2", "eps")]
                + [(n, c_vp) for n in ("side_stream", "ev_fork", "ev_join", "graph_t", "g_side_dense", "g_ego_dense")] + [("flags", c_i32)])


class NGCFDeepStepDesc(ctypes.Structure):
    """spex_ngcf_deep_step_t (include/spex_hip.h)."""
    _fields_ = ([(n, c_vp) for n in ("graph", "graph_t", "E0", "mE", "vE", "W", "mW", "vW", "gW", "all_emb", "g_all", "sides", "egos", "g_slots",
                                     "g_side_c", "g_ego_c", "gW_parts", "g_side", "g_ego", "g_next", "p_drop")]
                + [(n, c_i32) for n in ("L", "slot_capacity", "n_user_rows", "pad_row")] + [("slope", c_f32)]
                + [("seed", ctypes.c_uint64), ("dropout_step", c_i32), ("t", c_i32)]
                + [(n, c_f32) for n in ("lr", "beta1", "beta2", "eps")])


class DualTaskStepDesc(ctypes.Structure):
    """spex_dual_task_step_t (include/spex_hip.h)."""
    _fields_ = ([(n, c_vp) for n in ("graph", "graph_t", "params", "m", "v", "light", "ws_fwd", "lo_batch", "g_prop", "g_raw", "g_E0",
                                     "ws_bwd", "mixed_slots", "grad_slots", "g_prop_slots", "arange", "g_user", "g_small", "a2",
                                     "trust_ws", "dscore", "loss_b", "loss", "loss_acc", "precision")]
                + [(n, c_i32) for n in ("slot_capacity", "path_capacity", "path_len", "n_user_rows", "L", "d", "n_heads", "hybrid",
                                        "n_rec")]
                + [(n, c_f32) for n in ("lr", "beta1", "beta2", "eps")] + [("t", c_i32)]
                + [(n, c_vp) for n in ("side_stream", "ev_fork", "ev_join", "g_raw_slots", "att_parts", "loss_rows")] + [("flags", c_i32), ("side_pending", c_i32)])


class PartitionedStepDesc(ctypes.Structure):
    """spex_partitioned_step_t (include/spex_hip.h)."""
    _fields_ = ([(n, c_vp) for n in ("graph", "graph_t", "comm", "rows_per_rank", "E0", "m", "v", "light_out", "g_local", "gs", "grad_E0",
                                     "gathered1", "gathered", "rows", "grad_rows", "arange")]
                + [(n, c_i32) for n in ("n_local", "max_rows", "slot_capacity", "L", "d")]
                + [(n, c_f32) for n in ("lr", "beta1", "beta2", "eps")] + [("t", c_i32), ("flags", c_i32)]
                + [(n, c_vp) for n in ("graph_push", "gathered2")])


class PartitionedDualStepDesc(ctypes.Structure):
    """spex_partitioned_dual_step_t (include/spex_hip.h)."""
    _fields_ = ([(n, c_vp) for n in ("graph", "graph_t", "comm", "rows_per_rank", "params", "m", "v", "light", "g_prop", "g_raw", "gs", "g_E0",
                                     "gathered1", "gathered", "gathered0", "user_pos", "user_table", "rows", "mixed_slots", "grad_slots",
                                     "g_prop_slots", "g_raw_slots", "loss_rows", "att_parts", "arange", "g_user", "g_small", "a2", "trust_ws",
                                     "dscore", "loss_b", "loss", "loss_acc", "precision")]
                + [(n, c_i32) for n in ("n_local", "max_rows", "n_local_users", "user_lo", "slot_capacity", "path_capacity", "path_len",
                                        "n_user_rows", "L", "d", "n_heads", "hybrid", "n_rec")]
                + [(n, c_f32) for n in ("lr", "beta1", "beta2", "eps")] + [("t", c_i32)]
                + [(n, c_vp) for n in ("side_stream", "ev_fork", "ev_join")] + [("flags", c_i32)]
                + [(n, c_vp) for n in ("graph_push", "gathered2")])


COMM_ID_BYTES = 128             # spex_hip.h: SPEX_COMM_ID_BYTES


def release_step_events(desc):
    """Destroy the fork / join events a two-stream step descriptor holds (spex_step_events_release); safe to call twice."""
    if desc is not None and (desc.ev_fork or desc.ev_join):
        call("spex_step_events_release", ctypes.byref(desc, type(desc).ev_fork.offset), ctypes.byref(desc, type(desc).ev_join.offset))


_lib = None


class SpexError(RuntimeError):
    pass


def load():
    """Load libspexhip.so and bind every symbol of the ABI.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SpexError(
            f"{LIB_PATH} not found: the HIP extension is not built and spex_amd has no CPU fallback. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` from the repo root.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().spex_last_error().decode("utf-8", "replace")
        raise SpexError(f"{what or 'libspexhip'} failed with status {rc}: {msg}")


def call(name, *args):
    check(getattr(load(), name)(*args), name)
