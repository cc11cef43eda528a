"""ctypes binding of csrc/libazamd.so (symbols declared in include/az_engine.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libazamd.so")

CELL_PAD = 384
MAX_ACTIONS = 362
NPLANES = 17

ERR_ARG, ERR_HIP, ERR_ILLEGAL_MOVE, ERR_TREE_FULL = -1, -2, -3, -4
ERR_RNG_UNDERFLOW, ERR_DEPTH, ERR_STATE, ERR_NO_WEIGHTS = -5, -6, -7, -8

GAME_IDS = {"go": 0, "reversi": 1, "mnk": 2}


class AzConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("game", C.c_int32), ("board_height", C.c_int32), ("board_width", C.c_int32), ("k", C.c_int32),
        ("n_slots", C.c_int32), ("max_playout", C.c_int32), ("transform_types", C.c_int32),
        ("dirichlet_before", C.c_int32), ("max_turn", C.c_int32),
        ("d_alpha", C.c_double), ("d_epsilon", C.c_double),
        ("tree_edge_cap", C.c_int32), ("tree_block_cap", C.c_int32), ("max_depth", C.c_int32),
        ("rng_replay", C.c_int32), ("seed", C.c_uint64), ("record_trajectories", C.c_int32),
        ("num_blocks", C.c_int32), ("channels", C.c_int32), ("tower_dtype", C.c_int32), ("device", C.c_int32),
        ("leaves_per_round", C.c_int32), ("virtual_loss", C.c_double),
        ("tree_full_policy", C.c_int32), ("search_f64", C.c_int32), ("enforce_superko", C.c_int32),
        ("c_punt", C.c_double),
        ("tower_no_graphs", C.c_int32), ("tower_no_duo", C.c_int32), ("tower_group_boards", C.c_int32),
        ("match_parallel_towers", C.c_int32),
        ("tree_pool_edges", C.c_int64), ("tree_page_edges", C.c_int32), ("reserved0", C.c_int32),
    ]


class AzStateBlob(C.Structure):
    _fields_ = [
        ("board", C.c_int8 * CELL_PAD),
        ("history", (C.c_int8 * CELL_PAD) * 7),
        ("current_player", C.c_int32), ("ko", C.c_int32), ("turns", C.c_int32), ("n_history", C.c_int32),
        ("last1", C.c_int32), ("last2", C.c_int32),
        ("is_end_of_game", C.c_int32), ("winner", C.c_int32), ("winner_set", C.c_int32),
        ("passes_black", C.c_int32), ("passes_white", C.c_int32),
        ("num_black_prisoners", C.c_int32), ("num_white_prisoners", C.c_int32), ("stones_played", C.c_int32),
    ]


class AzCounters(C.Structure):
    _fields_ = [
        ("steps", C.c_uint64), ("evals", C.c_uint64), ("moves", C.c_uint64), ("games_finished", C.c_uint64),
        ("levels", C.c_uint64), ("children_scanned", C.c_uint64), ("edges_created", C.c_uint64),
        ("rule_moves", C.c_uint64),
        ("tower_ms", C.c_double), ("select_ms", C.c_double), ("finish_ms", C.c_double),
        ("tower_calls", C.c_uint64), ("tower_evals", C.c_uint64),
        ("conv_ms", C.c_double), ("conv_launches", C.c_uint64), ("conv_flops_per_launch", C.c_double),
        ("expansions_skipped", C.c_uint64),
        ("peak_tree_edges", C.c_uint64), ("peak_tree_blocks", C.c_uint64),
        ("tree_edge_cap", C.c_uint64), ("tree_block_cap", C.c_uint64),
        ("tree_page_edges", C.c_uint64), ("tree_pool_pages", C.c_uint64), ("tree_pool_pages_in_use", C.c_uint64),
        ("tree_pool_pages_peak", C.c_uint64),
    ]


_P = C.c_void_p
_I = C.c_int32
# name -> (restype, argtypes); every symbol include/az_engine.h declares
SIGNATURES = {
    "az_engine_create": (_P, [C.POINTER(AzConfig)]),
    "az_abi_sizes": (None, [_P]),
    "az_engine_destroy": (None, [_P]),
    "az_last_error": (C.c_char_p, [_P]),
    "az_engine_dims": (_I, [_P, _P, _P, _P]),
    "az_games_reset": (_I, [_P, _P, _I]),
    "az_state_copy": (_I, [_P, _I, _P, _I, _I]),
    "az_state_get": (_I, [_P, _I, C.POINTER(AzStateBlob)]),
    "az_state_set": (_I, [_P, _I, C.POINTER(AzStateBlob)]),
    "az_rules_legal_mask": (_I, [_P, _P, _I, _P]),
    "az_rules_do_move": (_I, [_P, _P, _I, _P, _P, _P, _P]),
    "az_rules_winner": (_I, [_P, _P, _I, _P]),
    "az_state_transform": (_I, [_P, _I, _I]),
    "az_features": (_I, [_P, _P, _I, _P, _P]),
    "az_search_begin": (_I, [_P, _P, _I, _P, _P]),
    "az_search_step_select": (_I, [_P, _P]),
    "az_leaf_batch": (_I, [_P, _P, _P]),
    "az_leaf_outputs": (_I, [_P, _P, _P]),
    "az_search_step_finish": (_I, [_P, _P, _P]),
    "az_search_root_stats": (_I, [_P, _I, _I, _P, _P, _P, _P, _P]),
    "az_search_choose": (_I, [_P, _I, _I, C.c_double, _P]),
    "az_search_advance_tree": (_I, [_P, _I, _I, _I]),
    "az_search_advance": (_I, [_P, _P, _I, _P]),
    "az_search_tree_size": (_I, [_P, _I, _I, _P, _P]),
    "az_search_grow_trees": (_I, [_P, _I, _I]),
    "az_search_grow_pool": (_I, [_P, C.c_int64]),
    "az_search_set_c_punt": (_I, [_P, C.c_double]),
    "az_selfplay_start": (_I, [_P]),
    "az_selfplay_run": (_I, [_P, _I]),
    "az_selfplay_select": (_I, [_P, _P]),
    "az_selfplay_finish": (_I, [_P, _P, _P]),
    "az_selfplay_status": (_I, [_P, _P, _P]),
    "az_trajectory_pop": (_I, [_P, _I, _P, _P, _P, _I]),
    "az_trajectory_pop_many": (_I, [_P, _P, _I, _P, _I, _P, _P, _P, _I]),
    "az_trajectory_pop_many_dev": (_I, [_P, _P, _I, _P, _I, _P, _P, _P, _I]),
    "az_trajectory_pop_many_static": (_I, [_P, _P, _I, _P, _I, _P, _P, _P, _I]),
    "az_trajectory_max_plies": (_I, [_P]),
    "az_rng_inject": (_I, [_P, _I, _P, C.c_int64, _P, C.c_int64, _P, C.c_int64]),
    "az_rng_cursor": (_I, [_P, _I, _P, _P, _P]),
    "az_tower_blob_floats": (C.c_size_t, [_P]),
    "az_tower_load_weights": (_I, [_P, _P, C.c_size_t]),
    "az_tower_load_weights_dev": (_I, [_P, _I, _P, C.c_size_t]),
    "az_search_step_finish_f64": (_I, [_P, _P, _P]),
    "az_selfplay_finish_f64": (_I, [_P, _P, _P]),
    "az_search_root_stats_f64": (_I, [_P, _I, _I, _P, _P, _P, _P, _P]),
    "az_tower_load_weights_net": (_I, [_P, _I, _P, C.c_size_t]),
    "az_match_assign": (_I, [_P, _P, _I, _P]),
    "az_match_leaf_nets": (_I, [_P, _P]),
    "az_tower_forward": (_I, [_P, _P, _I, _P, _P]),
    "az_tower_forward_dev": (_I, [_P, _P, _I, _P, _P]),
    "az_tower_flops_per_eval": (C.c_double, [_P]),
    "az_tower_conv_kernel": (C.c_char_p, [_P]),
    "az_search_run": (_I, [_P, _I]),
    "az_tower_profile": (_I, [_P, _I]),
    "az_tower_leaf_compaction": (_I, [_P, _I]),
    "az_counters_get": (_I, [_P, C.POINTER(AzCounters)]),
    "az_counters_reset": (_I, [_P]),
    "az_sync": (_I, [_P]),
}

_lib = None


def lib():
    """Load the HIP library; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "alphazero_amd: %s is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C alphazero_amd/csrc`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        sizes = (C.c_int32 * 3)()
        L.az_abi_sizes(sizes)
        mine = (C.sizeof(AzConfig), C.sizeof(AzStateBlob), C.sizeof(AzCounters))
        if tuple(sizes) != mine:
            raise ImportError("alphazero_amd: %s was built from another include/az_engine.h (struct sizes %s, binding %s): rebuild it"
                              % (LIB_PATH, tuple(sizes), mine))
        _lib = L
    return _lib
