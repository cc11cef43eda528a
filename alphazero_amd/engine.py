"""Python handle on one GPU's batched self-play engine (thin layer over the C ABI)."""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import AzConfig, AzCounters, AzStateBlob, GAME_IDS, NPLANES


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("az_engine error %d: %s" % (code, msg))
        self.code = code


class IllegalMove(Exception):
    """Raised by the env shims when the engine reports AZ_ERR_ILLEGAL_MOVE."""


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def game_config_to_engine_kwargs(game_config):
    """Map a reference game yaml dict (AlphaZero/config/<game>.yaml) to Engine kwargs."""
    env = game_config.get("env_path", "")
    name = env.rsplit(".", 1)[-1] if env else str(game_config.get("name", "go"))
    game = "go" if name.startswith("go") else ("reversi" if name.startswith("reversi") else "mnk")
    return dict(game=game, board_height=int(game_config["board_height"]), board_width=int(game_config["board_width"]),
                k=int(game_config.get("k", 0)), transform_types=int(game_config.get("transform_types", 0)),
                d_alpha=float(game_config.get("d_alpha", 0.03)), d_epsilon=float(game_config.get("d_epsilon", 0.25)))


def slots_for_whole_tile_rounds(n_max, board_height, board_width, compute_units=256, tile_points=256):
    """Largest slot count <= n_max whose board points fill a whole number of rounds of the tower's conv tiles
    (256 points each, one tile per compute unit at a time): a partly filled last round costs as much as a full one
    in every conv launch (4096 slots of 15x15 are 14.06 rounds).  The tower takes batches of four rounds and more in
    groups of boards that fill two rounds each, whole boards per group (582 boards of 15x15; 4074 slots are 7 such groups),
    and what is left over last.  n_max itself when it is below one round."""
    hw = board_height * board_width
    per_round = tile_points * compute_units
    rounds = (n_max * hw) // per_round
    if rounds < 1:
        return n_max
    if rounds < 4:
        return (rounds * per_round) // hw
    return (rounds // 2) * ((2 * per_round) // hw) + (rounds % 2) * (per_round // hw)


class Engine:
    def __init__(self, game, board_height, board_width, k=0, n_slots=1, max_playout=0, transform_types=0,
                 dirichlet_before=30, max_turn=400, d_alpha=0.03, d_epsilon=0.25, tree_edge_cap=0, tree_block_cap=0,
                 max_depth=0, rng_replay=False, seed=0, record_trajectories=False, num_blocks=0, channels=256,
                 tower_dtype="bf16", device=0, search_f64=False, tree_full_policy=0, leaves_per_round=1,
                 virtual_loss=1.0, enforce_superko=False, c_punt=5.0, tower_graphs=None, conv_duo=None,
                 tower_group_boards=None, match_parallel_towers=None, tree_pool_edges=0, tree_page_edges=0):
        """The last four arguments are measurement switches (A/B timing, tests).  Left at None they follow the
        environment variables the tools/ scripts set: AZ_NO_GRAPHS, AZ_CONV_DUO=0, AZ_TOWER_GROUP_BOARDS=N,
        AZ_MATCH_PARALLEL_TOWERS; the library itself reads no environment."""
        self._L = _lib.lib()
        env = os.environ
        if tower_graphs is None:
            tower_graphs = "AZ_NO_GRAPHS" not in env
        if conv_duo is None:
            conv_duo = not env.get("AZ_CONV_DUO", "1").startswith("0")
        if tower_group_boards is None:
            tower_group_boards = int(env.get("AZ_TOWER_GROUP_BOARDS", "0"))
        if match_parallel_towers is None:
            match_parallel_towers = "AZ_MATCH_PARALLEL_TOWERS" in env
        cfg = AzConfig(
            struct_size=C.sizeof(AzConfig), game=GAME_IDS[game] if isinstance(game, str) else int(game),
            board_height=board_height, board_width=board_width, k=k, n_slots=n_slots, max_playout=max_playout,
            transform_types=transform_types, dirichlet_before=dirichlet_before, max_turn=max_turn, d_alpha=d_alpha,
            d_epsilon=d_epsilon, tree_edge_cap=tree_edge_cap, tree_block_cap=tree_block_cap, max_depth=max_depth,
            rng_replay=int(rng_replay), seed=seed, record_trajectories=int(record_trajectories), num_blocks=num_blocks,
            channels=channels, tower_dtype={"fp32": 0, "bf16": 1, "f16": 2}[tower_dtype] if isinstance(tower_dtype, str) else int(tower_dtype),
            device=device, leaves_per_round=int(leaves_per_round), virtual_loss=float(virtual_loss),
            tree_full_policy=int(tree_full_policy), search_f64=int(bool(search_f64)), enforce_superko=int(bool(enforce_superko)),
            c_punt=float(c_punt), tower_no_graphs=int(not tower_graphs), tower_no_duo=int(not conv_duo),
            tower_group_boards=int(tower_group_boards), match_parallel_towers=int(bool(match_parallel_towers)),
            tree_pool_edges=int(tree_pool_edges), tree_page_edges=int(tree_page_edges))
        self.cfg = cfg
        # dtype of W / P / priors / values: float64 reproduces a search fed by a float64 evaluator (dummy_eval.py:26-28)
        self.real = np.float64 if search_f64 else np.float32
        self._h = self._L.az_engine_create(C.byref(cfg))
        if not self._h:
            raise EngineError(-2, self._L.az_last_error(None).decode())
        self.game = game
        self.H, self.W = board_height, board_width
        self.NC = board_height * board_width
        self.O = self.NC + (0 if cfg.game == 2 else 1)
        self.G = n_slots
        self.K = max(1, int(leaves_per_round))   # leaf rows per round = K * G, row j * G + slot

    def close(self):
        if getattr(self, "_h", None):
            self._L.az_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise EngineError(rc, self._L.az_last_error(self._h).decode())
        return rc

    # ---- state / rules ---------------------------------------------------
    def reset(self, slots):
        s = _i32(slots)
        self._ck(self._L.az_games_reset(self._h, s.ctypes.data, s.size))

    def state_copy(self, dst_slot, src_engine, src_slot, ref_semantics=True):
        self._ck(self._L.az_state_copy(self._h, dst_slot, src_engine._h, src_slot, int(ref_semantics)))

    def state_get(self, slot):
        b = AzStateBlob()
        self._ck(self._L.az_state_get(self._h, slot, C.byref(b)))
        return b

    def state_set(self, slot, blob):
        self._ck(self._L.az_state_set(self._h, slot, C.byref(blob)))

    def board(self, slot):
        b = self.state_get(slot)
        return np.frombuffer(b.board, dtype=np.int8, count=self.NC).reshape(self.H, self.W).copy()

    def legal_mask(self, slots):
        s = _i32(slots)
        out = np.zeros((s.size, self.O), np.uint8)
        self._ck(self._L.az_rules_legal_mask(self._h, s.ctypes.data, s.size, out.ctypes.data))
        return out

    def do_move(self, slots, actions, colors=None):
        s, a = _i32(slots), _i32(actions)
        c = _i32(colors) if colors is not None else None
        status = np.zeros(s.size, np.int32)
        ended = np.zeros(s.size, np.int32)
        self._ck(self._L.az_rules_do_move(self._h, s.ctypes.data, s.size, a.ctypes.data,
                                          c.ctypes.data if c is not None else None, status.ctypes.data, ended.ctypes.data))
        return status, ended

    def winner(self, slots):
        s = _i32(slots)
        out = np.zeros(s.size, np.int32)
        self._ck(self._L.az_rules_winner(self._h, s.ctypes.data, s.size, out.ctypes.data))
        return out

    def transform(self, slot, tid):
        self._ck(self._L.az_state_transform(self._h, slot, tid))

    def features(self, slots, tids=None):
        s = _i32(slots)
        t = _i32(tids) if tids is not None else None
        out = np.zeros((s.size, NPLANES, self.H, self.W), np.int8)
        self._ck(self._L.az_features(self._h, s.ctypes.data, s.size, t.ctypes.data if t is not None else None, out.ctypes.data))
        return out

    # ---- search ------------------------------------------------------------
    def search_begin(self, slots, sides, dirichlet):
        s, sd, d = _i32(slots), _i32(sides), _i32(dirichlet)
        self._ck(self._L.az_search_begin(self._h, s.ctypes.data, s.size, sd.ctypes.data, d.ctypes.data))

    def step_select(self):
        n = C.c_int32(0)
        self._ck(self._L.az_search_step_select(self._h, C.byref(n)))
        return n.value

    def leaf_batch(self):
        planes = np.zeros((self.K * self.G, NPLANES, self.H, self.W), np.int8)
        valid = np.zeros(self.K * self.G, np.uint8)
        self._ck(self._L.az_leaf_batch(self._h, planes.ctypes.data, valid.ctypes.data))
        return planes, valid

    def leaf_outputs(self):
        """(priors (K*G, O), values (K*G,)) float32 as the last finish step used them; rows that took no evaluation
        are stale."""
        p = np.zeros((self.K * self.G, self.O), np.float32); v = np.zeros(self.K * self.G, np.float32)
        self._ck(self._L.az_leaf_outputs(self._h, p.ctypes.data, v.ctypes.data))
        return p, v

    def step_finish(self, priors=None, values=None):
        if priors is None:
            self._ck(self._L.az_search_step_finish(self._h, None, None))
            return
        p = np.ascontiguousarray(priors, self.real)
        v = np.ascontiguousarray(values, self.real)
        assert p.shape == (self.K * self.G, self.O) and v.shape == (self.K * self.G,)
        fn = self._L.az_search_step_finish_f64 if self.real is np.float64 else self._L.az_search_step_finish
        self._ck(fn(self._h, p.ctypes.data, v.ctypes.data))

    def search_run(self, n_rounds):
        self._ck(self._L.az_search_run(self._h, int(n_rounds)))

    def root_stats(self, slot, side):
        a = np.zeros(self.O, np.int32); n = np.zeros(self.O, np.int32)
        w = np.zeros(self.O, self.real); p = np.zeros(self.O, self.real)
        rn = C.c_int32(0)
        fn = self._L.az_search_root_stats_f64 if self.real is np.float64 else self._L.az_search_root_stats
        c = self._ck(fn(self._h, slot, side, a.ctypes.data, n.ctypes.data, w.ctypes.data, p.ctypes.data, C.byref(rn)))
        return a[:c], n[:c], w[:c], p[:c], rn.value

    def choose(self, slot, side, u):
        m = C.c_int32(0)
        self._ck(self._L.az_search_choose(self._h, slot, side, float(u), C.byref(m)))
        return m.value

    def advance_tree(self, slot, side, move):
        self._ck(self._L.az_search_advance_tree(self._h, slot, side, int(move)))

    def advance(self, slots, moves):
        s, m = _i32(slots), _i32(moves)
        self._ck(self._L.az_search_advance(self._h, s.ctypes.data, s.size, m.ctypes.data))

    def grow_trees(self, edge_cap=0, block_cap=0):
        """Raise every tree's limits (0 = double) and the shared edge pool with them; contents are kept."""
        self._ck(self._L.az_search_grow_trees(self._h, int(edge_cap), int(block_cap)))

    def grow_pool(self, pool_edges=0):
        """Enlarge only the shared pool of tree edges (0 = double, never beyond every tree at its limit); trees are kept."""
        self._ck(self._L.az_search_grow_pool(self._h, int(pool_edges)))

    def set_c_punt(self, c_punt):
        """Assigning the module global c_punt of the reference's search module (search/mcts.py:9): used from the next
        selection step on."""
        self._ck(self._L.az_search_set_c_punt(self._h, float(c_punt)))
        self.cfg.c_punt = float(c_punt)

    def tree_size(self, slot, side):
        e, b = C.c_int32(0), C.c_int32(0)
        self._ck(self._L.az_search_tree_size(self._h, slot, side, C.byref(e), C.byref(b)))
        return e.value, b.value

    # ---- self-play machine ----------------------------------------------------
    def selfplay_start(self):
        self._ck(self._L.az_selfplay_start(self._h))

    def selfplay_run(self, n_rounds):
        self._ck(self._L.az_selfplay_run(self._h, int(n_rounds)))

    def selfplay_select(self):
        n = C.c_int32(0)
        self._ck(self._L.az_selfplay_select(self._h, C.byref(n)))
        return n.value

    def selfplay_finish(self, priors=None, values=None):
        if priors is None:
            self._ck(self._L.az_selfplay_finish(self._h, None, None))
            return
        p = np.ascontiguousarray(priors, self.real)
        v = np.ascontiguousarray(values, self.real)
        assert p.shape == (self.K * self.G, self.O) and v.shape == (self.K * self.G,)
        fn = self._L.az_selfplay_finish_f64 if self.real is np.float64 else self._L.az_selfplay_finish
        self._ck(fn(self._h, p.ctypes.data, v.ctypes.data))

    def selfplay_status(self):
        plies = np.zeros(self.G, np.int32); done = np.zeros(self.G, np.int32)
        self._ck(self._L.az_selfplay_status(self._h, plies.ctypes.data, done.ctypes.data))
        return plies, done

    def trajectory_pop(self, slot, restart=False):
        mp = self._L.az_trajectory_max_plies(self._h)
        moves = np.zeros(mp, np.int16); visits = np.zeros((mp, self.O), np.int32)
        w = C.c_int32(0)
        n = self._ck(self._L.az_trajectory_pop(self._h, slot, moves.ctypes.data, visits.ctypes.data, C.byref(w), int(restart)))
        return dict(moves=moves[:n].copy(), visits=visits[:n].copy(), winner=w.value)

    def trajectory_pop_many(self, slots, plies, restart=False, static=False):
        """Trajectories of several finished slots with one device round trip; `plies` are the slots' ply counts from
        selfplay_status().  Returns a list of dicts like trajectory_pop.  static=True goes through
        az_trajectory_pop_many_static (fixed staging array inside the library, no allocation on the device)."""
        s = _i32(slots)
        if s.size == 0:
            return []
        mp = self._L.az_trajectory_max_plies(self._h)
        n_rows = np.minimum(_i32(plies), mp).astype(np.int64)
        start = np.zeros(s.size, np.int32)
        start[1:] = np.cumsum(n_rows)[:-1]
        total = int(n_rows.sum())
        moves = np.empty(max(total, 1), np.int16); visits = np.empty((max(total, 1), self.O), np.int32)
        win = np.zeros(s.size, np.int32)
        fn = self._L.az_trajectory_pop_many_static if static else self._L.az_trajectory_pop_many
        self._ck(fn(self._h, s.ctypes.data, s.size, start.ctypes.data, total, moves.ctypes.data, visits.ctypes.data,
                    win.ctypes.data, int(restart)))
        return [dict(moves=moves[a:a + k].copy(), visits=visits[a:a + k].copy(), winner=int(w))
                for a, k, w in zip(start.tolist(), n_rows.tolist(), win.tolist())]

    def trajectory_pop_many_torch(self, slots, plies, restart=False):
        """Finished slots' trajectories as device tensors: (moves int16 (rows,), visits int32 (rows, O), plies int64
        (n,), winners int32 (n,)), the games' rows back to back; nothing crosses PCIe but the per-game scalars."""
        import torch
        s = _i32(slots)
        mp = self._L.az_trajectory_max_plies(self._h)
        n_rows = np.minimum(_i32(plies), mp).astype(np.int64)
        start = np.zeros(s.size, np.int32)
        if s.size:
            start[1:] = np.cumsum(n_rows)[:-1]
        total = int(n_rows.sum())
        dev = torch.device("cuda", self.cfg.device)
        moves = torch.empty(total, dtype=torch.int16, device=dev)
        visits = torch.empty((total, self.O), dtype=torch.int32, device=dev)
        win = np.zeros(s.size, np.int32)
        torch.cuda.current_stream(dev).synchronize()
        self._ck(self._L.az_trajectory_pop_many_dev(self._h, s.ctypes.data, s.size, start.ctypes.data, total, moves.data_ptr(),
                                                    visits.data_ptr(), win.ctypes.data, int(restart)))
        return moves, visits, n_rows, win

    # ---- rng ---------------------------------------------------------------------
    def rng_inject(self, slot, tids=(), etas=(), unis=()):
        t = np.ascontiguousarray(tids, np.int32); e = np.ascontiguousarray(etas, np.float64)
        u = np.ascontiguousarray(unis, np.float64)
        self._ck(self._L.az_rng_inject(self._h, slot, t.ctypes.data, t.size, e.ctypes.data, e.size, u.ctypes.data, u.size))

    def rng_cursor(self, slot):
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._ck(self._L.az_rng_cursor(self._h, slot, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # ---- tower ---------------------------------------------------------------------
    def tower_blob_floats(self):
        return self._L.az_tower_blob_floats(self._h)

    def tower_load(self, blob, net=0):
        b = np.ascontiguousarray(blob, np.float32)
        if net == 0:
            self._ck(self._L.az_tower_load_weights(self._h, b.ctypes.data, b.size))
        else:
            self._ck(self._L.az_tower_load_weights_net(self._h, int(net), b.ctypes.data, b.size))

    def tower_load_torch(self, blob, net=0):
        """The folded float32 blob as a CUDA tensor on this engine's GPU (what an RCCL broadcast delivers): converted and
        copied into the tower by kernels, no pass through host memory."""
        import torch
        assert blob.is_cuda and blob.dtype == torch.float32 and blob.is_contiguous() and blob.device.index == self.cfg.device
        torch.cuda.current_stream(blob.device).synchronize()   # the engine reads it on its own stream
        self._ck(self._L.az_tower_load_weights_dev(self._h, int(net), blob.data_ptr(), blob.numel()))

    def match_assign(self, slots, net_of_first):
        """Two-network match: in slots[i] the side moving first is evaluated by network net_of_first[i]
        (evaluator.py:81-84).  Empty `slots` switches back to one network."""
        s = np.ascontiguousarray(slots, np.int32)
        nf = np.ascontiguousarray(net_of_first, np.int32)
        self._ck(self._L.az_match_assign(self._h, s.ctypes.data if s.size else None, s.size, nf.ctypes.data if nf.size else None))

    def match_leaf_nets(self):
        """Network (0 / 1) that evaluates this round's leaf of every slot, -1 where no evaluation is needed."""
        out = np.zeros(self.G, np.int32)
        self._ck(self._L.az_match_leaf_nets(self._h, out.ctypes.data))
        return out

    def tower_forward(self, planes):
        x = np.ascontiguousarray(planes, np.int8)
        B = x.shape[0]
        p = np.zeros((B, self.O), np.float32); v = np.zeros(B, np.float32)
        self._ck(self._L.az_tower_forward(self._h, x.ctypes.data, B, p.ctypes.data, v.ctypes.data))
        return p, v

    def tower_forward_torch(self, planes):
        """planes: int8 CUDA tensor (B,17,H,W) on this engine's GPU -> (priors (B,O), values (B,)) float32 CUDA tensors;
        nothing crosses PCIe."""
        import torch
        assert planes.is_cuda and planes.dtype == torch.int8 and planes.is_contiguous()
        B = planes.shape[0]
        p = torch.empty((B, self.O), dtype=torch.float32, device=planes.device)
        v = torch.empty((B,), dtype=torch.float32, device=planes.device)
        torch.cuda.current_stream(planes.device).synchronize()   # the engine runs on its own stream
        self._ck(self._L.az_tower_forward_dev(self._h, planes.data_ptr(), B, p.data_ptr(), v.data_ptr()))
        return p, v

    def tower_profile(self, on=True):
        """False/0: off; True/1: per-launch conv timing on every round; N >= 8: on every N-th round only."""
        self._ck(self._L.az_tower_profile(self._h, int(on)))

    def tower_leaf_compaction(self, on=True):
        self._ck(self._L.az_tower_leaf_compaction(self._h, int(on)))

    def tower_conv_kernel(self):
        """Name and description of the conv kernel the residual layers of a self-play round run with."""
        return self._L.az_tower_conv_kernel(self._h).decode()

    def tower_flops_per_eval(self):
        return self._L.az_tower_flops_per_eval(self._h)

    # ---- measurement -----------------------------------------------------------------
    def counters(self):
        c = AzCounters()
        self._ck(self._L.az_counters_get(self._h, C.byref(c)))
        return {k: getattr(c, k) for k, _ in AzCounters._fields_}

    def counters_reset(self):
        self._ck(self._L.az_counters_reset(self._h))

    def sync(self):
        self._ck(self._L.az_sync(self._h))
